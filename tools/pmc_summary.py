#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: mean counter value per dispatch, per kernel. Usage: pmc_summary.py DIR..."""
import collections
import csv
import glob
import os
import sys


def main(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        if not any(t in k for t in ("k_march", "k_shade", "k_composite", "k_bwd", "k_")):
            continue
        print(k)
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"    {c:40s} mean/dispatch {sum(v) / len(v):18.1f}   dispatches {len(v)}")


if __name__ == "__main__":
    main(sys.argv[1:] or ["."])
