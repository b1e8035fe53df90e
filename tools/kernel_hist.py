"""Per-kernel launch counts and total time of a rocprofv3 --kernel-trace CSV (python tools/kernel_hist.py kernel_trace.csv)."""
import csv, sys, collections
c = collections.Counter(); t = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t2n::", "")[:50]
    c[n] += 1; t[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, k in c.most_common():
    print(f"{k:6d}  {t[n] / k / 1e3:9.1f} us  {n}")
