#!/usr/bin/env python3
"""Per-iteration view of a rocprofv3 *_kernel_stats.csv: kstats.py FILE [ITERS] [TOPN]."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tot = 0.0
for r in rows:
    tot += float(r["TotalDurationNs"]) / iters / 1e3
for r in rows[:top]:
    per = float(r["TotalDurationNs"]) / iters / 1e3
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} avg {float(r['AverageNs']) / 1e3:8.1f} us  per-iter {per:7.1f} us")
print(f"all kernels: {tot:.1f} us per iteration")
