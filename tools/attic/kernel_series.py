"""Duration series of the kernels matching argv[2:] in a rocprofv3 --kernel-trace CSV, averaged over consecutive groups of argv[2] launches:
python tools/kernel_series.py kernel_trace.csv GROUP name [name ...]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
grp = int(sys.argv[2])
for pat in sys.argv[3:]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
    out = [sum(d[i:i + grp]) / len(d[i:i + grp]) for i in range(0, len(d), grp)]
    print(pat, len(d), " ".join(f"{x:.1f}" for x in out))
