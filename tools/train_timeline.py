"""Timeline of ONE fused train iteration from a rocprofv3 --kernel-trace CSV: kernels in start order with start offset, duration, the
gap to the previous kernel's end on the same queue, and the iteration's idle time (python tools/train_timeline.py kernel_trace.csv [iter])."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
names = [r["Kernel_Name"] for r in rows]
# iteration boundaries: every k_train_loss marks one iteration; take the requested one from the k_march<true> before it to the next one
import os
mark = os.environ.get("T2N_TIMELINE_MARK", "k_march<true")     # first kernel of an iteration (render frames: k_march_tiles)
idx = [i for i, n in enumerate(names) if mark in n.replace(" ", "") or (mark == "k_march<true" and "k_marchILb1" in n)]
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[it], idx[it + 1]
t0 = rows[a]["s"]
qend = collections.defaultdict(lambda: t0)
busy_until, idle = t0, 0
print(f"iteration {it}: {len(rows[a:b])} kernels, wall {(rows[b]['s'] - t0) / 1e3:.1f} us")
for r in rows[a:b]:
    q = r.get("Queue_Id", "0")
    gap = r["s"] - qend[q]
    qend[q] = r["e"]
    if r["s"] > busy_until:
        idle += r["s"] - busy_until
    busy_until = max(busy_until, r["e"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t2n::", "")[:46]
    print(f"  +{(r['s'] - t0) / 1e3:8.1f} us  {(r['e'] - r['s']) / 1e3:7.1f} us  q{q[-2:]}  gap {gap / 1e3:7.1f}  {nm}")
print(f"idle (no kernel running) inside the iteration: {idle / 1e3:.1f} us")
