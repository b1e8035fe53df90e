#!/bin/bash
# L1 (TCP) / texture-addresser counters of the frame's kernels: is k_app_features bound by addresser issue or by L1 misses?
OUT=${1:-gpurun_out/pmc_tcp}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --quick "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
  tail -2 $OUT/pass$i.log | cut -c1-300
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(x in k for x in ("k_app_features", "k_mlp_ss", "k_march_tiles", "k_composite")):
        continue
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:44s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
