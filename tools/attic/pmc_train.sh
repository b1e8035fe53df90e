#!/bin/bash
# VALU / LDS / MFMA instruction counts per kernel of the fused train step (one PMC pass; counters only)
OUT=${1:-gpurun_out/pmc_train}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pass1 -o p -- python3 tools/experiments/train_only.py 2 20 > $OUT/pass1.log 2>&1 || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("t2n::", "")[:44]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    n = max(len(v) for v in c.values())
    act = m.get("GRBM_GUI_ACTIVE", 0) / 8.0   # cycles per launch (8 XCDs)
    rows.append((act * n, k, n, m, act))
for tot, k, n, m, act in sorted(rows, reverse=True)[:22]:
    valu = m.get("SQ_INSTS_VALU", 0) - m.get("SQ_INSTS_MFMA", 0)
    per_simd = valu / 1024.0
    print(f"{k:46s} n={n:4d} cycles/launch {act:10.0f}  VALU {valu/1e6:8.2f} M  ({per_simd/max(act,1):5.2f} per SIMD-cycle)  LDS {m.get('SQ_INSTS_LDS',0)/1e6:7.2f} M  SALU {m.get('SQ_INSTS_SALU',0)/1e6:7.2f} M")
PY
