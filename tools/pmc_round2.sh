#!/bin/bash
# PMC passes of the bench command (one 800x800 C2 frame per step) -> gpurun_out/<dir>/ + a per-kernel JSON (tools/pmc_round2.py).
# FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: 3 + 2 TCC slots); counters only, no trace domains.
OUT=${1:-gpurun_out/pmc_r2}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --quick "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
python3 tools/pmc_round2.py $OUT > $OUT/round2_pmc.json
cat $OUT/round2_pmc.json
