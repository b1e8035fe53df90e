#!/bin/bash
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in \
  "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" \
  "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A14 "k_march\|k_shade" $OUT/summary.txt
