#!/bin/bash
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
grep -A22 "k_march\|k_density_tiles" $OUT/summary.txt
