#!/bin/bash
# Collect PMC counters for bench.py kernels in separate rocprofv3 passes (--pmc only; no trace domains).
# Usage (on the GPU box, from the repo root): tools/pmc.sh OUTDIR [bench args]
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
