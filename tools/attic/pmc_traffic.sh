#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md: 3 + 2 TCC slots).
OUT=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $REPO
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pass$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-train "$@" > $OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
