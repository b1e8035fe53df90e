#!/bin/bash
out=gpurun_out/${1:-r2tp}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o t -- python3 tools/experiments/train_only.py 2 40 > $out/train.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1)
cp $f $out/train_kernel_stats.csv
python3 tools/kstats.py $out/train_kernel_stats.csv 43 45
tail -2 $out/train.log
