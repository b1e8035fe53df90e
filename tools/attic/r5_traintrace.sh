#!/bin/bash
# round 5 (from tools/attic/r3_traintrace.sh): kernel timeline of one C3 train iteration (mode 2 = fused step); further arguments: NAME=VALUE environment for the traced run
out=gpurun_out/${1:-r5_traintrace}; shift; mkdir -p $out
for v in "$@"; do export $v; done
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/experiments/train_only.py ${T2N_TRACE_MODE:-2} ${T2N_TRACE_ITERS:-30} ${T2N_TRACE_BATCH:-16384} ${T2N_TRACE_SPEC:-} > $out/train.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/train_timeline.py $f ${T2N_TRACE_AT:-} > $out/timeline.txt 2>&1
cat $out/timeline.txt
tail -1 $out/train.log | cut -c1-300
rm -rf $out/prof
