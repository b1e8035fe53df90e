#!/bin/bash
# round 6: kernel timeline of one train iteration. r6_traintrace.sh NAME FORM [BATCH] ; FORM = legacy | fused_eager | fused_graph
out=gpurun_out/${1:-r6_traintrace}; form=${2:-fused_eager}; batch=${3:-16384}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/experiments/train_only.py 2 30 $batch $form > $out/train.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/train_timeline.py $f ${T2N_TRACE_AT:-} > $out/timeline.txt 2>&1
cat $out/timeline.txt
tail -1 $out/train.log | cut -c1-300
python3 tools/kernel_hist.py $f > $out/hist.txt 2>&1
rm -rf $out/prof
