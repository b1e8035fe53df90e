#!/bin/bash
# kernel timeline of one C2 render frame
out=gpurun_out/${1:-r3_frametrace}; shift; mkdir -p $out
for v in "$@"; do export $v; done
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --quick --no-train > $out/bench.log 2>&1
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
T2N_TIMELINE_MARK=k_march_tiles python3 tools/train_timeline.py $f 8 > $out/timeline.txt 2>&1
cat $out/timeline.txt
rm -rf $out/prof
