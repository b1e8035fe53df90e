#!/bin/bash
# What the driver runs at round end, in its order, on a fresh box: GPU suite (-x), smoke(), default bench; then kernel stats.
out=gpurun_out/${1:-r2_driver}
mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -15 $out/gpu_tests.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1
tail -3 $out/smoke.log
( time python bench.py ) > $out/bench.json 2> $out/bench.err
tail -c 7000 $out/bench.json
tail -5 $out/bench.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --quick > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; python3 tools/kstats.py $f 23 > $out/kernel_stats.txt 2>&1
head -40 $out/kernel_stats.txt
