#!/bin/bash
# whole GPU suite (no -x: every failure is listed) + bench
out=gpurun_out/${1:-r2k}
mkdir -p $out
( time python -m pytest tests -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -40 $out/gpu_tests.log
python bench.py ${BENCH_ARGS:---no-train} --steps 30 > $out/bench.json 2> $out/bench.err
tail -c 6000 $out/bench.json
tail -5 $out/bench.err
