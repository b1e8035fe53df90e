#!/bin/bash
# round 3, first GPU call: microbenchmark for the head's shape, the GPU suite, smoke, default bench
out=gpurun_out/${1:-r3_first}
mkdir -p $out
timeout 300 ./tools/experiments/mfma_two_tile.bin > $out/mfma_two_tile.txt 2>&1
cat $out/mfma_two_tile.txt
( time python -m pytest tests/ -x -q -m gpu -s ) > $out/gpu_tests.log 2>&1
tail -25 $out/gpu_tests.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1
tail -3 $out/smoke.log
( time python bench.py ) > $out/bench.json 2> $out/bench.err
tail -c 9000 $out/bench.json
tail -5 $out/bench.err
