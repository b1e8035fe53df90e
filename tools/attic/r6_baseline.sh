#!/bin/bash
# round 6 baseline on one fresh box: GPU suite, the driver's bench command, the default bench
tag=${1:-base}
out=gpurun_out/r6_$tag
mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_driverlike.json 2> $out/bench_driverlike.err
python3 - <<PY
import json
d=json.loads(open("$out/bench_driverlike.json").read().strip().splitlines()[-1]); c=d["config"]
print("bench ms_per_step", round(d["ms_per_step"],4), "|", c["ms_per_step_blocks"], "| fused step", c.get("train_ms_per_iter_fused_step"), "| frac", d["roofline"].get("frac"))
print({k:v for k,v in c.items() if k.startswith("train_")})
PY
