#!/bin/bash
# GPU suite (-x) + smoke + a quick bench line
out=gpurun_out/${1:-r3_suite}
mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -25 $out/gpu_tests.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1
tail -2 $out/smoke.log
python bench.py --steps 50 --quick --no-cpu-baseline --no-train > $out/bench_quick.json 2> $out/bench_quick.err
python - <<PY
import json
d = json.loads(open("$out/bench_quick.json").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in d["config"]["kernel_ms_per_frame"].items()})
PY
