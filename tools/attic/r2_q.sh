#!/bin/bash
out=gpurun_out/${1:-r2q}
mkdir -p $out
( python -m pytest tests/test_hip_range.py tests/test_hip_parity.py tests/test_hip_bf16.py -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -12 $out/gpu_tests.log
python bench.py --no-train --steps 30 --quick --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d = json.loads([l for l in open("$out/bench.json") if l.startswith("{")][0])
print("ms/step", round(d["ms_per_step"], 3), d["config"]["kernel_ms_per_frame"])
PY
