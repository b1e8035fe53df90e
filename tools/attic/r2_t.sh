#!/bin/bash
out=gpurun_out/${1:-r2t}
mkdir -p $out
( python -m pytest tests/test_train_step.py tests/test_hip_parity.py tests/test_zz_bench_multirank.py -q -m gpu -x ) > $out/gpu_tests.log 2>&1
tail -30 $out/gpu_tests.log
python bench.py --steps 20 --quick --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - <<PY
import json
d = json.loads([l for l in open("$out/bench.json") if l.startswith("{")][0])
print("ms/step", round(d["ms_per_step"], 3), d["config"]["kernel_ms_per_frame"])
print({k: v for k, v in d["config"].items() if k.startswith("train")})
PY
tail -3 $out/bench.err
