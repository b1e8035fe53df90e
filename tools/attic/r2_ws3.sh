#!/bin/bash
out=gpurun_out/${1:-r2d}
mkdir -p $out
( time python -m pytest tests/test_hip_parity.py -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
python bench.py --no-train --steps 30 > $out/bench_ws.json 2> $out/bench_ws.err
python - <<PY
import json
try:
    d = json.loads([l for l in open("$out/bench_ws.json") if l.startswith("{")][0])
    print("ms/step", round(d["ms_per_step"], 3), d["config"]["kernel_ms_per_frame"], d.get("cpu_baseline", {}).get("parity_vs_oracle"))
except Exception as e:
    print("failed", e); print(open("$out/bench_ws.err").read()[-2000:])
PY
T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_phase.so python tools/experiments/ss_phase.py 2>&1 | tee $out/ws_phase.txt
