#!/bin/bash
out=gpurun_out/${1:-r2x}
mkdir -p $out
( T2N_SHADE_NO_WS=1 python -m pytest tests/test_hip_parity.py -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -3 $out/gpu_tests.log
T2N_SHADE_NO_WS=1 python bench.py --no-train --steps 30 > $out/bench_coop.json 2> $out/bench_coop.err
python bench.py --no-train --no-cpu-baseline --steps 30 > $out/bench_ws.json 2> $out/bench_ws.err
python - <<PY
import json
for n in ("coop", "ws"):
    try:
        d = json.loads([l for l in open("$out/bench_%s.json" % n) if l.startswith("{")][0])
        print(n, "ms/step", round(d["ms_per_step"], 3), d["config"]["kernel_ms_per_frame"], d.get("cpu_baseline", {}).get("parity_vs_oracle"))
    except Exception as e:
        print(n, "failed", e); print(open("$out/bench_%s.err" % n).read()[-2000:])
PY
