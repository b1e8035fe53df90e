#!/bin/bash
# round 4: three-tile head kernel (default) against the two-waves-per-SIMD one (T2N_SS_TWO_WAVE=1) on one box:
# the head's parity tests on the default build, then the quick bench line of each variant, interleaved, three times
out=gpurun_out/${1:-r4_head_ab}
mkdir -p $out
( python -m pytest tests/test_hip_parity.py tests/test_hip_range.py tests/test_hip_fullsize.py tests/test_list_budget.py tests/test_kept_rows_marker.py tests/test_hip_bf16.py -x -q -m gpu -k "g5 or range or whole_frame or big300 or budget or kept or g6 or g7 or bf16" ) > $out/tests.log 2>&1
tail -8 $out/tests.log
for rep in 1 2 3; do
  for v in 0 1; do
    T2N_SS_TWO_WAVE=$v python bench.py --steps 100 --quick --no-cpu-baseline --no-train > $out/bench_$v.json 2> $out/bench_$v.err
    python - <<PY
import json
d = json.loads(open("$out/bench_$v.json").read().strip().splitlines()[-1])
print("two_wave=$v rep=$rep ms/step", round(d["ms_per_step"], 3), {k: round(x, 4) for k, x in d["config"]["kernel_ms_per_frame"].items()})
PY
  done
done 2>&1 | tee $out/ab.txt
