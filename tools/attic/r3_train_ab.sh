#!/bin/bash
# gradient parity subset, then the C3 train step (modes 2 = fused step, 0 = reference form) under the environment variants given as arguments
out=gpurun_out/${1:-r3_train_ab}; shift
mkdir -p $out
( python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_train_step.py tests/test_heads.py -x -q -m gpu -k "grad or train or backward or c3" ) > $out/tests.log 2>&1
tail -3 $out/tests.log
for rep in 1 2 3; do
  for v in "$@"; do
    for mode in 3 2 0; do
      echo -n "rep $rep [$v] mode $mode: "
      ( export $v; python3 tools/experiments/train_only.py $mode 60 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if 'ms' in k or 'it' in k})" )
    done
  done
done 2>&1 | tee $out/ab.txt
