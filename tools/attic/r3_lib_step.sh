#!/bin/bash
# fused train step (resident set, 60 iterations) per library variant, three repetitions, + the default-stream timeline of the last variant
for rep in 1 2 3; do
  for v in "$@"; do
    lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
    echo -n "rep $rep $v: "
    T2N_LIB=$lib python3 tools/experiments/train_only.py 3 60 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['train_ms_per_iter_fused_step_resident'], 4))"
  done
done
