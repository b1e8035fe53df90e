#!/bin/bash
# round 5: same-box A/B of library variants on the fused train step (bench.py's loop, three blocks of 20): tools/r5_ab_train.sh REPS BATCH variant...
reps=$1; batch=$2; shift; shift
for i in $(seq 1 $reps); do
  for v in "$@"; do
    lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
    T2N_LIB=$lib python3 tools/experiments/train_only.py 3 20 $batch 2>&1 | grep "train blocks" | sed "s/^/$v rep $i resident $batch: /"
    T2N_LIB=$lib python3 tools/experiments/train_only.py 2 20 $batch 2>&1 | grep "train blocks" | sed "s/^/$v rep $i host     $batch: /"
  done
done
