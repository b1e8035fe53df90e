#!/bin/bash
# round 6: same-box A/B of the fused train step between library builds: r6_ab_train.sh OUT "LIBS (space separated, '' = shipped)" [BATCH] [FORM]
out=gpurun_out/${1:-r6_ab}; libs=${2:-""}; batch=${3:-16384}; form=${4:-fused_eager}; mkdir -p $out
for rep in $(seq 1 ${REPS:-2}); do
for lib in main $libs; do
  if [ "$lib" = main ]; then unset T2N_LIB; else export T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_$lib.so; fi
  python3 tools/experiments/train_only.py 2 20 $batch $form 2>&1 | grep "train blocks" | sed "s/^/$lib rep$rep: /" | tee -a $out/ab.txt
done
done
