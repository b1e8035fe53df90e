#!/bin/bash
# isolated (T2N_BWD_SERIAL=1) durations of the density scatter kernels per library variant
for v in "$@"; do
  lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
  echo "== $v"
  T2N_LIB=$lib bash tools/r3_traintrace.sh r3_den_$v T2N_BWD_SERIAL=1 | grep -i "k_bwd_den\|k_bwd_bin\|k_bin_scan\|wall"
done
