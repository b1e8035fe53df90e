#!/bin/bash
# isolated (T2N_BWD_SERIAL=1) kernel durations of one train iteration per library variant; $1 = grep pattern, rest = variants
pat=$1; shift
for v in "$@"; do
  lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
  echo "== $v"
  T2N_LIB=$lib bash tools/r3_traintrace.sh r3_lib_$v T2N_BWD_SERIAL=1 | grep -i "$pat\|wall"
done
