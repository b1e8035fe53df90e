#!/bin/bash
# isolated (T2N_BWD_SERIAL=1) durations of the scatter kernels per environment variant
i=0
for v in "$@"; do
  i=$((i+1)); echo "== $v"
  bash tools/r3_traintrace.sh r3_denenv_$i T2N_BWD_SERIAL=1 $v | grep -i "k_bwd_den\|k_bwd_bin\|k_bin_scan\|k_bwd_march\|k_app_bin\|k_bwd_tile\|wall"
done
