#!/bin/bash
# round 5: same-box A/B of library variants (T2N_LIB): kernel ms per frame of the quick bench, alternating, $1 repetitions
#   tools/r5_ab.sh 3 base main [other ...]   (variants: text2nerf_amd/libt2n_hip_NAME.so, main = the shipped library)
reps=$1; shift
for i in $(seq 1 $reps); do
  for v in "$@"; do
    lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
    T2N_LIB=$lib python3 bench.py --quick --no-train --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['config']['kernel_ms_per_frame']
print('$v rep $i: frame %.4f  march %.4f  app_features %.4f  shade %.4f  composite %.4f' % (d['ms_per_step'], k.get('march',0), k.get('app_features',0), k.get('shade',0), k.get('composite',0)))"
  done
done
