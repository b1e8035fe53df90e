#!/bin/bash
# r2_variants.sh NAME...: quick bench (kernel ms per C2 frame) with text2nerf_amd/libt2n_hip_NAME.so selected through T2N_LIB
for v in "$@"; do
  lib=$PWD/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
  T2N_LIB=$lib python bench.py --no-train --steps 30 --quick --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$v', 'ms/step', round(d['ms_per_step'], 3), {k: round(x, 3) for k, x in d['config']['kernel_ms_per_frame'].items()})"
done
