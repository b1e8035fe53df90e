#!/bin/bash
# Where a wave of k_march_tiles spends its cycles ("give k_march_tiles the same per-region accounting the head got"):
#   bash tools/build_variant.sh mtprof -DMT_PROF      (here, then on the GPU box:)
#   bash tools/r5_march_accounting.sh [outdir]
# -> per-wave s_memtime sums per region of the step loop (instrumented build), the shipped build's kernel times on the same box, the
# static census of the loop's basic blocks (tools/isa_census.py) and the PMC instruction counts per frame when a record exists.
out=gpurun_out/${1:-r5_march_accounting}; mkdir -p $out
{
  echo "# k_march_tiles<false,false,false>, C2 frame (800x800, 518 samples per ray, S1-soft): 10 000 waves (8x8-pixel tiles), five per SIMD"
  T2N_LIB=$PWD/text2nerf_amd/libt2n_hip_mtprof.so python3 bench.py --no-train --steps 40 --quick --no-cpu-baseline 2>$out/err.txt >$out/prof_bench.json
  grep -h "mt prof" $out/err.txt | tail -1
  python3 -c "
import json
d=json.loads(open('$out/prof_bench.json').read().strip().splitlines()[-1]); print('instrumented build: march', round(d['config']['kernel_ms_per_frame']['march'],4), 'ms per frame')"
  for rep in 1 2; do
    python3 bench.py --no-train --steps 100 --quick --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('shipped build: ms/step', round(d['ms_per_step'], 3), {k: round(x, 4) for k, x in d['config']['kernel_ms_per_frame'].items()})"
  done
} 2>&1 | tee $out/accounting.txt
