#!/bin/bash
# r4_ss3_variants.sh build|run NAME=FLAGS... : timing-only ablation builds of the three-tile head kernel (only t2n_mlp_ss.hip is
# recompiled per variant; the other objects come from build/obj). `run` prints the head's ms per C2 frame for each (results invalid).
set -e
mode=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-function -DNDEBUG"
if [ "$mode" = build ]; then
  mkdir -p $root/build/ss3
  pids=()
  for spec in "$@"; do
    name=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
    ( /opt/rocm/bin/hipcc $FLAGS $flags -c $root/text2nerf_amd/csrc/t2n_mlp_ss.hip -o $root/build/ss3/mlp_ss_$name.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined $(ls $root/build/obj/*.o | grep -v t2n_mlp_ss.o) $root/build/ss3/mlp_ss_$name.o -o $root/text2nerf_amd/libt2n_hip_$name.so ) &
    pids+=($!)
    [ ${#pids[@]} -ge 4 ] && { wait ${pids[0]}; pids=("${pids[@]:1}"); }
  done
  wait
  ls -la $root/text2nerf_amd/libt2n_hip_*.so
else
  out=gpurun_out/${R4_OUT:-r4_ss3_variants}; mkdir -p $out
  for rep in 1 2; do
    for spec in "$@"; do
      name=${spec%%=*}
      lib=$PWD/text2nerf_amd/libt2n_hip_$name.so; [ "$name" = main ] && lib=$PWD/text2nerf_amd/libt2n_hip.so
      T2N_LIB=$lib python bench.py --no-train --steps 40 --quick --no-cpu-baseline 2>$out/err_$name.txt | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$name', 'ms/step', round(d['ms_per_step'], 3), {k: round(x, 4) for k, x in d['config']['kernel_ms_per_frame'].items()})"
      grep -h 'ss3 prof' $out/err_$name.txt || true
    done
  done 2>&1 | tee $out/variants.txt
fi
