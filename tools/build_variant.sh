#!/bin/bash
# build_variant.sh NAME [extra hipcc flags]: text2nerf_amd/libt2n_hip_NAME.so from per-file objects (parallel, cached by flags);
# select it with T2N_LIB. NAME = "main" builds the shipped text2nerf_amd/libt2n_hip.so the same way.
set -e
name=$1; shift
extra="$*"
root=$(cd $(dirname $0)/.. && pwd)
src=$root/text2nerf_amd/csrc
tag=$(echo "$extra" | md5sum | cut -c1-8)
obj=$root/build/obj_$tag
mkdir -p $obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics -Wall -Wno-unused-function -DNDEBUG $extra"
pids=()
for f in $src/*.hip; do
  o=$obj/$(basename $f .hip).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find $src $root/include -name '*.h' -newer $o)" ]; then
    per=""; [ "$(basename $f)" = t2n_mlp_bwd_ss.hip ] && per="-fno-slp-vectorize"   # as text2nerf_amd/build.py PER_FILE
    /opt/rocm/bin/hipcc $FLAGS $per -c $f -o $o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
# the AccVGPR / spill guard of text2nerf_amd/build.py, on this variant's flags; a refused object is removed so that it is never linked
if [ ! -f $obj/t2n_mlp_ss.checked ] || [ $obj/t2n_mlp_ss.o -nt $obj/t2n_mlp_ss.checked ]; then
  (cd $root && python3 -m text2nerf_amd.build --check-ss3 $obj $extra) || { rm -f $obj/t2n_mlp_ss.o $obj/t2n_mlp_ss.checked; exit 1; }
  touch $obj/t2n_mlp_ss.checked
fi
out=$root/text2nerf_amd/libt2n_hip_$name.so
[ "$name" = main ] && out=$root/text2nerf_amd/libt2n_hip.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined $obj/*.o -o $out.tmp
mv $out.tmp $out
echo $out
