for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/text2nerf_amd/libt2n_hip_$v.so; [ "$v" = main ] && lib=$GRAFT_REPO_ROOT/text2nerf_amd/libt2n_hip.so
  out=$GRAFT_REPO_ROOT/gpurun_out/ks_$v; rm -rf $out; mkdir -p $out
  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  T2N_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py --steps 20 --no-cpu-baseline --quick --no-train > $out/log.txt 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1); echo "== $v"; python3 tools/kstats.py $f 23 14 | grep "k_ray_stats\|all kernels\|k_shade\|fillBuffer\|k_compact"
done
