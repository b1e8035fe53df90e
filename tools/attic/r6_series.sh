out=gpurun_out/r6_series; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T2N_TRAIN_BLOCKS=4 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/experiments/train_only.py 2 20 16384 fused_eager > $out/train.log 2>&1
grep "train blocks" $out/train.log
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_series.py $f 23 k_adam_cl_multi k_tv_seed k_march k_shade_coop k_mlp_bwd_ss k_bwd_tile_accum k_bwd_den_block k_gemm_tn_b k_bwd_march k_wgrad_reduce
rm -rf $out/prof
