out=gpurun_out/r6_genprof; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/experiments/general_probe.py 128 400 > $out/log.txt 2>&1
tail -3 $out/log.txt
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_hist.py $f | grep "k_gen\|k_dense" 
rm -rf $out/prof
