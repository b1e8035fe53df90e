out=gpurun_out/r6_tl2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T2N_TRAIN_BLOCKS=4 rocprofv3 --kernel-trace --output-format csv -d $out/prof -o t -- python3 tools/experiments/train_only.py 2 20 16384 fused_eager > $out/train.log 2>&1
grep "train blocks" $out/train.log
f=$(find $out/prof -name "*kernel_trace.csv" | head -1)
python3 tools/train_timeline.py $f 35 > $out/timeline_block2.txt 2>&1
python3 tools/train_timeline.py $f 58 > $out/timeline_block3.txt 2>&1
python3 tools/train_timeline.py $f 36 | head -1; python3 tools/train_timeline.py $f 37 | head -1; python3 tools/train_timeline.py $f 59 | head -1; python3 tools/train_timeline.py $f 60 | head -1
rm -rf $out/prof
