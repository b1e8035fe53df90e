#!/bin/bash
# r2_profile_all.sh TAG: default bench line, kernel stats of the render bench, PMC passes, kernel stats of the fused train step
tag=${1:-v17}; out=gpurun_out/r2_$tag; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
tail -c 400 $out/bench.json | head -c 10 > /dev/null
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --quick --no-train > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; python3 tools/kstats.py $f 23 12
bash tools/pmc_round2.sh $out/pmc > /dev/null 2>&1
cp $out/pmc/round2_pmc.json $out/round2_pmc.json; cp $out/pmc/summary.txt $out/pmc_summary.txt
bash tools/r2_trainprof.sh r2_${tag}_train | head -45
