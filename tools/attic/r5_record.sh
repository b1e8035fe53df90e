#!/bin/bash
# Round record on one fresh box: GPU suite (-x), smoke, the default bench, kernel stats (rocprofv3 --kernel-trace --stats), PMC passes.
#   tools/r5_record.sh TAG   -> gpurun_out/r5_TAG/{gpu_tests.log, smoke.log, bench.json, kernel_stats.csv/.txt, pmc.json, pmc_summary.txt}
tag=${1:-v1}
out=gpurun_out/r5_$tag
mkdir -p $out
( time python -m pytest tests/ -x -q -m gpu ) > $out/gpu_tests.log 2>&1
tail -4 $out/gpu_tests.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1
tail -2 $out/smoke.log | head -1
( time python bench.py ) > $out/bench.json 2> $out/bench.err
python3 - <<PY
import json
d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1]); c=d["config"]
print("bench ms_per_step", round(d["ms_per_step"],4), "|", c["ms_per_step_blocks"], "| sustained", c.get("sustained_ms_per_step"), "| fused step", c.get("train_ms_per_iter_fused_step"), "| frac", d["roofline"].get("frac"))
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o p -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --quick > $out/prof.log 2>&1
f=$(find $out/prof -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv; python3 tools/kstats.py $f 23 > $out/kernel_stats.txt 2>&1
head -14 $out/kernel_stats.txt
bash tools/pmc_round2.sh $out/pmc > $out/pmc.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $out/pmc/pass_ea -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --quick > $out/pmc/pass_ea.log 2>&1 || echo "EA pass failed"
timeout 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum --output-format csv -d $out/pmc/pass_eaw -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train --quick > $out/pmc/pass_eaw.log 2>&1 || echo "EA write pass failed"
python3 tools/pmc_summary.py $out/pmc > $out/pmc/summary.txt
cp $out/pmc/round2_pmc.json $out/pmc.json; cp $out/pmc/summary.txt $out/pmc_summary.txt
rm -rf $out/prof $out/pmc/pass*    # keep the merge small
tail -3 $out/pmc.log | cut -c1-300
