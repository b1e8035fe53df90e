#!/bin/bash
# round 5: the driver's exact bench command on a fresh box, N times in fresh processes (first = what the driver sees after smoke)
tag=${1:-a}; n=${2:-2}; extra=${3:-}   # extra: e.g. --no-inregion-timing
out=gpurun_out/r5_driverlike_$tag
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -1 $out/smoke.log
for i in $(seq 1 $n); do
  ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 $extra ) > $out/bench_$i.json 2> $out/bench_$i.err
  python3 - <<PY
import json
d=json.loads(open("$out/bench_$i.json").read().strip().splitlines()[-1])
c=d["config"]
print("run $i ms_per_step", round(d["ms_per_step"],4), "blocks", c.get("ms_per_step_blocks"), "sustained", c.get("sustained_ms_per_step"), "gap", c.get("host_gap_ms_per_step"), "head", d["roofline"].get("avg_launch_ms"))
PY
  tail -3 $out/bench_$i.err
done
