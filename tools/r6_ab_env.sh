#!/bin/bash
# round 6: same-box A/B of the fused train step between ENVIRONMENT settings of one library build:
#   r6_ab_env.sh OUT "name1:VAR=val,VAR2=val name2: ..." [BATCHES="16384 2048"] [REPS=2]      ("base:" = no variables)
out=gpurun_out/${1:-r6_abenv}; variants=${2:-"base:"}; batches=${3:-"16384 2048"}; reps=${4:-2}; mkdir -p $out
for batch in $batches; do
for rep in $(seq 1 $reps); do
for v in $variants; do
  name=${v%%:*}; envs=${v#*:}
  ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
    python3 tools/experiments/train_only.py 2 20 $batch fused_eager 2>&1 | grep "train blocks" | sed "s/^/$batch $name rep$rep: /" | tee -a $out/ab.txt )
done
done
done
