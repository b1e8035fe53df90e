#!/bin/bash
# the GPU suite several times over (no -x): lists every test that fails in any run
out=gpurun_out/${1:-r2_flake}; n=${2:-3}
mkdir -p $out
for i in $(seq 1 $n); do
  ( time python -m pytest tests/ -q -m gpu -p no:cacheprovider ) > $out/run$i.log 2>&1
  tail -4 $out/run$i.log | head -2; grep -E "^FAILED|^ERROR" $out/run$i.log
done
