#!/bin/bash
# Run the train loop with each libt2n_var_*.so swapped in as libt2n_hip.so (on the GPU box).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp text2nerf_amd/libt2n_hip.so /tmp/libt2n_hip.so.orig
for v in text2nerf_amd/libt2n_var_*.so; do
  cp $v text2nerf_amd/libt2n_hip.so
  echo "== $v"; python tools/experiments/train_only.py 1 40 2>&1 | tail -1
done
cp /tmp/libt2n_hip.so.orig text2nerf_amd/libt2n_hip.so
echo "== baseline"; python tools/experiments/train_only.py 1 40 2>&1 | tail -1
