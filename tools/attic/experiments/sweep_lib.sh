#!/bin/bash
# Run the train loop with each libt2n_var_*.so selected through T2N_LIB (text2nerf_amd/_lib.py) — the shipped libt2n_hip.so is never
# overwritten, so an interrupted sweep cannot leave a variant installed (on the GPU box).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in text2nerf_amd/libt2n_var_*.so; do
  echo "== $v"; T2N_LIB=$PWD/$v python tools/experiments/train_only.py 1 40 2>&1 | tail -1
done
echo "== baseline"; python tools/experiments/train_only.py 1 40 2>&1 | tail -1
