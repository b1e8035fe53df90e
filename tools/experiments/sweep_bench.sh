#!/bin/bash
# Render bench (fp32 + bf16 storage) with each libt2n_var_*.so swapped in as libt2n_hip.so (on the GPU box).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
cp text2nerf_amd/libt2n_hip.so /tmp/libt2n_hip.so.orig
for v in text2nerf_amd/libt2n_var_*.so; do
  cp $v text2nerf_amd/libt2n_hip.so
  echo "== $v"; python bench.py --no-train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['kernel_ms_per_frame'], d['config'].get('bf16_factor_storage_ms_per_step'))"
done
cp /tmp/libt2n_hip.so.orig text2nerf_amd/libt2n_hip.so
