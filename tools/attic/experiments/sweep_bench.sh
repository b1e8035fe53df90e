#!/bin/bash
# Render bench (fp32 + bf16 storage) with each libt2n_var_*.so selected through T2N_LIB (the shipped library is never overwritten).
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in text2nerf_amd/libt2n_var_*.so; do
  echo "== $v"; T2N_LIB=$PWD/$v python bench.py --no-train --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['kernel_ms_per_frame'], d['config'].get('bf16_factor_storage_ms_per_step'))"
done
