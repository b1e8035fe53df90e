#!/bin/bash
# A/B of the feature kernels: pair-phased (default) against the 144-row kernel (T2N_APPFEAT_WHOLE=1), fp32 and bf16 storage
run() { python bench.py --no-train --steps 30 --quick --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('ms/step', round(d['ms_per_step'], 3), {k: round(x, 3) for k, x in d['config']['kernel_ms_per_frame'].items()})"; }
echo "pair-phased fp32"; run
echo "whole fp32"; T2N_APPFEAT_WHOLE=1 run
echo "pair-phased bf16"; run --factor-storage bf16
echo "whole bf16"; T2N_APPFEAT_WHOLE=1 run --factor-storage bf16
echo "pair-phased fp32"; run
