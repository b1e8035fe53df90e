"""Counted vs speculative (T2N_FLAG_DEVICE_ROWS) fused train step on one box: bench.py's own loop, three blocks each."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

dev = torch.device("cuda:0")
out = {}
for batch in (16384, 2048):
    for resident in (False, True):
        a = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=batch, resident=resident)
        b = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=batch, resident=resident, speculative=True)
        ms = a.get("ms_per_iter") or a.get("train_ms_per_iter_fused_step") or a.get("train_ms_per_iter_fused_step_resident")
        out[f"{batch}_{'resident' if resident else 'host'}"] = {"counted_ms": ms, "speculative": b}
        print(batch, resident, ms, b, flush=True)
print(json.dumps(out))
