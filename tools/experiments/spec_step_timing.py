"""Counted vs speculative (T2N_FLAG_DEVICE_ROWS) fused train step on one box, with and without the TV seeding of small batches:
bench.py's own loop, three blocks each.   argv[1:]: batches (default 16384 2048)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from text2nerf_amd import tensorf as tf

dev = torch.device("cuda:0")
out = {}
batches = [int(a) for a in sys.argv[1:]] or [16384, 2048]
for batch in batches:
    for seed_min in (8192, 0):
        if batch >= 8192 and seed_min == 0:
            continue
        tf._SEED_MIN_RAYS = seed_min
        for resident in (False, True):
            a = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=batch, resident=resident)
            b = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=batch, resident=resident, speculative=True)
            ms = a.get("ms_per_iter") or a.get("train_ms_per_iter_fused_step") or a.get("train_ms_per_iter_fused_step_resident")
            out[f"{batch}_{'resident' if resident else 'host'}_seedmin{seed_min}"] = {"counted_ms": ms, "speculative_ms": b["ms_per_iter"], "speculative": b}
            print(batch, seed_min, resident, ms, b["ms_per_iter"], flush=True)
print(json.dumps(out))
