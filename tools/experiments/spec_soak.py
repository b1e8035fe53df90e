"""Speculative train step through the fog phase of bench.py's loop (noisy targets: the appearance rows grow ~6x between step 40 and 100):
how often the capacity (1.25 x the largest need of the last eight recorded steps) is exceeded."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
for resident in (False, True):
    r = bench.train_bench(dev, iters=110, warmup=3, fused_step=True, resident=resident, speculative=True)
    print("resident" if resident else "host data", {k: r[k] for k in ("ms_per_iter", "blocks_ms", "device_rows_steps", "overflows", "unanswered_polls", "loss")}, flush=True)
