"""Host threads of the fused train loop (CPU row gathers of 16 384 rays): bench.py's loop at 1 / 2 / 4 / 8 torch threads, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device("cuda:0")
print("host cores", bench.HOST_CORES, "torch threads", torch.get_num_threads())
for rep in range(2):
    for t in (8, 4, 2, 1):
        bench.TRAIN_THREADS = t
        r = bench.train_bench(dev, iters=20, warmup=3, fused_step=True)
        print("threads", t, "fused step (host data) blocks", r["train_ms_per_iter_fused_step_blocks"], flush=True)
