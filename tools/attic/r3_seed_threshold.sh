#!/bin/bash
# fused train step with the TV terms seeded on a side stream (T2N_SEED_MIN_RAYS=0) against the TV pass inside the optimiser step (=1000000),
# alternating processes; arguments: batch sizes
for rep in 1 2 3; do
for v in 0 1000000; do
  echo -n "rep $rep T2N_SEED_MIN_RAYS=$v: "
  T2N_SEED_MIN_RAYS=$v python3 - "$@" <<'PY' 2>/dev/null
import sys; sys.path.insert(0, "/root/repo")
import torch, bench
dev = torch.device("cuda:0")
out = {}
for b in [int(x) for x in sys.argv[1:]]:
    x = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=b)
    out[b] = round(x.get("ms_per_iter", x.get("train_ms_per_iter_fused_step", 0)), 3)
print(out)
PY
done
done
