"""Per-rank compute of a data-parallel step at 16384 / G rays on one GPU, exchanges left out: flat all-reduce form vs sharded optimiser."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class NoExchange:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def reduce(self, fs):
        pass

    def gather(self, fs):
        pass


dev = torch.device("cuda:0")
out = {}
for name, kw in (("one_call", dict(fused=True, graph=False)), ("flat", dict(fused=True, graph=False, all_reduce=lambda: None)),
                 ("sharded_rank0", dict(fused=True, graph=False, all_reduce=NoExchange(G, 0))),
                 ("sharded_rank_last", dict(fused=True, graph=False, all_reduce=NoExchange(G, G - 1)))):
    r = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=16384 // G, step_kw=kw)
    out[name] = (r["ms_per_iter"], r.get("ms_per_iter_blocks"))
    print(name, out[name], flush=True)
print(json.dumps(out))
