"""Run only the C3-shaped train loop of bench.py (for rocprofv3 --kernel-trace --stats). argv[1]: 0 = torch TV+Adam, 1 = fused TV+Adam,
2 = autograd-free train_step, 3 = the same with the training set resident in HBM. argv[3]: rays per batch (16384); argv[4] = "spec": speculative step."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
mode = sys.argv[1] if len(sys.argv) > 1 else "0"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
spec = len(sys.argv) > 4 and sys.argv[4] == "spec"
# round 6: argv[4] = legacy | fused_eager | fused_graph selects the train_step form (default: the product default)
kw = {"legacy": dict(fused=False), "fused_eager": dict(fused=True, graph=False), "fused_graph": dict(fused=True, graph=True)}.get(sys.argv[4] if len(sys.argv) > 4 else "", None)
print(json.dumps(bench.train_bench(torch.device("cuda:0"), iters=iters, warmup=3, fused_optim=mode == "1", fused_step=mode in ("2", "3"), resident=mode == "3",
                                   batch=batch, speculative=spec, step_kw=kw)))
