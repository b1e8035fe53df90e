"""Where do the ~90 ms stalls of the data-parallel train step on a single-rank RCCL group come from? Variants: collectives stubbed
out (group still initialised), no group at all."""
import datetime, os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("T2N_TRAIN_TRACE", "1")
import bench  # noqa: E402
torch.set_num_threads(16)  # as bench.main does (train_bench halves it for its loop): with one OpenMP thread per host core the process outruns its cgroup CPU quota and is
                            # throttled until the next 100-ms period (stalls of ~90 ms every few iterations)
import text2nerf_amd.parallel as par  # noqa: E402
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
dev = torch.device("cuda", 0); torch.cuda.set_device(0); torch.zeros(1, device=dev).add_(1); torch.cuda.synchronize()
mode = sys.argv[1]
if mode == "nogroup":
    print(mode, bench.train_bench(dev, iters=30, warmup=2, fused_step=True).get("train_ms_per_iter_fused_step"))
    sys.exit(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=120))
if mode == "stub_allreduce":
    def stub(params, group=None, average=True, field=None):
        if field is not None and getattr(field, "defer_factor_grads", False):
            field._gbuf_dirty = True; field._gbuf_reduced = True
    par.allreduce_gradients = stub
if mode == "stub_broadcast":
    par.broadcast_parameters = lambda *a, **k: None
if mode == "group_but_local":   # group exists, train step does not know about it
    print(mode, bench.train_bench(dev, iters=30, warmup=2, fused_step=True).get("train_ms_per_iter_fused_step"))
else:
    print(mode, bench.train_bench(dev, iters=30, warmup=2, fused_step=True, dist=dist).get("train_dp_ms_per_iter"))
dist.destroy_process_group()
