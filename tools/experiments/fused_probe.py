"""Round 6 probe of the fused training step (t2n_train_step): tiny-field trajectory against the composed step, then C3-shaped timing
(eager / graph). python tools/experiments/fused_probe.py [iters] [batch]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

def tiny_check():
    from tests.conftest import TINY
    from tests.test_hip_parity import make_field
    from tests.test_train_step import batch, assert_same_trajectory
    from text2nerf_amd.optim import TVAdam
    from text2nerf_amd import synth
    params = synth.make_field_params(11, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"])
    rays, rgb_t, dep_t = batch()
    res = {}
    for mode in ("eager", "graph"):
        fa = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
        fb = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
        oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
        ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
        steps = 7
        for it in range(steps):
            tva = [(fa.density_plane, 0.1 * 0.9 ** it), (fa.app_plane, 0.01)]
            tvb = [(fb.density_plane, 0.1 * 0.9 ** it), (fb.app_plane, 0.01)]
            for g in oa.param_groups: g["lr"] *= 0.97
            for g in ob.param_groups: g["lr"] *= 0.97
            torch.manual_seed(100 + it)
            la = fa.train_step(rays, rgb_t, dep_t, oa, N_samples=-1, white_bg=True, tv=tva, fused=False)
            torch.manual_seed(100 + it)
            lb = fb.train_step(rays, rgb_t, dep_t, ob, N_samples=-1, white_bg=True, tv=tvb, fused=True, graph=mode == "graph").clone()
            torch.cuda.synchronize()
            print(mode, it, la.tolist(), lb.tolist(), flush=True)
            assert torch.allclose(la, lb, rtol=1e-5, atol=1e-9), (it, la, lb)
        fb._fused_step.sync()
        assert_same_trajectory(fa, fb, steps=steps)
        fs = fb._fused_step
        res[mode] = dict(eager=fs.eager_launches, graph=fs.graph_launches, captures=fs.graph_captures, nodes=getattr(fs, "graph_nodes", None),
                         replays=fs.replays, overflows=getattr(fb, "device_rows_overflows", 0), cap=fs.rows_cap, needs=fs.needs)
    print("tiny ok", res, flush=True)

def c3_timing(iters, batch_n):
    import bench
    dev = torch.device("cuda:0")
    out = {}
    for name, kw in (("legacy", dict(fused=False)), ("fused_eager", dict(fused=True, graph=False)), ("fused_graph", dict(fused=True, graph=True))):
        r = bench.train_bench(dev, iters=iters, warmup=8, fused_step=True, batch=batch_n, step_kw=kw)
        out[name] = {k: r[k] for k in r if "ms" in k or "blocks" in k or "loss" in k}
        print(name, out[name], flush=True)
    return out

if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    batch_n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
    if os.environ.get("T2N_PROBE_TINY", "1") == "1":
        tiny_check()
    if os.environ.get("T2N_PROBE_C3", "1") == "1":
        c3_timing(iters, batch_n)
