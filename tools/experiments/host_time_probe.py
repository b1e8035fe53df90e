"""Host-side cost of one fused train step: wall time of the enqueue loop (no drain) against the drained time, C3 shape.
python tools/experiments/host_time_probe.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, cProfile, pstats
import bench
from text2nerf_amd import synth
from text2nerf_amd.optim import TVAdam
dev = torch.device("cuda:0")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
torch.set_num_threads(2)
field, params, aabb = bench.build_field(dev)
N = 259
poses = bench.reference_poses("local_fixed")
allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))
g = np.random.Generator(np.random.PCG64(1024))
with torch.no_grad():
    sub = allrays[::4].to(dev)
    rgb_s, dep_s, _, _ = field(sub, white_bg=True, is_train=False, N_samples=N)
allrgb = (rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0], 3)).astype(np.float32))).clamp(0, 1)
alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0],)).astype(np.float32))
perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))
batches = []
for k in range(40):
    idx = perm[k * batch:(k + 1) * batch]
    batches.append((allrays[idx].contiguous(), allrgb[idx].contiguous(), alldepth[idx].contiguous()))
tv = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
for mode in ("eager", "graph"):
    field.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    opt = TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
    kw = dict(fused=True, graph=mode == "graph")
    for k in range(8):
        field.train_step(*batches[k], opt, N_samples=N, white_bg=True, tv=tv, **kw)
    torch.cuda.synchronize()
    fs = field._fused_step
    fs.events.clear()       # (no run-ahead limit: the loop below measures the host alone)
    import text2nerf_amd.trainer as T
    old = T._RUN_AHEAD
    T._RUN_AHEAD = 10 ** 6
    t0 = time.perf_counter()
    for k in range(8, 28):
        field.train_step(*batches[k], opt, N_samples=N, white_bg=True, tv=tv, **kw)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    T._RUN_AHEAD = old
    print(f"{mode}: host enqueue {1e3 * (t1 - t0) / 20:.3f} ms/step, drained {1e3 * (t2 - t0) / 20:.3f} ms/step", flush=True)
    if True:
        pr = cProfile.Profile()
        pr.enable()
        for k in range(28, 38):
            field.train_step(*batches[k], opt, N_samples=N, white_bg=True, tv=tv, **kw)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
