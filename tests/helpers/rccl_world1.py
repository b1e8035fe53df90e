"""Child process of tests/test_rccl_single.py: the N > 1 code paths on a world-size-1 "nccl" (= RCCL on ROCm) process group on the one
GPU of the box. The collectives degenerate to copies, but they run through RCCL's communicator, streams, work handles and
device-pointer checks — the same calls bench.py --gpus N and parallel.py make. Prints one line per check; the
last line is `RCCL_WORLD1_OK <version>`."""
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from tests.conftest import TINY  # noqa: E402
from text2nerf_amd import TensorVMSplit, synth  # noqa: E402
from text2nerf_amd.optim import TVAdam  # noqa: E402
from text2nerf_amd.parallel import all_gather_tiles, allreduce_gradients, broadcast_parameters, render_sharded  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
torch.zeros(1, device=dev).add_(1)
torch.cuda.synchronize()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=90))
assert dist.get_backend() == "nccl"
version = ".".join(str(x) for x in torch.cuda.nccl.version())
print("process group up, rccl", version, flush=True)


def make():
    params = synth.make_field_params(7, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"])
    m = TensorVMSplit(torch.tensor(TINY["aabb"], dtype=torch.float32), list(TINY["grid"]), dev, density_n_comp=[16] * 3,
                      appearance_n_comp=[48] * 3, app_dim=27, near_far=TINY["near_far"], shadingMode="MLP_Fea_noview",
                      alphaMask_thres=1e-4, density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128,
                      step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    return m


field = make()
field.materialize_weights = False
W = 64
rays = torch.from_numpy(synth.frame_rays_np(W, W, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)))).to(dev)
render = lambda r: field(r, white_bg=True, is_train=False, N_samples=-1)[:2]  # noqa: E731
with torch.no_grad():
    rgb, depth = render(rays)
    # 1. render_sharded: contiguous tiles and interleaved 8-row bands, ONE all_gather_into_tensor each; bitwise == the direct render
    for fw in (0, W):
        s_rgb, s_depth = render_sharded(rays, render, frame_width=fw)
        assert torch.equal(s_rgb, rgb) and torch.equal(s_depth, depth), fw
    print("render_sharded == direct render (contiguous, banded)", flush=True)
    # 2. bench.py's weak-mode step: asynchronous all-gather of frame k's tile while frame k + 1 renders, then wait
    tile = torch.cat([rgb, depth[:, None]], 1)
    out = torch.empty_like(tile)
    work = dist.all_gather_into_tensor(out, tile, async_op=True)
    rgb2, _ = render(rays)
    work.wait()
    torch.cuda.synchronize()
    assert torch.equal(out, tile) and torch.equal(rgb2, rgb) and torch.equal(all_gather_tiles(tile), tile)
    print("async all_gather_into_tensor + wait ok", flush=True)

# 3. data-parallel fused train step: in-place all-reduce of the field's gradient buffer in two buckets, the density one on the side
#    stream behind t2n_field_wait_density_grads (overlap=True), against the single flat message (overlap=False): bitwise
g = np.random.Generator(np.random.PCG64(3))
b_rays = torch.from_numpy(synth.frame_rays_np(16, 24, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))
rgb_t = torch.from_numpy(g.uniform(0, 1, (b_rays.shape[0], 3)).astype(np.float32))
dep_t = torch.from_numpy(g.uniform(2, 7, (b_rays.shape[0],)).astype(np.float32))
fields = []
for overlap in (True, False):
    f = make()
    broadcast_parameters(f.parameters())
    opt = TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f)
    ps = [p for p in f.parameters() if p.requires_grad]
    for it in range(3):
        torch.manual_seed(100 + it)
        f.train_step(b_rays, rgb_t, dep_t, opt, N_samples=-1, white_bg=True, tv=[(f.density_plane, 0.1), (f.app_plane, 0.01)],
                     all_reduce=lambda: allreduce_gradients(ps, average=True, field=f, overlap=overlap))
    torch.cuda.synchronize()
    fields.append(f)
# (two runs of the step differ at rounding level by themselves: the backward's scatters flush with floating-point atomics)
for (ka, a), (kb, b) in zip(fields[0].state_dict().items(), fields[1].state_dict().items()):
    assert ka == kb and torch.allclose(a, b, rtol=1e-3, atol=2e-5), (ka, float((a - b).abs().max()))
print("3 fused data-parallel steps: bucketed (side stream) and flat all-reduce trajectories agree", flush=True)
# on ONE rank a sum over the group is the identity: both forms must hand the buffer back bit for bit (average=False)
f = fields[0]
ps = [p for p in f.parameters() if p.requires_grad]
buf = f.factor_grad_buffer()
for overlap in (True, False):
    buf.copy_(torch.randn_like(buf))
    snap = buf.clone()
    hp = f._all_params()[12:]                      # the head tensors (one small flat message)
    for p in hp:
        p.grad = torch.randn_like(p)
    head = [p.grad.clone() for p in hp]
    allreduce_gradients(ps, average=False, field=f, overlap=overlap)
    torch.cuda.synchronize()
    assert torch.equal(f.factor_grad_buffer(), snap), overlap
    assert all(torch.equal(p.grad, h) for p, h in zip(hp, head)), overlap
print("allreduce_gradients(overlap=True / False) on one rank: identity, bitwise", flush=True)
# 4. the sharded optimiser's exchange (parallel.ShardedExchange) through RCCL: on one rank a plane has no body (everything is
#    "replicated"), so the step must equal the plain fused step; the three collectives are then driven directly on device tensors
#    IN PLACE (output aliasing the input's own chunk: what reduce() / gather() do on larger groups)
from text2nerf_amd.parallel import ShardedExchange  # noqa: E402
fa, fb = make(), make()
oa = TVAdam(fa.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fa)
ob = TVAdam(fb.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=fb)
ex = None
for it in range(3):
    torch.manual_seed(100 + it)
    fa.train_step(b_rays, rgb_t, dep_t, oa, N_samples=-1, white_bg=True, tv=[(fa.density_plane, 0.1), (fa.app_plane, 0.01)], fused=True, graph=False)
    torch.manual_seed(100 + it)
    if ex is None:
        fb.train_step(b_rays, rgb_t, dep_t, ob, N_samples=-1, white_bg=True, tv=[(fb.density_plane, 0.1), (fb.app_plane, 0.01)], fused=True,
                      graph=False, all_reduce=lambda: allreduce_gradients([p for p in fb.parameters() if p.requires_grad], average=True, field=fb))
        ex = ShardedExchange.for_field(fb)
        assert ex.world == 1 and all(l[1] == 0 for l in ex.layout)
        # the channel-last parameter views wrap the library's own allocations: same values as the reference-layout tensors
        pl = fb.density_plane[0].detach()
        assert torch.equal(ex.params[0].view(pl.shape[2], pl.shape[3], pl.shape[1]).permute(2, 0, 1), pl[0])
    else:
        fb.train_step(b_rays, rgb_t, dep_t, ob, N_samples=-1, white_bg=True, tv=[(fb.density_plane, 0.1), (fb.app_plane, 0.01)], fused=True,
                      graph=False, all_reduce=ex)
for o in (fa, fb):
    o.__dict__["_fused_step"].sync()
for (ka, a), (kb, b) in zip(fa.state_dict().items(), fb.state_dict().items()):
    assert ka == kb and torch.allclose(a, b, rtol=1e-3, atol=2e-5), (ka, float((a - b).abs().max()))
x = torch.randn(4096, device=dev)
snap = x.clone()
ex._reduce_scatter_avg(x, x)          # reduce_scatter_tensor(AVG), output == the input's chunk of this rank
ex._all_reduce_avg(x)
ex._all_gather(x, x)                  # all_gather_into_tensor, input == the output's chunk of this rank
torch.cuda.synchronize()
assert torch.equal(x, snap)
print("ShardedExchange: fused step through reduce() / gather() == plain fused step; in-place reduce_scatter / all_gather ok", flush=True)
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_WORLD1_OK", version, flush=True)
