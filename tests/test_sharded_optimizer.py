"""Sharded optimiser of the data-parallel train step (text2nerf_amd/parallel.py ShardedExchange): gloo world-2 / world-4
groups on CPU against the single-rank trajectory. The kernels are stood in for by torch arithmetic on the same channel-last layout (there is
no GPU here); what is under test is the partition, the seeding rule, the two exchanges and that a rank never touches state it does not own.
The HIP kernels' side of the same rule is tests/test_train_step.py::test_sharded_optimizer_virtual_ranks (GPU)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["T2N_ROOT"])
import torch, torch.distributed as dist
from text2nerf_amd.parallel import ShardedExchange, shard_layout
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(5)
grid = [24, 20, 16]
MAT, VEC = ((0, 1), (0, 2), (1, 2)), (2, 1, 0)
CD, CA = 16, 48
lay = shard_layout(grid, world, CD, CA)
assert any(l[1] > 0 for l in lay) and any(l[2] > world * l[1] > 0 for l in lay), "the case must have bodies AND replicated plane blocks"
shapes = []
for q in range(4):
    for k in range(3):
        C = CD if q < 2 else CA
        shapes.append((grid[MAT[k][1]], grid[MAT[k][0]], C) if q % 2 == 0 else (grid[VEC[k]], 1, C))
TVW = [0.1 if t < 3 else (0.01 if 6 <= t < 9 else 0.0) for t in range(12)]
LR, B1, B2, EPS = 0.02, 0.9, 0.99, 1e-15

def tv_grad(p, shape, w):
    if w == 0.0:
        return torch.zeros_like(p)
    H, W, C = shape
    x = p.detach().view(H, W, C).clone().requires_grad_(True)
    loss = w * 2.0 * (((x[1:] - x[:-1]) ** 2).sum() / (C * (H - 1) * W) + ((x[:, 1:] - x[:, :-1]) ** 2).sum() / (C * H * (W - 1)))
    loss.backward()
    return x.grad.reshape(-1)

def local_grad(p, r, step):      # stands in for the rank's backward over ITS ray shard
    return torch.sin(p * 3.0 + (r + 1) * (step + 1)) * 0.01

def adam(p, g, m, v, step, sel):
    m[sel] = m[sel] + (g[sel] - m[sel]) * (1 - B1)
    v[sel] = v[sel] * B2 + (1 - B2) * g[sel] * g[sel]
    bc1, bc2 = 1 - B1 ** step, 1 - B2 ** step
    p[sel] = p[sel] - (LR / bc1) * (m[sel] / (v[sel].sqrt() / bc2 ** 0.5 + EPS))

def make():
    g = torch.Generator().manual_seed(9)
    ps = [0.2 * torch.randn(s[0] * s[1] * s[2], generator=g) for s in shapes]
    head = 0.1 * torch.randn(257, generator=g)
    return ps, head

# ---- single-rank trajectory: full-batch gradient = mean of the shards' gradients, TV on top, Adam on everything
ref_p, ref_h = make()
ref_m = [torch.zeros_like(p) for p in ref_p]; ref_v = [torch.zeros_like(p) for p in ref_p]
ref_hm, ref_hv = torch.zeros_like(ref_h), torch.zeros_like(ref_h)
STEPS = 3
for step in range(1, STEPS + 1):
    for t in range(12):
        g = sum(local_grad(ref_p[t], r, step) for r in range(world)) / world + tv_grad(ref_p[t], shapes[t], TVW[t])
        adam(ref_p[t], g, ref_m[t], ref_v[t], step, slice(None))
    hg = sum(local_grad(ref_h, r, step) for r in range(world)) / world
    adam(ref_h, hg, ref_hm, ref_hv, step, slice(None))

# ---- sharded: this rank's buffers
ps, head = make()
ms = [torch.zeros_like(p) for p in ps]; vs = [torch.zeros_like(p) for p in ps]
hm, hv = torch.zeros_like(head), torch.zeros_like(head)
nfl = lay[-1][0] + (lay[-1][2] * 4 + 255) // 256 * 64
gbuf = torch.zeros(nfl)
head_g = torch.zeros(258)           # head gradients + the vote word
ex = ShardedExchange(lay, gbuf, ps, extra=lambda: [head_g])
assert ex.world == world and ex.rank == rank
for step in range(1, STEPS + 1):
    gbuf.fill_(float("nan"))        # (whatever a rank does not own may hold anything after the exchange)
    own_sel = []
    for t in range(12):
        off, sl, total = lay[t]
        tv = tv_grad(ps[t], shapes[t], TVW[t])
        seed = tv.clone()                                         # replicated blocks: the plain TV gradient on every rank
        seed[: world * sl] = 0.0                                  # body: zero ...
        seed[rank * sl:(rank + 1) * sl] = world * tv[rank * sl:(rank + 1) * sl]   # ... but world x TV on the owned slice
        gbuf[off:off + total] = seed + local_grad(ps[t], rank, step)
        sel = torch.zeros(total, dtype=torch.bool)
        sel[rank * sl:(rank + 1) * sl] = True
        sel[world * sl:] = True
        own_sel.append(sel)
    head_g[:257] = local_grad(head, rank, step)
    head_g[257] = 1.0 if (rank == 1 and step == 2) else 0.0       # one rank votes to withhold step 2
    ex.reduce()
    vote = float(head_g[257])
    assert (vote != 0.0) == (step == 2), (step, vote)            # every rank sees the vote
    for t in range(12):
        off, sl, total = lay[t]
        g = gbuf[off:off + total]
        assert not torch.isnan(g[own_sel[t]]).any()
        adam(ps[t], torch.nan_to_num(g), ms[t], vs[t], step, own_sel[t])
    adam(head, head_g[:257], hm, hv, step, slice(None))
    ex.gather()
for t in range(12):
    off, sl, total = lay[t]
    tol = 2e-6 * float(ref_p[t].abs().max())
    assert float((ps[t] - ref_p[t]).abs().max()) <= tol, (t, float((ps[t] - ref_p[t]).abs().max()), tol)
    other = ~own_sel[t]
    assert float(ms[t][other].abs().max() if other.any() else 0.0) == 0.0 and float(vs[t][other].abs().max() if other.any() else 0.0) == 0.0, \
        "moments of blocks owned by other ranks must never be touched"
    assert float((ms[t][own_sel[t]] - ref_m[t][own_sel[t]]).abs().max()) <= 1e-5 * float(ref_m[t].abs().max()) + 1e-12
assert float((head - ref_h).abs().max()) <= 2e-6 * float(ref_h.abs().max())
# every rank ends with the same parameters, bitwise (the owner's values travel; the replicated parts saw identical inputs)
chk = torch.cat([p.double().sum().reshape(1) for p in ps])
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert torch.equal(lo, hi)
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_optimizer_matches_single_rank_gloo(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   T2N_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_shard_layout_partition():
    sys.path.insert(0, ROOT)
    from text2nerf_amd.parallel import shard_layout
    for grid in ([300, 300, 300], [24, 20, 16], [65, 64, 63]):
        for world in (1, 2, 3, 4, 8):
            lay = shard_layout(grid, world)
            end = 0
            for t, (off, sl, total) in enumerate(lay):
                plane = (t // 3) % 2 == 0
                C = 16 if t < 6 else 48
                assert off * 4 % 256 == 0 and off >= end and total % C == 0
                assert sl % (64 * C) == 0 and world * sl <= total
                if not plane or world == 1:
                    assert sl == 0
                else:
                    assert total - world * sl < (world + 1) * 64 * C      # what stays replicated: less than a block per rank + the partial one
                end = off + total
