"""Diagnose gradient errors of embedded shapes: HIP gradients of the EMBEDDED tensors vs the oracle's autograd on the same tensors."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_shapes_cases import SHAPES
from oracle import oracle_torch as O
from tests.conftest import TINY
from tests.test_shapes import _field, _cfg, KEYS
tiny = dict(np.load(os.path.join(ROOT, "tests", "golden", "tiny.npz")))
gs = dict(np.load(os.path.join(ROOT, "tests", "golden", "shapes.npz")))
dev = torch.device("cuda:0")
for tag in sys.argv[1:] or list(SHAPES):
    kw = SHAPES[tag]
    m = _field(kw, dev, gs[f"{tag}_seed"])
    rays = torch.from_numpy(tiny["tiny_rays"])
    torch.manual_seed(55)
    rgb, depth, z, w = m(rays, is_train=True, white_bg=True, N_samples=36)
    emb = m._autograd_params()
    for t in emb:
        if not t.is_leaf: t.retain_grad()
    ca = torch.from_numpy(gs[f"{tag}_ca"]).to(dev)
    ((rgb * ca).sum() + 0.1 * depth.sum() + (w ** 2).sum()).backward()
    print(tag, "train rgb err", float((rgb.detach().cpu() - torch.from_numpy(gs[f"{tag}_train_rgb"])).abs().max()), "app samples", m.stats()["appearance"])
    # oracle on the embedded tensors
    kd, ka, kdim, kpe, kfc = m._kernel_shape()
    P = {k: t.detach().cpu().clone().requires_grad_(True) for k, t in zip(KEYS, emb)}
    torch.manual_seed(55); jit = torch.rand(rays.shape[0], 1)
    o = O.forward(_cfg(kw, fea_pe=kpe), P, rays, white_bg=True, is_train=True, n_samples=36, jitter=jit)
    ((o[0] * ca.cpu()).sum() + 0.1 * o[1].sum() + (o[3] ** 2).sum()).backward()
    for k, t in zip(KEYS, emb):
        if k not in P: continue
        g = P[k].grad
        if g is None or t.grad is None: continue
        e = float((t.grad.cpu() - g).abs().max()) / (float(g.abs().max()) + 1e-12)
        print(f"   embedded {k:28s} rel err {e:.2e}  max|g| {float(g.abs().max()):.3e}")
    for k, p in m.named_parameters():
        g = gs[f"{tag}_grad_" + k]
        e = float(np.abs(p.grad.cpu().numpy() - g).max()) / (float(np.abs(g).max()) + 1e-12)
        print(f"   real     {k:28s} rel err {e:.2e}  max|g| {float(np.abs(g).max()):.3e}")
