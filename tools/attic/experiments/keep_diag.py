"""Gradients of one tiny train-mode render through the kept-rows two-kernel forward vs the one-kernel form (T2N_EXP_OLDKEEP=1 in a child)."""
import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from tests.conftest import TINY
    from tests.test_hip_parity import make_field
    from text2nerf_amd import synth
    d = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "tiny.npz"))
    tp = synth.make_field_params(11, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"])
    f = make_field(tp, TINY["grid"], TINY["aabb"], TINY["near_far"])
    rays = torch.from_numpy(d["tiny_rays"])
    torch.manual_seed(5)
    rgb, depth, z, w = f(rays, is_train=True, white_bg=True, N_samples=40)
    ((rgb ** 2).sum() + depth.sum()).backward()
    np.savez(sys.argv[1], **{k: p.grad.cpu().numpy() for k, p in f.named_parameters()})
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "/tmp/g_new.npz"], env=env)
env["T2N_EXP_OLDKEEP"] = "1"
subprocess.check_call([sys.executable, __file__, "/tmp/g_old.npz"], env=env)
a, b = np.load("/tmp/g_new.npz"), np.load("/tmp/g_old.npz")
for k in a.files:
    e = np.abs(a[k] - b[k]).max() / (np.abs(b[k]).max() + 1e-12)
    print(k, a[k].shape, "rel err %.2e" % e)
k = "basis_mat.weight"
dlt = np.abs(a[k] - b[k]) / (np.abs(b[k]).max() + 1e-12)
print("per column-block (pair) max:", [float(dlt[:, 48 * q:48 * q + 48].max()) for q in range(3)])
print("per column max (first pair):", np.round(dlt[:, :48].max(0) * 1e4, 1))
print("per row max:", np.round(dlt.max(1) * 1e4, 1))
