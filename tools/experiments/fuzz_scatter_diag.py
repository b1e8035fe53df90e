"""A failed gradient-fuzz seed: the HIP gradients with the binned scatters against the sliding-window global-atomic scatter
(T2N_BWD_ATOMIC_SCATTER=1, an independent implementation of the same sums), against the oracle's autograd in float32 and against the
oracle in FLOAT64 (what tells fp32 summation noise in a nearly empty batch from a wrong sum), per tensor."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
seed = int(sys.argv[1])
if len(sys.argv) > 2:
    import torch
    from tests import test_hip_fuzz as F
    from tests.test_hip_fuzz import _field, synth, O, dev
    g = np.random.Generator(np.random.PCG64(5000 + seed))
    grid = [int(g.integers(9, 40)) for _ in range(3)]
    lo = (-g.uniform(2.0, 9.0, 3)).astype(np.float32); hi = g.uniform(2.0, 9.0, 3).astype(np.float32)
    aabb = [lo.tolist(), hi.tolist()]
    near_far = [float(g.uniform(0.05, 1.0)), float(g.uniform(4.0, 12.0))]
    params = synth.make_field_params(6000 + seed, grid, density_scale=float(g.uniform(0.5, 1.4)), aabb=aabb)
    f = _field(params, grid, aabb, near_far, 1.0)
    cfg = O.FieldConfig(aabb=aabb, grid_size=grid, near_far=near_far)
    centre = tuple(float(v) for v in (lo + (hi - lo) * g.uniform(0.3, 0.7, 3)))
    rays = torch.from_numpy(synth.frame_rays_np(12, 14, c2w=synth.look_pose(float(g.uniform(-3, 3)), float(g.uniform(-1, 1)), centre)))
    is_train = bool(seed % 2 == 0)
    n = int(g.integers(20, 80))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    torch.manual_seed(seed)
    jit = torch.rand(rays.shape[0], 1) if is_train else None
    torch.manual_seed(seed)
    out = f(rays, is_train=is_train, white_bg=True, N_samples=n)
    ((out[0] * ca.to(dev())).sum() + 0.1 * out[1].sum() + (out[3] ** 2).sum()).backward()
    res = {k: p.grad.cpu().numpy() for k, p in f.named_parameters()}
    if sys.argv[2] in ("oracle", "oracle64"):
        dt = torch.float64 if sys.argv[2] == "oracle64" else torch.float32
        P = O.params_from_numpy(params, requires_grad=True, dtype=dt)
        o = O.forward(cfg, P, rays.to(dt), white_bg=True, is_train=is_train, n_samples=n, jitter=None if jit is None else jit.to(dt))
        ((o[0] * ca.to(dt)).sum() + 0.1 * o[1].sum() + (o[3] ** 2).sum()).backward()
        res = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}
        print("grid", grid, "train", is_train, "n", n)
    np.savez(sys.argv[3], **res)
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, str(seed), "hip", "/tmp/fz_bin.npz"], env=env)
subprocess.check_call([sys.executable, __file__, str(seed), "oracle", "/tmp/fz_or.npz"], env=env)
subprocess.check_call([sys.executable, __file__, str(seed), "oracle64", "/tmp/fz_or64.npz"], env=env)
env["T2N_BWD_ATOMIC_SCATTER"] = "1"
subprocess.check_call([sys.executable, __file__, str(seed), "hip", "/tmp/fz_at.npz"], env=env)
a, b, o, o64 = np.load("/tmp/fz_bin.npz"), np.load("/tmp/fz_at.npz"), np.load("/tmp/fz_or.npz"), np.load("/tmp/fz_or64.npz")
print("seed", seed, ": errors as max |difference| / max |float64 gradient| per tensor")
for k in a.files:
    s = np.abs(o64[k]).max() + 1e-300
    print("%-28s binned-vs-atomic %.1e   binned-vs-f32-oracle %.1e   binned-vs-f64 %.1e   atomic-vs-f64 %.1e   f32-oracle-vs-f64 %.1e   max|g| %.2e" % (
        k, np.abs(a[k] - b[k]).max() / s, np.abs(a[k] - o[k]).max() / s, np.abs(a[k] - o64[k]).max() / s, np.abs(b[k] - o64[k]).max() / s,
        np.abs(o[k] - o64[k]).max() / s, np.abs(o64[k]).max()))
