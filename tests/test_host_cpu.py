"""CPU-only checks: the C-ABI library loads and exports every symbol include/t2n.h declares (no compute calls),
host-side logic of the Python mirror, synthetic-scene recipes, and the N>1 ray-tile sharding over gloo (world_size 2)."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from text2nerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "t2n.h")).read()
    declared = set(re.findall(r"\b(t2n_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"t2n_field_desc", "t2n_field_params", "t2n_field_grads"}
    assert declared, "header parse failed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/t2n.h but not exported"
    assert declared == set(_lib.SIGNATURES), "ctypes SIGNATURES table is out of sync with include/t2n.h"
    assert lib.t2n_version() >= 100
    # argument validation happens before any HIP call, so it is testable without a GPU
    assert lib.t2n_field_create(None, None) == -1
    assert b"NULL" in lib.t2n_last_error()
    d = _lib.FieldDesc()
    d.density_n_comp, d.app_n_comp = 8, 24
    import ctypes as C
    h = C.c_void_p()
    assert lib.t2n_field_create(C.byref(d), C.byref(h)) == -2      # unsupported, loudly
    assert b"unsupported" in lib.t2n_last_error()
    assert lib.t2n_render_workspace_bytes(0, 10) == 0
    assert lib.t2n_render_workspace_bytes(16384, 259) > 16384 * 259 * 36


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the host."""
    from text2nerf_amd import TensorVMSplit
    from text2nerf_amd._lib import T2NError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    m = TensorVMSplit(torch.tensor([[-1.0] * 3, [1.0] * 3]), [8, 8, 8], "cpu", density_n_comp=[16] * 3,
                      appearance_n_comp=[48] * 3, shadingMode="MLP_Fea_noview", fea_pe=6, step_ratio=1.0)
    with pytest.raises(T2NError):
        m(torch.zeros(4, 6))
    # the image-space entry points (f-2 / f-3) and the other field variants refuse to run on the host as well
    import numpy as np
    from text2nerf_amd import TensorCP, TensorVM, postprocess_frame
    from text2nerf_amd.warp import bilinear_splat_warping_multiview, sparse_bilateral_filtering
    with pytest.raises(T2NError):
        postprocess_frame(torch.zeros(4, 4, 3), torch.zeros(4, 4), [0.5, 8.0])
    with pytest.raises(T2NError):
        sparse_bilateral_filtering(np.ones((8, 8), np.float32), np.zeros((8, 8, 3), np.float32))
    with pytest.raises(T2NError):
        bilinear_splat_warping_multiview([np.zeros((8, 8, 3), np.float32)], [np.ones((8, 8), np.float32)], np.eye(4)[None], np.eye(4),
                                         8, 8, [8.0, 8.0, 4, 4])
    for cls, kw in ((TensorVM, dict(density_n_comp=16, appearance_n_comp=48)),
                    (TensorCP, dict(density_n_comp=[16] * 3, appearance_n_comp=[48] * 3))):
        v = cls(torch.tensor([[-1.0] * 3, [1.0] * 3]), [8, 8, 8], "cpu", shadingMode="MLP_Fea_noview", fea_pe=6, step_ratio=1.0, **kw)
        with pytest.raises(T2NError):
            v(torch.zeros(4, 6))
    src = ""
    for f in sorted(os.listdir(os.path.join(ROOT, "text2nerf_amd"))):
        if f.endswith(".py"):
            src += open(os.path.join(ROOT, "text2nerf_amd", f)).read()
    assert "oracle" not in src, "the shipped package must never import the oracle"


def test_module_surface_matches_reference(tiny):
    from text2nerf_amd import TensorVMSplit
    m = TensorVMSplit(torch.tensor([[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]]), [24, 20, 16], "cpu",
                      density_n_comp=[16] * 3, appearance_n_comp=[48] * 3, app_dim=27, near_far=[0.5, 8.0],
                      shadingMode="MLP_Fea_noview", density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6,
                      featureC=128, step_ratio=1.0)
    ref = dict(zip(map(str, tiny["g10_keys"]), map(str, tiny["g10_shapes"])))
    assert {k: str(tuple(v.shape)) for k, v in m.state_dict().items()} == ref
    groups = m.get_optparam_groups(0.02, 1e-3)
    assert [g["lr"] for g in groups] == tiny["g10_group_lr"].tolist()
    assert [sum(p.numel() for p in g["params"]) for g in groups] == tiny["g10_group_sizes"].tolist()
    assert sorted(m.get_kwargs()) == list(tiny["g10_kwargs_keys"])
    assert (float(m.stepSize), m.nSamples) == (tiny["tiny_step"][0], int(tiny["tiny_step"][1]))
    m300 = TensorVMSplit(torch.tensor([[-8.0] * 3, [8.0] * 3]), [8, 8, 8], "cpu", density_n_comp=[16] * 3,
                         appearance_n_comp=[48] * 3, shadingMode="SH", step_ratio=1.0)
    m300.update_stepSize([300] * 3)
    assert (float(m300.stepSize), m300.nSamples) == (tiny["g9_step_300"][0], 518)
    # TV regulariser surface works on the reference-layout parameters with a plain callable
    from oracle import oracle_torch as O
    assert float(m.TV_loss_density(O.tv_loss)) > 0 and float(m.TV_loss_app(O.tv_loss)) > 0
    # checkpoint round trip keeps keys and kwargs
    import io
    buf = io.BytesIO()
    m.save(buf)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    m2 = TensorVMSplit(**{**ck["kwargs"], "device": "cpu"})
    m2.load(ck)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))


def test_simple_sampler_and_shards():
    from text2nerf_amd import SimpleSampler
    from text2nerf_amd.parallel import shard_bounds, tile_capacity
    np.random.seed(1024)
    s = SimpleSampler(100, 32)
    ids = [s.nextids() for _ in range(4)]
    assert all(len(i) == 32 for i in ids[:3]) and len(torch.cat(ids[:3]).unique()) == 96
    for R, W in ((640000, 8), (10, 4), (7, 8), (0, 2)):
        b = [shard_bounds(R, W, r) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == R and all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        assert max(h - l for l, h in b) == tile_capacity(R, W) if R else True


def test_batch_prefetcher_returns_the_rows_in_submission_order():
    """The training loop's gathers one batch ahead on a worker thread: same rows as `tensor[ids]` (text2nerf_main.py:550-553), FIFO,
    errors of the worker surface at get()."""
    from text2nerf_amd import BatchPrefetcher, SimpleSampler
    g = torch.Generator().manual_seed(3)
    rays, rgb, dep = torch.randn(1000, 6, generator=g), torch.rand(1000, 3, generator=g), torch.rand(1000, generator=g)
    pf = BatchPrefetcher([rays, rgb, dep])
    np.random.seed(7)
    s = SimpleSampler(1000, 64)
    ids = [s.nextids() for _ in range(5)]
    pf.submit(ids[0])
    for k in range(5):
        if k + 1 < 5:
            pf.submit(ids[k + 1])
        b = pf.get()
        assert torch.equal(b[0], rays[ids[k]]) and torch.equal(b[1], rgb[ids[k]]) and torch.equal(b[2], dep[ids[k]])
    pf.submit(torch.tensor([1000]))
    with pytest.raises(IndexError):
        pf.get()
    pf.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["T2N_ROOT"])
import numpy as np, torch, torch.distributed as dist
from text2nerf_amd.parallel import render_sharded
from text2nerf_amd import synth
from oracle import oracle_torch as O          # stand-in renderer on CPU: the test checks the sharding, not the kernels
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
grid, aabb = [24, 20, 16], [[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]]
cfg = O.FieldConfig(aabb=aabb, grid_size=grid)
P = O.params_from_numpy(synth.make_field_params(11, grid, density_scale=0.9, aabb=aabb))
rays = torch.from_numpy(synth.frame_rays_np(9, 11, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))   # 99 rays: ragged tiles
def fn(r):
    rgb, depth, _, _ = O.forward(cfg, P, r)
    return rgb, depth
rgb, depth = render_sharded(rays, fn)
full_rgb, full_depth = fn(rays)
assert torch.equal(rgb, full_rgb) and torch.equal(depth, full_depth), "gathered tiles must equal the unsharded render bitwise"
# a row-major frame whose rows divide into world x 8-row bands: interleaved bands, same result bitwise
world = int(os.environ["WORLD_SIZE"])
rows = 16 * world
frame = torch.from_numpy(synth.frame_rays_np(rows, 11, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))
seen = []
def fn2(r):     # a per-ray function (the CPU oracle's matmuls are not batch-invariant to the last bit; the HIP kernels are)
    seen.append(r.shape[0])
    return torch.sin(r[:, 3:6] * 3.0 + r[:, :3]), (r[:, 3] * 2.0 + r[:, 5]).contiguous()
rgb, depth = render_sharded(frame, fn2, frame_width=11)
full_rgb, full_depth = fn2(frame)
assert seen[0] == rows * 11 // world and torch.equal(rgb, full_rgb) and torch.equal(depth, full_depth), "interleaved bands"
rgb, depth = render_sharded(frame[: 24 * 11], fn2, frame_width=11)          # 24 rows: not world x 8 x k -> contiguous tiles
assert torch.equal(rgb, fn2(frame[: 24 * 11])[0])
dist.barrier()
dist.destroy_process_group()
print("OK", os.environ["RANK"])
'''


@pytest.mark.parametrize("world", [2, 4])
def test_ray_tile_sharding_gloo_world2(tmp_path, world):
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


DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["T2N_ROOT"])
import numpy as np, torch, torch.distributed as dist
from text2nerf_amd.parallel import shard_batch, allreduce_gradients, broadcast_parameters
from text2nerf_amd import synth
from oracle import oracle_torch as O          # CPU stand-in for the HIP forward/backward: the test checks the DP algebra
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
grid, aabb = [24, 20, 16], [[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]]
cfg = O.FieldConfig(aabb=aabb, grid_size=grid)
P = O.params_from_numpy(synth.make_field_params(11 + rank, grid, density_scale=0.9, aabb=aabb))   # ranks start DIFFERENT
names = sorted(P.keys())
params = [P[k].requires_grad_(True) for k in names]
broadcast_parameters(params, src=0)
ref = O.params_from_numpy(synth.make_field_params(11, grid, density_scale=0.9, aabb=aabb))
assert all(torch.equal(P[k].detach(), ref[k]) for k in names), "broadcast must reproduce rank 0's parameters"
rays = torch.from_numpy(synth.frame_rays_np(8, 8, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))   # 64-ray batch
g = torch.Generator().manual_seed(3)
target = torch.rand(rays.shape[0], 3, generator=g)
def loss_of(p, lo, hi):
    rgb, depth, _, _ = O.forward(cfg, p, rays[lo:hi])
    return torch.mean((rgb - target[lo:hi]) ** 2) + 0.005 * torch.mean(depth ** 2)
lo, hi = shard_batch(rays.shape[0], world, rank)
loss_of(P, lo, hi).backward()
allreduce_gradients(params, average=True)
# single-process reference: the whole batch
refp = [ref[k].requires_grad_(True) for k in names]
loss_of(ref, 0, rays.shape[0]).backward()
for k, a, b in zip(names, params, refp):
    gb = b.grad if b.grad is not None else torch.zeros_like(b)
    tol = 1e-6 * float(gb.abs().max()) + 1e-12
    assert float((a.grad - gb).abs().max()) <= tol, (k, float((a.grad - gb).abs().max()), tol)
# the factor-gradient buffer's two buckets (density prefix | appearance rest, parallel.allreduce_buckets) against ONE flat message:
# bitwise the same sums, for a split in the middle, at the ends and for a first bucket that is handed a "ready" hook
from text2nerf_amd.parallel import allreduce_buckets
gen = torch.Generator().manual_seed(100 + rank)
base = torch.randn(100003, generator=gen)
flat = base.clone()
dist.all_reduce(flat)
for split in (0, 1, 25001, 100002, 100003):
    b = base.clone()
    allreduce_buckets(b, split)
    assert torch.equal(b, flat), split
called = []
b = base.clone()
allreduce_buckets(b, 25001, first_ready=lambda st: called.append(st))     # CPU tensors: the hook is for device streams only
assert torch.equal(b, flat) and not called
dist.barrier()
dist.destroy_process_group()
print("OK", rank)
'''


def test_data_parallel_gradient_allreduce_gloo_world2(tmp_path):
    """SURVEY 8(e) train partitioning: shard the batch, local backward, one flat all-reduce == the full-batch gradient."""
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   T2N_ROOT=ROOT, OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"OK {r}" in o, o


def test_reference_pose_fixture():
    """tests/golden/poses.npz (tests/golden/make_golden_poses.py: the reference's own trajectory generators): shapes, rotations, the
    first local_fixed pose is the reference pose, the circle keeps its radius."""
    import numpy as np
    P = dict(np.load(os.path.join(ROOT, "tests", "golden", "poses.npz")))
    assert P["local_fixed"].shape == (9, 4, 4) and P["circle_train_96"].shape == (48, 4, 4)
    assert P["eval_spiral_120"].shape == (120, 4, 4) and P["circle_eval_360"].shape == (360, 4, 4)
    for k in ("local_fixed", "circle_train_96", "eval_spiral_120"):
        R = P[k][:, :3, :3]
        assert np.allclose(np.einsum("vij,vkj->vik", R, R), np.eye(3)[None], atol=1e-5), k
    assert np.allclose(P["local_fixed"][0], np.eye(4))
    assert np.abs(P["local_fixed"][1:, :3, 3]).max() <= 0.2 + 1e-6          # range_center = 0.2


def test_raster_width_detection_for_drop_in_evaluation():
    """renderer.detect_frame_width: the width of a row-major pinhole raster (what the reference's evaluation passes to the renderer,
    renderer.py:85-89) from its first rays — narrow and wide fields of view, normalised and unnormalised directions, ragged sizes;
    0 for shuffled rays, several origins, too few rows, and widths that do not divide the ray count."""
    import numpy as np
    import torch
    from text2nerf_amd import synth
    from text2nerf_amd.renderer import detect_frame_width

    def raster(H, W, focal, c2w, normalise=True):
        j, i = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
        d = np.stack([(i - W / 2 + 0.5) / focal, -(j - H / 2 + 0.5) / focal, -np.ones_like(i)], -1).reshape(-1, 3)
        if normalise:
            d = d / np.linalg.norm(d, axis=1, keepdims=True)
        d = d @ c2w[:3, :3].T
        o = np.broadcast_to(c2w[:3, 3], d.shape)
        return torch.from_numpy(np.concatenate([o, d], 1).astype(np.float32))

    pose = synth.look_pose(0.7, -0.3, (0.4, -0.2, 1.5))
    for H, W, focal, norm in ((20, 24, 30.0, True), (64, 48, 500.0, True), (100, 800, 400.0, True), (800, 800, 800.0, True),
                              (512, 512, 128.0, True), (37, 51, 40.0, False), (8, 8, 6.0, True)):
        assert detect_frame_width(raster(H, W, focal, pose, norm)) == W, (H, W, focal, norm)
    r = raster(64, 64, 60.0, pose)
    assert detect_frame_width(r[torch.randperm(r.shape[0], generator=torch.Generator().manual_seed(0))]) == 0
    r2 = r.clone(); r2[100, :3] += 1.0
    assert detect_frame_width(r2) == 0                       # not one camera
    assert detect_frame_width(raster(7, 64, 60.0, pose)) == 0   # fewer than 8 rows
    assert detect_frame_width(r[: 64 * 10 + 5]) == 0         # the width does not divide the ray count
    assert detect_frame_width(r[:40]) == 0


def test_band_interleaved_sharding_round_trip():
    """parallel.band_layout / band_shard / band_unshard (C4: every rank takes every world-th 8-row band of the frame)."""
    from text2nerf_amd.parallel import band_layout, band_shard, band_unshard
    assert band_layout(1600 * 1600, 1600, 8) == (25, 8 * 1600)
    assert band_layout(1600 * 1600, 1600, 7) is None and band_layout(100, 0, 2) is None and band_layout(101, 10, 2) is None
    W, H, world = 16, 64, 4
    rays = torch.arange(W * H * 6, dtype=torch.float32).view(W * H, 6)
    parts = [band_shard(rays, W, world, r) for r in range(world)]
    assert all(p.shape == (W * H // world, 6) for p in parts)
    assert torch.equal(parts[1][:8 * W], rays[8 * W:16 * W])               # rank 1's first band = image rows 8..15
    assert torch.equal(band_unshard(torch.cat(parts), W, world), rays)


def test_bench_gpus_n_without_launcher_refuses_cleanly_without_gpus():
    """`python bench.py --gpus 2` with no WORLD_SIZE self-launches; on a node with fewer GPUs it must say so and exit 2 — from
    the parent, before any rank exists."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "T2N_BENCH_SAME_DEVICE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2 and "one rank per GPU" in p.stderr and p.stdout.strip() == ""


class _FakeDataset:
    """What renderer.evaluation / evaluation_path read from a SceneGenDataset (dataLoader/scene_gen.py)."""

    def __init__(self, split, n_views, H, W):
        from text2nerf_amd import synth
        self.split, self.img_wh, self.near_far = split, (W, H), [0.5, 8.0]
        g = torch.Generator().manual_seed(0)
        rays = torch.stack([torch.from_numpy(synth.frame_rays_np(H, W, c2w=synth.look_pose(0.1 * v, 0.0, (0.0, 0.0, 0.0))))
                            for v in range(n_views)])
        self.all_rays_split = rays
        self.all_rays_sprt_split = rays[:2]
        self.all_rays_gen_split = rays
        self.all_rgbs_gen_split = torch.rand(n_views, H * W, 3, generator=g)
        self.directions = torch.from_numpy(synth.ray_directions_np(H, W, float(W), float(W), W // 2, H // 2))
        self.focal = [float(W), float(W)]


def test_evaluation_and_evaluation_path_drop_in(tmp_path, monkeypatch):
    """renderer.evaluation / evaluation_path with the reference's signatures (renderer.py:44-197): view selection by split /
    preview / N_iter / N_vis, one renderer call per view with all rays of the image, file names, PSNR list. The renderer and the
    post-processing kernel are stood in for on the CPU (fake renderer; the oracle's postprocess_frame) — the GPU forms are covered
    by tests/test_postprocess.py and the parity suite."""
    from types import SimpleNamespace
    from PIL import Image
    from oracle import oracle_torch as O
    from text2nerf_amd import renderer as Rm
    H, W = 12, 16
    calls = []

    def fake_renderer(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False, device="cuda"):
        calls.append((tuple(rays.shape), chunk, N_samples, white_bg, tensorf.materialize_weights))
        t = torch.linspace(0, 1, rays.shape[0])
        return torch.stack([t, 1 - t, 0.5 * t], 1) * 1.2 - 0.1, None, 2.0 + 5.0 * t, None, None

    def fake_post(rgb, depth, near_far, push_depth=None, gt_rgb=None):
        r8, d8, ps = O.postprocess_frame(rgb.numpy(), depth.numpy(), near_far, push_depth=push_depth,
                                         gt_rgb=None if gt_rgb is None else gt_rgb.numpy())
        return torch.from_numpy(r8), torch.from_numpy(d8), ps

    monkeypatch.setattr(Rm, "postprocess_frame", fake_post)
    monkeypatch.setattr(Rm, "get_rays_for_path", None, raising=False)
    field = SimpleNamespace(materialize_weights=True, device="cpu")
    args = SimpleNamespace(batch_size=4096, push_depth=2.0)
    ds = _FakeDataset("train", 5, H, W)
    out = tmp_path / "inpaint"
    psnrs = Rm.evaluation(ds, field, args, fake_renderer, str(out), N_vis=-1, prtx="e01_", N_samples=77, white_bg=True,
                          compute_extra_metrics=True, device="cpu", N_iter=2, preview=False)
    assert len(psnrs) == 3 and all(np.isfinite(p) for p in psnrs)          # views [: N_iter + 1] with ground truth
    assert calls == [((H * W, 6), 4096, 77, True, False)] * 3 and field.materialize_weights is True
    assert sorted(os.listdir(out / "rgbs")) == [f"e01_{i:03d}_rgb.png" for i in range(3)]
    assert sorted(os.listdir(out / "depths")) == [f"e01_{i:03d}_depth.png" for i in range(3)]
    img = np.asarray(Image.open(out / "rgbs" / "e01_000_rgb.png"))
    t = torch.linspace(0, 1, H * W)
    want = ((torch.stack([t, 1 - t, 0.5 * t], 1) * 1.2 - 0.1).clamp(0, 1).numpy() * 255).astype("uint8").reshape(H, W, 3)
    assert img.shape == (H, W, 3) and np.array_equal(img, want)
    # PSNR of view 0 as the reference computes it (renderer.py:100-101)
    gt = ds.all_rgbs_gen_split[0].view(H, W, 3)
    ref = -10.0 * np.log(float(torch.mean((torch.from_numpy(want.astype(np.float32)) * 0 + (torch.stack([t, 1 - t, 0.5 * t], 1) * 1.2 - 0.1)
                                           .clamp(0, 1).view(H, W, 3) - gt) ** 2))) / np.log(10.0)
    assert abs(psnrs[0] - ref) < 1e-4
    # preview = support views, no ground truth -> no PSNR; compute_extra_metrics False (the driver's per-epoch call) likewise
    calls.clear()
    assert Rm.evaluation(ds, field, args, fake_renderer, str(tmp_path / "sprt"), N_vis=-1, compute_extra_metrics=False, device="cpu",
                         preview=True) == []
    assert len(calls) == 2
    assert Rm.evaluation(ds, field, args, fake_renderer, None, N_vis=-1, compute_extra_metrics=False, device="cpu", N_iter=4) == []
    # test split: every view, N_vis subsampling (5 views, N_vis = 2 -> interval 2 -> views 0, 2, 4)
    calls.clear()
    dt = _FakeDataset("test", 5, H, W)
    Rm.evaluation(dt, field, args, fake_renderer, str(tmp_path / "test"), N_vis=2, device="cpu", video_gen=True)
    assert len(calls) == 3 and len(os.listdir(tmp_path / "test" / "rgbs")) == 3

    # evaluation_path: rays from get_rays(directions, c2w) (a HIP kernel in the product; the oracle's here)
    def cpu_get_rays(directions, c2w):
        return O.get_rays(directions, c2w)
    import text2nerf_amd.ray_utils as RU
    monkeypatch.setattr(RU, "get_rays", cpu_get_rays)
    calls.clear()
    c2ws = [np.eye(4, dtype=np.float32)[:3], np.eye(4, dtype=np.float32)[:3]]
    res = Rm.evaluation_path(dt, field, c2ws, fake_renderer, str(tmp_path / "path"), prtx="p_", N_samples=-1, white_bg=True, device="cpu")
    assert res == [] and calls == [((H * W, 6), 8192, -1, True, False)] * 2
    assert sorted(os.listdir(tmp_path / "path")) == ["p_000.png", "p_001.png", "rgbd"]
    rgbd = np.asarray(Image.open(tmp_path / "path" / "rgbd" / "p_001.png"))
    assert rgbd.shape == (H, 2 * W, 3) and np.array_equal(rgbd[:, :W], want)


def test_signatures_match_the_reference():
    """Every callable of the reference's hot-path surface (tests/golden/signatures.json <- make_golden_signatures.py, which imports the
    reference) exists in the mirror under the same name with the same parameters — names, order, kinds, defaults. The mirror may ADD
    trailing keyword parameters with defaults (device=, normalize=, frame_width=, ...): a call written for the reference binds the same."""
    import inspect
    import json
    import text2nerf_amd as T
    from text2nerf_amd import ray_utils, renderer, sh, tensorf
    from tests.conftest import GOLDEN
    want = json.load(open(os.path.join(GOLDEN, "signatures.json")))
    bases = want.pop("_bases")
    mods = {"renderer": renderer, "tensorBase": tensorf, "sh": sh, "ray_utils": ray_utils}
    # what the mirror deliberately differs in, each with its reason
    allowed = {
        # list-valued defaults print differently but are equal; checked by value below
    }
    problems = []
    for key, params in sorted(want.items()):
        owner, name = key.split(".")
        obj = getattr(mods[owner], name, None) if owner in mods else getattr(getattr(T, owner, None), name, None)
        if obj is None:
            problems.append(f"{key}: missing")
            continue
        got = [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)]
               for p in inspect.signature(obj).parameters.values()]
        for i, w in enumerate(params):
            if i >= len(got):
                problems.append(f"{key}: parameter {w[0]!r} missing")
                break
            g = got[i]
            same_default = g[2] == w[2] or (g[2] is not None and w[2] is not None and eval(g[2]) == eval(w[2]))   # (2, 6) vs [2, 6] etc. aside
            if g[0] != w[0] or g[1] != w[1] or not same_default:
                if key not in allowed:
                    problems.append(f"{key}: parameter {i} is {g}, the reference has {w}")
        for g in got[len(params):]:
            if g[2] is None and g[1] not in ("VAR_POSITIONAL", "VAR_KEYWORD"):
                problems.append(f"{key}: extra parameter {g[0]!r} without a default")
    for cname, bs in bases.items():
        cls = getattr(T, cname, None) or getattr(renderer, cname)
        have = [b.__name__ for b in cls.__mro__[1:-1]]
        for b in bs:
            if b not in have:
                problems.append(f"{cname}: the reference derives from {b}, the mirror's bases are {have}")
    assert not problems, "\n".join(problems)
