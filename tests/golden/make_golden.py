#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only) on CPU.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite consume
the committed .npz files, never this script's imports. Packages the reference imports but the image
lacks (cv2, imageio, kornia, torchvision, ...) are replaced by inert stubs; the one function of
those the hot path touches, ``kornia.create_meshgrid``, gets a stand-in with kornia 0.6.7's
documented semantics (x = linspace(0, W-1, W), y likewise, shape [1,H,W,2], last dim (x, y)).

    python tests/golden/make_golden.py

Inputs are rebuilt from seeds by text2nerf_amd.synth (numpy PCG64), so fixtures hold only what a
test cannot regenerate: the reference's OUTPUTS (plus small inputs such as jitter draws).
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

for name in ["cv2", "imageio", "imageio.v2", "configargparse", "torchvision", "torchvision.transforms",
             "statsmodels", "statsmodels.api", "lpips", "plyfile", "skimage", "skimage.metrics",
             "skimage.measure", "scripts.Warper"]:
    sys.modules.setdefault(name, MagicMock())
kornia = types.ModuleType("kornia")


def create_meshgrid(height, width, normalized_coordinates=True, device=None, dtype=torch.float32):
    xs = torch.linspace(0, width - 1, width, dtype=dtype)
    ys = torch.linspace(0, height - 1, height, dtype=dtype)
    assert not normalized_coordinates
    base = torch.stack(torch.meshgrid([xs, ys], indexing="ij"), dim=-1)  # [W,H,2]
    return base.permute(1, 0, 2).unsqueeze(0)  # [1,H,W,2]


kornia.create_meshgrid = create_meshgrid
sys.modules["kornia"] = kornia

from text2nerf_amd import synth  # noqa: E402

torch.set_num_threads(8)
import io  # noqa: E402
import contextlib  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from models.tensoRF import TensorVMSplit, raw2alpha  # noqa: E402
    from models.tensorBase import positional_encoding, SHRender  # noqa: E402
    from models.sh import eval_sh_bases  # noqa: E402
    from dataLoader.ray_utils import get_ray_directions, get_rays  # noqa: E402
    import renderer as ref_renderer  # noqa: E402
    import utils as ref_utils  # noqa: E402


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


TINY = dict(grid=[24, 20, 16], aabb=[[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]], near_far=[0.5, 8.0])


def build_ref(seed, grid, aabb, near_far, scene="random", density_scale=0.1, shading="MLP_Fea_noview",
              step_ratio=1.0):
    sd = synth.make_field_params(seed, grid, scene=scene, density_scale=density_scale, aabb=aabb,
                                 shading_mode=shading)
    m = quiet(TensorVMSplit, torch.tensor(aabb, dtype=torch.float32), list(grid), "cpu",
              density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48], app_dim=27, near_far=near_far,
              shadingMode=shading, alphaMask_thres=1e-4, density_shift=-10, distance_scale=25,
              pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=step_ratio, fea2denseAct="softplus")
    missing = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m, sd


def tiny_rays():
    """192 camera rays from a non-identity pose + hand-made edge rays (zero component, outside, miss)."""
    c2w = synth.look_pose(yaw=0.35, pitch=-0.2, center=(0.4, -0.3, -1.0))
    H, W = 12, 16
    d = get_ray_directions(H, W, [float(max(H, W))] * 2, center=[W // 2, H // 2])
    d = d / torch.norm(d, dim=-1, keepdim=True)
    ro, rd = get_rays(d, torch.from_numpy(c2w))
    rays = torch.cat([ro, rd], 1)
    extra = torch.tensor([
        [0.0, 0.0, 0.0, 0.0, 0.0, 1.0],          # two zero direction components
        [0.0, 0.0, -20.0, 0.0, 0.0, 1.0],        # starts outside, enters at z=-7 (t_min=13 -> clamped to far=8)
        [0.0, 0.0, -12.0, 0.1, 0.0, 0.995],      # outside, t_min=5.0x
        [30.0, 0.0, 0.0, 0.0, 1.0, 0.0],         # misses the box entirely
        [1.0, 1.0, 1.0, 0.6, 0.0, 0.8],          # zero y component
        [-7.9, 6.9, 2.5, -0.7, -0.7, 0.14],      # grazing, near corners
        [0.0, 0.0, 1.9, 0.0, 0.3, 0.954],        # crosses the z>2 gate late
        [0.0, 0.0, 6.0, 0.0, 0.0, -1.0],         # moving toward -z: gate switches off
    ], dtype=torch.float32)
    return torch.cat([rays, extra], 0), c2w, (H, W)


def main():
    out = {}

    # ---- G9: host arithmetic --------------------------------------------------------------
    aabb300 = torch.tensor([[-8.0] * 3, [8.0] * 3])
    out["g9_n_to_reso_27e6"] = np.array(ref_utils.N_to_reso(27000000, aabb300))
    out["g9_n_to_reso_2097156"] = np.array(ref_utils.N_to_reso(2097156, aabb300))
    out["g9_cal_n_samples_300_1"] = np.array(ref_utils.cal_n_samples([300, 300, 300], 1.0))
    for g in (300, 128):
        m, _ = build_ref(0, [g] * 3 if g == 128 else [8, 8, 8], [[-8.0] * 3, [8.0] * 3], [0.5, 8.0])
        quiet(m.update_stepSize, [g] * 3)
        out[f"g9_step_{g}"] = np.array([m.stepSize.item(), m.nSamples], np.float64)
    x = torch.from_numpy(synth._randn(np.random.Generator(np.random.PCG64(5)), (1, 4, 9, 7), 1.0))
    out["g9_tv_in"] = x.numpy()
    out["g9_tv_out"] = ref_utils.TVLoss()(x).numpy()

    # ---- G1: ray generation ---------------------------------------------------------------
    H, W = 6, 8
    c2w = synth.look_pose(yaw=0.5, pitch=0.25, center=(0.3, -0.2, 0.1))
    dirs = get_ray_directions(H, W, [9.0, 7.5], center=[W // 2, H // 2])
    out["g1_dirs_raw"] = dirs.numpy()
    dn = dirs / torch.norm(dirs, dim=-1, keepdim=True)
    ro, rd = get_rays(dn, torch.from_numpy(c2w))
    out["g1_c2w"] = c2w
    out["g1_rays"] = torch.cat([ro, rd], 1).numpy()

    # ---- tiny model -----------------------------------------------------------------------
    m, sd = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9)
    out["tiny_step"] = np.array([m.stepSize.item(), m.nSamples], np.float64)
    rays, c2w_t, _ = tiny_rays()
    out["tiny_rays"] = rays.numpy()

    # G2 sample_ray eval/train
    with torch.no_grad():
        pts, z, valid = m.sample_ray(rays[:, :3], rays[:, 3:6], is_train=False, N_samples=-1)
        out["g2_eval_pts"], out["g2_eval_z"], out["g2_eval_valid"] = pts.numpy(), z.numpy(), valid.numpy()
        torch.manual_seed(77)
        jit = torch.rand(rays.shape[0], 1)
        torch.manual_seed(77)
        pts, z, valid = m.sample_ray(rays[:, :3], rays[:, 3:6], is_train=True, N_samples=40)
        out["g2_train_jitter"] = jit.numpy()
        out["g2_train_pts"], out["g2_train_z"], out["g2_train_valid"] = pts.numpy(), z.numpy(), valid.numpy()

    # G3 density at points incl. exact borders
    rng = np.random.Generator(np.random.PCG64(3))
    p = rng.uniform(-1, 1, size=(4096, 3)).astype(np.float32)
    p[:8] = np.array([[-1, -1, -1], [1, 1, 1], [-1, 1, 0.3], [0.2, -1, 1], [1, 0, 0], [0, 0, 0],
                      [0.99999994, -0.99999994, 0.5], [-1, 0.5, 1]], np.float32)
    out["g3_xyz"] = p
    with torch.no_grad():
        f = m.compute_densityfeature(torch.from_numpy(p))
        out["g3_feat"] = f.numpy()
        out["g3_sigma"] = m.feature2density(f).numpy()
        xw = torch.from_numpy(p) * 7.0
        out["g3_norm_in"] = xw.numpy()
        out["g3_norm_out"] = m.normalize_coord(xw).numpy()

    # G4 raw2alpha incl. saturation
    sg = torch.from_numpy(np.abs(rng.standard_normal((16, 50)).astype(np.float32)) * 0.3)
    sg[0, 10] = 40.0
    sg[1, :] = 0.0
    sg[2, 5:9] = 5.0
    dist = torch.full((16, 50), 0.76 * 25)
    dist[:, -1] = 0
    a, w, bg = raw2alpha(sg, dist)
    out["g4_sigma"], out["g4_dist"] = sg.numpy(), dist.numpy()
    out["g4_alpha"], out["g4_weight"], out["g4_bg"] = a.numpy(), w.numpy(), bg.numpy()

    # G5 appearance + shading heads
    with torch.no_grad():
        pa = torch.from_numpy(p[:1024])
        af = m.compute_appfeature(pa)
        out["g5_appfeat"] = af.numpy()
        out["g5_pe"] = positional_encoding(af, 6).numpy()
        out["g5_rgb_mlp"] = m.renderModule(pa, None, af).numpy()
        vd = torch.from_numpy(rng.standard_normal((1024, 3)).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        out["g5_viewdirs"] = vd.numpy()
        out["g5_sh_basis"] = eval_sh_bases(2, vd).numpy()
        out["g5_rgb_sh"] = SHRender(pa, vd, af).numpy()

    # G6 full forward: eval / train, white_bg on/off, N=-1 (33) and N=70
    with torch.no_grad():
        for tag, kw in [("eval", dict(is_train=False, white_bg=True, N_samples=-1)),
                        ("eval70", dict(is_train=False, white_bg=True, N_samples=70)),
                        ("evalblack", dict(is_train=False, white_bg=False, N_samples=-1))]:
            rgb, depth, zv, wt = m(rays, ndc_ray=False, **kw)
            out[f"g6_{tag}_rgb"], out[f"g6_{tag}_depth"] = rgb.numpy(), depth.numpy()
            out[f"g6_{tag}_z"], out[f"g6_{tag}_w"] = zv.numpy(), wt.numpy()
        torch.manual_seed(123)
        jit = torch.rand(rays.shape[0], 1)
        torch.manual_seed(123)
        rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=40)
        out["g6_train_jitter"] = jit.numpy()
        out["g6_train_rgb"], out["g6_train_depth"] = rgb.numpy(), depth.numpy()
        out["g6_train_z"], out["g6_train_w"] = zv.numpy(), wt.numpy()
        # train + white_bg=False: the reference then flips a coin on the CPU generator AFTER the jitter draw (:497)
        for seed in (7, 8, 9, 10):
            torch.manual_seed(seed)
            _ = torch.rand(rays.shape[0], 1)
            coin = bool(torch.rand((1,)) < 0.5)
            torch.manual_seed(seed)
            rgb, depth, zv, wt = m(rays, is_train=True, white_bg=False, ndc_ray=False, N_samples=40)
            out[f"g6_trainblack{seed}_coin"] = np.array(coin)
            out[f"g6_trainblack{seed}_rgb"], out[f"g6_trainblack{seed}_depth"] = rgb.numpy(), depth.numpy()
        # filtering_rays(bbox_only=True) on the edge rays (:385-391)
        frays = torch.cat([rays, rays[:, [3, 4, 5, 0, 1, 2]] * 3.0], 0)
        kept = quiet(m.filtering_rays, frays, torch.zeros(frays.shape[0], 3), bbox_only=True)
        out["g16_rays"] = frays.numpy()
        out["g16_kept"] = kept[0].numpy()

    # a-7 alpha mask branch (models/tensorBase.py:41-59,451-456): synthetic occupancy volume on a slightly smaller box
    from models.tensorBase import AlphaGridMask
    ag = np.random.Generator(np.random.PCG64(99))
    vol = (ag.uniform(0, 1, (9, 11, 13)) < 0.30).astype(np.float32)          # [D(z), H(y), W(x)]
    vol[:, 4:6, 6] = 1.0
    mask_aabb = torch.tensor([[-7.5, -5.5, -6.5], [7.5, 6.5, 6.0]])
    out["g7a_volume"], out["g7a_aabb"] = vol, mask_aabb.numpy()
    m.alphaMask = AlphaGridMask("cpu", mask_aabb, torch.from_numpy(vol))
    with torch.no_grad():
        pw = torch.from_numpy(p[:2048]) * torch.tensor([8.5, 7.0, 7.5])
        out["g7a_pts"] = pw.numpy()
        out["g7a_alpha"] = m.alphaMask.sample_alpha(pw).numpy()
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["g7a_eval_rgb"], out["g7a_eval_depth"], out["g7a_eval_w"] = rgb.numpy(), depth.numpy(), wt.numpy()
        torch.manual_seed(321)
        rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=40)
        out["g7a_train_rgb"], out["g7a_train_depth"], out["g7a_train_w"] = rgb.numpy(), depth.numpy(), wt.numpy()
    m.alphaMask = None

    # G7 renderer harness: R not divisible by chunk
    with torch.no_grad():
        r5 = ref_renderer.OctreeRender_trilinear_fast(rays, m, chunk=64, N_samples=-1, ndc_ray=False,
                                                      white_bg=True, is_train=False, device="cpu")
    assert r5[1] is None
    out["g7_rgb"], out["g7_depth"], out["g7_w"], out["g7_z"] = [t.numpy() for t in (r5[0], r5[2], r5[3], r5[4])]

    # G8 gradients through the whole path (train mode, captured jitter)
    gr = np.random.Generator(np.random.PCG64(8))
    ca = torch.from_numpy(gr.standard_normal((rays.shape[0], 3)).astype(np.float32))
    cb = torch.from_numpy(gr.standard_normal((rays.shape[0],)).astype(np.float32))
    cc = torch.from_numpy(gr.standard_normal((rays.shape[0], 40)).astype(np.float32) * 0.1)
    out["g8_ca"], out["g8_cb"], out["g8_cc"] = ca.numpy(), cb.numpy(), cc.numpy()
    m.zero_grad()
    torch.manual_seed(123)
    rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=40)
    loss = (rgb * ca).sum() + (depth * cb).sum() + (wt * cc).sum()
    loss.backward()
    out["g8_loss"] = loss.detach().numpy()
    for k, prm in m.named_parameters():
        out["g8_grad." + k] = prm.grad.numpy()

    # G10 state_dict layout / kwargs / optimiser groups
    out["g10_keys"] = np.array(list(m.state_dict().keys()))
    out["g10_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
    groups = m.get_optparam_groups(0.02, 1e-3)
    out["g10_group_lr"] = np.array([g["lr"] for g in groups])
    out["g10_group_sizes"] = np.array([sum(p.numel() for p in g["params"]) for g in groups])
    kw = m.get_kwargs()
    out["g10_kwargs_keys"] = np.array(sorted(kw.keys()))

    # SH-head field forward (shadingMode='SH')
    msh, _ = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9, shading="SH")
    with torch.no_grad():
        rgb, depth, zv, wt = msh(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
    out["g6_sh_rgb"], out["g6_sh_depth"] = rgb.numpy(), depth.numpy()

    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **out)
    napp = (out["g6_eval_w"] > 1e-4).sum(1)
    print("tiny: rays", rays.shape[0], "nSamples", m.nSamples, "app/ray mean", napp.mean(), "max", napp.max(),
          "valid frac", out["g2_eval_valid"].mean())

    # ---- 300^3 production-shaped spot checks (params rebuilt from seed by the tests) ----------
    big = {}
    for scene, seed in (("S1-soft", 0), ("S2", 1)):
        mb, _ = build_ref(seed, [300] * 3, [[-8.0] * 3, [8.0] * 3], [0.5, 8.0], scene=scene)
        full = synth.frame_rays_np(800, 800)
        idx = np.random.Generator(np.random.PCG64(42)).choice(full.shape[0], 96, replace=False)
        idx.sort()
        r = torch.from_numpy(full[idx])
        with torch.no_grad():
            rgb, depth, zv, wt = mb(r, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
            torch.manual_seed(5)
            jit = torch.rand(r.shape[0], 1)
            torch.manual_seed(5)
            rgb_t, depth_t, zv_t, wt_t = mb(r, is_train=True, white_bg=True, ndc_ray=False, N_samples=259)
        big[f"{scene}_idx"] = idx
        big[f"{scene}_step"] = np.array([mb.stepSize.item(), mb.nSamples], np.float64)
        big[f"{scene}_rgb"], big[f"{scene}_depth"] = rgb.numpy(), depth.numpy()
        big[f"{scene}_acc"] = wt.sum(-1).numpy()
        big[f"{scene}_napp"] = (wt > 1e-4).sum(-1).numpy()
        big[f"{scene}_w_first16"] = wt[:16].numpy()
        big[f"{scene}_train_jitter"] = jit.numpy()
        big[f"{scene}_train_rgb"], big[f"{scene}_train_depth"] = rgb_t.numpy(), depth_t.numpy()
        big[f"{scene}_train_acc"] = wt_t.sum(-1).numpy()
        print(scene, "step", mb.stepSize.item(), "N", mb.nSamples, "app/ray", big[f"{scene}_napp"].mean(),
              "rgb mean", rgb.mean().item())
    np.savez_compressed(os.path.join(HERE, "big300.npz"), **big)


if __name__ == "__main__":
    main()
