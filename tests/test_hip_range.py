"""Range stress of the split-f16 appearance stage (default render path): every fp32 product of basis_mat and the MLP runs as three
f16 MFMA products of hi/lo halves, so magnitudes matter — f16 has 5 exponent bits. The reference's arithmetic is fp32 nn.Linear
(models/tensorBase.py:94-106, models/tensoRF.py:147,239). Each case renders a frame of the tiny field with rescaled appearance
factors / head weights and compares with the oracle at the north-star tolerance (1e-4 RGB); cases built to leave the f16 range
must be caught by the kernels' range flag and come out of the exact-fp32 redo (stats()["f16_redo"] > 0) — still within 1e-4.

Scales are applied where they change magnitudes seen by the split path: appearance planes / lines (-> plane x line products),
basis_mat, the three MLP layers. Density factors stay put so the sample set is the same."""
import numpy as np
import pytest
import torch

from text2nerf_amd import synth
from tests.conftest import TINY
from tests.test_hip_parity import RGB_ATOL, dev, make_field

pytestmark = pytest.mark.gpu


def rays():
    return torch.from_numpy(synth.frame_rays_np(20, 24, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0))))


def render_both(params):
    from oracle import oracle_torch as O
    f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    r = rays()
    with torch.no_grad():
        rgb, depth, _, _ = f(r.to(dev()), white_bg=True, is_train=False, N_samples=-1)
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    o_rgb, o_depth, _, _ = O.forward(cfg, O.params_from_numpy(params), r)
    st = f.stats()
    assert st["appearance"] > 500, st          # the case must actually exercise the appearance stage
    return rgb.cpu().numpy(), o_rgb.numpy(), st


def scaled(base, **scale):
    """copy of the parameter dict with every tensor whose name starts with a key of `scale` multiplied by that factor"""
    out = {}
    for k, v in base.items():
        s = 1.0
        for pre, fac in scale.items():
            if k.startswith(pre):
                s *= fac
        out[k] = (v * np.float32(s)).astype(np.float32)
    return out


CASES = {
    # name: (scales, expect_redo)  expect_redo None = either
    "baseline": ({}, False),
    "app factors x8 each (products x64)": ({"app_plane": 8.0, "app_line": 8.0}, None),
    "app factors x1/32 each (products x1/1024)": ({"app_plane": 1 / 32.0, "app_line": 1 / 32.0}, False),
    "basis_mat x64": ({"basis_mat": 64.0}, None),
    "basis_mat x1/1024": ({"basis_mat": 1 / 1024.0}, False),
    "layer 0 x64": ({"renderModule.mlp.0": 64.0}, None),
    "layer 1 x64": ({"renderModule.mlp.2": 64.0}, None),
    "all MLP layers x1/1024": ({"renderModule.mlp": 1 / 1024.0}, False),
    "all MLP layers x8": ({"renderModule.mlp": 8.0}, None),
}


@pytest.mark.parametrize("name", list(CASES))
def test_scaled_fields_match_oracle(tiny_params, name):
    scales, expect_redo = CASES[name]
    got, ref, st = render_both(scaled(tiny_params, **scales))
    err = float(np.abs(got - ref).max())
    assert err <= RGB_ATOL, (name, err, st)
    if expect_redo is not None:
        assert (st["f16_redo"] > 0) == expect_redo, (name, st)


def test_weight_beyond_256_and_large_activations(tiny_params):
    """|w| > 256 (the round-1 path pre-scaled by a fixed 2^8 and overflowed there) and hidden activations beyond 2^15 / 2^8"""
    p = {k: v.copy() for k, v in tiny_params.items()}
    w1 = p["renderModule.mlp.2.weight"]
    w1[3, 5] = 300.0
    w1[17, 9] = -411.0
    got, ref, st = render_both(p)
    assert float(np.abs(got - ref).max()) <= RGB_ATOL, st
    p = scaled(tiny_params, **{"renderModule.mlp.0.bias": 1.0})
    p["renderModule.mlp.0.bias"] = p["renderModule.mlp.0.bias"] + np.float32(400.0)       # h0 ~ 400 on every unit
    got, ref, st = render_both(p)
    assert float(np.abs(got - ref).max()) <= RGB_ATOL, st


def test_values_outside_f16_range_take_the_exact_redo(tiny_params):
    """hidden activations of ~1e5 cannot be split into f16 halves: the range flag must fire and the frame must come from the exact
    fp32 kernels (same tolerance)"""
    p = {k: v.copy() for k, v in tiny_params.items()}
    p["renderModule.mlp.0.bias"] = p["renderModule.mlp.0.bias"] + np.float32(1.0e5)
    p["renderModule.mlp.2.weight"] = p["renderModule.mlp.2.weight"] * np.float32(1e-3)   # keeps layer 1 in a sane range
    got, ref, st = render_both(p)
    assert st["f16_redo"] > 0, st
    assert float(np.abs(got - ref).max()) <= RGB_ATOL, st
    # features beyond the range (basis_mat x 1e7): caught on the raw-feature check
    got, ref, st = render_both(scaled(tiny_params, **{"basis_mat": 1.0e7, "renderModule.mlp.0": 1.0e-7}))
    assert st["f16_redo"] > 0, st
    assert float(np.abs(got - ref).max()) <= RGB_ATOL, st


def test_field_after_optimisation_steps_matches_oracle(tiny_params):
    """a trained checkpoint is where magnitudes drift: 200 fused TV + Adam steps on random targets, then a frame vs the oracle on the
    trained parameters"""
    from oracle import oracle_torch as O
    from text2nerf_amd import OctreeRender_trilinear_fast
    from text2nerf_amd.optim import TVAdam
    f = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    opt = TVAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=f)
    r = rays()
    g = torch.Generator().manual_seed(7)
    tgt = torch.rand(r.shape[0], 3, generator=g).to(dev())
    torch.manual_seed(11)
    for _ in range(200):
        rgb, _, depth, w, z = OctreeRender_trilinear_fast(r, f, chunk=r.shape[0], N_samples=-1, white_bg=True, is_train=True, device=dev())
        loss = torch.mean((rgb - tgt) ** 2)
        opt.zero_grad()
        loss.backward()
        opt.step(tv=[(f.density_plane, 0.1), (f.app_plane, 0.01)])
    params = {k: v.detach().cpu().numpy() for k, v in f.state_dict().items()}
    with torch.no_grad():
        rgb, depth, _, _ = f(r.to(dev()), white_bg=True, is_train=False, N_samples=-1)
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    o_rgb, _, _, _ = O.forward(cfg, O.params_from_numpy(params), r)
    st = f.stats()
    assert st["appearance"] > 100, st
    assert float((rgb.cpu() - o_rgb).abs().max()) <= RGB_ATOL, st
    assert float(loss.detach()) < 0.2            # the steps did optimise something


def _train_pass(params, n_samples=48, seed=5, check_grads=True):
    """training-mode forward + backward (the activation-keeping one-kernel appearance path, then the backward on its rows) and the
    oracle's forward + autograd on the same rays and jitter"""
    from oracle import oracle_torch as O
    from tests.test_hip_parity import _grad_check
    f = make_field(params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    r = rays()
    g = np.random.Generator(np.random.PCG64(seed))
    ca = torch.from_numpy(g.uniform(-1, 1, (r.shape[0], 3)).astype(np.float32))
    out = {}
    for it in range(2):   # iteration 0: the backward recomputes the activations; iteration 1: the forward keeps them
        for p in f.parameters():
            p.grad = None
        torch.manual_seed(seed)
        jit = torch.rand(r.shape[0], 1)
        torch.manual_seed(seed)
        rgb, depth, z, w = f(r, is_train=True, white_bg=True, N_samples=n_samples)
        ((rgb * ca.to(dev())).sum() + 0.1 * depth.sum()).backward()
        out[it] = (rgb.detach().cpu().numpy(), {k: p.grad.detach().cpu().numpy().copy() for k, p in f.named_parameters()})
    cfg = O.FieldConfig(aabb=TINY["aabb"], grid_size=TINY["grid"], near_far=TINY["near_far"])
    P = O.params_from_numpy(params, requires_grad=True)
    o = O.forward(cfg, P, r, white_bg=True, is_train=True, n_samples=n_samples, jitter=jit)
    ((o[0] * ca).sum() + 0.1 * o[1].sum()).backward()
    ref = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}
    st = f.stats()
    assert st["appearance"] > 500, st
    for it in (0, 1):
        assert float(np.abs(out[it][0] - o[0].detach().numpy()).max()) <= RGB_ATOL, (it, st)
    if check_grads:
        _grad_check(f, ref, rel=5e-4)
    return st


def test_training_forward_range_guard(tiny_params):
    """The training forward and the backward's recompute run the one-kernel split path, whose weights carry a fixed 2^8 pre-scale:
    a weight beyond 2^16 / 2^8 is found at upload (the split launch defers to the exact one behind it), an activation or feature
    beyond the f16 range at run time (range flag -> the exact launch rewrites every entry and kept activation row). Forward and
    gradients against the oracle in both cases, on the recompute iteration and on the kept-rows iteration."""
    p = {k: v.copy() for k, v in tiny_params.items()}
    p["renderModule.mlp.2.weight"][3, 5] = 300.0
    p["renderModule.mlp.2.weight"][17, 9] = -411.0
    _train_pass(p)
    p = {k: v.copy() for k, v in tiny_params.items()}
    p["renderModule.mlp.0.bias"] = p["renderModule.mlp.0.bias"] + np.float32(1.0e5)
    p["renderModule.mlp.2.weight"] = p["renderModule.mlp.2.weight"] * np.float32(1e-3)
    _train_pass(p)
    # features of ~1e5: forward only (an fp32 feature of that size carries an absolute error of ~1e-2, i.e. radians of error in
    # sin(2^5 f): the encoding's gradient is ill-conditioned there, in the reference as much as here)
    _train_pass(scaled(tiny_params, **{"basis_mat": 1.0e7, "renderModule.mlp.0": 1.0e-7}), check_grads=False)
    _train_pass(tiny_params)   # and the in-range field still takes the split launch (same bounds)


def test_explicit_point_shade_range_guard(tiny_params):
    """compute_appfeature / shade at explicit points (models/tensoRF.py:223-239 + renderModule) under the same three stresses"""
    from oracle import oracle_torch as O
    g = np.random.Generator(np.random.PCG64(9))
    xyz = torch.from_numpy(g.uniform(-1, 1, (777, 3)).astype(np.float32))
    cases = []
    p = {k: v.copy() for k, v in tiny_params.items()}
    p["renderModule.mlp.2.weight"][3, 5] = 300.0
    cases.append(p)
    p = {k: v.copy() for k, v in tiny_params.items()}
    p["renderModule.mlp.0.bias"] = p["renderModule.mlp.0.bias"] + np.float32(1.0e5)
    p["renderModule.mlp.2.weight"] = p["renderModule.mlp.2.weight"] * np.float32(1e-3)
    cases.append(p)
    cases.append(scaled(tiny_params, **{"basis_mat": 1.0e7, "renderModule.mlp.0": 1.0e-7}))
    cases.append(tiny_params)
    for p in cases:
        f = make_field(p, TINY["grid"], TINY["aabb"], TINY["near_far"])
        feat, rgb = f.shade(xyz.to(dev()))
        P = O.params_from_numpy(p)
        o_feat = O.app_feature(P, xyz)
        o_rgb = O.mlp_fea_noview(P, o_feat, 6)
        fscale = float(o_feat.abs().max()) + 1e-12
        assert float((feat.cpu() - o_feat).abs().max()) <= 2e-5 * fscale
        assert float((rgb.cpu() - o_rgb).abs().max()) <= RGB_ATOL


def test_one_kernel_appearance_paths_stay_correct():
    """The appearance forms beside the default two-kernel one are all production paths and are reached here without switches: the
    cooperative one-kernel form (`f.shade` at explicit points: t2n_shade_at; tiles beyond the feature-row capacity of a render) and
    the same cooperative form with kept activation rows (training forward; the per-wave form keeps them under bf16 factor storage)
    — the range stresses and goldens of this file and of test_hip_parity.py cover the first through f.shade and the second through
    the gradient tests; here the overflow-tile route: a
    render whose feature-row capacity is forced down to a few tiles must equal the default render bitwise-close (<= 2e-7: the two
    forms order their f16-split sums differently)."""
    from text2nerf_amd import _lib, tensorf as tf
    lib = _lib.load()
    p = synth.make_field_params(11, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"])
    f = make_field(p, TINY["grid"], TINY["aabb"], TINY["near_far"])
    f.materialize_weights = False
    rays = torch.from_numpy(synth.frame_rays_np(96, 128, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)))).to(dev())
    with torch.no_grad():
        ref = f(rays)[0]
        n_app = f.stats()["appearance"]
        R, N = rays.shape[0], f.nSamples
        # a workspace with worst-case lists but room for only ~1/8 of the rows the frame needs: the rest takes k_shade_coop
        full = int(lib.t2n_render_workspace_bytes(R, N))
        f.workspace_bytes_override = full - (R * 32 + 1024) * 128 + (n_app // 8 // 128 * 128 + 128) * 128
        got = f(rays)[0]
        f.workspace_bytes_override = None
        tf._WORKSPACE.clear()
    assert n_app > 20000
    assert float((got - ref).abs().max()) <= 2e-7


def test_nan_density_feature_renders_as_empty_space(tiny_params):
    """ADVICE r5, deliberate deviation (csrc/t2n_device.h exp_finite): the marchers clamp the activation's argument with a raw v_max_f32,
    which returns the other operand for a NaN — a NaN density feature becomes sigma = 0 / alpha = 0 where the reference propagates NaN
    into the ray. Pinned here so that the behaviour is a decision: a field with one NaN density texel renders FINITE, and the rays that
    do not touch the texel's footprint are bit-identical to the clean field's."""
    r = rays().to(dev())
    clean = make_field(tiny_params, TINY["grid"], TINY["aabb"], TINY["near_far"])
    bad = {k: v.copy() for k, v in tiny_params.items()}
    pl = bad["density_plane.0"]
    pl[0, :, pl.shape[2] // 2, pl.shape[3] // 2] = np.nan
    f = make_field(bad, TINY["grid"], TINY["aabb"], TINY["near_far"])
    with torch.no_grad():
        a = clean(r, white_bg=True, is_train=False, N_samples=-1)
        b = f(r, white_bg=True, is_train=False, N_samples=-1)
    assert bool(torch.isfinite(b[0]).all()) and bool(torch.isfinite(b[1]).all())
    same = (a[0] == b[0]).all(dim=1)
    assert 0 < int(same.sum()) < r.shape[0] or bool(same.all())     # (rays through the texel's column differ, the rest are untouched)
