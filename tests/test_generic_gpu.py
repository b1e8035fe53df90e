"""The general-shape path (csrc/t2n_generic.hip: field / head shapes beyond the tuned kernels) beyond the tiny goldens of tests/test_shapes.py: a
128^3-class wide field rendered through the reference's call (OctreeRender_trilinear_fast, chunked, rays on the host) against the PyTorch
oracle on the same parameters — eval and train mode, white / black background, the 5-tuple's shapes — and its gradients against the
oracle's autograd."""
import numpy as np
import pytest
import torch

from oracle import oracle_torch as O
from tests.test_hip_parity import DEPTH_ATOL, RGB_ATOL, W_ATOL, W_RTOL, close, dev
from text2nerf_amd import synth

pytestmark = pytest.mark.gpu

GRID, AABB, NF = [41, 37, 33], [[-3.0, -2.5, -2.0], [3.0, 2.5, 4.0]], [0.3, 7.0]
KW = dict(density_n_comp=[24, 20, 32], appearance_n_comp=[64, 72, 56], app_dim=27, shadingMode="MLP_Fea_noview", fea_pe=6, featureC=160,
          view_pe=6, pos_pe=6)


def _build(seed=3, density_shift=-10):
    from text2nerf_amd import TensorVMSplit
    params = synth.make_field_params(seed, GRID, density_n_comp=KW["density_n_comp"], app_n_comp=KW["appearance_n_comp"], app_dim=27,
                                     feature_c=KW["featureC"], fea_pe=6, shading_mode="MLP_Fea_noview", density_scale=0.8, aabb=AABB)
    m = TensorVMSplit(torch.tensor(AABB), GRID, dev(), near_far=NF, alphaMask_thres=1e-4, density_shift=density_shift, distance_scale=25,
                      step_ratio=1.0, fea2denseAct="softplus", **KW)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
    cfg = O.FieldConfig(aabb=AABB, grid_size=GRID, near_far=NF, shading_mode="MLP_Fea_noview", fea_pe=6, density_shift=float(density_shift))
    return m, params, cfg


def test_wide_field_through_the_reference_call_vs_oracle():
    from text2nerf_amd import OctreeRender_trilinear_fast
    m, params, cfg = _build()
    assert m._is_general()
    rays = torch.from_numpy(synth.frame_rays_np(40, 56, c2w=synth.look_pose(0.2, -0.1, (0.1, 0.2, -2.5))))       # 2240 rays, on the host
    P = O.params_from_numpy(params)
    for white in (True, False):
        with torch.no_grad():
            rgb, none, depth, w, z = OctreeRender_trilinear_fast(rays, m, chunk=512, N_samples=-1, white_bg=white, is_train=False, device=dev())
            o_rgb, o_depth, o_z, o_w = O.forward(cfg, P, rays, white_bg=white)
        assert none is None and w.shape == (rays.shape[0], m.nSamples) == tuple(o_w.shape)
        close(z, o_z.numpy(), atol=0)
        close(w, o_w.numpy(), atol=W_ATOL, rtol=W_RTOL)
        close(rgb, o_rgb.numpy(), atol=RGB_ATOL)
        close(depth, o_depth.numpy(), atol=DEPTH_ATOL)
    st = m.stats()
    assert st["evaluated"] > 0 and st["appearance"] > 0


def test_wide_field_train_gradients_vs_oracle_autograd():
    from tests.test_hip_parity import _grad_check
    m, params, cfg = _build(seed=5)
    rays = torch.from_numpy(synth.frame_rays_np(12, 16, c2w=synth.look_pose(0.2, -0.1, (0.1, 0.2, -2.5))))
    g = np.random.Generator(np.random.PCG64(9))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    torch.manual_seed(7)
    jit = torch.rand(rays.shape[0], 1)
    torch.manual_seed(7)
    out = m(rays, is_train=True, white_bg=True, N_samples=48)
    ((out[0] * ca.to(dev())).sum() + 0.1 * out[1].sum() + (out[3] ** 2).sum()).backward()
    P = O.params_from_numpy(params, requires_grad=True)
    o = O.forward(cfg, P, rays, white_bg=True, is_train=True, n_samples=48, jitter=jit)
    close(out[0], o[0].detach().numpy(), atol=RGB_ATOL)
    ((o[0] * ca).sum() + 0.1 * o[1].sum() + (o[3] ** 2).sum()).backward()
    _grad_check(m, {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}, rel=5e-4)


def test_clamp_passes_the_gradient_on_the_closed_interval():
    """rgb_map.clamp(0, 1) passes its gradient at exactly 0.0 and 1.0 (torch.clamp's autograd; the tuned path does the same:
    k_bwd_march). A nearly empty field (density_shift -40: sigma ~ 1e-13, no weight reaches the appearance threshold) under a white
    background composites to exactly 1.0f on every ray, while d rgb / d sigma = -(1 - acc)' is not zero: the density gradients of the
    general-shape path against the oracle's autograd (the upstream gradient scaled so that they are O(1))."""
    from tests.test_hip_parity import _grad_check
    m, params, cfg = _build(seed=5, density_shift=-40)
    rays = torch.from_numpy(synth.frame_rays_np(12, 16, c2w=synth.look_pose(0.2, -0.1, (0.1, 0.2, -2.5))))
    g = np.random.Generator(np.random.PCG64(9))
    ca = torch.from_numpy((g.uniform(0.5, 1.5, (rays.shape[0], 3)) * 1e14).astype(np.float32))
    torch.manual_seed(7)
    jit = torch.rand(rays.shape[0], 1)
    torch.manual_seed(7)
    out = m(rays, is_train=True, white_bg=True, N_samples=48)
    assert bool((out[0] == 1.0).all())                      # every pre-clamp colour sits ON the boundary
    (out[0] * ca.to(dev())).sum().backward()
    P = O.params_from_numpy(params, requires_grad=True)
    o = O.forward(cfg, P, rays, white_bg=True, is_train=True, n_samples=48, jitter=jit)
    (o[0] * ca).sum().backward()
    ref = {k: (v.grad if v.grad is not None else torch.zeros_like(v)).numpy() for k, v in P.items()}
    assert max(float(np.abs(ref[f"density_plane.{k}"]).max()) for k in range(3)) > 1e-3     # the reference does pass a gradient there
    _grad_check(m, ref, rel=5e-4)
