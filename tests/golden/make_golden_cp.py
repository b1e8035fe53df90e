#!/usr/bin/env python3
"""Golden vectors for the CP (line-only) field TensorCP (models/tensoRF.py:306-434), produced by IMPORTING the reference on
CPU (see make_golden.py for the stubbing). Lines are rebuilt in the tests from text2nerf_amd.synth (the VM-split lines of
seed 17, anisotropic grid). Writes tests/golden/cp.npz.    python tests/golden/make_golden_cp.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, quiet, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from models.tensoRF import TensorCP  # noqa: E402
from text2nerf_amd import synth  # noqa: E402


def cp_state(sd):
    out = {k: v for k, v in sd.items() if k.startswith(("density_line", "app_line", "renderModule"))}
    for k in range(3):                       # CP lines carry the whole signal: larger amplitude than the VM init
        out[f"density_line.{k}"] = (out[f"density_line.{k}"] * 1.5).astype(np.float32)
        out[f"app_line.{k}"] = (out[f"app_line.{k}"] * 3.0).astype(np.float32)
    g = np.random.Generator(np.random.PCG64(23))
    out["basis_mat.weight"] = g.uniform(-0.14, 0.14, (27, 48)).astype(np.float32)
    return out


def main():
    sd = cp_state(synth.make_field_params(17, TINY["grid"], density_scale=0.9, aabb=TINY["aabb"]))
    m = quiet(TensorCP, torch.tensor(TINY["aabb"]), TINY["grid"], "cpu", density_n_comp=[16, 16, 16], appearance_n_comp=[48, 48, 48],
              app_dim=27, near_far=TINY["near_far"], shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4, density_shift=-10,
              distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rays, _, _ = tiny_rays()
    out = {}
    with torch.no_grad():
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["cp_eval_rgb"], out["cp_eval_depth"], out["cp_eval_w"] = rgb.numpy(), depth.numpy(), wt.numpy()
    g = np.random.Generator(np.random.PCG64(6))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    cb = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0],)).astype(np.float32))
    torch.manual_seed(78)
    rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=30)
    cw = torch.from_numpy(g.uniform(-1, 1, tuple(wt.shape)).astype(np.float32))
    out["cp_train_rgb"], out["cp_train_z"] = rgb.detach().numpy(), zv.numpy()
    out["cp_ca"], out["cp_cb"], out["cp_cw"] = ca.numpy(), cb.numpy(), cw.numpy()
    ((rgb * ca).sum() + (depth * cb).sum() + (wt * cw).sum()).backward()
    for name, p in m.named_parameters():
        out["cp_grad_" + name] = p.grad.numpy()
    out["cp_state_keys"] = np.array(sorted(m.state_dict().keys()))
    # shrink (models/tensoRF.py:387-416): the reference dereferences self.alphaMask, so it gets one (all ones, a grid that differs
    # from the field's: the corrected-aabb branch); the render after it runs without the mask
    from models.tensoRF import AlphaGridMask
    m.alphaMask = AlphaGridMask("cpu", m.aabb, torch.ones(8, 8, 8))
    new_aabb = torch.tensor([[-5.1, -3.3, -4.2], [4.4, 5.2, 3.9]])
    quiet(m.shrink, new_aabb)
    m.alphaMask = None
    out["cp_shrink_in"], out["cp_shrink_aabb"] = new_aabb.numpy(), m.aabb.numpy()
    out["cp_shrink_grid"], out["cp_shrink_nsamples"] = m.gridSize.numpy(), np.array(m.nSamples)
    for k in range(3):
        out[f"cp_shrink_density_line.{k}"], out[f"cp_shrink_app_line.{k}"] = m.density_line[k].detach().numpy(), m.app_line[k].detach().numpy()
    with torch.no_grad():
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
    out["cp_shrink_rgb"], out["cp_shrink_depth"] = rgb.numpy(), depth.numpy()
    np.savez_compressed(os.path.join(HERE, "cp.npz"), **out)
    print({k: v.shape for k, v in out.items() if not k.startswith("cp_grad")}, "napp", int((wt > 1e-4).sum()))


if __name__ == "__main__":
    main()
