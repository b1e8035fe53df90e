#!/usr/bin/env python3
"""Golden vectors for the stacked-coefficient TensorVM field (models/tensoRF.py:4-136), produced by IMPORTING the reference
on CPU (see make_golden.py for the stubbing). The reference class only constructs with scalar n_comp (upstream TensoRF
style). Parameters are rebuilt in the tests from text2nerf_amd.synth (cubic 20^3 grid): plane_coef[k] = [app_plane.k |
density_plane.k]. Writes tests/golden/vm.npz.    python tests/golden/make_golden_vm.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, quiet, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from models.tensoRF import TensorVM  # noqa: E402
from text2nerf_amd import synth  # noqa: E402

GRID = [20, 20, 20]


def vm_state(sd):
    out = {k: v for k, v in sd.items() if not k.startswith(("density_", "app_"))}
    out["plane_coef"] = np.concatenate([np.concatenate([sd[f"app_plane.{k}"], sd[f"density_plane.{k}"]], 1) for k in range(3)], 0)
    out["line_coef"] = np.concatenate([np.concatenate([sd[f"app_line.{k}"], sd[f"density_line.{k}"]], 1) for k in range(3)], 0)
    return out


def main():
    sd = vm_state(synth.make_field_params(13, GRID, density_scale=0.9, aabb=TINY["aabb"]))
    m = quiet(TensorVM, torch.tensor(TINY["aabb"]), GRID, "cpu", density_n_comp=16, appearance_n_comp=48, app_dim=27,
              near_far=TINY["near_far"], shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4, density_shift=-10, distance_scale=25,
              pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    rays, _, _ = tiny_rays()
    out = {}
    with torch.no_grad():
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["vm_eval_rgb"], out["vm_eval_depth"], out["vm_eval_w"], out["vm_eval_z"] = rgb.numpy(), depth.numpy(), wt.numpy(), zv.numpy()
    g = np.random.Generator(np.random.PCG64(5))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    cb = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0],)).astype(np.float32))
    torch.manual_seed(77)
    rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=30)
    cw = torch.from_numpy(g.uniform(-1, 1, tuple(wt.shape)).astype(np.float32))
    out["vm_train_rgb"], out["vm_train_z"] = rgb.detach().numpy(), zv.numpy()
    out["vm_ca"], out["vm_cb"], out["vm_cw"] = ca.numpy(), cb.numpy(), cw.numpy()
    ((rgb * ca).sum() + (depth * cb).sum() + (wt * cw).sum()).backward()
    for name, p in m.named_parameters():
        out["vm_grad_" + name] = p.grad.numpy()
    out["vm_state_keys"] = np.array(sorted(m.state_dict().keys()))
    # upsample_volume_grid (models/tensoRF.py:126-136): F.interpolate of the stacked tensors (bilinear, align_corners=True; the
    # planes by scale factor, the lines to an explicit size). The reference then calls an undefined self.compute_stepSize (an
    # AttributeError upstream); the resized parameters are in place by then, and so is what the method evidently means: update_stepSize
    try:
        quiet(m.upsample_volume_grid, [27, 27, 27])
    except AttributeError:
        pass
    out["vm_up27_plane"], out["vm_up27_line"] = m.plane_coef.detach().numpy(), m.line_coef.detach().numpy()
    m.update_stepSize([27, 27, 27])
    with torch.no_grad():
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
    out["vm_up27_rgb"], out["vm_up27_depth"], out["vm_up27_nsamples"] = rgb.numpy(), depth.numpy(), np.array(m.nSamples)
    np.savez_compressed(os.path.join(HERE, "vm.npz"), **out)
    print({k: v.shape for k, v in out.items() if not k.startswith("vm_grad")}, [k for k in out if k.startswith("vm_grad")])


if __name__ == "__main__":
    main()
