#!/usr/bin/env python3
"""Golden vectors for the NDC branch (ndc_ray=True): TensorBase.sample_ray_ndc + the ndc lines of forward
(models/tensorBase.py:293-302,441-446) and ndc_rays_blender / ndc_rays (dataLoader/ray_utils.py:88-124), produced by IMPORTING
the reference on CPU. Writes tests/golden/ndc.npz.    python tests/golden/make_golden_ndc.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, build_ref, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from dataLoader.ray_utils import ndc_rays, ndc_rays_blender  # noqa: E402


def main():
    out = {}
    rays, _, _ = tiny_rays()
    rays = rays.clone()
    rays[:, 3:6] *= torch.linspace(0.6, 1.7, rays.shape[0])[:, None]      # non-unit directions: the |d| scaling matters
    out["ndc_rays_in"] = rays.numpy()
    for shading in ("MLP_Fea_noview", "SH"):
        m, sd = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9, shading=shading)
        tag = "mlp" if shading != "SH" else "sh"
        with torch.no_grad():
            rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=True, N_samples=-1)
            out[f"ndc_{tag}_eval_rgb"], out[f"ndc_{tag}_eval_depth"], out[f"ndc_{tag}_eval_w"], out[f"ndc_{tag}_eval_z"] = (
                rgb.numpy(), depth.numpy(), wt.numpy(), zv.numpy())
            torch.manual_seed(99)
            rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=True, N_samples=40)
            torch.manual_seed(99)
            out[f"ndc_{tag}_jit"] = torch.rand(1, 40).numpy()
            out[f"ndc_{tag}_train_rgb"], out[f"ndc_{tag}_train_depth"], out[f"ndc_{tag}_train_w"], out[f"ndc_{tag}_train_z"] = (
                rgb.numpy(), depth.numpy(), wt.numpy(), zv.numpy())
    g = np.random.Generator(np.random.PCG64(3))
    ro = torch.from_numpy(g.uniform(-1, 1, (300, 3)).astype(np.float32))
    rd = torch.from_numpy(g.uniform(-1, 1, (300, 3)).astype(np.float32))
    rd[:, 2] = -rd[:, 2].abs() - 0.2
    out["nr_o"], out["nr_d"] = ro.numpy(), rd.numpy()
    a, b = ndc_rays_blender(378, 504, 400.0, 1.0, ro, rd)
    out["nr_blender_o"], out["nr_blender_d"] = a.numpy(), b.numpy()
    a, b = ndc_rays(378, 504, 400.0, 1.0, ro, -rd)
    out["nr_cv_o"], out["nr_cv_d"] = a.numpy(), b.numpy()
    np.savez_compressed(os.path.join(HERE, "ndc.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print("eval weight sums:", float(out["ndc_mlp_eval_w"].sum()), "train:", float(out["ndc_mlp_train_w"].sum()))


if __name__ == "__main__":
    main()
