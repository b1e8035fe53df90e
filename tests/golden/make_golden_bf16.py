#!/usr/bin/env python3
"""Golden vectors G11: the reference run with its 12 plane / line factor tensors rounded to bf16 (the
tolerance basis of BASELINE.json configs[4], "bf16 factor tensors"), produced by IMPORTING the reference on CPU (see
make_golden.py for the stubbing). basis_mat and the MLP stay fp32. Writes tests/golden/bf16.npz.

    python tests/golden/make_golden_bf16.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, build_ref, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)

FACTOR_KEYS = ("density_plane", "density_line", "app_plane", "app_line")


def main():
    out = {}
    m, sd = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith(FACTOR_KEYS):
                p.copy_(p.to(torch.bfloat16).to(torch.float32))     # round-to-nearest-even, like the upload kernel
    rays, _, _ = tiny_rays()
    with torch.no_grad():
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["g11_eval_rgb"], out["g11_eval_depth"], out["g11_eval_w"] = rgb.numpy(), depth.numpy(), wt.numpy()
        torch.manual_seed(123)
        rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=40)
        out["g11_train_rgb"], out["g11_train_depth"], out["g11_train_w"], out["g11_train_z"] = (
            rgb.numpy(), depth.numpy(), wt.numpy(), zv.numpy())
    # gradients at the rounded point (scalar of golden G8's form) w.r.t. all parameters
    g = np.random.Generator(np.random.PCG64(99))
    ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
    cb = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0],)).astype(np.float32))
    torch.manual_seed(321)
    rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=40)
    cw = torch.from_numpy(g.uniform(-1, 1, tuple(wt.shape)).astype(np.float32))
    out["g11_ca"], out["g11_cb"], out["g11_cw"] = ca.numpy(), cb.numpy(), cw.numpy()
    ((rgb * ca).sum() + (depth * cb).sum() + (wt * cw).sum()).backward()
    for name, p in m.named_parameters():
        out["g11_grad_" + name] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    np.savez_compressed(os.path.join(HERE, "bf16.npz"), **out)
    print("wrote bf16.npz:", {k: v.shape for k, v in out.items() if not k.startswith("g11_grad")})


if __name__ == "__main__":
    main()
