#!/usr/bin/env python3
"""Golden vectors for field / head shapes other than the driver's (models/tensoRF.py:144-160, models/tensorBase.py:88-109: the
reference is generic in n_lamb_sigma, n_lamb_sh, data_dim_color, fea_pe, featureC), produced by IMPORTING the reference on CPU:
eval render, train render and the autograd gradients of a scalar w.r.t. every parameter. Writes tests/golden/shapes.npz.
    python tests/golden/make_golden_shapes.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, quiet, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from make_golden_shapes_cases import SHAPES  # noqa: E402
from models.tensoRF import TensorVMSplit  # noqa: E402
from text2nerf_amd import synth  # noqa: E402


def main():
    rays, _, _ = tiny_rays()
    out = {}
    for tag, kw in SHAPES.items():
        sd = synth.make_field_params(41, TINY["grid"], density_n_comp=kw["density_n_comp"], app_n_comp=kw["appearance_n_comp"],
                                     app_dim=kw["app_dim"], feature_c=kw["featureC"], fea_pe=kw["fea_pe"],
                                     shading_mode=kw["shadingMode"], density_scale=0.9, aabb=TINY["aabb"], view_pe=kw["view_pe"],
                                     pos_pe=kw["pos_pe"])
        m = quiet(TensorVMSplit, torch.tensor(TINY["aabb"]), TINY["grid"], "cpu", near_far=TINY["near_far"], alphaMask_thres=1e-4,
                  density_shift=-10, distance_scale=25, step_ratio=1.0, fea2denseAct="softplus", **kw)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
            out[f"{tag}_eval_rgb"], out[f"{tag}_eval_depth"], out[f"{tag}_eval_acc"] = rgb.numpy(), depth.numpy(), wt.sum(-1).numpy()
        g = np.random.Generator(np.random.PCG64(8))
        ca = torch.from_numpy(g.uniform(-1, 1, (rays.shape[0], 3)).astype(np.float32))
        torch.manual_seed(55)
        rgb, depth, zv, wt = m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=36)
        out[f"{tag}_train_rgb"], out[f"{tag}_ca"] = rgb.detach().numpy(), ca.numpy()
        ((rgb * ca).sum() + 0.1 * depth.sum() + (wt ** 2).sum()).backward()
        for name, p in m.named_parameters():
            out[f"{tag}_grad_" + name] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "shapes.npz"), **out)
    print({k: v.shape for k, v in out.items() if "grad" not in k})


if __name__ == "__main__":
    main()
