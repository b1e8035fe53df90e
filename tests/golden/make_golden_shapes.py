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


MARGIN = 5e-6   # no ReLU input of the train pass closer to zero than this


def build(tag, kw, seed):
    sd = synth.make_field_params(seed, TINY["grid"], density_n_comp=kw["density_n_comp"], app_n_comp=kw["appearance_n_comp"],
                                 app_dim=kw["app_dim"], feature_c=kw["featureC"], fea_pe=kw["fea_pe"],
                                 shading_mode=kw["shadingMode"], density_scale=0.9, aabb=TINY["aabb"], view_pe=kw["view_pe"],
                                 pos_pe=kw["pos_pe"])
    m = quiet(TensorVMSplit, torch.tensor(TINY["aabb"]), TINY["grid"], "cpu", near_far=TINY["near_far"], alphaMask_thres=1e-4,
              density_shift=-10, distance_scale=25, step_ratio=1.0, fea2denseAct="softplus", **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m


def relu_margin(m, rays):
    """Smallest |x| any ReLU of the head sees in the train pass (seed 55, N = 36): the hidden layers of the MLP heads (hooks on the
    Linear modules in front of the in-place ReLUs), torch.relu of SHRender. A gradient comparison is only meaningful when no such
    input is within rounding of zero: an implementation whose products are ordered differently (split-f16 MFMA sums, a padded
    contraction) may land on the other side and that ONE sample's whole upstream gradient flips — ~1e-3 of a tensor's largest
    gradient on this 200-ray batch. The generator therefore walks seeds until the margin is MARGIN."""
    lo = [float("inf")]
    hooks = []
    if isinstance(m.renderModule, torch.nn.Module):
        for i in (0, 2):
            hooks.append(m.renderModule.mlp[i].register_forward_hook(lambda mod, inp, out: lo.__setitem__(0, min(lo[0], float(out.detach().abs().min())))))
    real_relu = torch.relu

    def spy(x):
        lo[0] = min(lo[0], float(x.detach().abs().min())) if x.numel() else lo[0]
        return real_relu(x)
    torch.relu = spy
    try:
        torch.manual_seed(55)
        with torch.no_grad():
            m(rays, is_train=True, white_bg=True, ndc_ray=False, N_samples=36)
    finally:
        torch.relu = real_relu
        for h in hooks:
            h.remove()
    return lo[0]


def main():
    rays, _, _ = tiny_rays()
    out = {}
    for tag, kw in SHAPES.items():
        seed = 41
        while True:
            m = build(tag, kw, seed)
            mg = relu_margin(m, rays)
            if mg >= MARGIN:
                break
            seed += 1
        print(tag, "seed", seed, "relu margin", mg)
        out[f"{tag}_seed"] = np.array(seed)
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
