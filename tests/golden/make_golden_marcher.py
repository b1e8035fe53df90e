#!/usr/bin/env python3
"""Golden vectors for dda / ray_marcher (dataLoader/ray_utils.py:174-228), produced by IMPORTING the reference on CPU (see
make_golden.py for the stubbing). Writes tests/golden/marcher.npz.    python tests/golden/make_golden_marcher.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from dataLoader.ray_utils import dda, ray_marcher  # noqa: E402


def main():
    rays, _, _ = tiny_rays()
    bbox = torch.tensor(TINY["aabb"], dtype=torch.float32)
    out = {}
    tmin, tmax = dda(rays[:, :3], rays[:, 3:6], bbox)
    out["dda_tmin"], out["dda_tmax"] = tmin.numpy(), tmax.numpy()
    for tag, kw in (("lin", dict(N_samples=17, lindisp=False)), ("disp", dict(N_samples=9, lindisp=True))):
        xyz, ro, rd, z = ray_marcher(rays, perturb=0, bbox_3D=bbox, **kw)
        out[f"m_{tag}_xyz"], out[f"m_{tag}_z"] = xyz.numpy(), z.numpy()
    r8 = torch.cat([rays, torch.full((rays.shape[0], 1), 0.5), torch.full((rays.shape[0], 1), 8.0)], 1)
    xyz, ro, rd, z = ray_marcher(r8, N_samples=12, lindisp=False, perturb=0, bbox_3D=None)
    out["m_nf_xyz"], out["m_nf_z"] = xyz.numpy(), z.numpy()
    torch.manual_seed(11)
    xyz, ro, rd, z = ray_marcher(rays, N_samples=10, lindisp=False, perturb=1.0, bbox_3D=bbox)
    torch.manual_seed(11)
    out["m_pert_u"] = torch.rand(z.shape).numpy()
    out["m_pert_z"], out["m_pert_xyz"] = z.numpy(), xyz.numpy()
    np.savez_compressed(os.path.join(HERE, "marcher.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
