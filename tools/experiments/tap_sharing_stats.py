"""How often consecutive appearance-list entries (consecutive steps of one ray) share the texel cell of a plane / line of the C2 frame:
the fraction of entries that are 'heads' (their cell differs from the previous entry's, or they start a ray's run)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from text2nerf_amd import synth

dev = torch.device("cuda", 0)
f = bench.build_field(dev, scene="S1-soft", seed=0)[0]
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
G = 300
tot = 0
heads = {k: 0 for k in ("p_xy", "p_xz", "p_yz", "l_x", "l_y", "l_z", "ray")}
aabb = f.aabb.to(dev)
with torch.no_grad():
    for c0 in range(0, rays.shape[0], 80000):
        r = rays[c0:c0 + 80000]
        rgb, depth, z, w = f(r)
        m = w > float(f.rayMarch_weight_thres)
        pts = r[:, None, :3] + r[:, None, 3:6] * z[..., None]
        xn = (pts - aabb[0]) / (aabb[1] - aabb[0])          # [0, 1]
        cell = torch.floor(xn * (G - 1)).clamp(0, G - 2).to(torch.int32)
        ridx = torch.arange(r.shape[0], device=dev)[:, None].expand_as(m)
        cm = cell[m]                      # entries in (ray, step) order
        rm = ridx[m]
        n = cm.shape[0]
        tot += n
        new_ray = torch.ones(n, dtype=torch.bool, device=dev)
        new_ray[1:] = rm[1:] != rm[:-1]
        d = torch.ones(n, 3, dtype=torch.bool, device=dev)
        d[1:] = cm[1:] != cm[:-1]
        heads["ray"] += int(new_ray.sum())
        heads["p_xy"] += int((new_ray | d[:, 0] | d[:, 1]).sum())
        heads["p_xz"] += int((new_ray | d[:, 0] | d[:, 2]).sum())
        heads["p_yz"] += int((new_ray | d[:, 1] | d[:, 2]).sum())
        heads["l_x"] += int((new_ray | d[:, 0]).sum())
        heads["l_y"] += int((new_ray | d[:, 1]).sum())
        heads["l_z"] += int((new_ray | d[:, 2]).sum())
print("appearance entries", tot)
for k, v in heads.items():
    print(f"{k}: heads {v}  fraction {v / tot:.3f}")
pl = (heads["p_xy"] + heads["p_xz"] + heads["p_yz"]) * 4 + (heads["l_x"] + heads["l_y"] + heads["l_z"]) * 2
print("tap loads with run sharing / tap loads today: %.3f" % (pl / (tot * 18)))
