#!/usr/bin/env python3
"""Golden vectors for the coarse-to-fine / occupancy-mask methods, produced by IMPORTING the reference
on CPU (see make_golden.py for the stubbing). Writes tests/golden/grid_ops.npz.

    python tests/golden/make_golden_grid.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, build_ref, quiet, tiny_rays  # noqa: E402  (also seeds sys.path / module stubs)
from text2nerf_amd import synth  # noqa: E402
from models.tensorBase import AlphaGridMask  # noqa: E402

DENSE_GRID = (12, 10, 14)
UP_RES = [30, 26, 22]


def main():
    out = {}
    m, sd = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9)
    rays, _, _ = tiny_rays()
    g = np.random.Generator(np.random.PCG64(77))
    lo, hi = np.array(TINY["aabb"][0], np.float32), np.array(TINY["aabb"][1], np.float32)
    pts = (lo + (hi - lo) * g.uniform(-0.05, 1.05, (3000, 3))).astype(np.float32)
    out["pts"] = pts
    with torch.no_grad():
        # compute_alpha without / with an occupancy mask
        out["alpha_nomask"] = m.compute_alpha(torch.from_numpy(pts), length=0.37).numpy()
        vol = (np.random.Generator(np.random.PCG64(99)).uniform(0, 1, (9, 11, 13)) < 0.30).astype(np.float32)
        mask_aabb = torch.tensor([[-7.5, -5.5, -6.5], [7.5, 6.5, 6.0]])
        out["mask_volume"], out["mask_aabb"] = vol, mask_aabb.numpy()
        m.alphaMask = AlphaGridMask("cpu", mask_aabb, torch.from_numpy(vol))
        out["alpha_mask"] = m.compute_alpha(torch.from_numpy(pts), length=0.37).numpy()
        # a sparse mask so that the ray filter rejects a fair share of the rays
        vol2 = (np.random.Generator(np.random.PCG64(98)).uniform(0, 1, (9, 11, 13)) < 0.004).astype(np.float32)
        out["filter_volume"] = vol2
        m.alphaMask = AlphaGridMask("cpu", mask_aabb, torch.from_numpy(vol2))
        # filtering_rays(bbox_only=False)
        kept = quiet(m.filtering_rays, rays, torch.zeros(rays.shape[0], 3), N_samples=24, bbox_only=False)
        out["filter_rays"] = rays.numpy()
        keep_mask = np.zeros(rays.shape[0], bool)
        kr = kept[0].numpy()
        j = 0
        for i in range(rays.shape[0]):
            if j < kr.shape[0] and np.array_equal(rays[i].numpy(), kr[j]):
                keep_mask[i] = True
                j += 1
        assert j == kr.shape[0]
        out["filter_keep"] = keep_mask
        m.alphaMask = None

        # getDenseAlpha / updateAlphaMask on a field whose density lives in a sub-box (synth.concentrate_density)
        csd = synth.concentrate_density({k: v.copy() for k, v in sd.items()})
        m.load_state_dict({k: torch.from_numpy(v) for k, v in csd.items()}, strict=True)
        alpha, dense_xyz = m.getDenseAlpha(DENSE_GRID)
        out["dense_alpha"] = alpha.numpy()
        thres = float(np.float32(np.quantile(alpha.numpy(), 0.97)))   # keeps ~3 % of the nodes before dilation
        m.alphaMask_thres = thres
        new_aabb = quiet(m.updateAlphaMask, DENSE_GRID)
        out["mask_thres"] = np.array(thres, np.float32)
        out["upd_volume"] = m.alphaMask.alpha_volume[0, 0].numpy()
        out["upd_new_aabb"] = new_aabb.numpy()
        # render with the rebuilt mask
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["upd_rgb"], out["upd_depth"] = rgb.numpy(), depth.numpy()

        # shrink to a hand-picked box (the mask grid differs from the field grid -> corrected aabb branch)
        box = new_aabb.clone()
        quiet(m.shrink, box)
        out["shrink_box"] = box.numpy()
        out["shrink_aabb"] = m.aabb.numpy()
        out["shrink_grid"] = m.gridSize.numpy()
        out["shrink_step"] = np.array([m.stepSize.item(), m.nSamples], np.float64)
        out["shrink_dplane0"] = m.density_plane[0].numpy()
        out["shrink_aline2"] = m.app_line[2].numpy()
        m.alphaMask = None
        rgb, depth, zv, wt = m(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["shrink_rgb"], out["shrink_depth"] = rgb.numpy(), depth.numpy()

    # upsample_volume_grid on a fresh tiny field
    m2, _ = build_ref(11, TINY["grid"], TINY["aabb"], TINY["near_far"], density_scale=0.9)
    with torch.no_grad():
        quiet(m2.upsample_volume_grid, UP_RES)
        out["up_res"] = np.array(UP_RES)
        out["up_step"] = np.array([m2.stepSize.item(), m2.nSamples], np.float64)
        for i in range(3):
            out[f"up_density_plane{i}"] = m2.density_plane[i].numpy()
            out[f"up_density_line{i}"] = m2.density_line[i].numpy()
        out["up_app_plane1"] = m2.app_plane[1].numpy()
        out["up_app_line0"] = m2.app_line[0].numpy()
        rgb, depth, zv, wt = m2(rays, is_train=False, white_bg=True, ndc_ray=False, N_samples=-1)
        out["up_rgb"], out["up_depth"] = rgb.numpy(), depth.numpy()
    np.savez_compressed(os.path.join(HERE, "grid_ops.npz"), **out)
    print("grid_ops: kept voxels", out["upd_volume"].sum(), "of", out["upd_volume"].size, "new aabb", out["upd_new_aabb"],
          "filter keep", keep_mask.sum(), "/", keep_mask.size, "shrink grid", out["shrink_grid"],
          "alpha>0.01 frac", (out["alpha_nomask"] > 0.01).mean())


if __name__ == "__main__":
    main()
