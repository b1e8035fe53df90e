#!/usr/bin/env python3
"""Golden vectors for SURVEY.md 8 f-3 — the steps either side of the renderer in `render_warping_inapinting`
(text2nerf_main.py:102-141): `sparse_bilateral_filtering` (dataLoader/bilateral_filtering.py:5-35,138-228) and
`bilinear_splat_warping_multiview` (utils.py:83-119) with `Warper.forward_warp` (scripts/Warper.py:21-186) — produced by
IMPORTING the reference on CPU. cv2 / skimage / imageio are inert stubs (neither function touches them). Inputs are rebuilt
from seeds by text2nerf_amd.synth.rgbd_frame; only outputs are stored. Writes tests/golden/warp.npz.

    python tests/golden/make_golden_warp.py
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, REF)
for name in ["cv2", "imageio", "imageio.v2", "torchvision", "torchvision.transforms", "statsmodels", "statsmodels.api",
             "skimage", "skimage.io", "skimage.metrics", "skimage.measure", "lpips", "plyfile", "kornia", "configargparse"]:
    sys.modules.setdefault(name, MagicMock())

from text2nerf_amd import synth  # noqa: E402
from dataLoader.bilateral_filtering import sparse_bilateral_filtering  # noqa: E402
from scripts.Warper import Warper  # noqa: E402
import utils as ref_utils  # noqa: E402

from make_golden_warp_cases import FILTER_CASES, H, W, pose44, warp_poses  # noqa: E402


def main():
    out = {}
    for tag, c in FILTER_CASES.items():
        rgb, depth = synth.rgbd_frame(c["seed"], H, W, holes=c["holes"])
        photos, depths = sparse_bilateral_filtering(depth.copy(), rgb.copy(), filter_size=c["filter_size"], depth_threshold=0.02,
                                                    num_iter=c["num_iter"], HR=False, mask=None)
        out[f"filt_{tag}_depth"] = depths[-1]          # what the driver keeps (:119)
        out[f"filt_{tag}_photo"] = photos[-1]          # (:120)
        out[f"filt_{tag}_depth1"] = depths[1]          # state after one pass
    # DIBR: three source views -> one target
    poses = [pose44(p) for p in warp_poses()]
    intrinsic = [float(max(H, W)), float(max(H, W)), W // 2, H // 2]
    frames = [synth.rgbd_frame(31 + v, H, W) for v in range(3)]
    mask, img, dep = ref_utils.bilinear_splat_warping_multiview([f[0] for f in frames], [f[1] for f in frames],
                                                                np.stack(poses[:3]), poses[3], H, W, intrinsic, masks=None)
    out["warp_mask"], out["warp_image"], out["warp_depth"] = mask.astype(np.uint8), img, dep
    # one plain forward_warp with its flow, for the stage-level check
    K = np.eye(3, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intrinsic
    f2, m2, d2, flow = Warper().forward_warp((frames[1][0] * 255).astype(np.uint8), None, frames[1][1],
                                             np.linalg.inv(poses[1]), np.linalg.inv(poses[3]), K, None)
    out["fw_frame"], out["fw_mask"], out["fw_depth"], out["fw_flow"] = f2, m2, d2, flow
    # hole filling (utils.py:393-409, the use_filter_filling branch of text2nerf_main.py:134-135) on the merged warp with
    # extra random holes punched in (seeded), so that the raster-order dependence between fills is exercised
    g = np.random.Generator(np.random.PCG64(41))
    holes = g.uniform(0, 1, mask.shape) < 0.18
    hm, hi, hd = mask.copy(), img.copy(), dep.copy()
    hm[holes] = 0
    hi[holes] = 1.0
    hd[holes] = 0.0
    out["fill_in_mask"] = hm.astype(np.uint8)
    f_img, f_map, f_dep = ref_utils.dibr_filter_mask2(hi.copy(), hm.copy(), output_depth=hd.copy())
    out["fill_image"], out["fill_mask"], out["fill_depth"] = f_img, f_map.astype(np.uint8), f_dep
    print("holes", int((hm == 0).sum()), "filled", int((f_map != hm).sum()))
    np.savez_compressed(os.path.join(HERE, "warp.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})
    print("filled fraction:", float(mask.mean()), "filter changed px:",
          int((out["filt_a_depth"] != synth.rgbd_frame(21, H, W)[1]).sum()))


if __name__ == "__main__":
    main()
