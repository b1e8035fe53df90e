"""Timing of the f-3 image-space kernels at the driver's frame size (512x512; SceneGen renders 512^2 support views) on the
MI355X, next to the oracle's numpy restatement on the host. Prints one JSON line. `python tools/bench_image_ops.py [H W]`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from text2nerf_amd import synth
from text2nerf_amd.warp import bilinear_splat_warping_multiview, dibr_filter_mask2, sparse_bilateral_filtering

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
from make_golden_warp_cases import pose44, warp_poses

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)
dev = torch.device("cuda:0")
rgb, depth = synth.rgbd_frame(7, H, W, n_boxes=12, holes=40)
rgb_t, depth_t = torch.from_numpy(rgb).to(dev), torch.from_numpy(depth).to(dev)
poses = [pose44(p) for p in warp_poses()]
frames = [synth.rgbd_frame(60 + v, H, W, n_boxes=12) for v in range(3)]
frames_t = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in frames]
intr = [float(max(H, W)), float(max(H, W)), W // 2, H // 2]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


filt_ms = timed(lambda: sparse_bilateral_filtering(depth_t, rgb_t, filter_size=[7, 5, 5, 3, 3], depth_threshold=0.02, num_iter=5))
warp_ms = timed(lambda: bilinear_splat_warping_multiview([f[0] for f in frames_t], [f[1] for f in frames_t], np.stack(poses[:3]),
                                                         poses[3], H, W, intr))
wm, wi, wd = bilinear_splat_warping_multiview([f[0] for f in frames_t], [f[1] for f in frames_t], np.stack(poses[:3]), poses[3], H, W, intr)
fill_ms = timed(lambda: dibr_filter_mask2(wi, wm, output_depth=wd))
out = {"frame": [H, W], "sparse_bilateral_filtering_ms": filt_ms, "warp_3_views_ms": warp_ms, "dibr_filter_mask2_ms": fill_ms}
if "--cpu" in sys.argv:
    from oracle import oracle_warp as OW
    t0 = time.perf_counter(); OW.sparse_bilateral_filtering(depth, rgb, [7, 5, 5, 3, 3], 0.02, 5); out["oracle_filter_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    OW.bilinear_splat_warping_multiview([f[0] for f in frames], [f[1] for f in frames], np.stack(poses[:3]), poses[3], H, W, intr)
    out["oracle_warp_s"] = time.perf_counter() - t0
print(json.dumps(out))
