"""Wall time per view of render_views (device-side ray generation + render) and evaluation_frames (+ post-processing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth
from text2nerf_amd.renderer import render_views, evaluation_frames
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
H = W = 800
poses = [synth.look_pose(0.02 * k, -0.01 * k, (0.02 * k, 0.0, 0.0)) for k in range(20)]
intr = [float(W), float(W), W // 2, H // 2]
for name, fn in (("render_views", lambda: render_views(field, poses, intr, H, W)),
                 ("evaluation_frames", lambda: evaluation_frames(field, poses, intr, H, W, [0.5, 8.0]))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / len(poses) * 1e3:.2f} ms per 800x800 view", flush=True)
