"""Render the C2 frame (S1-soft, 300^3, 800x800) several times and report where two renders differ (head-kernel hazard hunting)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from text2nerf_amd import synth
from tests.test_hip_parity import make_field

aabb = [[-8.0] * 3, [8.0] * 3]
f = make_field(synth.make_field_params(0, [300] * 3, scene="S1-soft", aabb=aabb), [300] * 3, aabb, [0.5, 8.0])
f.materialize_weights = False
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).cuda()
with torch.no_grad():
    ref, dref, _, _ = f(rays)
    for it in range(6):
        rgb, depth, _, _ = f(rays)
        d = (rgb - ref).abs().max(-1).values
        bad = torch.nonzero(d > 0).flatten()
        print(f"run {it}: {bad.numel()} rays differ, max |d rgb| {float(d.max()):.3e}, depth differs on {int((depth != dref).sum())}; stats {f.stats()}",
              "first rays:", bad[:12].tolist(), flush=True)
