"""Render three C2 frames (profiling target)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import build_field
from text2nerf_amd import synth
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
field.materialize_weights = False
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
with torch.no_grad():
    for _ in range(3): field(rays)
torch.cuda.synchronize()
