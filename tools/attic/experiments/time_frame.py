import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
field.materialize_weights = False
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
for coh in (800, 0):
  field.frame_width = coh
  with torch.no_grad():
    for _ in range(2): field(rays)
    field.timing(True); field.read_timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): field(rays)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    tm = field.read_timing(True); field.timing(False)
  print(coh, {k: round(v[0]/5,3) for k,v in tm.items()}); print(f"frame {dt*1e3:.2f} ms  march {tm['march'][0]/5:.2f} ms shade {tm['shade'][0]/5:.2f} ms", flush=True)
