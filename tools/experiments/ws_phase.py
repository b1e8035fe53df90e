"""Per-phase cycles of k_mlp_ws (wave 0 of every workgroup; needs a -DT2N_PHASE_TIMING build selected with T2N_LIB)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth, _lib
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
field.materialize_weights = False
field.frame_width = 800
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
with torch.no_grad():
    for _ in range(2): field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_ws_phase_read(buf, 1)
    field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_ws_phase_read(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["octet 0 + barrier", "L0 steps 0-4", "L0 step 5", "h0 store + barrier", "L1 + barrier", "h1 store + barrier", "L2 + output", "end barrier",
         "-", "-", "-", "-", "-", "-", "-", "loop top"]
tot = v.sum()
groups = field.stats()["appearance"] / 128.0
for n, x in zip(names, v):
    if x: print(f"{n:36s} {x/1e6:10.1f} Mcyc  {100*x/tot:5.1f} %   {x/max(groups,1):8.0f} cyc/group")
print("total cyc/group", tot / groups, "groups", groups)
