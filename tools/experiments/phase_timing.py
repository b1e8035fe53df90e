"""Per-phase wave cycles of k_shade_coop (debug build with -DT2N_PHASE_TIMING swapped in as libt2n_hip.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth, _lib
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
field.materialize_weights = False
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
L = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    for _ in range(2): field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_phase_read(buf, 1)
    field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_phase_read(buf, 0)
v = np.array(list(buf), dtype=np.float64)
names = ["gather", "basis", "barrier", "layer0", "h0 write", "layer1", "layer2+out", "-"]
tot = v.sum()
for n, x in zip(names, v): print(f"{n:12s} {x/1e6:10.1f} Mcyc  {100*x/tot:5.1f} %")
print("tiles", field.last_stats)
