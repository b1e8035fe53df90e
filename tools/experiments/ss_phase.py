"""Per-phase cycles of k_mlp_ss (wave 0 of every workgroup; needs a -DT2N_PHASE_TIMING build selected with T2N_LIB)."""
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
buf = (ctypes.c_ulonglong * 128)()
with torch.no_grad():
    for _ in range(2): field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_ss_phase_read(buf, 1)
    field(rays)
    torch.cuda.synchronize()
    L.t2n_debug_ss_phase_read(buf, 0)
v = np.array(list(buf), dtype=np.float64).reshape(8, 16)
names = ["L0 slots", "h0 conv chunk 0", "L1 (4 chunks)", "h1 conv + L2", "output", "L0 barrier wait", "L0 ring store + load issue", "-",
         "-", "-", "-", "-", "-", "-", "-", "loop top"]
rounds = field.stats()["appearance"] / 256.0
print(f"{'cycles per round, by wave':36s}" + "".join(f"{w:8d}" for w in range(8)))
for i, n in enumerate(names):
    if v[:, i].any(): print(f"{n:36s}" + "".join(f"{x / rounds:8.0f}" for x in v[:, i]))
print(f"{'total':36s}" + "".join(f"{x / rounds:8.0f}" for x in v.sum(1)))
