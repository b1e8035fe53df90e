import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth, _lib
dev = torch.device("cuda:0")
field, params, aabb = build_field(dev)
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
field.frame_width = 800; field.materialize_weights = False
lib = ctypes.CDLL(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else "/root/repo/text2nerf_amd/libt2n_hip.so")
out = (ctypes.c_ulonglong * 24)()
with torch.no_grad():
    field(rays)
    lib.t2n_debug_tile_counters(out, 1)
    rgb = field(rays)[0]
    print("rc", lib.t2n_debug_tile_counters(out, 1), float(rgb.mean()), list(out))
o = list(out)
print("waves", o[6], "steps", o[0], "clk/step: coords+reduce %.0f  table step %.0f  table+alpha %.0f | loop clk/wave %.0f  epilogue clk/wave %.0f  max wave %d" % (
    o[1]/o[0], o[2]/o[0], o[3]/o[0], o[4]/o[6], o[5]/o[6], o[7]))
print("span histogram (0 = out of range):", o[8:24])
