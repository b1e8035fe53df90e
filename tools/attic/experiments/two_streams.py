"""Experiment: do the march (TA/VALU-bound) and shade (matrix-core) kernels of two half-frames overlap usefully when issued
on two HIP streams?  (each stream gets its own workspace)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bench import build_field
from text2nerf_amd import synth, tensorf as tf
dev = torch.device("cuda:0")
f, params, aabb = build_field(dev)
f.materialize_weights = False
f.collect_stats = False
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
_orig = tf.workspace
def ws_per_stream(device, nbytes):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    buf = tf._WORKSPACE.get(key)
    if buf is None or buf.numel() < nbytes:
        tf._WORKSPACE[key] = buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return buf
tf.workspace = ws_per_stream
def seq():
    with torch.no_grad():
        f(rays)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def par(nparts):
    parts = rays.chunk(nparts)
    with torch.no_grad():
        for i, p in enumerate(parts):
            with torch.cuda.stream(streams[i % 2]):
                f(p)
for name, fn in [("sequential", seq), ("2 streams x 2 parts", lambda: par(2)), ("2 streams x 4 parts", lambda: par(4)), ("2 streams x 8 parts", lambda: par(8))]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)/10*1e3:.2f} ms/frame", flush=True)
