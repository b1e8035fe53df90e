"""One C2 frame rendered as n bands of image rows on n HIP streams (renderer._FramePipe) against the same frame in one call:
ms per frame and bitwise equality."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from text2nerf_amd import synth
from text2nerf_amd.renderer import _FramePipe
dev = torch.device("cuda:0")
field = bench.build_field(dev)[0]
field.materialize_weights, field.frame_width = False, 800
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)

def whole():
    return field(rays, white_bg=True, is_train=False, N_samples=-1)[:2]

def banded(n, splits):
    pipe = _FramePipe(dev, n)
    outs = []
    for k in range(len(splits) - 1):
        with torch.cuda.stream(pipe.next()):
            r = rays[splits[k] * 800: splits[k + 1] * 800]
            outs.append(field(r, white_bg=True, is_train=False, N_samples=-1)[:2])
    pipe.hand_over([t for o in outs for t in o])
    return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])

def timeit(fn, n=20):
    with torch.no_grad():
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    ref = whole()
for rep in range(2):
    print("whole frame: %.3f ms" % timeit(whole), flush=True)
    for n, splits in ((2, [0, 400, 800]), (3, [0, 272, 536, 800]), (4, [0, 200, 400, 600, 800]), (3, [0, 136, 272, 408, 536, 672, 800]), (2, [0, 200, 400, 600, 800])):
        with torch.no_grad():
            got = banded(n, splits)
        same = torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        print("%d streams, %d bands: %.3f ms, bitwise equal %s" % (n, len(splits) - 1, timeit(lambda: banded(n, splits)), same), flush=True)
