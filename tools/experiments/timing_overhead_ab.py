"""What the per-kernel HIP events cost a frame: C2 frame time with no events, events around the head only, events around every kernel
group (alternating blocks, medians)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from text2nerf_amd import synth
dev = torch.device("cuda:0")
field = bench.build_field(dev)[0]
field.materialize_weights, field.frame_width = False, 800
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
def block(n=30):
    with torch.no_grad():
        for _ in range(3): field(rays, white_bg=True, is_train=False, N_samples=-1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): field(rays, white_bg=True, is_train=False, N_samples=-1)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
res = {"off": [], "head": [], "all": []}
for rep in range(5):
    field.timing(False); res["off"].append(block())
    field.timing(True, kernels=("shade",)); res["head"].append(block()); field.read_timing(reset=True)
    field.timing(True); res["all"].append(block()); field.read_timing(reset=True)
for k, v in res.items():
    print(k, [round(x, 3) for x in v], "median %.3f" % sorted(v)[2])
