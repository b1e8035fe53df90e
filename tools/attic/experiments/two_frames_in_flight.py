"""Two frames in flight: frame k on stream k % 2 (each stream has its own scratch). Does the march of frame k + 1 (VALU-bound) overlap
the feature gather (L1 / addresser-bound) or the head (MFMA-bound) of frame k?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from text2nerf_amd import generate_rays
dev = torch.device("cuda", 0)
field, params, aabb = bench.build_field(dev)
field.materialize_weights = False
field.frame_width = 800
rays = generate_rays(800, 800, [800.0, 800.0, 400, 400], np.eye(4, dtype=np.float32), device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2)]
def run(n, ns):
    outs = []
    with torch.no_grad():
        for k in range(n):
            s = streams[k % ns]
            with torch.cuda.stream(s):
                outs.append(field(rays, white_bg=True, is_train=False, N_samples=-1)[0])
            if len(outs) > 4: outs.pop(0)
    return outs
for ns in (1, 2, len(streams)):
    for s in streams: s.wait_stream(torch.cuda.current_stream(dev))
    run(6, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = run(200, ns); torch.cuda.synchronize()
    print(f"{ns} stream(s): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per frame")
ref = None
with torch.no_grad():
    ref = field(rays, white_bg=True, is_train=False, N_samples=-1)[0]
torch.cuda.synchronize()
print("last frame equals a single-stream frame:", bool(torch.equal(o[-1], ref)))
