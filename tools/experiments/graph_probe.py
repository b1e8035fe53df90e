"""Can a whole frame (t2n_render_forward: ~14 launches on one stream, no host wait since round 5) be captured into a hipGraph through
torch.cuda.graph and replayed? Probe for the train-step work of the next round: the same mechanism, without the
device-side launch bounds the backward still lacks. Prints eager vs replay time per frame and whether the replayed frame equals the
eager one bit for bit."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import synth  # noqa: E402

dev = torch.device("cuda:0")
field = bench.build_field(dev)[0]
field.materialize_weights = False
field.frame_width = 800
field.collect_stats = True
rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
with torch.no_grad():
    for _ in range(5):
        ref = field(rays)[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ref = field(rays)[0]
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50 * 1e3
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            for _ in range(2):
                field(rays)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g, capture_error_mode=os.environ.get("T2N_CAPTURE_MODE", "relaxed")):
            out = field(rays)[0]
        torch.cuda.synchronize()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        rep = (time.perf_counter() - t0) / 50 * 1e3
        print(f"eager {eager:.4f} ms per frame, graph replay {rep:.4f} ms per frame, bitwise equal: {bool(torch.equal(out, ref))}")
    except Exception as e:  # noqa: BLE001
        print("capture / replay failed:", repr(e)[:600])
