"""Per-region cycle accounting of k_bwd_tile_accum (a -DTA_PROF build: tools/build_variant.sh taprof -DTA_PROF; T2N_LIB=.../libt2n_hip_taprof.so).
python tools/experiments/ta_prof_probe.py [rays ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from text2nerf_amd import _lib
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.t2n_debug_ta_prof
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
names = ["stage + zero", "barrier", "record tables", "gradient loads (forced wait)", "accumulate loop", "barrier", "flush"]
for rays in [int(x) for x in sys.argv[1:]] or [16384, 2048]:
    buf = (C.c_uint64 * 16)()
    r = bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=rays, step_kw=dict(fused=True, graph=False))
    torch.cuda.synchronize()
    fn(buf, 1)       # (warm-up + timed blocks together: shares, not absolute times)
    tot = sum(buf[i] for i in range(7))
    segw, batches, recs = buf[7], buf[8], buf[9]
    print(f"{rays} rays: {segw} segment-waves, {batches} batches, {recs} records ({recs / max(batches, 1):.1f} per batch, "
          f"{batches / max(segw, 1):.2f} batches per segment-wave); cycles per segment-wave {tot / max(segw, 1):.0f}")
    for i, n in enumerate(names):
        print(f"    {n:32s} {100.0 * buf[i] / max(tot, 1):5.1f} %   {buf[i] / max(segw, 1):9.0f} cycles per segment-wave")
