"""Per-region cycle accounting of k_bwd_den_block (a -DDB_PROF build: tools/build_variant.sh dbprof -DDB_PROF; T2N_LIB=.../libt2n_hip_dbprof.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from text2nerf_amd import _lib
torch.set_num_threads(max(1, min(bench.HOST_CORES, 16)))
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.t2n_debug_db_prof
fn.restype = C.c_int
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
names = ["zero + line rows", "barrier", "splat", "barrier", "plane contractions + flush", "line contractions + flush"]
for rays in [int(x) for x in sys.argv[1:]] or [16384, 2048]:
    buf = (C.c_uint64 * 16)()
    bench.train_bench(dev, iters=20, warmup=3, fused_step=True, batch=rays, step_kw=dict(fused=True, graph=False))
    torch.cuda.synchronize()
    fn(buf, 1)
    tot = sum(buf[i] for i in range(6))
    segw, recs = buf[6], buf[7]
    print(f"{rays} rays: {segw} segment-waves ({segw // 8} segments), {recs} records ({recs / max(segw // 8, 1):.0f} per segment); cycles per segment-wave {tot / max(segw, 1):.0f}")
    for i, n in enumerate(names):
        print(f"    {n:32s} {100.0 * buf[i] / max(tot, 1):5.1f} %   {buf[i] / max(segw, 1):9.0f} cycles per segment-wave")
