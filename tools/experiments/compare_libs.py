"""Bitwise comparison of two builds of libt2n_hip.so (T2N_LIB): whole C2 frames of S1-soft and S2 (rgb, depth, sample counts) and the
density at 1 M random points, each library in a fresh child process.   python tools/experiments/compare_libs.py base main"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(tag):
    import torch
    import bench
    from text2nerf_amd import synth
    dev = torch.device("cuda", 0)
    out = {}
    rays = torch.from_numpy(synth.frame_rays_np(800, 800)).to(dev)
    g = np.random.Generator(np.random.PCG64(5))
    pts = torch.from_numpy(g.uniform(-1, 1, (1 << 20, 3)).astype(np.float32)).to(dev)
    for sc, sd in (("S1-soft", 0), ("S2", 1)):
        f = bench.build_field(dev, scene=sc, seed=sd)[0]
        f.materialize_weights = False
        for fw in (800, 0):           # tile marcher / per-ray marcher
            f.frame_width = fw
            with torch.no_grad():
                rgb, depth, _, _ = f(rays)
            st = f.stats()
            out[f"{sc}_{fw}_rgb"], out[f"{sc}_{fw}_depth"] = rgb.cpu().numpy(), depth.cpu().numpy()
            out[f"{sc}_{fw}_counts"] = np.array([st["evaluated"], st["appearance"]])
        out[f"{sc}_sigma"] = f.compute_sigma(pts).cpu().numpy()
    np.savez(os.path.join(ROOT, "gpurun_out", f"cmp_{tag}.npz"), **out)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    names = sys.argv[1:3]
    for n in names:
        lib = os.path.join(ROOT, "text2nerf_amd", "libt2n_hip.so" if n == "main" else f"libt2n_hip_{n}.so")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=dict(os.environ, T2N_LIB=lib), check=True)
    a, b = (np.load(os.path.join(ROOT, "gpurun_out", f"cmp_{n}.npz")) for n in names)
    ok = True
    for k in a.files:
        same = np.array_equal(a[k], b[k])
        d = float(np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max())
        print(f"{k:22s} {'bitwise equal' if same else 'DIFFERENT'}  max |diff| {d:.3e}")
        ok &= same
    print("ALL BITWISE EQUAL" if ok else "NOT EQUAL")
