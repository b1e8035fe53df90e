"""How slow is the general-shape path? One eval frame of a wide field ([32,20,24] / [96,64,72] components, featureC 256) against the tuned
shape (16 / 48, featureC 128) on the same grid and camera: ms per frame, ns per evaluated sample, ns per appearance sample.
python tools/experiments/general_probe.py [grid] [H]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from text2nerf_amd import TensorVMSplit, synth
dev = torch.device("cuda:0")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 400
aabb = [[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]]
nf = [0.5, 8.0]
rays = torch.from_numpy(synth.frame_rays_np(H, H, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)))).to(dev)
def run(tag, dn, an, fc):
    params = synth.make_field_params(11, [G] * 3, density_n_comp=dn, app_n_comp=an, app_dim=27, feature_c=fc, fea_pe=6,
                                     shading_mode="MLP_Fea_noview", density_scale=0.9, aabb=aabb)
    m = TensorVMSplit(torch.tensor(aabb), [G] * 3, dev, density_n_comp=dn, appearance_n_comp=an, app_dim=27, near_far=nf,
                      shadingMode="MLP_Fea_noview", density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=fc,
                      step_ratio=1.0, fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    from text2nerf_amd import OctreeRender_trilinear_fast
    with torch.no_grad():
        for _ in range(2):
            out = OctreeRender_trilinear_fast(rays, m, chunk=65536, N_samples=-1, white_bg=True, is_train=False, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 3
        for _ in range(n):
            out = OctreeRender_trilinear_fast(rays, m, chunk=65536, N_samples=-1, white_bg=True, is_train=False, device=dev)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
    st = m.stats()     # (last chunk only)
    print(f"{tag}: general={m._is_general()} N={m.nSamples} frame {ms:.2f} ms, {ms * 1e6 / (rays.shape[0] * m.nSamples):.2f} ns per nominal sample, stats {st}", flush=True)
    return ms
a = run("tuned 16/48/128", [16] * 3, [48] * 3, 128)
b = run("wide [32,20,24]/[96,64,72]/256", [32, 20, 24], [96, 64, 72], 256)
print(f"ratio {b / a:.1f}x")
