import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from text2nerf_amd import TensorVMSplit, synth
dev = torch.device("cuda:0")
aabb = [[-8.0]*3, [8.0]*3]
def mk(shading):
    params = synth.make_field_params(0, [300]*3, scene="S1-soft", aabb=aabb, shading_mode=shading)
    m = TensorVMSplit(torch.tensor(aabb), [300]*3, dev, density_n_comp=[16]*3, appearance_n_comp=[48]*3, app_dim=27, near_far=[0.5, 8.0], shadingMode=shading, fea_pe=6, featureC=128, step_ratio=1.0)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return m
n = 4_280_000
g = torch.Generator().manual_seed(0)
# coherent-ish points: walk along a surface so gathers are local like real appearance samples
xyz = (torch.rand(n, 3, generator=g) * 2 - 1)
xyz[:, 2] = 0.875 + 0.01 * xyz[:, 2]
xyz = xyz[torch.argsort(xyz[:, 1] * 1000 + xyz[:, 0])].contiguous().to(dev)
vd = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(dev)
for shading in ["MLP_Fea_noview", "SH"]:
    m = mk(shading)
    for _ in range(2): m.shade(xyz, viewdirs=vd, want_features=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.shade(xyz, viewdirs=vd, want_features=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{shading}: {dt*1e3:.2f} ms for {n} points", flush=True)
