"""Is torch.optim.Adam's fused implementation available on this ROCm build, does a per-group 'fused' key select it, and what does a step
over the field's 69.6 MB of parameters cost against the default (foreach) one?"""
import time
import torch
dev = torch.device("cuda", 0)
shapes = [(1, 16, 300, 300)] * 3 + [(1, 16, 300, 1)] * 3 + [(1, 48, 300, 300)] * 3 + [(1, 48, 300, 1)] * 3 + [(27, 144), (128, 351), (128,), (128, 128), (128,), (3, 128), (3,)]
def make():
    return [torch.nn.Parameter(torch.randn(s, device=dev) * 0.1) for s in shapes]
def run(name, opt, ps, n=30):
    for p in ps: p.grad = torch.randn_like(p) * 1e-3
    for _ in range(3): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): opt.step()
    torch.cuda.synchronize(); print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:7.3f} ms/step", flush=True)
torch.manual_seed(0); pa = make(); torch.manual_seed(0); pb = make(); torch.manual_seed(0); pc = make()
oa = torch.optim.Adam([{"params": pa[:12], "lr": 0.02}, {"params": pa[12:], "lr": 1e-3}], betas=(0.9, 0.99))
run("default (foreach)", oa, pa)
try:
    ob = torch.optim.Adam([{"params": pb[:12], "lr": 0.02}, {"params": pb[12:], "lr": 1e-3}], betas=(0.9, 0.99), fused=True)
    run("fused=True (constructor)", ob, pb)
except Exception as e:
    print("fused=True failed:", repr(e)[:300])
try:
    oc = torch.optim.Adam([{"params": pc[:12], "lr": 0.02, "fused": True}, {"params": pc[12:], "lr": 1e-3, "fused": True}], betas=(0.9, 0.99))
    run("'fused': True in the param groups", oc, pc)
    # same gradients -> same parameters as the default implementation?
    torch.manual_seed(1)
    for a, c in zip(pa, pc):
        g = torch.randn_like(a) * 1e-3; a.grad = g.clone(); c.grad = g.clone()
    with torch.no_grad():
        for a, c in zip(pa, pc): c.copy_(a)
    oa2 = torch.optim.Adam([{"params": pa[:12], "lr": 0.02}, {"params": pa[12:], "lr": 1e-3}], betas=(0.9, 0.99))
    oc2 = torch.optim.Adam([{"params": pc[:12], "lr": 0.02, "fused": True}, {"params": pc[12:], "lr": 1e-3, "fused": True}], betas=(0.9, 0.99))
    for _ in range(5): oa2.step(); oc2.step()
    print("max |foreach - fused| after 5 steps:", max(float((a - c).abs().max()) for a, c in zip(pa, pc)))
except Exception as e:
    print("per-group fused failed:", repr(e)[:300])
