#!/usr/bin/env python3
"""bench.py — render throughput of the HIP TensoRF VM-split ray marcher on MI355X.

Workload (BASELINE.json configs[1], "C2"): TensorVMSplit 300^3, one 800x800 view (640 000 rays), N_samples=-1 => 518
samples/ray, render_only, synthetic scene S1-soft (seed 0), white background. A step = one full frame through the hot
path (march -> shade -> composite) with rays and weights resident in HBM. With --gpus N every rank renders its own view
of a small-baseline trajectory (weak scaling: independent ray tiles) and the rgb+depth tiles are all-gathered with
RCCL inside the timed region.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json plus `roofline` for the dominant kernel and
`cpu_baseline` (the oracle — a PyTorch CPU restatement of the reference path — on a bounded sample of the workload).
"""
import argparse
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def host_cores():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        pass
    return n


if os.environ.get("T2N_BENCH_DEADLINE_S"):
    # a rank that has not finished by then dumps every thread's Python stack to stderr and exits non-zero: a hang becomes a
    # diagnosable failure of THIS process (never a re-exec of a process that has touched the GPU)
    faulthandler.dump_traceback_later(float(os.environ["T2N_BENCH_DEADLINE_S"]), exit=True)

HOST_CORES = host_cores()
# host-side thread pools sized to the usable cores BEFORE torch / OpenMP start: oversubscribed pools stall the training
# loop's CPU indexing for tens of milliseconds at a time and make the CPU baseline meaningless
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_CORES))

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
L1_PEAK_GBS = 256 * 64 * 2.4   # 256 CUs x 64 B/clk vector-L1 data path x 2.4 GHz max clock = 39 321 GB/s
BYTES_PER_EVAL = 1152   # 3 planes x 4 taps x 64 B + 3 lines x 2 taps x 64 B
BYTES_PER_APP = 3456
BYTES_PER_RAY = 40
FLOP_PER_APP = 131168    # 2*(144*27 + 351*128 + 128*128 + 128*3)
FLOP_HEAD = 123392       # 2*(351*128 + 128*128 + 128*3): the MLP head without basis_mat
PMC_FILES = ("round6_pmc.json", "round5_pmc.json", "round4_pmc.json", "round3_pmc.json")   # the newest committed counter record is used (profiles/)
MFMA_F32_PEAK_TF = 157.3
MFMA_F16_PEAK_TF = 2500.0  # dense f16/bf16 MFMA peak (MI355X_MICROARCH.md)


def kernel_source_sha16():
    """sha256 over the kernel sources (csrc/*.hip, *.h, include/t2n.h), first 16 hex digits: ties a counter record to the code it was
    taken on (tools/pmc_round2.py stores the same value in the record)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "text2nerf_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "text2nerf_amd", "csrc", "*.h")) +
                    [os.path.join(ROOT, "include", "t2n.h")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc():
    """(counters, record): the per-launch PMC means of a SEPARATE profiled run of this command (rocprofv3 --pmc cannot ride along with a
    timed run) and what identifies that record: file, git blob id, the kernel-source hash it was taken on and whether that still is the
    hash of the sources this process runs."""
    import hashlib
    for name in PMC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            raw = open(path, "rb").read()
            pmc = json.loads(raw)
        except Exception:
            continue
        blob = hashlib.sha1(b"blob %d\0" % len(raw) + raw).hexdigest()
        now = kernel_source_sha16()
        rec = {"file": "profiles/" + name, "git_blob": blob, "taken_on_kernel_source_sha16": pmc.get("_kernel_source_sha16"),
               "current_kernel_source_sha16": now, "stale": pmc.get("_kernel_source_sha16") != now}
        return pmc, rec
    return {}, {"file": None}


def reference_poses(name, n=None):
    """Camera poses captured from the reference's own generators (tests/golden/poses.npz <- tests/golden/make_golden_poses.py:
    `local_fixed` = get_local_fixed_poses2, `circle_train_96` = cam_traj_gen circle, dataLoader/scene_util.py); repeated cyclically
    when more than the fixture holds are asked for."""
    P = np.load(os.path.join(ROOT, "tests", "golden", "poses.npz"))[name].astype(np.float32)
    if n is None or n <= P.shape[0]:
        return P if n is None else P[:n]
    return P[np.arange(n) % P.shape[0]]


def build_field(dev, scene="S1-soft", seed=0, grid=300):
    from text2nerf_amd import TensorVMSplit, synth
    aabb = [[-8.0] * 3, [8.0] * 3]
    params = synth.make_field_params(seed, [grid] * 3, scene=scene, aabb=aabb)
    m = TensorVMSplit(torch.tensor(aabb), [grid] * 3, dev, density_n_comp=[16] * 3, appearance_n_comp=[48] * 3,
                      app_dim=27, near_far=[0.5, 8.0], shadingMode="MLP_Fea_noview", alphaMask_thres=1e-4,
                      density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=128, step_ratio=1.0,
                      fea2denseAct="softplus")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
    return m, params, aabb


def cpu_baseline(params, aabb, grid, n_samples, budget_s=12.0, check=None):
    """The oracle's plain-C port (oracle/oracle_c.c, OpenMP over rays) timed on the host cores on the SAME workload: whole
    800x800 frames, repeated until ~budget_s of host work (at most 4 frames; fewer rays if one frame would take minutes)."""
    from oracle import oracle_torch as O
    from oracle.oracle_c import COracle
    from text2nerf_amd import synth
    cfg = O.FieldConfig(aabb=aabb, grid_size=[grid] * 3)
    co = COracle(cfg, params)
    cores = co.threads()
    probe = synth.frame_rays_np(800, 800, stride=8)            # 10 000 rays: calibrates the sample size
    t0 = time.time()
    co.render(probe, n_samples=n_samples, want_weights=False)
    rate = probe.shape[0] / max(time.time() - t0, 1e-6)          # rays / s
    stride = 1 if 640000 / rate < 2 * budget_s else (2 if 160000 / rate < 2 * budget_s else 4)
    rays = synth.frame_rays_np(800, 800, stride=stride)
    done, frames, t0 = 0, 0, time.time()
    while frames < 4 and (frames == 0 or time.time() - t0 < budget_s):
        o_rgb, o_depth, _, _ = co.render(rays, n_samples=n_samples, want_weights=False)
        done += rays.shape[0]
        frames += 1
    dt = time.time() - t0
    out = {"value": done * n_samples / dt, "unit": "ray-samples/s", "cores": cores, "kind": "port",
           "sample": f"{frames} x {rays.shape[0]} rays (800x800 frame, pixel stride {stride}) x {n_samples} samples, "
                     f"oracle_c (plain C, OpenMP {cores} threads), {dt:.1f} s"}
    # the PyTorch restatement (oracle_torch, the form BASELINE.md's reference-on-CPU figures were taken in) on a bounded sample
    try:
        P = O.params_from_numpy(params)
        sub = torch.from_numpy(synth.frame_rays_np(800, 800, stride=8))      # 10 000 rays
        t0 = time.time()
        with torch.no_grad():
            O.forward(cfg, P, sub[:2000], n_samples=n_samples)
        per = (time.time() - t0) / 2000
        n_t = int(min(10000, max(2000, 8.0 / max(per, 1e-6))))
        t0 = time.time()
        with torch.no_grad():
            for i in range(0, n_t, 2000):
                O.forward(cfg, P, sub[i:i + 2000], n_samples=n_samples)
        dt_t = time.time() - t0
        out["oracle_torch"] = {"value": n_t * n_samples / dt_t, "unit": "ray-samples/s", "cores": torch.get_num_threads(),
                               "sample": f"{n_t} rays (800x800 frame, pixel stride 8) x {n_samples} samples in chunks of 2000, {dt_t:.1f} s"}
    except Exception as e:  # noqa: BLE001
        out["oracle_torch"] = {"error": repr(e)[:200]}
    if check is not None:   # the oracle as the checker: the HIP render of the SAME rays against it, whole frame (untimed)
        h_rgb, h_depth = check(torch.from_numpy(rays))
        e = np.abs(h_rgb - o_rgb)
        out["parity_vs_oracle"] = {"rays": int(rays.shape[0]), "max_abs_rgb_err": float(e.max()),
                                   "rays_over_1e-4": int((e.max(1) > 1e-4).sum()),
                                   "max_abs_depth_err": float(np.abs(h_depth - o_depth).max()),
                                   "evaluated_samples_equal": None}
        out["parity_vs_oracle"]["oracle_evaluated"] = co.last_stats["evaluated"]
    return out


TRAIN_THREADS = None   # (experiments: host threads of the train loop)


def train_bench(dev, iters=20, warmup=3, fused_optim=False, dist=None, fused_step=False, batch=16384, resident=False, speculative=False, step_kw=None):
    keep_threads = torch.get_num_threads()
    try:
        return _train_bench(dev, iters, warmup, fused_optim, dist, fused_step, batch, resident, speculative, step_kw or {})
    finally:
        torch.set_num_threads(keep_threads)     # the loop below runs on half the cores; callers (CPU baseline, other legs) get theirs back


def _train_bench(dev, iters, warmup, fused_optim, dist, fused_step, batch, resident=False, speculative=False, step_kw={}):
    """C3-shaped optimisation step (text2nerf_main.py:547-601): 16 384 random rays of 9 small-baseline 512x512 views,
    N=259, is_train, MSE(rgb)+0.005 MSE(depth)+1e3 transmittance+TV(density 0.1, app 0.01), Adam(0.02/1e-3).
    With `dist` (world > 1): data-parallel — the SAME 16 384-ray batch is split into equal shards (strong scaling), local
    forward/backward, one flat gradient all-reduce (text2nerf_amd.parallel.allreduce_gradients), identical optimiser step."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    # the loop's CPU work is row gathers of 16 384 rays / targets: half the usable cores leaves the quota headroom for the HIP runtime's
    # own threads (a process that outruns its cgroup CPU quota is throttled for the rest of a 100-ms period: one such stall inside a
    # 40-ms timed region turned 1.87 ms per iteration into 2.4)
    # (the fused step's own host work is launches and one staging copy; its gathers run on the prefetch thread: two threads, the rest of
    # the quota stays with the HIP runtime — 0.97-0.98 ms at 1 or 2 threads against 0.97-1.07 at 4 or 8, tools/experiments/train_threads_ab.py)
    torch.set_num_threads(TRAIN_THREADS or (2 if fused_step else max(1, min(8, torch.get_num_threads() // 2))))
    from text2nerf_amd import OctreeRender_trilinear_fast, synth, to_device_async
    from text2nerf_amd.losses import TVLoss, TransMittanceLoss_mask
    field, params, aabb = build_field(dev)
    n_samples = min(int(1e6), int(synth.cal_n_samples([300] * 3, 1.0) / 2))          # text2nerf_main.py:439 -> 259
    poses = reference_poses("local_fixed")      # the 9 training poses of the driver's default trajectory (scene_gen.py:242)
    allrays = torch.from_numpy(np.concatenate([synth.frame_rays_np(512, 512, c2w=p) for p in poses]))   # CPU, like the driver
    # targets: the scene's own colours/depths (eval render of every 4th ray, nearest-assigned) + noise, so that the step
    # has realistic gradients but the field stays a scene; i.i.d. uniform targets turn the field into fog within ~10
    # Adam steps (2 M appearance samples per batch instead of ~1e5), which benchmarks nothing the driver ever does
    g = np.random.Generator(np.random.PCG64(1024))
    with torch.no_grad():
        sub = allrays[::4].to(dev)
        rgb_s, dep_s, _, _ = field(sub, white_bg=True, is_train=False, N_samples=n_samples)
    allrgb = (rgb_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] +
              torch.from_numpy(g.normal(0, 0.05, (allrays.shape[0], 3)).astype(np.float32))).clamp(0, 1)
    alldepth = dep_s.cpu().repeat_interleave(4, 0)[: allrays.shape[0]] + torch.from_numpy(
        g.normal(0, 0.05, (allrays.shape[0],)).astype(np.float32))
    def make_opt():
        if fused_optim or fused_step:   # TV gradient + Adam as HIP streaming kernels (text2nerf_amd/optim.py)
            from text2nerf_amd.optim import TVAdam
            # the plane / line tensors are stepped on the device's channel-last copies from device-side gradients; data-parallel runs
            # all-reduce that contiguous gradient buffer in place
            return TVAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), field=field)
        return torch.optim.Adam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
    opt = make_opt()
    init_state = {k: v.detach().clone() for k, v in field.state_dict().items()}   # every timed block starts from these parameters
    tv, tl = TVLoss(), TransMittanceLoss_mask(dev)
    np.random.seed(1024)
    torch.manual_seed(1024)
    perm = torch.from_numpy(np.random.permutation(allrays.shape[0]))

    if dist is not None:
        from text2nerf_amd.parallel import allreduce_gradients, broadcast_parameters, shard_batch
        broadcast_parameters(field.parameters())
        all_params = [p for p in field.parameters() if p.requires_grad]
        lo, hi = shard_batch(batch, world, rank)

    tv_terms = [(field.density_plane, 0.1), (field.app_plane, 0.01)]
    if resident:   # the whole training set (9 views: 94 MB) and the permutation live in HBM: the batch is three device gathers
        allrays_d, allrgb_d, alldepth_d, perm_d = allrays.to(dev), allrgb.to(dev), alldepth.to(dev), perm.to(dev)

    def idx_of(k):
        idx = perm[(k * batch) % (perm.numel() - batch):][:batch]
        return idx[lo:hi] if dist is not None else idx

    # fused step with the training set on the host: the row gathers of batch k+1 run on a worker thread while step k is enqueued
    # (text2nerf_amd.BatchPrefetcher; what a data loader does). The reference-form legs keep the driver's synchronous gather.
    from text2nerf_amd import BatchPrefetcher
    pf = BatchPrefetcher([allrays, allrgb, alldepth]) if (fused_step and not resident) else None
    pf_next = [None]

    def batch_of(k):
        if pf_next[0] != k:
            if pf_next[0] is not None:
                pf.get()            # (a block restarted the sequence: drop the batch gathered ahead)
            pf.submit(idx_of(k))
        b = pf.get()
        pf.submit(idx_of(k + 1))
        pf_next[0] = k + 1
        return b

    def it(k):
        if resident:
            idx = perm_d[(k * batch) % (perm.numel() - batch):][:batch]
            return field.train_step(allrays_d[idx], allrgb_d[idx], alldepth_d[idx], opt, N_samples=n_samples, white_bg=True, tv=tv_terms,
                                    speculative=speculative, **step_kw)[3]
        idx = perm[(k * batch) % (perm.numel() - batch):][:batch]
        if dist is not None:
            idx = idx[lo:hi]
        if fused_step:   # autograd-free: render -> loss kernel (emits d_rgb / d_depth / d_weights) -> backward -> TV + Adam
            ar = (lambda: allreduce_gradients(all_params, average=True, field=field)) if dist is not None else None
            b_rays, b_rgb, b_dep = batch_of(k)      # the rows allrays[idx] / allrgb[idx] / alldepth[idx] (text2nerf_main.py:550-553)
            kw = step_kw
            if "all_reduce" in step_kw:             # (scaling_prediction: a rank's compute with the exchange left out)
                ar, kw = step_kw["all_reduce"], {a: b for a, b in step_kw.items() if a != "all_reduce"}
            return field.train_step(b_rays, b_rgb, b_dep, opt, N_samples=n_samples, white_bg=True, tv=tv_terms, all_reduce=ar,
                                    speculative=speculative, **kw)[3]
        # targets go host -> device like text2nerf_main.py:550-553, through the pinned staging ring (a pageable .to(device)
        # drains the stream first and idles the GPU for the rest of the host-side batch preparation)
        rays, rgb_t, dep_t = allrays.index_select(0, idx), to_device_async(allrgb.index_select(0, idx), dev), to_device_async(alldepth.index_select(0, idx), dev)
        rgb, _, depth, w, z = OctreeRender_trilinear_fast(rays, field, chunk=max(int(rays.shape[0]), 1), N_samples=n_samples, white_bg=True,
                                                          ndc_ray=False, device=dev, is_train=True)
        loss = torch.mean((rgb - rgb_t) ** 2) + 0.005 * torch.mean((depth - dep_t) ** 2)
        loss = loss + 1e3 * tl(w, (z - dep_t[:, None] + 0.1) < 0)
        if not fused_optim:
            loss = loss + field.TV_loss_density(tv) * 0.1 + field.TV_loss_app(tv) * 0.01
        opt.zero_grad()
        loss.backward()
        if dist is not None:
            allreduce_gradients(all_params, average=True, field=field if fused_optim else None)
        if fused_optim:
            opt.step(tv=tv_terms)
        else:
            opt.step()
        return loss

    for k in range(warmup):
        it(k)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trace = [] if os.environ.get("T2N_TRAIN_TRACE") else None   # per-iteration wall times (drains the stream every iteration)
    for k in range(iters):
        loss = it(warmup + k)
        if trace is not None:
            if os.environ.get("T2N_TRAIN_TRACE") != "host":   # "host": host-side timestamps only, no drain
                torch.cuda.synchronize()
            trace.append(round((time.perf_counter() - t0) * 1e3, 2))
    if trace is not None:
        print("[bench] train iteration end times (ms):", trace, file=sys.stderr, flush=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    blocks_ms = None
    if dist is None and trace is None:
        # the loop's host side runs under a cgroup CPU quota: one throttled 100-ms period inside a 25-50 ms timed block doubles it.
        # FIVE blocks (three until round 6: one multi-millisecond host stall in a block — seen on some boxes once per few hundred steps —
        # plus the first block's warm-up cost was enough to move a median of three), each from the same initial parameters and a fresh optimiser (the noisy targets turn the field into fog after
        # ~45 steps — 114 000 appearance samples until step 40, 674 000 at 60, tools/experiments/train_sample_growth.py — so later
        # steps of one trajectory would be another workload), warm-up included; the MEDIAN is reported and all of them are in the line
        blocks = [dt]
        for _ in range(max(2, int(os.environ.get("T2N_TRAIN_BLOCKS", "5")) - 1)):
            field.load_state_dict(init_state)
            opt = make_opt()
            for k in range(warmup):
                it(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(iters):
                loss = it(warmup + k)
            torch.cuda.synchronize()
            blocks.append(time.perf_counter() - t0)
            if os.environ.get("T2N_TRAIN_DEBUG") and field.__dict__.get("_fused_step") is not None:
                fs_ = field._fused_step
                print("[bench] fused step driver:", dict(issued=fs_.issued, eager=fs_.eager_launches, graph=fs_.graph_launches, pipelined=fs_.pipelined_launches,
                                                          replays=fs_.replays, cap=fs_.rows_cap, needs=fs_.needs[-3:], ws=hex(fs_.ws.data_ptr()), ws_bytes=fs_.ws.numel(),
                                                          m=[hex(t.data_ptr()) for t in fs_._moments()[1][:12:3]], v=[hex(t.data_ptr()) for t in fs_._moments()[2][:12:3]],
                                                          gbuf=hex(field._gbuf.data_ptr()), inbuf=[hex(b.data_ptr()) for b in fs_.inbuf if b is not None], hg=hex(fs_.head_grads.data_ptr())),
                      file=sys.stderr, flush=True)
        blocks_ms = [round(b / iters * 1e3, 4) for b in blocks]
        print("[bench] train blocks of %d iterations (ms/iter): %s" % (iters, blocks_ms), file=sys.stderr, flush=True)
        dt = sorted(blocks)[len(blocks) // 2]
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        return {"train_dp_iters_per_s": iters / dt, "train_dp_ms_per_iter": dt / iters * 1e3,
                "train_dp_step": f"data-parallel x{world}: {batch} rays split into {hi - lo}/GPU (strong scaling), in-place all-reduce of the "
                                 f"{field.factor_grad_buffer().numel() * 4 / 1e6:.1f} MB channel-last factor-gradient buffer + one small flat "
                                 f"message for the head, fused loss + TV + Adam on the device copies, loss {float(loss.detach()):.4f}"}
    if speculative:
        return {"ms_per_iter": dt / iters * 1e3, "blocks_ms": blocks_ms, "rays": batch, "resident": bool(resident),
                "device_rows_steps": int(getattr(field, "device_rows_steps", 0)), "overflows": int(getattr(field, "device_rows_overflows", 0)),
                "unanswered_polls": int(getattr(field, "device_rows_unanswered", 0)), "loss": float(loss.detach())}
    if fused_step and batch != 16384:
        return {"ms_per_iter": dt / iters * 1e3, "rays": batch}
    if resident:
        return {"train_iters_per_s_fused_step_resident": iters / dt, "train_ms_per_iter_fused_step_resident": dt / iters * 1e3,
                "train_ms_per_iter_fused_step_resident_blocks": blocks_ms,
                "train_step_fused_resident": "train_step with the training set (rays, colours, depths of the 9 views) resident in HBM: the "
                                             "batch is gathered on the device, only the jitter draws (CPU generator, like the reference) "
                                             "cross PCIe"}
    if fused_step:
        # kernel groups of one fused iteration (HIP events on the launch streams, a short pass of its own: the brackets cost the
        # streams a few us each) and what bounds the largest of them. In the one-call form (t2n_train_step) the groups run side by side on
        # four streams: each figure is the group's own start-to-end time UNDER that concurrency, their sum exceeds the step
        kt, roofs = {}, {}
        fs = field.__dict__.get("_fused_step") if step_kw.get("fused", True) is not False else None
        try:
            field.timing(True)
            field.read_timing(reset=True)
            n_t = 10
            for k in range(n_t):
                it(warmup + iters + k)
            torch.cuda.synchronize()
            kt = {k: round(v[0] / n_t, 5) for k, v in field.read_timing(reset=True).items()}
            field.timing(False)
            # sample counts of one such batch (a train-mode render of its own: the one-call step keeps no statistics)
            with torch.no_grad():
                b_rays = allrays[idx_of(warmup + iters)]
                torch.manual_seed(7)
                field(b_rays.to(dev), white_bg=True, is_train=True, N_samples=n_samples)
            st_ = field.stats()
            A, V = st_["appearance"], st_["evaluated"]
            lds_peak = 256 * 64 / 9.0 * 2.1e9     # a CU retires one wave-wide ds_add_f64 per ~9 clocks (tools/experiments/lds_atomic_bench.hip)
            if kt.get("bwd_scatter"):
                # appearance scatter (k_bwd_tile_accum<48> [+ its binning in the composed form]): per appearance row and plane 48 channels x
                # (4 plane + 2 line taps) double-precision LDS atomics
                atom = A * 3 * 48 * 6
                roofs["bwd_scatter"] = {"bound": "lds-f64-atomics", "kernel": "k_bwd_tile_accum<48>", "unit": "G lane-atomics/s",
                                        "achieved": atom / (kt["bwd_scatter"] * 1e-3) / 1e9, "peak": lds_peak / 1e9,
                                        "frac": atom / (kt["bwd_scatter"] * 1e-3) / lds_peak, "ms_per_iter": kt["bwd_scatter"]}
            if kt.get("bwd_density"):
                # density scatter (k_bin_reduce + k_bin_scan + k_bwd_bin + k_bwd_den_block): per evaluated sample ONE record, eight ds_add_f64
                # (trilinear splat into the block's 16^3 corner field); the six contractions per block segment run on fp32 MFMA and are
                # flushed with 13 824 global fp32 atomics per segment — the group is bound by that flush and by the binning passes, not by
                # the LDS atomics this fraction prices
                atom = V * 8
                roofs["bwd_density"] = {"bound": "lds-f64-atomics", "kernel": "k_bwd_bin + k_bwd_den_block (+ scans)", "unit": "G lane-atomics/s",
                                        "achieved": atom / (kt["bwd_density"] * 1e-3) / 1e9, "peak": lds_peak / 1e9,
                                        "frac": atom / (kt["bwd_density"] * 1e-3) / lds_peak, "ms_per_iter": kt["bwd_density"],
                                        "records_per_iter": V}
            if kt.get("bwd_mlp"):
                # input-gradient chain (k_mlp_bwd_ss; 3 f16 products per fp32 product) [+ the weight-gradient GEMMs in the composed form]:
                # the reference's MACs per row against the dense f16 peak
                both = "bwd_wgrad" not in kt
                flop = (2.0 if both else 1.0) * 2.0 * (351 * 128 + 128 * 128 + 128 * 3 + 144 * 27) * A
                roofs["bwd_mlp"] = {"bound": "mfma", "kernel": "k_mlp_bwd_ss" + (" + k_gemm_tn_b + k_bwd_l2" if both else ""), "unit": "TFLOP/s",
                                    "achieved": flop / (kt["bwd_mlp"] * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TF,
                                    "frac": flop / (kt["bwd_mlp"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, "ms_per_iter": kt["bwd_mlp"]}
            if kt.get("bwd_wgrad"):
                # weight-gradient GEMMs (k_gemm_tn_b x 2: six bf16 products per fp32 product; k_gemm_tn fp32; one reduce): VALU-bound on the
                # operand splits — with half the MFMAs the step is 20 us shorter (profiles/round6_gemm_half_mfma_ab.txt)
                flop = 2.0 * (351 * 128 + 128 * 128 + 144 * 27) * A
                roofs["bwd_wgrad"] = {"bound": "mfma", "kernel": "k_gemm_tn_b x 2 + k_gemm_tn + k_wgrad_reduce", "unit": "TFLOP/s",
                                      "achieved": flop / (kt["bwd_wgrad"] * 1e-3) / 1e12, "peak": MFMA_F16_PEAK_TF,
                                      "frac": flop / (kt["bwd_wgrad"] * 1e-3) / 1e12 / MFMA_F16_PEAK_TF, "ms_per_iter": kt["bwd_wgrad"]}
            if kt.get("adam") or kt.get("tv_seed"):
                # TV seed + Adam on the 70 MB of factors: streaming (seed: read p, write g; Adam: read g, p, m, v, write p, m, v + the reference-layout copy)
                nb = field.factor_grad_buffer(_raw=True).numel() * 4
                roofs["tv_adam"] = {"bound": "hbm", "kernel": "k_tv_seed_cl_multi + k_adam_cl_multi (appearance part; the density part runs beside the appearance scatter)",
                                    "unit": "GB/s", "bytes_per_iter": 10 * nb,
                                    "note": "2 + 8 passes over the factor bytes per step = 0.70 GB: ~0.12 ms at the ~6 TB/s these kernels reach alone"}
        except Exception as e:  # noqa: BLE001  (reporting only)
            kt = {"error": repr(e)[:200]}
        form = "one C call (t2n_train_step, eager, pipelined)" if fs is not None and not step_kw.get("graph") else (
            "one hipGraph launch (t2n_train_graph_*)" if fs is not None else "composed: render -> loss kernel -> backward -> TVAdam, separate calls")
        extra = {}
        if fs is not None:
            fs.sync()
            extra = {"train_step_driver": dict(issued=fs.issued, eager=fs.eager_launches, pipelined=fs.pipelined_launches, graph_launches=fs.graph_launches,
                                               graph_captures=fs.graph_captures, graph_nodes=getattr(fs, "graph_nodes", None), replays=fs.replays,
                                               rows_capacity=fs.rows_cap, rows_needed=fs.needs[-3:], workspace_GiB=round(fs.ws.numel() / 2 ** 30, 3))}
        return {"train_iters_per_s_fused_step": iters / dt, "train_ms_per_iter_fused_step": dt / iters * 1e3,
                "train_ms_per_iter_fused_step_blocks": blocks_ms,
                "train_kernel_ms_per_iter": kt, "train_kernel_rooflines": roofs, **extra,
                "train_step_fused": f"TensorVMSplit.train_step, {form}: no autograd graph, the driver's loss inside the per-ray backward kernel, "
                                    f"device-side row plan, TV seed + Adam on the device copies; loss {float(loss.detach()):.4f}"}
    if fused_optim:
        return {"train_iters_per_s_fused_optim": iters / dt, "train_ms_per_iter_fused_optim": dt / iters * 1e3,
                "train_ms_per_iter_fused_optim_blocks": blocks_ms}
    return {"train_iters_per_s": iters / dt, "train_ms_per_iter": dt / iters * 1e3, "train_iters": iters,
            "train_ms_per_iter_blocks": blocks_ms,
            "train_timing": "MEDIAN of three blocks of train_iters iterations, each from the same initial parameters with a fresh optimiser "
                            "(every train_* figure; the three block times are in *_blocks)",
            "train_step": f"C3-shaped: {batch} rays x {n_samples} samples, fwd+bwd HIP, TV+Adam torch (reference-form "
                          f"step), loss {float(loss.detach()):.4f}; *_fused_optim: TV gradient + Adam as HIP kernels",
            "train_appearance_samples": field.stats()["appearance"]}


def general_shape_ms(dev, G=128, H=400):
    """One eval frame of a WIDE field on the general-shape path (csrc/t2n_generic.hip: [32,20,24] density / [96,64,72] appearance
    components, featureC 256 — beyond the tuned kernels' 16 / 48 / 128) through the reference's call, next to the tuned shape on the
    same 128^3 grid and 400 x 400 camera: what leaving the tuned shape costs."""
    from text2nerf_amd import TensorVMSplit, OctreeRender_trilinear_fast, synth
    aabb, nf = [[-8.0, -6.0, -7.0], [8.0, 7.0, 6.5]], [0.5, 8.0]
    rays = torch.from_numpy(synth.frame_rays_np(H, H, c2w=synth.look_pose(0.3, -0.1, (0.2, 0.1, -1.0)))).to(dev)
    out = {}
    for tag, dn, an, fc in (("tuned_16_48_128", [16] * 3, [48] * 3, 128), ("wide_32.20.24_96.64.72_256", [32, 20, 24], [96, 64, 72], 256)):
        params = synth.make_field_params(11, [G] * 3, density_n_comp=dn, app_n_comp=an, app_dim=27, feature_c=fc, fea_pe=6,
                                         shading_mode="MLP_Fea_noview", density_scale=0.9, aabb=aabb)
        m = TensorVMSplit(torch.tensor(aabb), [G] * 3, dev, density_n_comp=dn, appearance_n_comp=an, app_dim=27, near_far=nf,
                          shadingMode="MLP_Fea_noview", density_shift=-10, distance_scale=25, pos_pe=0, view_pe=0, fea_pe=6, featureC=fc,
                          step_ratio=1.0, fea2denseAct="softplus")
        m.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})
        with torch.no_grad():
            for _ in range(2):
                OctreeRender_trilinear_fast(rays, m, chunk=65536, N_samples=-1, white_bg=True, is_train=False, device=dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                OctreeRender_trilinear_fast(rays, m, chunk=65536, N_samples=-1, white_bg=True, is_train=False, device=dev)
            torch.cuda.synchronize()
        out[tag] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        out["n_samples"] = int(m.nSamples)
        del m
    out["ratio"] = round(out["wide_32.20.24_96.64.72_256"] / max(out["tuned_16_48_128"], 1e-9), 1)
    out["frame"] = f"{G}^3 grid, {H} x {H} rays, ms per frame; round 5's plain form of the general path: 3 396 ms (profiles/round6_general_path.txt)"
    return out


def dropin_eval_ms(field, dev, H, W, n=12):
    """The reference's own evaluation call, unchanged (renderer.py:85-89): ALL rays of one image as a HOST tensor into
    OctreeRender_trilinear_fast(rays, tensorf, chunk=..., N_samples=-1, ...) — H2D through the pinned ring, raster width detected
    from the rays, (a) the full 5-tuple with weights / z_vals [R, N] materialised like the reference returns them, (b) what
    `evaluation` keeps (rgb + depth; text2nerf_amd.renderer.evaluation switches the two [R, N] tensors off). Every call is timed on
    its own (drained before and after); the MEDIAN is reported next to the mean: the 15.4 MB host-side staging copy of a call can run
    into the container's CPU quota, and one throttled period inside a short loop would otherwise dominate the figure."""
    from text2nerf_amd import OctreeRender_trilinear_fast, synth
    rays = torch.from_numpy(synth.frame_rays_np(H, W))
    keep = field.frame_width, field.materialize_weights
    out = {}
    try:
        field.frame_width = 0
        for name, mat in (("full_5tuple_host_rays", True), ("rgb_depth_only_host_rays", False)):
            field.materialize_weights = mat
            ts = []
            with torch.no_grad():
                for i in range(n + 2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = OctreeRender_trilinear_fast(rays, field, chunk=16384, N_samples=-1, white_bg=True, device=dev)
                    torch.cuda.synchronize()
                    if i >= 2:
                        ts.append((time.perf_counter() - t0) * 1e3)
                    del r
            ts.sort()
            out[name] = ts[len(ts) // 2]
            out[name + "_mean"] = sum(ts) / len(ts)
            out[name + "_list_retry"] = field.stats()["list_retry"]
    finally:
        field.frame_width, field.materialize_weights = keep
    return out


def scaling_prediction(field, dev, fused_ms, G=8):
    """What ONE GPU can say about N = 8 (no multi-GPU node is available to the builder): (i) C4 — the 8 tiles of the ONE 1600x1600
    frame timed separately, as contiguous row blocks and as interleaved 8-row bands: the frame is done when the slowest tile is, so
    predicted efficiency = mean / max of the tile times (the all-gather of 8 x 5.1 MB is priced from the xGMI link rate);
    (ii) data-parallel training — the fused step at 16384 / G rays on one GPU is the per-rank time, plus the in-place all-reduce of
    the 69.6 MB gradient buffer priced from the link rate."""
    from text2nerf_amd import generate_rays
    from text2nerf_amd.parallel import band_shard, shard_bounds
    H = W = 1600
    fl = float(W)
    keep = field.frame_width, field.materialize_weights
    field.frame_width, field.materialize_weights = W, False
    pred = {"n_gpus": G}
    try:
        rays = generate_rays(H, W, [fl, fl, W // 2, H // 2], np.eye(4, dtype=np.float32), device=dev)
        R = rays.shape[0]

        def t_ms(r, n=5):
            with torch.no_grad():
                for _ in range(2):
                    field(r, white_bg=True, is_train=False, N_samples=-1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    field(r, white_bg=True, is_train=False, N_samples=-1)
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        full = t_ms(rays)
        cont = [t_ms(rays[slice(*shard_bounds(R, G, r))].contiguous()) for r in range(G)]
        inter = [t_ms(band_shard(rays, W, G, r).contiguous()) for r in range(G)]
        # all-gather of G equal [R / G, 4] tiles: each rank's 5.12 MB goes to its 7 peers over 7 separate xGMI links (~153 GB/s
        # each, priced at 70 %) + ~30 us of launch / synchronisation
        ag_ms = (R // G) * 16 / (153e9 * 0.7) * 1e3 + 0.03
        pred["c4"] = {
            "single_gpu_frame_ms": full, "tile_ms_contiguous": cont, "tile_ms_interleaved": inter,
            "balance_contiguous": sum(cont) / G / max(cont), "balance_interleaved": sum(inter) / G / max(inter),
            "all_gather_estimate_ms": ag_ms,
            "predicted_speedup_contiguous": full / (max(cont) + ag_ms), "predicted_speedup_interleaved": full / (max(inter) + ag_ms),
            "predicted_8gpu_efficiency": full / (max(inter) + ag_ms) / G,
            "note": "bench.py --mode c4 uses interleaved bands when the rows divide evenly (--c4-tiles); a tile's time includes its "
                    "fixed per-frame costs (launches, counter read-back), which do not shrink with 1 / G"}
    except Exception as e:  # noqa: BLE001
        pred["c4"] = {"error": repr(e)[:300]}
    finally:
        field.frame_width, field.materialize_weights = keep
    if fused_ms is not None:
        try:
            # five more train legs (smaller batches, the two data-parallel forms): a CHILD process — what they say is a prediction beside
            # the headline, and the one GPU fault this script ever died of (once in 32 runs, never reproduced) was inside them
            import subprocess
            cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--aux-train-dp", str(G)], stdout=subprocess.PIPE, timeout=900)
            if cp.returncode != 0:
                raise RuntimeError(f"auxiliary train legs exited with {cp.returncode}")
            aux = json.loads(cp.stdout.decode().strip().splitlines()[-1])
            steps = {16384: fused_ms}
            for b in (8192, 4096, 2048):
                steps[b] = aux["steps"][str(b)]
            dp_flat, dp_shard = aux["dp_flat"], aux["dp_shard"]
            # direct reduce-scatter + all-gather of the 69.6 MB buffer over 7 links per GPU: 2 x (bytes / 8) per link
            ar_ms = 2 * (69.6e6 / G) / (153e9 * 0.7) * 1e3 + 0.05
            pred["train_dp"] = {"fused_step_ms_by_rays_per_gpu": {str(k): v for k, v in steps.items()},
                                "all_reduce_estimate_ms": ar_ms,
                                "per_rank_compute_ms_flat_all_reduce": dp_flat,
                                "per_rank_compute_ms_sharded_optimizer": dp_shard,
                                "exchange_estimate_ms_sharded": ar_ms,     # reduce-scatter of the gradients + all-gather of the parameters: the same bytes per link
                                "predicted_8gpu_step_ms_flat": dp_flat + ar_ms,
                                "predicted_8gpu_step_ms": dp_shard + ar_ms,
                                "predicted_speedup": fused_ms / (dp_shard + ar_ms),
                                "predicted_speedup_flat": fused_ms / (dp_flat + ar_ms),
                                # the other way to use 8 GPUs (the driver's --batch_size x 8): every GPU keeps its 16 384 rays
                                "predicted_8gpu_weak_step_ms": fused_ms + ar_ms,
                                "predicted_weak_rays_per_s_ratio": G * fused_ms / (fused_ms + ar_ms),
                                "note": "strong scaling of ONE 16 384-ray batch: the step at 2 048 rays / GPU is a chain of 28 launches whose "
                                        "fixed cost does not shrink. per_rank_compute_*: measured on this GPU with the exchange left out — "
                                        "flat: every rank runs TV + Adam over all 70 MB; sharded (parallel.ShardedExchange, t2n_train_step "
                                        "shard_world / shard_rank): 1 / 8 of the planes per rank + the write-back of the gathered 7 / 8. The "
                                        "exchange (reduce-scatter of gradients + all-gather of parameters, priced from the link rate) is not "
                                        "overlapped with compute in either form. *_weak_*: 16 384 rays per GPU (8 x the batch), rays/s against one GPU"}
        except Exception as e:  # noqa: BLE001
            pred["train_dp"] = {"error": repr(e)[:300]}
    pred["weak_c2"] = {"note": "default --gpus N mode: one independent 800x800 view per GPU, the 10 MB all-gather of frame k overlaps the "
                               "render of frame k + 1 (async): no shared resource but the host; predicted efficiency ~1.0"}
    return pred


def aux_train_dp(G):
    """`bench.py --aux-train-dp G` (child of scaling_prediction): the fused step at 8 192 / 4 096 / 2 048 rays and the per-rank COMPUTE of a
    data-parallel step at 16 384 / G rays with the exchanges left out (one GPU here) — the two-call form every rank of the flat all-reduce
    runs (TV + Adam over all 70 MB on every rank), and rank 0 of the sharded optimiser (parallel.ShardedExchange: seeds / steps 1 / G of
    the planes, then writes the gathered 7 / 8 into the reference layout). One JSON line."""
    torch.set_num_threads(max(1, min(HOST_CORES, 16)))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)

    class _NoExchange:
        def __init__(self, world, rank):
            self.world, self.rank = world, rank

        def reduce(self, fs):
            pass

        def gather(self, fs):
            pass
    out = {"steps": {}}
    for b in (8192, 4096, 2048):
        out["steps"][str(b)] = train_bench(dev, iters=20, warmup=3, fused_step=True, batch=b)["ms_per_iter"]
    out["dp_flat"] = train_bench(dev, iters=20, warmup=3, fused_step=True, batch=16384 // G,
                                 step_kw=dict(fused=True, graph=False, all_reduce=lambda: None))["ms_per_iter"]
    out["dp_shard"] = train_bench(dev, iters=20, warmup=3, fused_step=True, batch=16384 // G,
                                  step_kw=dict(fused=True, graph=False, all_reduce=_NoExchange(G, 0)))["ms_per_iter"]
    print(json.dumps(out), flush=True)


def count_gpus_sysfs():
    """GPUs this process can USE, without touching the HIP runtime: KFD topology nodes with SIMDs (CPU nodes have simd_count 0) whose
    render node (/dev/dri/renderD<drm_render_minor>) is readable and writable here — a container's device cgroup may expose fewer
    render nodes than the host topology lists —, then narrowed the way the runtime composes the visibility variables:
    ROCR_VISIBLE_DEVICES filters first, HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES index into what is left. 0 without a KFD topology,
    None (unknown: the child ranks decide) when it cannot be parsed."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return 0       # no KFD topology: no AMD GPU is visible to this process
    n = 0
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
            node = f"/dev/dri/renderD{minor}"
            if minor >= 0 and os.path.exists("/dev/dri") and not os.access(node, os.R_OK | os.W_OK):
                continue   # listed by the host's topology, not usable from here
            n += 1
        except Exception:
            return None

    def listed(var):
        v = os.environ.get(var)
        return None if v is None or v.strip() == "" else len([x for x in v.split(",") if x.strip() != ""])
    rocr = listed("ROCR_VISIBLE_DEVICES")
    if rocr is not None:
        n = min(n, rocr)
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):     # (indices into the ROCR-filtered list: never more than it holds)
        k = listed(var)
        if k is not None:
            n = min(n, k)
    return n


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process — which makes NO GPU call (it imports torch but never asks the
    runtime anything: the device count comes from sysfs) — starts N fresh child ranks of the same command (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays rank 0's stdout (the ONE JSON line), lets the ranks' stderr through and
    returns the first non-zero exit code (the other ranks are then killed by PID). Nothing is exec'ed and no process that has
    touched the GPU is re-launched. The `python -m torch.distributed.run ... bench.py --gpus N` form (WORLD_SIZE already set)
    never comes through here."""
    import socket
    import subprocess
    import threading
    same = bool(os.environ.get("T2N_BENCH_SAME_DEVICE"))
    have = count_gpus_sysfs()
    if have is not None and have < n and not same:
        print(f"bench.py --gpus {n}: this node shows {have} GPU(s); one rank per GPU is required "
              f"(T2N_BENCH_SAME_DEVICE=1 T2N_BENCH_BACKEND=gloo runs the N>1 code path on one device for testing)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), T2N_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env["OMP_NUM_THREADS"] = str(max(1, HOST_CORES // n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout (the JSON line, plus whatever a library prints there) is drained WHILE the ranks run: a full 64-KiB pipe would
    # block rank 0 in write() with its peers waiting in a collective
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("T2N_BENCH_LAUNCH_DEADLINE_S", "1500"))
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() is not None and p.returncode != 0]
        if bad or time.time() > deadline:
            rc = bad[0] if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.kill()          # the exact children started above
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
        if rc == 0 and p.returncode != 0:
            rc = p.returncode
    reader.join(timeout=10)
    sys.stdout.write("".join(chunks))
    sys.stdout.flush()
    return rc


def device_identity(dev):
    """What tells two GPUs apart (for the rank evidence in the JSON line): index, name, uuid / PCI ids where torch exposes them."""
    pr = torch.cuda.get_device_properties(dev)
    d = {"index": dev.index, "name": pr.name}
    for k in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        v = getattr(pr, k, None)
        if v is not None:
            d[k] = str(v)
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scene", default="S1-soft")
    ap.add_argument("--weights", type=int, default=0, help="1: also materialise weights/z_vals [R,N] like the reference")
    ap.add_argument("--factor-storage", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: BASELINE configs[4] storage mode (render == fp32 render of the bf16-rounded factor tensors)")
    ap.add_argument("--per-ray-marcher", action="store_true",
                    help="disable the 8x8-tile marcher (default for whole row-major frames: field.frame_width = W)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick", action="store_true", help="skip the secondary measurements (sustained, bf16, materialised, exact fp32)")
    ap.add_argument("--no-train", action="store_true", help="skip the C3-shaped train-step timing (iters/s)")
    ap.add_argument("--mode", default="weak", choices=["weak", "c4"],
                    help="weak: one 800x800 view per GPU (C2, the headline); c4: BASELINE configs[3] — ONE 1600x1600 frame split "
                         "into contiguous ray tiles over the ranks, all-gather of the rgb+depth tiles inside every step "
                         "(strong scaling)")
    ap.add_argument("--c4-tiles", default="auto", choices=["auto", "contiguous", "interleaved"],
                    help="c4 mode: how the frame's rows go to the ranks — interleaved 8-row bands (balanced: every rank sees the same mix of "
                         "border and centre rays) or contiguous row blocks; auto = interleaved when the rows divide evenly")
    ap.add_argument("--check-c4", action="store_true",
                    help="c4 mode: rank 0 also renders the whole frame alone and compares it bitwise with the gathered one")
    ap.add_argument("--no-inregion-timing", action="store_true",
                    help="no HIP timing events inside the timed region (the head's launch time then comes from the untimed pass behind it)")
    ap.add_argument("--train-iters", type=int, default=20)
    ap.add_argument("--train-warmup", type=int, default=3)
    ap.add_argument("--aux-train-dp", type=int, default=0, help=argparse.SUPPRESS)   # (internal: child of scaling_prediction)
    args = ap.parse_args()
    if args.aux_train_dp:
        aux_train_dp(args.aux_train_dp)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))      # parent of N fresh ranks; has not touched the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (plain `python bench.py --gpus N` does)")
    torch.set_num_threads(max(1, min(HOST_CORES // max(world, 1), 16)))
    same_device = bool(os.environ.get("T2N_BENCH_SAME_DEVICE"))   # functional test of the N>1 path on a 1-GPU box (gloo)
    if same_device:
        local_rank = 0
    dev = torch.device("cuda", local_rank)

    def note(msg):   # progress markers on stderr: the integration test tells a device bring-up stall from a hung collective
        print(f"[bench] rank {rank}: {msg}", file=sys.stderr, flush=True)

    def gpu_up():
        assert torch.cuda.is_available(), "bench.py needs an MI355X"
        if torch.cuda.device_count() <= local_rank:      # (a rank without a device of its own leaves at once instead of hanging its peers)
            print(f"[bench] rank {rank}: local rank {local_rank} but {torch.cuda.device_count()} visible device(s)", file=sys.stderr, flush=True)
            os._exit(3)
        torch.cuda.set_device(local_rank)
        torch.zeros(1, device=dev).add_(1)
        torch.cuda.synchronize()

    dist = None
    # T2N_BENCH_FORCE_GROUP=1: initialise the process group even for ONE rank, so that the N > 1 code of this file (RCCL group with
    # device_id, object gather of the rank evidence, tile all-gather inside the timed region, max-over-ranks all-reduce, the
    # data-parallel train step) executes on a 1-GPU box (tests/test_rccl_single.py)
    grouped = world > 1 or bool(os.environ.get("T2N_BENCH_FORCE_GROUP"))
    if grouped:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("T2N_BENCH_BACKEND", "nccl")   # "nccl" IS RCCL on ROCm
        tmo = datetime.timedelta(seconds=float(os.environ.get("T2N_BENCH_COLLECTIVE_TIMEOUT_S", "300")))
        if backend == "nccl":
            gpu_up()
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
            # several ranks on ONE device: bring the device up one process at a time
            for r in range(world):
                if r == rank or not same_device:
                    gpu_up()
                if same_device:
                    dist.barrier()
                else:
                    break
        note("gpu and process group ready")
        ids = [None] * world
        dist.all_gather_object(ids, dict(device_identity(dev), rank=rank, local_rank=local_rank, pid=os.getpid()))
        key = lambda d: d.get("uuid") or d.get("pci_bus_id") or d["index"]   # noqa: E731
        rccl = {"backend": dist.get_backend(), "is_rccl": dist.get_backend() == "nccl", "world_size": dist.get_world_size(),
                "launcher": "bench.py self-launch" if os.environ.get("T2N_BENCH_SELF_LAUNCHED") else "external (torch.distributed.run)",
                "ranks": ids, "distinct_devices": len({key(d) for d in ids})}
    else:
        gpu_up()
        rccl = None

    from text2nerf_amd import generate_rays, synth
    from text2nerf_amd.parallel import all_gather_tiles

    c4 = args.mode == "c4"
    H = W = 1600 if c4 else 800
    field, params, aabb = build_field(dev, scene=args.scene, seed=0 if args.scene.startswith("S1") else 1)
    field.materialize_weights = bool(args.weights)
    field.factor_storage = args.factor_storage
    # the rays of a step are one whole row-major frame: the width hint lets the renderer use the tile marcher (rays of an 8x8
    # pixel tile share per-step dot-product tables); results stay within the parity tolerances of the per-ray marcher
    field.frame_width = 0 if args.per_ray_marcher else W
    N = field.nSamples
    poses = reference_poses("local_fixed", max(world, 9))
    pose = poses[rank % len(poses)] if world > 1 and not c4 else np.eye(4, dtype=np.float32)
    f = float(max(H, W))
    rays = generate_rays(H, W, [f, f, W // 2, H // 2], pose, device=dev)   # resident in HBM before the timed region
    bands = False
    if c4:   # this rank's part of the ONE frame: every world-th 8-row band (1600 rows / 8 ranks = 25 bands each), or a contiguous block
        from text2nerf_amd.parallel import band_layout, band_shard, band_unshard, shard_bounds, tile_capacity
        frame_rays, R_frame = rays, rays.shape[0]
        bands = args.c4_tiles != "contiguous" and band_layout(R_frame, W, world) is not None
        if args.c4_tiles == "interleaved" and not bands:
            raise SystemExit(f"--c4-tiles interleaved: {H} rows do not divide into {world} x 8-row bands")
        if bands:
            rays = band_shard(frame_rays, W, world, rank).contiguous()
            cap = rays.shape[0]
        else:
            lo, hi = shard_bounds(R_frame, world, rank)
            cap = tile_capacity(R_frame, world)
            rays = frame_rays[lo:hi].contiguous()
        if not args.check_c4:
            del frame_rays
    R = rays.shape[0]

    # N > 1: the all-gather of frame k's [rays, 4] tile runs asynchronously (RCCL's own stream) while frame k+1 is rendered;
    # a step waits for the PREVIOUS frame's gather, and the timed region ends only after the last gather has completed.
    pending = []

    def step():
        with torch.no_grad():
            rgb, depth, z, w = field(rays, white_bg=True, is_train=False, N_samples=-1)
            if c4 and grouped:   # strong scaling: the frame is complete only when every tile has arrived
                tile = torch.zeros((cap, 4), dtype=rgb.dtype, device=rgb.device)
                tile[:R, :3], tile[:R, 3] = rgb, depth
                out = torch.empty((world * cap, 4), dtype=tile.dtype, device=tile.device)
                dist.all_gather_into_tensor(out, tile)
                return out
            if grouped:
                tile = torch.cat([rgb, depth[:, None]], 1)
                out = torch.empty((world * tile.shape[0], 4), dtype=tile.dtype, device=tile.device)
                work = dist.all_gather_into_tensor(out, tile, async_op=True)
                pending.append((work, out, tile))
                if len(pending) > 1:
                    pending.pop(0)[0].wait()
                return out
            return rgb

    def drain():
        while pending:
            pending.pop(0)[0].wait()

    for _ in range(max(args.warmup, 1)):   # at least one untimed frame: allocations + the per-frame sample counters
        step()
    drain()
    st = field.stats()
    # HIP events on the launch stream around the DOMINANT kernel only inside the timed region (the roofline's kernel: the head); an event
    # pair costs the stream ~10 us of bubble per bracketed kernel group — all four groups of a frame were 1.8 % of it
    # (profiles/round4_frame_timeline.txt). The other kernels' times come from a short untimed pass below.
    if args.no_inregion_timing:
        field.timing(False)
    else:
        field.timing(True, kernels=("shade",))
    field.read_timing(reset=True)
    # THREE back-to-back blocks of `steps` frames, each bracketed by barrier + synchronize on both sides; ms_per_step is the MEDIAN block
    # (max over ranks per block). One 20-frame block is ~50 ms of wall clock: a single descheduling of this process (the boxes run
    # under a cgroup CPU quota) inside it would be the whole figure.
    block_dt = []
    for _ in range(3):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        block_dt.append(time.perf_counter() - t0)
    timing = {k: (v[0] / 3.0, v[1] / 3.0) for k, v in field.read_timing(reset=True).items()}   # three blocks -> per block of `steps` frames
    field.timing(True)
    aux_steps = max(min(args.steps, 30), 1)
    with torch.no_grad():
        for _ in range(min(aux_steps, 10)):   # (the clocks sag while the host reads the timed region's events: not part of this pass)
            field(rays, white_bg=True, is_train=False, N_samples=-1)
        torch.cuda.synchronize()
        field.read_timing(reset=True)
        for _ in range(aux_steps):
            field(rays, white_bg=True, is_train=False, N_samples=-1)
    torch.cuda.synchronize()
    for k, v in field.read_timing(reset=True).items():   # scaled to the timed region's step count (what the per-step figures below divide by)
        if k not in timing:
            timing[k] = (v[0] * args.steps / aux_steps, v[1] * args.steps / aux_steps)
    field.timing(False)
    if dist is not None:
        t = torch.tensor(block_dt, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        block_dt = [float(x) for x in t.tolist()]
    dt = sorted(block_dt)[1]

    c4_equal = None
    if c4 and args.check_c4:   # the gathered frame against the whole frame rendered by rank 0 alone (untimed; every rank gathers)
        g = step()
        if rank == 0:
            with torch.no_grad():
                s_rgb, s_depth, _, _ = field(frame_rays, white_bg=True, is_train=False, N_samples=-1)
            if grouped:
                if bands:
                    g = band_unshard(g, W, world)
                else:
                    g = g.view(world, cap, 4)
                    g = torch.cat([g[r, : shard_bounds(R_frame, world, r)[1] - shard_bounds(R_frame, world, r)[0]]
                                   for r in range(world)], 0)
                c4_equal = bool(torch.equal(g[:, :3], s_rgb) and torch.equal(g[:, 3], s_depth))
            else:
                c4_equal = bool(torch.equal(g, s_rgb))
            del s_rgb, s_depth
    dp = {}
    if dist is not None and not args.no_train:   # every rank takes part in the data-parallel train step
        try:
            dp = train_bench(dev, iters=args.train_iters, warmup=args.train_warmup, fused_step=True, dist=dist)
        except BaseException:
            # a rank that fails inside the data-parallel section must not leave its peers blocked in a collective it will
            # never join: report and leave non-zero at once (the launcher tears the job down; the collectives carry a timeout)
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            os._exit(13)
    if rank == 0:
        V, A = st["evaluated"], st["appearance"]
        ms_step = dt / args.steps * 1e3
        nominal = (R_frame if c4 else world * R) * N * args.steps / dt
        k_ms = {k: v[0] / max(v[1], 1) for k, v in timing.items()}        # avg ms per launch
        k_per_step = {k: v[1] / args.steps for k, v in timing.items()}     # launches per step
        frame_ms = {k: v[0] / args.steps for k, v in timing.items()}       # kernel ms per frame
        pmc, pmc_rec = load_pmc()   # per-launch PMC figures of the same command (tools/pmc_round2.sh), committed under profiles/
        split = not field.mlp_exact_fp32
        two_kernel = "app_features" in frame_ms    # default path: gather + basis kernel, then the sample-stationary head
        roofs = {}
        if "shade" in k_ms:
            # algorithmic work of the head kernel: the reference's fp32 MACs, 2 flop each; basis_mat's 7 776 flop per sample belong
            # to the gather + basis kernel on the two-kernel path
            flop_per = FLOP_HEAD if two_kernel else FLOP_PER_APP
            launches = max(k_per_step["shade"], 1.0)
            alg = flop_per * A / launches
            t = k_ms["shade"] * 1e-3
            peak = MFMA_F16_PEAK_TF if split else MFMA_F32_PEAK_TF
            kname = "k_mlp_ss3" if two_kernel else ("k_shade_coop" if split else "k_shade<exact>")
            roofs["shade"] = {
                "bound": "mfma", "kernel": kname, "achieved": alg / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                "frac": alg / t / 1e12 / peak, "traffic": pmc.get(kname, {}).get("hbm_bytes_per_launch"),
                "algorithmic_flop_per_launch": alg, "avg_launch_ms": k_ms["shade"],
                # every fp32 product runs as three f16 MFMA products of hi/lo splits: what the matrix pipe actually executes
                "executed_frac": (3.0 if split else 1.0) * alg / t / 1e12 / peak,
                "mfma_busy_frac_pmc": pmc.get(kname, {}).get("mfma_busy_frac"),
                # what a bare stream of v_mfma_f32_32x32x16_f16 sustains on this part (no other instructions,
                # tools/experiments/mfma_fillers.hip: 16.2 ns per MFMA and SIMD = 2.07 PFLOP/s at the clock the chip holds under
                # that load) - the kernel's practical ceiling; profiles/round4_head_issue_accounting.txt says where the rest goes
                "executed_frac_of_sustained_mfma_rate": ((3.0 * alg / t / 1e12 / 2070.0) if split else None),
                "note": ("frac = algorithmic fp32-equivalent flop (123 392 per appearance sample: 351x128 + 128x128 + 128x3 MACs) / time / "
                         "dense f16 MFMA peak; executed_frac counts the 3 f16 products per fp32 product" if split else "exact fp32 MFMA")}
        if "march" in k_ms:
            # the tile marcher replaces the per-sample 1152-B gather by dot-product tables on the matrix cores: it is bound by VALU
            # issue, not by bytes. achieved = VALU instructions per launch (PMC SQ_INSTS_VALU of this command) / measured time;
            # peak = 1024 SIMDs x one wave64 VALU instruction per 2 cycles x 2.4 GHz
            kname = "k_march" if args.per_ray_marcher else "k_march_tiles"
            insts = pmc.get(kname, {}).get("valu_insts_per_launch")
            t = k_ms["march"] * 1e-3
            peak = 1024 * 0.5 * 2.4
            roofs["march"] = {"bound": "valu-issue", "kernel": kname, "unit": "Ginst/s", "peak": peak,
                              "achieved": (insts / t / 1e9) if insts else None, "frac": (insts / t / 1e9 / peak) if insts else None,
                              "traffic": pmc.get(kname, {}).get("hbm_bytes_per_launch"), "avg_launch_ms": k_ms["march"],
                              "gather_work_rate_GBps": (BYTES_PER_EVAL * V + BYTES_PER_RAY * R) / max(k_per_step["march"], 1.0) / t / 1e9,
                              "note": "gather_work_rate = 1152 B per evaluated sample / time: a work rate for comparison across versions "
                                      "(the field is cache-resident and the tile marcher does not move those bytes), not a bandwidth"}
        if "app_features" in k_ms:
            # 216 16-B lane loads per appearance sample (18 taps x 12 channel quads) through the texture addresser, which retires
            # ~4 lane addresses per clock and CU (measured): peak = 256 x 4 x 2.4 GHz
            t = k_ms["app_features"] * 1e-3
            lane_loads = 216.0 * A / max(k_per_step["app_features"], 1.0)
            peak = 256 * 4 * 2.4
            roofs["app_features"] = {"bound": "texture-addresser", "kernel": "k_app_features_p", "unit": "G lane-loads/s",
                                     "achieved": lane_loads / t / 1e9, "peak": peak, "frac": lane_loads / t / 1e9 / peak,
                                     "traffic": (pmc.get("k_app_features_p") or pmc.get("k_app_features") or {}).get("hbm_bytes_per_launch"), "avg_launch_ms": k_ms["app_features"],
                                     "algorithmic_GBps": BYTES_PER_APP * A / max(k_per_step["app_features"], 1.0) / t / 1e9}
        dom = max(roofs, key=lambda k: frame_ms.get(k, 0.0)) if roofs else None
        roof = dict(roofs[dom]) if dom else {}
        if dom and roof.get("bound") not in ("hbm", "mfma"):
            roof["bound_contract"] = "hbm"     # the contract's two classes: everything that is not matrix work
        # whole-frame SIMD-issue model: the counter record's instruction counts per kernel x the measured issue costs
        # (profiles/round4_issue_cost_microbench.txt: an MFMA ~21 cycles of its SIMD's issue, a wave64 VALU instruction ~5 on the 16-lane
        # pipe, an LDS instruction ~8) over 1 024 SIMDs at the ~2.1 GHz the part holds under these kernels. What the frame would take if
        # issue were the only limit and nothing overlapped: the distance to ms_per_step is latency the occupancy does not hide
        issue = {}
        for kn in ("k_march_tiles", "k_app_features_p", "k_mlp_ss3", "k_composite"):
            c = pmc.get(kn)
            if c:
                cyc = 21.0 * c.get("mfma_insts_per_launch", 0.0) + 5.0 * c.get("valu_insts_per_launch", 0.0) + 8.0 * c.get("lds_insts_per_launch", 0.0)
                issue[kn] = cyc / (1024 * 2.1e9) * 1e3
        kernel_sum = sum(frame_ms.values())
        cfg = {
            # the driver's record keeps the first 24 keys of `config` and only their scalar values: the figures a reader needs come first
            "workload": (f"C4 300^3 ONE 1600x1600 frame in {world} ray tiles" if c4 else "C2 300^3 800x800 view/GPU") +
                        f", N={N}, S1-soft seed 0, white_bg, weights={int(bool(args.weights))}, {'per-ray' if args.per_ray_marcher else 'tile'} marcher",
            "ms_per_step_blocks": " / ".join(f"{b / args.steps * 1e3:.4f}" for b in block_dt) + " (three blocks of `steps` frames; ms_per_step = median)",
            "sustained_ms_per_step": None,
            "kernel_ms_per_frame_sum": kernel_sum,
            "host_gap_ms_per_step": ms_step - kernel_sum,
            "issue_model_ms": sum(issue.values()) if issue else None,
            "exact_fp32_ms_per_step": None,
            "weights_materialised_ms_per_step": None,
            "bf16_factor_storage_ms_per_step": None,
            "c5_circle_48_views_bf16_ms_per_view": None,
            "train_ms_per_iter": None,
            "train_ms_per_iter_fused_optim": None,
            "train_ms_per_iter_fused_step": None,
            "train_ms_per_iter_fused_step_2048_rays": None,
            "train_iters_per_s_fused_step": None,
            "evaluated_samples_per_s": world * V * args.steps / dt,
            "rays_per_s": (R_frame if c4 else world * R) * args.steps / dt,
            "evaluated_samples_per_frame": V,
            "appearance_samples_per_frame": A,
            "memory_reserved_GiB_render": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 3),
            "memory_reserved_GiB_train": None,
            "pmc_blob": (pmc_rec.get("git_blob") or "")[:12] + (" STALE" if pmc_rec.get("stale") else ""),
            "rccl_version": None,
            "early_termination_eps": float(getattr(field, "early_termination", 0.0) or 0.0),
            # ---- past the driver's cut: details ----
            "workload_detail": (f"C4: TensorVMSplit 300^3, ONE 1600x1600 frame in {world} ray tiles "
                                f"({'interleaved 8-row bands' if bands else 'contiguous row blocks'}), "
                                if c4 else "C2: TensorVMSplit 300^3, 800x800 view/GPU, ") +
                               f"{N} samples/ray, render_only, scene {args.scene} seed 0, white_bg, weights/z_vals materialised: "
                               f"{bool(args.weights)}, marcher: {'per-ray' if args.per_ray_marcher else '8x8-pixel tiles'}",
            "rays_per_gpu": R, "samples_per_ray": N,
            "ms_per_step_block_list": [round(b / args.steps * 1e3, 4) for b in block_dt],
            "inregion_timing": "none" if args.no_inregion_timing else "head only",
            "parallelism": f"ray-tile x{world}" + ((" + RCCL all-gather of rgb+depth tiles inside every step" if c4 else
                                                    " + RCCL all-gather of rgb+depth tiles (async, overlapped with the "
                                                    "next frame's render)") if grouped else ""),
            "kernel_ms_per_frame": frame_ms,
            "issue_model_ms_by_kernel": issue,
            "kernel_rooflines": roofs,
            # which counter record fed roofline.traffic / mfma_busy_frac_pmc / the march's `achieved` (not measured in THIS run)
            "pmc_record": pmc_rec,
        }
        out = {
            "metric": "ray-samples/s (render) + iters/s (train), 300^3 VM-split, 800x800",
            "value": nominal, "unit": "ray-samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong" if c4 else "weak", "vs_baseline": None,
            "dtype": "f32" + (" (basis/MLP products as f16x2-split MFMA, fp32 accumulate)" if split else "") +
                     ("; factor tensors stored as bf16" if args.factor_storage == "bf16" else ""),
            "data": "synthetic",
            "config": cfg,
            "roofline": roof,
            # `value` counts the reference's tensors (rays x samples per ray, what BASELINE.json's metric names); 76.6 % of those samples lie
            # outside the box or behind the z gate and are evaluated by nobody (the reference masks them too): config.evaluated_samples_per_s
            # is the rate of samples that actually read the field
            "notes": {"scaling_measured": "the builder has 1-GPU boxes only: see scaling_prediction (N = 1 line) for what one GPU can "
                                          "say about N = 8, and config.rccl (N > 1 lines) for the devices the ranks actually ran on"},
        }
        out["config"].update(dp)
        if rccl is not None:
            try:
                rccl["version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001
                rccl["version"] = repr(e)[:80]
            out["config"]["rccl_version"] = rccl["version"] if rccl.get("is_rccl") else f"(backend {rccl.get('backend')})"
            rccl["all_gather_bytes_per_rank_per_step"] = int((cap if c4 else R) * 16)
            rccl["all_gather_bytes_total_per_step"] = int((cap if c4 else R) * 16 * world)
            if dp:
                rccl["all_reduce_bytes_per_train_step"] = int(sum(p.numel() for p in field._all_params()[:12]) * 4)
            out["config"]["rccl"] = rccl
        try:   # render scratch actually reserved (appearance lists budgeted from the previous frame) next to the worst-case figure
            from text2nerf_amd import _lib as _L, tensorf as _tf
            _l = _L.load()
            out["config"]["workspace"] = {
                "reserved_GiB": round(_tf.workspace_reserved(dev) / 2 ** 30, 2),
                "worst_case_GiB": round(int(_l.t2n_render_workspace_bytes(R, N)) / 2 ** 30, 2),
                "steady_state_hint_GiB": round(int(_l.t2n_render_workspace_bytes_hint(field.sync_params(), R, N)) / 2 ** 30, 2),
                "list_retries": int(_l.t2n_field_list_retries(field.sync_params())),
                "note": "image-ordered frames run as one launch with appearance lists sized from earlier frames' counters (read back "
                        "without waiting); rays that find no room are finished on the device (frames with such rays: list_retries)"}
        except Exception as e:   # reporting only
            out["config"]["workspace"] = {"error": repr(e)}
        if c4_equal is not None:
            out["config"]["c4_gathered_equals_single_rank"] = c4_equal

        def timed_frames(n):
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        if not grouped and not c4 and not args.quick:
            # sustained clocks: the same frame for >= 2 s
            n_sus = max(int(2200.0 / max(ms_step, 0.1)), args.steps)
            out["config"]["sustained_ms_per_step"] = timed_frames(n_sus)
            out["config"]["sustained_frames"] = n_sus
            # frames of a trajectory are independent: frame k on HIP stream k % 2 (each stream has its own scratch), so the march of
            # frame k + 1 (VALU issue) runs beside the feature gather (L1) and the head (matrix cores) of frame k. Not the headline:
            # `value` / `ms_per_step` and the per-kernel timings above are taken one frame at a time on one stream
            try:
                from text2nerf_amd.renderer import _FramePipe
                for nfl in (2, 3):
                    pipe = _FramePipe(dev, nfl)

                    def piped(n):
                        keep_alive = []
                        with torch.no_grad():
                            for _ in range(n):
                                with torch.cuda.stream(pipe.next()):
                                    keep_alive.append(field(rays, white_bg=True, is_train=False, N_samples=-1)[0])
                                if len(keep_alive) > 2 * nfl:
                                    keep_alive.pop(0)
                        pipe.hand_over(keep_alive)
                    piped(6)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    piped(args.steps)
                    torch.cuda.synchronize()
                    out["config"][f"frames_in_flight_{nfl}_ms_per_step"] = (time.perf_counter() - t0) / args.steps * 1e3
            except Exception as e:  # noqa: BLE001
                out["config"]["frames_in_flight_error"] = repr(e)[:200]
            if args.factor_storage == "fp32":
                # bf16 factor storage (configs[4] mode): half the appearance gather's instructions and bytes (the feature kernel is L1-bound,
                # DESIGN.md); not the headline value: it renders the ROUNDED field
                field.factor_storage = "bf16"
                out["config"]["bf16_factor_storage_ms_per_step"] = timed_frames(args.steps)
                out["notes"]["bf16_factor_storage"] = "appearance factors read as bf16 (26 MB instead of 52 MB), taps fetched as 16-B octets of 8 channels: half the gather instructions; the tile marcher reads the fp32 copy of the rounded density factors (17 MB), the per-ray marcher their bf16 copy"
                field.factor_storage = "fp32"
            # BASELINE configs[4] on one GPU: the 48 training views of the reference's circle trajectory (cam_traj_gen, fixture), bf16
            # factor storage, rays generated on the device, one frame per view
            try:
                from text2nerf_amd import render_views
                c5 = reference_poses("circle_train_96")
                field.factor_storage = "bf16"
                render_views(field, c5, [f, f, W // 2, H // 2], H, W)   # untimed pass: the 48 views' output tensors (0.5 GB) come out of the allocator's pool afterwards
                c5_ms = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    render_views(field, c5, [f, f, W // 2, H // 2], H, W)
                    torch.cuda.synchronize()
                    c5_ms.append((time.perf_counter() - t0) / c5.shape[0] * 1e3)
                out["config"]["c5_circle_48_views_bf16_ms_per_view"] = sorted(c5_ms)[1]   # median of three passes
                out["config"]["c5_circle_48_views_bf16_ms_per_view_passes"] = [round(x, 4) for x in c5_ms]
            except Exception as e:  # noqa: BLE001
                out["config"]["c5_circle_error"] = repr(e)[:200]
            field.factor_storage = "fp32"
            field.frame_width = 0 if args.per_ray_marcher else W
            if not args.weights:
                # the reference's actual return: weights and z_vals [R, N] materialised (renderer.py:39-42), 2.65 GB per frame
                field.materialize_weights = True
                out["config"]["weights_materialised_ms_per_step"] = timed_frames(max(args.steps // 2, 3))
                field.materialize_weights = False
            if split:
                # the reference's own arithmetic: exact fp32 products (nn.Linear, models/tensorBase.py:94-106) on the f32 matrix cores
                field.mlp_exact_fp32 = True
                field.timing(True)
                field.read_timing(reset=True)
                ms_exact = timed_frames(max(args.steps // 2, 3))
                tx = field.read_timing(reset=True)
                field.timing(False)
                field.mlp_exact_fp32 = False
                sh = tx.get("shade")
                ex = {"ms_per_step": ms_exact}
                if sh and sh[1]:
                    t = sh[0] / sh[1] * 1e-3
                    ex.update({"kernel": "k_shade<exact> (gather + basis + MLP, one kernel)", "shade_avg_launch_ms": sh[0] / sh[1],
                               "achieved_TFLOPs": FLOP_PER_APP * A / t / 1e12, "peak_TFLOPs": MFMA_F32_PEAK_TF,
                               "frac": FLOP_PER_APP * A / t / 1e12 / MFMA_F32_PEAK_TF})
                out["exact_fp32"] = ex
                # the strict same-arithmetic figure travels in `config` (the driver keeps config, unknown top-level keys only by name)
                out["config"]["exact_fp32_ms_per_step"] = ms_exact
                out["config"]["exact_fp32_value_ray_samples_per_s"] = R * N / (ms_exact * 1e-3)
                out["config"]["exact_fp32_shade_frac_of_f32_mfma_peak"] = ex.get("frac")
        if not grouped and not c4 and not args.quick:
            # early ray termination (eval renders without weights / z_vals; the mirror's default eps 1e-6): frame time and evaluated
            # samples with it on and off, on the bench scene (soft walls: T ~ 4e-4 behind them, nothing to skip), on fog (S2) and on
            # opaque walls (S1-sharp)
            et = {}
            keep_eps = field.early_termination
            for sc, sd in (("S1-soft", 0), ("S2", 1), ("S1-sharp", 0)):
                try:
                    fld = field if sc == args.scene else build_field(dev, scene=sc, seed=sd)[0]
                    fld.materialize_weights, fld.frame_width = False, W
                    row, blocks = {}, {"on": [], "off": []}
                    for rep in range(3):   # alternating blocks, medians: the first block behind a field build runs 1-4 % slow whatever its mode
                        for tag, eps in (("on", 1e-6), ("off", 0.0)):
                            fld.early_termination = eps
                            with torch.no_grad():
                                for _ in range(3):
                                    fld(rays, white_bg=True, is_train=False, N_samples=-1)
                                torch.cuda.synchronize()
                                t0 = time.perf_counter()
                                for _ in range(10):
                                    fld(rays, white_bg=True, is_train=False, N_samples=-1)
                                torch.cuda.synchronize()
                            blocks[tag].append((time.perf_counter() - t0) / 10 * 1e3)
                            row[f"evaluated_samples_{tag}"] = fld.stats()["evaluated"]
                    for tag in ("on", "off"):
                        row[f"ms_per_frame_{tag}"] = sorted(blocks[tag])[1]
                        row[f"ms_per_frame_{tag}_blocks"] = [round(b, 4) for b in blocks[tag]]
                    et[sc] = row
                    if fld is not field:
                        del fld
                except Exception as e:  # noqa: BLE001
                    et[sc] = {"error": repr(e)[:200]}
            field.early_termination = keep_eps
            out["config"]["early_termination"] = et
            try:
                out["config"]["dropin_eval_call_ms"] = dropin_eval_ms(field, dev, H, W)
            except Exception as e:  # noqa: BLE001
                out["config"]["dropin_eval_call_ms"] = {"error": repr(e)[:300]}
            try:
                out["config"]["general_shape_ms"] = general_shape_ms(dev)
            except Exception as e:  # noqa: BLE001
                out["config"]["general_shape_ms"] = {"error": repr(e)[:300]}
        if not grouped and not args.no_train:
            # the fused step first, from a clean allocator: its peak reserved memory is the train figure (the render legs above
            # leave 2.65-GB weight tensors and worst-case workspaces in torch's cache)
            try:
                from text2nerf_amd import tensorf as _tf2
                _tf2.release_workspaces(keep_current=False)
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats(dev)
                base_res = torch.cuda.memory_reserved(dev)
            except Exception:  # noqa: BLE001
                base_res = 0
            out["config"].update(train_bench(dev, iters=args.train_iters, warmup=args.train_warmup, fused_step=True))
            # (what the fused train leg added on top of the resident render state: the 300^3 field, its ray tensor, outputs)
            out["config"]["memory_reserved_GiB_train"] = round((torch.cuda.max_memory_reserved(dev) - base_res) / 2 ** 30, 3)
            for tag, kw in (("graph", dict(fused=True, graph=True)), ("composed", dict(fused=False))):
                try:   # the same step as ONE hipGraph launch, and in round 5's composed form (separate C calls), same loop
                    r_ = train_bench(dev, iters=args.train_iters, warmup=args.train_warmup, fused_step=True, step_kw=kw)
                    out["config"][f"train_ms_per_iter_fused_step_{tag}"] = r_["train_ms_per_iter_fused_step"]
                    out["config"][f"train_ms_per_iter_fused_step_{tag}_blocks"] = r_["train_ms_per_iter_fused_step_blocks"]
                    if tag == "graph":
                        out["config"]["train_step_graph_nodes"] = (r_.get("train_step_driver") or {}).get("graph_nodes")
                except Exception as e:  # noqa: BLE001  (reporting only)
                    out["config"][f"train_ms_per_iter_fused_step_{tag}"] = {"error": repr(e)[:200]}
            out["config"].update(train_bench(dev, iters=args.train_iters, warmup=args.train_warmup))
            out["config"].update(train_bench(dev, iters=args.train_iters, warmup=args.train_warmup, fused_optim=True))
            out["config"].update(train_bench(dev, iters=args.train_iters, warmup=args.train_warmup, fused_step=True, resident=True))
        if not grouped and not c4 and not args.quick:
            out["scaling_prediction"] = sp = scaling_prediction(field, dev, out["config"].get("train_ms_per_iter_fused_step"))
            out["config"]["train_ms_per_iter_fused_step_2048_rays"] = sp.get("train_dp", {}).get("fused_step_ms_by_rays_per_gpu", {}).get("2048")
            out["config"]["scaling_prediction_headline"] = {
                "c4_predicted_8gpu_speedup": sp.get("c4", {}).get("predicted_speedup_interleaved"),
                "c4_tile_balance": sp.get("c4", {}).get("balance_interleaved"),
                "train_dp_predicted_8gpu_speedup": sp.get("train_dp", {}).get("predicted_speedup"),
                "train_dp_step_ms_at_2048_rays": sp.get("train_dp", {}).get("fused_step_ms_by_rays_per_gpu", {}).get("2048"),
                "train_dp_all_reduce_estimate_ms": sp.get("train_dp", {}).get("all_reduce_estimate_ms"),
                "note": "one-GPU predictions (tile times, link-rate estimates), not measurements: see scaling_prediction"}
        if not grouped and not args.no_cpu_baseline:
            def hip_render(r):
                with torch.no_grad():
                    a, b, _, _ = field(r.to(dev), white_bg=True, is_train=False, N_samples=-1)
                hip_render.evaluated = field.stats()["evaluated"]
                return a.cpu().numpy(), b.cpu().numpy()
            field.factor_storage = args.factor_storage
            out["cpu_baseline"] = cpu_baseline(params, aabb, 300, N, check=hip_render)
            pv = out["cpu_baseline"].get("parity_vs_oracle")
            if pv:
                oe = pv.pop("oracle_evaluated")
                pv["evaluated_samples_equal"] = bool(oe == hip_render.evaluated)
                pv["evaluated_samples_hip_over_oracle"] = hip_render.evaluated / max(oe, 1)   # < 1 only through early termination (config.early_termination_eps)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
