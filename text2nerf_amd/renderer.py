"""Renderer harness with the reference's call surface (renderer.py:14-42)."""
from __future__ import annotations

import os

import numpy as np
import torch


class SimpleSampler:
    """Permutation batcher (renderer.py:14-26); host-side, unchanged semantics."""

    def __init__(self, total, batch):
        self.total, self.batch = total, batch
        self.curr = total
        self.ids = None

    def nextids(self):
        self.curr += self.batch
        if self.curr + self.batch > self.total:
            self.ids = torch.LongTensor(np.random.permutation(self.total))
            self.curr = 0
        return self.ids[self.curr:self.curr + self.batch]


class BatchPrefetcher:
    """The training loop's row gathers (text2nerf_main.py:550-553: `allrays[ray_idx]`, `allrgbs[ray_idx]`, depths) one batch ahead on a
    worker thread: `submit(ids)` starts gathering rows `ids` of every tensor, `get()` returns the oldest submitted batch. The gathers
    are `index_select` (per-row copies; the advanced-indexing kernel costs 3x the host time and collapses when its threads are
    oversubscribed) and release the GIL, so they run beside the main thread's kernel launches."""

    def __init__(self, tensors):
        import queue
        import threading
        self.tensors = list(tensors)
        self._in, self._out = queue.Queue(), queue.Queue()
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def _run(self):
        while True:
            ids = self._in.get()
            if ids is None:
                return
            try:
                self._out.put([t.index_select(0, ids) for t in self.tensors])
            except Exception as e:  # noqa: BLE001  (handed to the consumer)
                self._out.put(e)

    def submit(self, ids):
        self._in.put(ids)

    def get(self):
        b = self._out.get()
        if isinstance(b, Exception):
            raise b
        return b

    def close(self):
        self._in.put(None)


_FW_CACHE = {}     # (data_ptr, shape, version, device) -> detected width: a loop that re-renders the same ray tensor probes it once


def detect_frame_width(rays, probe=8192):
    """Width W of a row-major pinhole raster in `rays` [R, >=6] (what `evaluation` passes: all rays of one image,
    renderer.py:85-89), or 0. Looks at the first `probe` rays on the host: one origin, directions that change smoothly along a row
    and jump at a row's end (consecutive direction deltas differ by more than half their size only there); the jump must repeat at
    2W - 1 when visible, the second row must start like the first, R must be a multiple of W with at least 8 rows of at least 8
    pixels, and the first ray of every 8-row band must share the first ray's origin (several views concatenated into one tensor
    are not one raster). The hint only selects the tile marcher (8x8-pixel tiles of coherent rays); a wrong guess costs speed,
    never correctness. The result is cached per ray tensor (storage address, shape, version)."""
    R = int(rays.shape[0])
    if R < 64 or rays.shape[1] < 6:
        return 0
    key = (rays.data_ptr(), tuple(rays.shape), rays._version, str(rays.device))
    hit = _FW_CACHE.get(key)
    if hit is not None:
        return hit
    W = _detect_frame_width(rays, R, probe)
    if len(_FW_CACHE) >= 64:
        _FW_CACHE.clear()
    _FW_CACHE[key] = W
    return W


def _detect_frame_width(rays, R, probe):
    head = rays[: min(R, probe), :6].detach().float().cpu().numpy()
    o, d = head[:, :3], head[:, 3:6]
    tol = 1e-6 * (1.0 + np.abs(o[0]).max())
    if not np.all(np.abs(o - o[0]).max(axis=1) <= tol):
        return 0
    delta = d[1:] - d[:-1]
    n = np.linalg.norm(delta, axis=1)
    if not np.isfinite(n).all() or n[0] <= 0:
        return 0
    jump = np.linalg.norm(delta[1:] - delta[:-1], axis=1) > 0.5 * np.maximum(n[:-1], 1e-30)
    ks = np.nonzero(jump)[0]
    if ks.size == 0:
        return 0
    W = int(ks[0]) + 2          # jump[k] compares delta[k+1] (the wrap d[k+2] - d[k+1]) with delta[k]: the row ends at index k + 1
    if W < 8 or R % W or R // W < 8:
        return 0
    if 2 * W <= head.shape[0]:  # the second row: starts like the first, ends with the same jump
        if np.linalg.norm(delta[W] - delta[0]) > 0.5 * n[0] or not jump[2 * W - 2]:
            return 0
    if R > head.shape[0]:       # beyond the probe: one origin per 8-row band
        bo = rays[:: 8 * W, :3].detach().float().cpu().numpy()
        if not np.all(np.abs(bo - o[0]).max(axis=1) <= tol):
            return 0
    return W


def OctreeRender_trilinear_fast(rays, tensorf, chunk=4096, N_samples=-1, ndc_ray=False, white_bg=True, is_train=False,
                                device="cuda"):
    """Same signature and 5-tuple ``(rgb [R,3], None, depth [R], weights [R,N], z_vals [R,N])`` as renderer.py:28-42.

    Eval: the whole batch goes to the HIP renderer in one call (it sub-launches internally; ``chunk`` only bounds the
    reference's Python loop and is not needed here). Train: the reference draws one jitter vector per chunk from the CPU
    generator, so the chunk loop is kept to consume the RNG stream identically (the driver's batch is one chunk)."""
    if not is_train:
        # the driver's evaluation hands over all rays of one image: without an explicit hint the raster width is detected, so that
        # the frame takes the tile marcher (tensorf.auto_frame_width = False turns that off). The width travels as a call argument:
        # the module's own frame_width is not touched.
        fw = None
        if not tensorf.frame_width and not ndc_ray and getattr(tensorf, "auto_frame_width", True):
            fw = detect_frame_width(rays)
        rgb, depth, z, w = tensorf(rays, is_train=False, white_bg=white_bg, ndc_ray=ndc_ray, N_samples=N_samples, frame_width=fw)
        return rgb, None, depth, w, z
    outs = [[], [], [], []]
    n = rays.shape[0]
    for k in range(n // chunk + int(n % chunk > 0)):
        o = tensorf(rays[k * chunk:(k + 1) * chunk], is_train=True, white_bg=white_bg, ndc_ray=ndc_ray,
                    N_samples=N_samples)
        for lst, t in zip(outs, o):
            lst.append(t)
    # (one chunk — the driver's batch — is handed through as it is: torch.cat of a single tensor is a copy of 2 x 17 MB per step)
    rgb, depth, z, w = [(x[0] if len(x) == 1 else torch.cat(x)) if x[0] is not None else None for x in outs]
    return rgb, None, depth, w, z


class _FramePipe:
    """Frames of a trajectory are independent: frame k runs on HIP stream k % n (each stream has its own scratch buffer,
    tensorf.workspace), so the march of frame k + 1 — bound by VALU issue — runs beside the feature gather (L1 / texture addresser) and
    the MLP head (matrix cores) of frame k. Measured on the C2 frame: 2.62 -> 2.39 ms per frame with two frames in flight, 2.34 with
    three (tools/experiments/two_frames_in_flight.py), bitwise the same pictures. The kernels and their order inside a frame are
    unchanged; the library's per-field host state is only touched from the calling thread, call by call."""

    _STREAMS = {}     # device -> the side streams every pipe on that device shares

    def __init__(self, device, n=2):
        self.dev = torch.device(device)
        self.main = torch.cuda.current_stream(self.dev)
        # The scratch buffers are per (device, stream) and grow-only (tensorf.workspace: several GiB for a large frame): every pipe on
        # a device reuses the SAME side streams, so repeated evaluation calls during a training run hold n buffers, not one per
        # stream torch's pool hands out.
        pool = _FramePipe._STREAMS.setdefault(str(self.dev), [])
        while len(pool) < max(int(n), 1):
            pool.append(torch.cuda.Stream(self.dev))
        self.streams = pool[: max(int(n), 1)]
        for st in self.streams:
            st.wait_stream(self.main)          # whatever the caller queued (parameter updates, ray tensors) is visible
        self.k = 0

    def next(self):
        st = self.streams[self.k % len(self.streams)]
        self.k += 1
        return st

    def hand_over(self, tensors):
        """Make the frames' outputs safe to use (and to free) on the caller's stream."""
        for st in self.streams:
            self.main.wait_stream(st)
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(self.main)


@torch.no_grad()
def render_views(tensorf, poses, intrinsic, H, W, N_samples=-1, white_bg=True, frames_in_flight=2):
    """Device-side counterpart of the per-view loop of ``evaluation`` / ``evaluation_path`` (renderer.py:85-93,160-170)
    without its file I/O: for every camera-to-world pose generate the [H*W,6] rays on the GPU
    (dataLoader/scene_gen.py:44-45,92-94), render the whole frame in one call and return ``rgb [V,H,W,3]`` (clamped) and
    ``depth [V,H,W]``. Nothing crosses PCIe per view; ``frames_in_flight`` consecutive views run on alternating HIP streams
    (see _FramePipe; 1 = everything on the caller's stream). ``intrinsic`` = [fx, fy, cx, cy]."""
    from .ray_utils import generate_rays
    dev = tensorf.basis_mat.weight.device
    keep, keep_w = tensorf.materialize_weights, tensorf.frame_width
    tensorf.materialize_weights = False            # evaluation discards weights / z_vals (renderer.py:89)
    tensorf.frame_width = W                        # whole row-major frames: the 8x8-tile marcher applies
    rgbs, depths = [], []
    pipe = _FramePipe(dev, frames_in_flight) if frames_in_flight > 1 and len(poses) > 1 else None
    try:
        for c2w in poses:
            with torch.cuda.stream(pipe.next() if pipe else torch.cuda.current_stream(dev)):
                rays = generate_rays(H, W, intrinsic, c2w, device=dev)
                rgb, depth, _, _ = tensorf(rays, is_train=False, white_bg=white_bg, N_samples=N_samples)
                rgbs.append(rgb.clamp(0.0, 1.0).reshape(H, W, 3))
                depths.append(depth.reshape(H, W))
        if pipe:
            pipe.hand_over(rgbs + depths)
    finally:
        tensorf.materialize_weights, tensorf.frame_width = keep, keep_w
    return torch.stack(rgbs), torch.stack(depths)



@torch.no_grad()
def postprocess_frame(rgb, depth, near_far, push_depth=None, gt_rgb=None):
    """Device-side form of the per-view post-processing of ``evaluation`` (renderer.py:91-113) / ``evaluation_path``
    (:168-176) with ``visualize_depth_numpy`` (utils.py:241-257): returns ``(rgb8 [..,3] uint8, depth8 [..,3] uint8 JET in
    OpenCV's BGR order, psnr or None)`` as device tensors — one elementwise HIP kernel instead of a D2H copy + numpy per
    view. ``push_depth`` given: the ``evaluation`` form ``max(depth - push_depth + 0.8, 0)``; ``None``: ``evaluation_path``."""
    from . import _lib
    import ctypes as C
    lib = _lib.load()
    dev = rgb.device
    if dev.type != "cuda":
        raise _lib.T2NError("postprocess_frame runs on the MI355X only (no CPU fallback)")
    shape = tuple(depth.shape)
    rgb_c = rgb.reshape(-1, 3).contiguous().float()
    dep_c = depth.reshape(-1).contiguous().float()
    n = dep_c.numel()
    rgb8 = torch.empty(n, 3, dtype=torch.uint8, device=dev)
    dep8 = torch.empty(n, 3, dtype=torch.uint8, device=dev)
    sq = torch.zeros(1, dtype=torch.float64, device=dev) if gt_rgb is not None else None
    gt = gt_rgb.reshape(-1, 3).to(dev).contiguous().float() if gt_rgb is not None else None
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_frame_postprocess(_lib.ptr(rgb_c), _lib.ptr(dep_c), n, float(push_depth or 0.0), 0.8,
                                             1 if push_depth is not None else 0, float(near_far[0]), float(near_far[1]),
                                             _lib.ptr(rgb8), _lib.ptr(dep8), _lib.ptr(gt), _lib.ptr(sq),
                                             _lib.current_stream_ptr(dev)), "t2n_frame_postprocess")
    psnr = None
    if sq is not None:
        loss = float(sq.item()) / (3 * n)
        psnr = -10.0 * float(np.log(loss)) / float(np.log(10.0))
    return rgb8.reshape(shape + (3,)), dep8.reshape(shape + (3,)), psnr


@torch.no_grad()
def evaluation_frames(tensorf, poses, intrinsic, H, W, near_far, N_samples=-1, white_bg=True, push_depth=2.0, gt_rgbs=None):
    """``evaluation`` (renderer.py:45-140) without its file I/O, entirely on the device: rays from ``(c2w, intrinsics)``,
    one render call per view, post-processing kernel; returns ``(rgb8 [V,H,W,3], depth8 [V,H,W,3], PSNRs)``."""
    rgbs, depths = render_views(tensorf, poses, intrinsic, H, W, N_samples=N_samples, white_bg=white_bg)
    out_rgb, out_dep, psnrs = [], [], []
    for v in range(rgbs.shape[0]):
        r8, d8, p = postprocess_frame(rgbs[v], depths[v], near_far, push_depth=push_depth,
                                      gt_rgb=None if gt_rgbs is None else gt_rgbs[v])
        out_rgb.append(r8)
        out_dep.append(d8)
        if p is not None:
            psnrs.append(p)
    return torch.stack(out_rgb), torch.stack(out_dep), psnrs


# ---- the reference's evaluation loops with their own signatures (renderer.py:44-197) -----------------------------------------
# Only what the hot path produces is computed here: the render through `renderer`, the post-processing kernel (uint8 rgb, JET
# depth map, PSNR sum on the device) and the image files. SSIM / LPIPS are metric packages outside this path: they are called
# through the two hooks below when the integrator installs them (`text2nerf_amd.renderer.rgb_ssim = utils.rgb_ssim`), and skipped
# otherwise. Videos need imageio (absent from this image): written when it is importable, skipped with a notice when not.
rgb_ssim = None      # callable (img0 [H,W,3] np/tensor, img1, max_val) -> float           (utils.py:436-483)
rgb_lpips = None     # callable (np_gt [H,W,3], np_im [H,W,3], net_name, device) -> float  (utils.py:486-495)


def _imwrite(path, arr):
    try:
        import imageio
        imageio.imwrite(path, arr)
    except ImportError:
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(arr)).save(path)


def _mimwrite(path, frames, fps, quality):
    try:
        import imageio
    except ImportError:
        print(f"[text2nerf_amd] imageio is not installed: {path} not written ({len(frames)} frames were rendered)")
        return False
    imageio.mimwrite(path, np.stack(frames), fps=fps, quality=quality)
    return True


def _post_views(tensorf, rgb_map, depth_map, H, W, near_far, push_depth, gt_rgb):
    """One view's post-processing (renderer.py:91-113 / :168-176) on the field's device; returns host uint8 arrays + PSNR."""
    dev = tensorf.basis_mat.weight.device if hasattr(tensorf, "basis_mat") else rgb_map.device
    r8, d8, psnr = postprocess_frame(rgb_map.to(dev).reshape(H, W, 3), depth_map.to(dev).reshape(H, W), near_far,
                                     push_depth=push_depth, gt_rgb=gt_rgb)
    return r8.cpu().numpy(), d8.cpu().numpy(), psnr


def _eval_pipe(tensorf, n=2):
    """Two views in flight for the evaluation loops when the field lives on a GPU (see _FramePipe); None otherwise."""
    w = getattr(getattr(tensorf, "basis_mat", None), "weight", None)
    if w is None or w.device.type != "cuda":
        return None
    return _FramePipe(w.device, n)


class _on_stream:
    """``with torch.cuda.stream(st)`` that accepts None (no GPU / no pipelining: the caller's stream)."""

    def __init__(self, st):
        self.ctx = torch.cuda.stream(st) if st is not None else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)


class _no_materialised_weights:
    """`evaluation` discards weights / z_vals (renderer.py:89): do not write the two [R, N] tensors while it runs."""

    def __init__(self, tensorf):
        self.t, self.keep = tensorf, getattr(tensorf, "materialize_weights", None)

    def __enter__(self):
        if self.keep is not None:
            self.t.materialize_weights = False

    def __exit__(self, *exc):
        if self.keep is not None:
            self.t.materialize_weights = self.keep


@torch.no_grad()
def evaluation(test_dataset, tensorf, args, renderer, savePath=None, N_vis=5, prtx="", N_samples=-1, white_bg=True,
               ndc_ray=False, compute_extra_metrics=True, device="cuda", video_gen=False, N_iter=-1, preview=False,
               stitching=True):
    """renderer.py:44-140 with the same signature, view selection, file names and return value (the PSNR list; PSNR is taken
    only when ground truth exists AND ``compute_extra_metrics`` is set, as there). ``stitching`` is accepted and forced off like
    the reference does (:55). Per view: ONE ``renderer(...)`` call with all rays of the image, post-processing on the device."""
    PSNRs, rgb_maps, depth_maps = [], [], []
    ssims, l_alex, l_vgg = [], [], []
    stitching = False
    if savePath is not None:
        os.makedirs(savePath, exist_ok=True)
        os.makedirs(os.path.join(savePath, "rgbs"), exist_ok=True)
        os.makedirs(os.path.join(savePath, "depths"), exist_ok=True)
    if test_dataset.split == "train":
        if preview:
            all_rays, all_rgbs = test_dataset.all_rays_sprt_split, None
        else:
            all_rays = test_dataset.all_rays_gen_split[:N_iter + 1]
            all_rgbs = test_dataset.all_rgbs_gen_split[:N_iter + 1]
    elif test_dataset.split == "test" and N_iter >= 0:
        all_rays, all_rgbs = test_dataset.all_rays_split[:N_iter + 1], None
    else:
        all_rays, all_rgbs = test_dataset.all_rays_split, None
    near_far = test_dataset.near_far
    img_eval_interval = 1 if N_vis < 0 else max(all_rays.shape[0] // N_vis, 1)
    idxs = list(range(0, all_rays.shape[0], img_eval_interval))
    pipe = _eval_pipe(tensorf)

    def finish(p):
        """Post-processing, download and files of one rendered view, on the stream it was rendered on."""
        idx, st, rgb_map, depth_map, H, W = p
        with _on_stream(st):
            want_psnr = all_rgbs is not None and compute_extra_metrics
            gt_rgb = all_rgbs[idxs[idx]].view(H, W, 3) if want_psnr else None
            rgb8, depth8, psnr = _post_views(tensorf, rgb_map, depth_map, H, W, near_far, args.push_depth, gt_rgb)
            if want_psnr:
                PSNRs.append(psnr)
                if rgb_ssim is not None and rgb_lpips is not None:
                    rgb_f = rgb_map.clamp(0.0, 1.0).reshape(H, W, 3).cpu()
                    ssims.append(rgb_ssim(rgb_f, gt_rgb, 1))
                    l_alex.append(rgb_lpips(gt_rgb.numpy(), rgb_f.numpy(), "alex", tensorf.device))
                    l_vgg.append(rgb_lpips(gt_rgb.numpy(), rgb_f.numpy(), "vgg", tensorf.device))
        rgb_maps.append(rgb8)
        depth_maps.append(depth8)
        if savePath is not None:
            _imwrite(f"{savePath}/rgbs/{prtx}{idx:03d}_rgb.png", rgb8)
            _imwrite(f"{savePath}/depths/{prtx}{idx:03d}_depth.png", depth8)

    # view k + 1 is rendered (on the other stream) while view k is post-processed, downloaded and written
    pending = None
    with _no_materialised_weights(tensorf):
        for idx, samples in enumerate(all_rays[0::img_eval_interval]):
            W, H = test_dataset.img_wh
            rays = samples.view(-1, samples.shape[-1])
            st = pipe.next() if pipe else None
            with _on_stream(st):
                rgb_map, _, depth_map, _, _ = renderer(rays, tensorf, chunk=args.batch_size, N_samples=N_samples, ndc_ray=ndc_ray,
                                                       white_bg=white_bg, device=device)
            if pending is not None:
                finish(pending)
            pending = (idx, st, rgb_map, depth_map, H, W)
        if pending is not None:
            finish(pending)
    if video_gen:
        _mimwrite(f"{savePath}/{prtx}video.mp4", rgb_maps, fps=30, quality=9)
        _mimwrite(f"{savePath}/{prtx}depthvideo.mp4", depth_maps, fps=30, quality=9)
    return PSNRs


@torch.no_grad()
def evaluation_path(test_dataset, tensorf, c2ws, renderer, savePath=None, N_vis=5, prtx="", N_samples=-1, white_bg=False,
                    ndc_ray=False, compute_extra_metrics=True, device="cuda"):
    """renderer.py:142-197 with the same signature and files (``{prtx}NNN.png``, ``rgbd/{prtx}NNN.png`` = rgb | depth map side
    by side, the two videos when imageio exists). Rays come from ``get_rays(test_dataset.directions, c2w)`` (a HIP kernel here),
    optionally through ``ndc_rays_blender``; the render call passes ``chunk=8192`` like :166."""
    from .ray_utils import get_rays, ndc_rays_blender
    PSNRs, rgb_maps, depth_maps = [], [], []
    os.makedirs(savePath, exist_ok=True)
    os.makedirs(savePath + "/rgbd", exist_ok=True)
    near_far = test_dataset.near_far
    pipe = _eval_pipe(tensorf)

    def finish(p):
        idx, st, rgb_map, depth_map, H, W = p
        with _on_stream(st):
            rgb8, depth8, _ = _post_views(tensorf, rgb_map, depth_map, H, W, near_far, None, None)
        rgb_maps.append(rgb8)
        depth_maps.append(depth8)
        if savePath is not None:
            _imwrite(f"{savePath}/{prtx}{idx:03d}.png", rgb8)
            _imwrite(f"{savePath}/rgbd/{prtx}{idx:03d}.png", np.concatenate((rgb8, depth8), axis=1))

    pending = None
    with _no_materialised_weights(tensorf):
        for idx, c2w in enumerate(c2ws):
            W, H = test_dataset.img_wh
            c2w = torch.FloatTensor(np.asarray(c2w, dtype=np.float32))
            st = pipe.next() if pipe else None
            with _on_stream(st):
                rays_o, rays_d = get_rays(test_dataset.directions, c2w)
                if ndc_ray:
                    rays_o, rays_d = ndc_rays_blender(H, W, test_dataset.focal[0], 1.0, rays_o, rays_d)
                rays = torch.cat([rays_o, rays_d], 1)
                rgb_map, _, depth_map, _, _ = renderer(rays, tensorf, chunk=8192, N_samples=N_samples, ndc_ray=ndc_ray,
                                                       white_bg=white_bg, device=device)
            if pending is not None:
                finish(pending)
            pending = (idx, st, rgb_map, depth_map, H, W)
        if pending is not None:
            finish(pending)
    _mimwrite(f"{savePath}/{prtx}video.mp4", rgb_maps, fps=30, quality=8)
    _mimwrite(f"{savePath}/{prtx}depthvideo.mp4", depth_maps, fps=30, quality=8)
    return PSNRs     # always empty, as in the reference (no ground truth on a path; its mean.txt branch is unreachable)
