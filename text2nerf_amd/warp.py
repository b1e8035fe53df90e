"""the image-space steps either side of the renderer inside `render_warping_inapinting`
(text2nerf_main.py:102-141) with the reference's call surface, running as HIP kernels (csrc/t2n_image.hip):

* ``sparse_bilateral_filtering`` — dataLoader/bilateral_filtering.py:5-35 (an O(H W) Python loop in the reference)
* ``bilinear_splat_warping_multiview`` — utils.py:83-119 over ``Warper.forward_warp`` (scripts/Warper.py:21-186, numpy add.at)
* ``dibr_filter_mask2`` — utils.py:393-409, the raster-order hole filling (skewed wavefronts on the GPU); ``dibr_filter_mask`` —
  utils.py:345-392, its four-stage sibling (unused by the driver)

Inputs may be numpy arrays (as in the driver) or torch tensors; numpy in -> numpy out. No CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import T2NError


def _dev(device=None):
    if not torch.cuda.is_available():
        raise T2NError("text2nerf_amd.warp runs on an MI355X only (no CPU fallback)")
    return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def _to(x, dev, dtype):
    t = torch.from_numpy(np.ascontiguousarray(x)) if isinstance(x, np.ndarray) else x
    return t.to(device=dev, dtype=dtype).contiguous()


def sparse_bilateral_filtering(depth, image, filter_size=[7, 7, 5, 5, 5], depth_threshold=0.04, num_iter=5, HR=False, mask=None,
                               device=None):
    """Same signature / return as dataLoader/bilateral_filtering.py:5: ``(save_images, save_depths)``, lists of length
    ``num_iter``. ``save_depths[i]`` is the depth before pass i; every ``save_images`` entry is the SAME array holding the
    image after all passes (the reference appends one array and filters it in place). ``mask`` / ``HR`` are not used by the
    driver and are rejected."""
    if mask is not None or HR:
        raise T2NError("sparse_bilateral_filtering: mask / HR are not on the Text2NeRF path and are not implemented")
    lib = _lib.load()
    as_numpy = isinstance(depth, np.ndarray)
    dev = _dev(device if device is not None else (None if as_numpy else depth.device))
    d = _to(depth, dev, torch.float32)
    im = _to(image, dev, torch.float32)
    H, W = d.shape
    if im.shape != (H, W, 3):
        raise T2NError(f"image shape {tuple(im.shape)} does not match depth {(H, W)}")
    sizes = [int(filter_size[i]) if isinstance(filter_size, (list, tuple)) else int(filter_size) for i in range(num_iter)]
    arr = (C.c_int * num_iter)(*sizes)
    photo = torch.empty(H, W, 3, device=dev, dtype=torch.float32)
    states = torch.empty(num_iter, H, W, device=dev, dtype=torch.float32)
    ws = torch.empty(int(lib.t2n_image_filter_workspace_bytes(H, W)), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_sparse_bilateral_filtering(_lib.ptr(d), _lib.ptr(im), H, W, arr, num_iter, float(depth_threshold),
                                                      _lib.ptr(photo), _lib.ptr(states), _lib.ptr(ws), ws.numel(),
                                                      _lib.current_stream_ptr(dev)), "t2n_sparse_bilateral_filtering")
    if as_numpy:
        p = photo.cpu().numpy()
        s = states.cpu().numpy()
        return [p] * num_iter, [s[i] for i in range(num_iter)]
    return [photo] * num_iter, [states[i] for i in range(num_iter)]


def bilinear_splat_warping_multiview(rgbs, depths, poses, pose_tar, H, W, intrinsic, masks=None, device=None):
    """Same signature / return as utils.py:83: ``(mask_final [H,W] int, output_image [H,W,3] fp32 in [0,1], output_depth
    [H,W] fp64)``. rgbs in [0,1]; poses are camera-to-world 4x4 (inverted here exactly like the reference: numpy, the
    inputs' dtype)."""
    lib = _lib.load()
    as_numpy = isinstance(rgbs[0], np.ndarray)
    dev = _dev(device if device is not None else (None if as_numpy else rgbs[0].device))
    pose_tar = np.asarray(pose_tar.cpu() if isinstance(pose_tar, torch.Tensor) else pose_tar)
    T2 = np.linalg.inv(pose_tar)
    K = np.eye(3).astype(np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intrinsic[0], intrinsic[1], intrinsic[2], intrinsic[3]
    Ki = np.linalg.inv(K)
    filled = torch.zeros(H, W, dtype=torch.uint8, device=dev)
    img8 = torch.zeros(H, W, 3, dtype=torch.uint8, device=dev)
    dep = torch.zeros(H, W, dtype=torch.float64, device=dev)
    ws = torch.empty(int(lib.t2n_warp_workspace_bytes(H, W)), dtype=torch.uint8, device=dev)
    d9 = lambda m: (C.c_double * 9)(*np.asarray(m, np.float64).reshape(-1)[:9])       # noqa: E731
    with torch.cuda.device(dev):
        st = _lib.current_stream_ptr(dev)
        for v in range(len(rgbs)):
            pv = np.asarray(poses[v].cpu() if isinstance(poses[v], torch.Tensor) else poses[v])
            T = np.matmul(T2, np.linalg.inv(np.linalg.inv(pv)))       # transformation2 @ inv(transformation1), Warper.py:75
            T12 = (C.c_double * 12)(*np.asarray(T, np.float64)[:3, :4].reshape(-1))
            m1 = None if masks is None else _to(masks[v], dev, torch.uint8)
            rgb = _to(rgbs[v], dev, torch.float32)
            d = _to(depths[v], dev, torch.float32)
            _lib.check(lib.t2n_warp_view(_lib.ptr(rgb), _lib.ptr(d), _lib.ptr(m1), H, W, d9(Ki), T12, d9(K), _lib.ptr(filled),
                                         _lib.ptr(img8), _lib.ptr(dep), _lib.ptr(ws), ws.numel(), st), "t2n_warp_view")
        out_img = torch.empty(H, W, 3, dtype=torch.float32, device=dev)
        out_mask = torch.empty(H, W, dtype=torch.int64, device=dev)
        _lib.check(lib.t2n_warp_finish(_lib.ptr(filled), _lib.ptr(img8), H, W, _lib.ptr(out_img), _lib.ptr(out_mask), st),
                   "t2n_warp_finish")
    if as_numpy:
        return out_mask.cpu().numpy(), out_img.cpu().numpy(), dep.cpu().numpy()
    return out_mask, out_img, dep


def dibr_filter_mask2(output_image, myMap, output_depth=None, device=None):
    """Same signature / return as utils.py:393: ``(output_image, myMap[, output_depth])`` after the raster-order hole filling.
    The reference mutates its numpy arguments in place; here the (new) results are returned — the driver rebinds all three
    (text2nerf_main.py:135)."""
    lib = _lib.load()
    as_numpy = isinstance(output_image, np.ndarray)
    dev = _dev(device if device is not None else (None if as_numpy else output_image.device))
    img = _to(output_image, dev, torch.float32).clone()
    known = _to(myMap, dev, torch.int32).clone()
    H, W = known.shape
    dep = None if output_depth is None else _to(output_depth, dev, torch.float64).clone()
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_dibr_filter_mask2(_lib.ptr(img), _lib.ptr(known), _lib.ptr(dep), H, W, 0.65, _lib.current_stream_ptr(dev)),
                   "t2n_dibr_filter_mask2")
    if as_numpy:
        m = known.cpu().numpy().astype(np.asarray(myMap).dtype)
        out = (img.cpu().numpy(), m)
        return out if dep is None else out + (dep.cpu().numpy(),)
    out = (img, known.to(myMap.dtype))
    return out if dep is None else out + (dep,)


def dibr_filter_mask(output_image, myMap, device=None):
    """Same signature / return as utils.py:345: ``(output_image, myMap)`` after the 5x5 fill scan (threshold 0.6), the 3x3 fill scan, the
    border lines and the erase scan (pixels set to 255 / unknown). The reference's driver does not call it; it is here for scripts that
    do. Results are returned, not written into the numpy arguments."""
    lib = _lib.load()
    as_numpy = isinstance(output_image, np.ndarray)
    dev = _dev(device if device is not None else (None if as_numpy else output_image.device))
    img = _to(output_image, dev, torch.float32).clone()
    known = _to(myMap, dev, torch.int32).clone()
    H, W = known.shape
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_dibr_filter_mask(_lib.ptr(img), _lib.ptr(known), H, W, _lib.current_stream_ptr(dev)), "t2n_dibr_filter_mask")
    if as_numpy:
        return img.cpu().numpy(), known.cpu().numpy().astype(np.asarray(myMap).dtype)
    return img, known.to(myMap.dtype)


def align_depth_global(depth_rendered, depth_est, pixel_sample, push_depth=2.0, device=None):
    """The global stage of the depth alignment in ``render_warping_inapinting`` (text2nerf_main.py:241-270) on the device: scale and
    shift of the monocular estimate against the rendered depth over the caller's sampled pixel list (``random.sample`` of the filled
    pixels, :234-240 — host-side and the caller's, like the reference's). Returns ``(scale, shift, depth_shift)`` with ``depth_shift``
    a float32 [H,W] device tensor (= depth_est * scale - shift) and scale / shift Python floats (one 32-byte read-back)."""
    lib = _lib.load()
    dev = _dev(device)
    dr = _to(depth_rendered, dev, torch.float32)
    de = _to(depth_est, dev, torch.float32)
    ps = torch.as_tensor(np.asarray(pixel_sample, dtype=np.int32).reshape(-1, 2)).to(dev).contiguous()
    H, W = dr.shape
    out = torch.empty(H, W, device=dev, dtype=torch.float32)
    ss = torch.empty(4, device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        _lib.check(lib.t2n_depth_align_global(_lib.ptr(dr), _lib.ptr(de), H, W, _lib.ptr(ps), int(ps.shape[0]), float(push_depth),
                                              _lib.ptr(out), _lib.ptr(ss), _lib.current_stream_ptr(dev)), "t2n_depth_align_global")
    s = ss.cpu().tolist()
    return s[0], s[1], out
