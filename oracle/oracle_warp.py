"""Oracle for SURVEY.md 8 f-3: CPU restatements (numpy) of the two image-space steps around the renderer in
`render_warping_inapinting` (text2nerf_main.py:102-141). TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing else);
the product path is the HIP code behind text2nerf_amd/warp.py. Pinned against goldens produced by the reference itself
(tests/golden/make_golden_warp.py -> warp.npz; tests/test_oracle_warp.py).

  depth_discontinuities / median_filter_pass / sparse_bilateral_filtering
      dataLoader/bilateral_filtering.py:5-35 (driver), :64-136 (vis_depth_discontinuity), :138-228 (bilateral_filter,
      discontinuity branch with mask=None: an unweighted median over the window's non-discontinuity pixels)
  forward_warp / bilinear_splat_warping_multiview
      scripts/Warper.py:21-186 (compute_transformed_points, bilinear_splatting), utils.py:83-119
"""
import numpy as np
from numpy.lib.stride_tricks import sliding_window_view


# ---- sparse_bilateral_filtering ---------------------------------------------------------------------------------------------
def depth_discontinuities(vis_depth, orig_depth, thr):
    """bilateral_filtering.py:17-20,78-97,115-118: 4-neighbour disparity jumps > thr on interior pixels (border = 0), union
    clipped to [0,1], plus every pixel whose ORIGINAL depth is 0."""
    with np.errstate(divide="ignore", invalid="ignore"):
        disp = (1.0 / vis_depth).astype(vis_depth.dtype)
        H, W = disp.shape
        over = np.zeros((H, W), np.float32)
        c = disp[1:-1, 1:-1]
        for nb in (disp[:-2, 1:-1], disp[2:, 1:-1], disp[1:-1, :-2], disp[1:-1, 2:]):   # up, below, left, right
            over[1:-1, 1:-1] += (np.abs(c - nb) > thr).astype(np.float32)
    over = np.clip(over, 0.0, 1.0)
    over[orig_depth == 0] = 1
    return over


def median_filter_pass(x, disc, window):
    """bilateral_filtering.py:138-196 with discontinuity_map given and mask=None. The border ring of `x` and `disc` is first
    replaced by its inner neighbours (:149-154); pixels whose window holds no discontinuity keep that value; the others take
    the weighted median of the window with weight 1/n on the n non-discontinuity pixels: values sorted ascending, fp32
    running sum of the weights, first element whose running sum exceeds 0.5 (np.digitize(0.5, cumsum))."""
    m = window // 2
    xi = np.pad(x[1:-1, 1:-1], 1, mode="edge")
    di = np.pad(disc[1:-1, 1:-1], 1, mode="edge")
    px = sliding_window_view(np.pad(xi, m, mode="edge"), (window, window)).reshape(x.shape + (-1,))
    pd = sliding_window_view(np.pad(di, m, mode="edge"), (window, window)).reshape(x.shape + (-1,))
    out = xi.copy()
    ys, xs = np.nonzero(pd.any(-1))
    for y, xx in zip(ys, xs):
        vals, hole = px[y, xx], (1.0 - pd[y, xx]).astype(np.float32)
        if hole.max() == 0:
            continue                                      # output = window centre = xi[y, xx]
        order = np.argsort(vals, kind="stable")
        w = (hole / hole.sum())[order]
        k = int(np.searchsorted(np.cumsum(w), 0.5, side="right"))
        out[y, xx] = vals[order][k]
    return out


def sparse_bilateral_filtering(depth, image, filter_size, depth_threshold, num_iter):
    """bilateral_filtering.py:5-35 (mask=None). Returns (photo, depth_kept, depth_states): the reference appends the SAME
    image array every iteration and filters it in place, so `save_images[-1]` is the image after ALL num_iter passes, while
    the depth array is re-bound, so `save_depths[-1]` is the depth after num_iter - 1 passes (the last depth pass is
    discarded) — both quirks are what the driver consumes (text2nerf_main.py:119-120)."""
    vis_depth, vis_image = depth.copy(), image.copy()
    states = []
    for i in range(num_iter):
        window = filter_size[i] if isinstance(filter_size, (list, tuple)) else filter_size
        states.append(vis_depth)
        disc = depth_discontinuities(vis_depth, depth, depth_threshold)
        vis_depth = median_filter_pass(vis_depth, disc, window)
        for ch in range(3):
            vis_image[:, :, ch] = median_filter_pass(vis_image[:, :, ch], disc, window)
    return vis_image, states[-1], states


# ---- DIBR forward warp ------------------------------------------------------------------------------------------------------
def transformed_points(depth1, T1, T2, K1, K2):
    """Warper.py:64-95 per pixel: X = depth * K1^-1 (x, y, 1); X' = (T2 T1^-1) X; p = K2 X'. Returns (u, v, z) float64 maps.
    dtype rules of the reference: inverses / the 4x4 product stay in the inputs' dtype (fp32 from the driver), the per-pixel
    products run in fp64."""
    h, w = depth1.shape
    T = np.matmul(T2, np.linalg.inv(T1))
    Ki = np.linalg.inv(K1)
    x, y = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    Ki, T, K2 = Ki.astype(np.float64), T.astype(np.float64), K2.astype(np.float64)
    d = depth1.astype(np.float64)
    cam = [d * (Ki[r, 0] * x + Ki[r, 1] * y + Ki[r, 2] * 1.0) for r in range(3)]
    w2 = [T[r, 0] * cam[0] + T[r, 1] * cam[1] + T[r, 2] * cam[2] + T[r, 3] * 1.0 for r in range(3)]
    p = [K2[r, 0] * w2[0] + K2[r, 1] * w2[1] + K2[r, 2] * w2[2] for r in range(3)]
    return p[0] / p[2], p[1] / p[2], p[2]


def splat(values, u, v, z, mask1=None):
    """Warper.py:97-186: inverse-bilinear splat of `values` [h,w,c] to (u, v) with weights proximity x mask / depth weight,
    depth weight = exp(50 * log(1 + clip(z,0,1000)) / max(log(1 + ...))), fp64 accumulation on an (h+2, w+2) canvas whose
    outer ring is cropped. Returns (sum / weight where weight > 0 else 0, weight > 0)."""
    h, w, c = values.shape
    ox, oy = u + 1.0, v + 1.0
    fx, fy = np.floor(ox).astype(np.int64), np.floor(oy).astype(np.int64)
    cx, cy = np.ceil(ox).astype(np.int64), np.ceil(oy).astype(np.int64)
    ox, oy = np.clip(ox, 0, w + 1), np.clip(oy, 0, h + 1)
    fx, cx = np.clip(fx, 0, w + 1), np.clip(cx, 0, w + 1)
    fy, cy = np.clip(fy, 0, h + 1), np.clip(cy, 0, h + 1)
    logd = np.log(1 + np.clip(z, 0, 1000))
    dw = np.exp(logd / logd.max() * 50)
    m = np.ones((h, w)) if mask1 is None else mask1.astype(np.float64)
    canvas = np.zeros((h + 2, w + 2, c), np.float64)
    wsum = np.zeros((h + 2, w + 2), np.float64)
    for (iy, ix, py, px) in ((fy, fx, 1 - (oy - fy), 1 - (ox - fx)), (cy, fx, 1 - (cy - oy), 1 - (ox - fx)),
                             (fy, cx, 1 - (oy - fy), 1 - (cx - ox)), (cy, cx, 1 - (cy - oy), 1 - (cx - ox))):
        wt = py * px * m / dw
        np.add.at(canvas, (iy, ix), values * wt[:, :, None])
        np.add.at(wsum, (iy, ix), wt)
    canvas, wsum = canvas[1:-1, 1:-1], wsum[1:-1, 1:-1]
    known = wsum > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        out = np.where(known[:, :, None], canvas / wsum[:, :, None], 0)
    return out, known


def forward_warp(frame1_u8, depth1, T1, T2, K1, K2=None, mask1=None):
    """Warper.forward_warp (Warper.py:21-62): (warped uint8 frame, known mask, warped fp64 depth, flow)."""
    K2 = K1 if K2 is None else K2
    u, v, z = transformed_points(depth1, T1, T2, K1, K2)
    img, known = splat(frame1_u8.astype(np.float64), u, v, z, mask1)
    img8 = np.round(np.clip(img, 0, 255)).astype(np.uint8)
    dep, _ = splat(z[:, :, None], u, v, z, mask1)
    h, w = depth1.shape
    x, y = np.meshgrid(np.arange(w), np.arange(h))
    return img8, known, dep[:, :, 0], np.stack([u - x, v - y], -1)


def bilinear_splat_warping_multiview(rgbs, depths, poses, pose_tar, H, W, intrinsic, masks=None):
    """utils.py:83-119: warp every known view into the target, earlier views win, white where nothing landed.
    Returns (mask_final [H,W] int, image [H,W,3] fp32 in [0,1], depth [H,W] fp64)."""
    T2 = np.linalg.inv(pose_tar)
    K = np.eye(3, dtype=np.float32)
    K[0, 0], K[1, 1], K[0, 2], K[1, 2] = intrinsic[0], intrinsic[1], intrinsic[2], intrinsic[3]
    filled = np.zeros((H, W), bool)
    image = np.zeros((H, W, 3), np.uint8)
    depth = np.zeros((H, W))
    for v in range(len(rgbs)):
        f8, known, d2, _ = forward_warp((rgbs[v] * 255).astype(np.uint8), depths[v], np.linalg.inv(poses[v]), T2, K, None,
                                        None if masks is None else masks[v])
        new = known & ~filled
        image[new] = f8[new]
        depth[new] = d2[new]
        filled |= known
    image[~filled] = 255
    return filled.astype(np.int64), (image / 255).astype(np.float32), depth


# ---- hole filling -------------------------------------------------------------------------------------------------------------
FILL_W5 = np.array([[1, 1, 1.5, 1, 1], [1, 1.5, 3, 1.5, 1], [1.5, 3, 0, 3, 1.5], [1, 1.5, 3, 1.5, 1], [1, 1, 1.5, 1, 1]], np.float32)


def dibr_filter_mask2(image, known, depth=None, thr=0.65):
    """utils.py:393-409: one raster scan (rows 2..H-3, cols 2..W-3) over the unknown pixels; a pixel whose 5x5 neighbourhood
    is known to more than `thr` (weights FILL_W5 / 36) takes the mean of its known 3x3 neighbours (colour and depth) and
    becomes known IMMEDIATELY, so later pixels of the scan see it. In-place semantics restated on copies; fp64 means."""
    image, known = image.copy(), known.copy()
    depth = None if depth is None else depth.copy()
    H, W, _ = image.shape
    tot = float(FILL_W5.sum())
    for i in range(2, H - 2):
        for j in range(2, W - 2):
            if known[i, j] != 0:
                continue
            if float((known[i - 2:i + 3, j - 2:j + 3] * FILL_W5).sum()) / tot <= thr:
                continue
            k3 = known[i - 1:i + 2, j - 1:j + 2].astype(np.float64)
            n = k3.sum()
            for c in range(3):
                image[i, j, c] = (image[i - 1:i + 2, j - 1:j + 2, c].astype(np.float64) * k3).sum() / n
            if depth is not None:
                depth[i, j] = (depth[i - 1:i + 2, j - 1:j + 2] * k3).sum() / n
            known[i, j] = 1
    return (image, known) if depth is None else (image, known, depth)


FILL_W3 = np.array([[1, 3, 1], [3, 0, 3], [1, 3, 1]], np.int64)


def dibr_filter_mask(image, known):
    """utils.py:345-392 (no call site in the driver): five in-place stages over the merged warp. (1) the scan of dibr_filter_mask2 with
    threshold 0.6 and no depth; (2) a second raster scan over rows 1..H-2, cols 1..W-2: an unknown pixel whose 3x3 neighbourhood is known
    to more than 0.5 (weights FILL_W3 / 16) takes the mean of its known 3x3 neighbours and becomes known immediately; (3) the four border
    lines — top row, bottom row, left column, right column, in that order — copy their inner neighbour where it is known; (4) an erase
    scan in raster order: a KNOWN pixel whose 3x3 neighbourhood is known to less than 0.45 is set to 255 and becomes unknown
    immediately. In-place semantics restated on copies; fp64 means."""
    image, known = dibr_filter_mask2(image, known, None, thr=0.6)
    H, W, _ = image.shape
    for i in range(1, H - 1):
        for j in range(1, W - 1):
            if known[i, j] != 0 or int((known[i - 1:i + 2, j - 1:j + 2] * FILL_W3).sum()) / 16.0 <= 0.5:
                continue
            k3 = known[i - 1:i + 2, j - 1:j + 2].astype(np.float64)
            n = k3.sum()
            for c in range(3):
                image[i, j, c] = (image[i - 1:i + 2, j - 1:j + 2, c].astype(np.float64) * k3).sum() / n
            known[i, j] = 1
    for i, ii in ((0, 1), (H - 1, H - 2)):
        for j in range(W):
            if known[i, j] == 0 and known[ii, j] > 0:
                image[i, j, :] = image[ii, j, :]
                known[i, j] = 1
    for j, jj in ((0, 1), (W - 1, W - 2)):
        for i in range(H):
            if known[i, j] == 0 and known[i, jj] > 0:
                image[i, j, :] = image[i, jj, :]
                known[i, j] = 1
    for i in range(1, H - 1):
        for j in range(1, W - 1):
            if known[i, j] == 1 and int((known[i - 1:i + 2, j - 1:j + 2] * FILL_W3).sum()) / 16.0 < 0.45:
                image[i, j, :] = 255
                known[i, j] = 0
    return image, known


def align_depth_global(depth_rendered, depth_est, pixel_sample, push_depth):
    """text2nerf_main.py:241-270 (the part after the pixel list has been drawn with random.sample, :234-240): global scale from the
    ratios of depth differences between CONSECUTIVE sampled pixels — kept when finite, non-negative and within 5 |thresh - 1| of 1,
    thresh = (max rendered - push) / (max estimate - push); fallback thresh — then the mean shift of the scaled estimate against the
    rendered depth over the samples within 2 |max scaled - max rendered|; fallback that difference. numpy dtypes as the driver has
    them: depth_rendered float32, depth_est float64. Returns (scale, shift, depth_shift)."""
    dr, de = np.asarray(depth_rendered), np.asarray(depth_est)
    ps = np.asarray(pixel_sample)
    thresh = (dr.max() - push_depth) / (de.max() - push_depth)
    scales = []
    for ii in range(len(ps) - 1):
        y1, x1 = ps[ii]
        y2, x2 = ps[ii + 1]
        dd1 = dr[y1, x1] - dr[y2, x2]
        dd2 = de[y1, x1] - de[y2, x2]
        with np.errstate(divide="ignore", invalid="ignore"):
            ss = dd1 / (dd2 + 1e-8)
        if not np.isfinite(ss) or abs(ss - 1) > 5 * abs(thresh - 1) or ss < 0:
            continue
        scales.append(ss)
    if len(scales) == 0:
        scales.append(thresh)
    scale = np.average(np.stack(scales))
    ds = de * scale
    thresh2 = ds.max() - dr.max()
    shifts = []
    for ii in range(len(ps)):
        y1, x1 = ps[ii]
        ss = ds[y1, x1] - dr[y1, x1]
        if abs(ss) > 2 * abs(thresh2):
            continue
        shifts.append(ss)
    if len(shifts) == 0:
        shifts.append(thresh2)
    shift = np.average(np.stack(shifts))
    return float(scale), float(shift), ds - shift
