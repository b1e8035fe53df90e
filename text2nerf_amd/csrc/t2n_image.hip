// Image-space steps either side of the renderer in `render_warping_inapinting` (SURVEY.md 8 f-3,
// text2nerf_main.py:102-141), as HBM-bound stencil / scatter kernels:
//   sparse_bilateral_filtering   dataLoader/bilateral_filtering.py:5-35 (driver), :64-136 (vis_depth_discontinuity),
//                                :138-228 (bilateral_filter, discontinuity branch, mask=None): an O(H W) Python loop in the
//                                reference (seconds per view)
//   forward warp + merge         scripts/Warper.py:21-186 (compute_transformed_points, bilinear_splatting: numpy add.at in
//                                fp64), utils.py:83-119 (bilinear_splat_warping_multiview)
#include "t2n_internal.h"

namespace t2n {

// ---- depth-aware median filter ------------------------------------------------------------------------------------------------
// planar state [4][H][W]: channel 0 = depth, 1..3 = r, g, b
__global__ __launch_bounds__(256) void k_img_pack(const float* __restrict__ depth, const float* __restrict__ image, int n, float* st) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    st[t] = depth[t];
    st[n + t] = image[t * 3]; st[2 * n + t] = image[t * 3 + 1]; st[3 * n + t] = image[t * 3 + 2];
}
__global__ __launch_bounds__(256) void k_img_unpack(const float* __restrict__ st, int n, float* image) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    image[t * 3] = st[n + t]; image[t * 3 + 1] = st[2 * n + t]; image[t * 3 + 2] = st[3 * n + t];
}

// bilateral_filtering.py:17-20,78-97,115-118: |1/d - 1/d_neighbour| > thr for any 4-neighbour (interior pixels only), or
// original depth == 0. inf - inf = nan compares false, like numpy.
__global__ __launch_bounds__(256) void k_img_disc(const float* __restrict__ vis, const float* __restrict__ orig, int H, int W, float thr,
                                                  unsigned char* disc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W) return;
    const int y = t / W, x = t - y * W;
    bool d = false;
    if (y >= 1 && y <= H - 2 && x >= 1 && x <= W - 2) {
        const float c = 1.f / vis[t];
        const float u = 1.f / vis[t - W], b = 1.f / vis[t + W], l = 1.f / vis[t - 1], r = 1.f / vis[t + 1];
        d = (fabsf(c - u) > thr) || (fabsf(c - b) > thr) || (fabsf(c - l) > thr) || (fabsf(c - r) > thr);
    }
    if (orig[t] == 0.f) d = true;
    disc[t] = d ? 1 : 0;
}

// One pass of the filter on all four channels (blockIdx.z). 16x16 pixels per workgroup, window <= 7: the (16+6)^2 halo tile
// of values and flags is staged in LDS. The border ring of the image is replaced by its inner neighbours before the
// edge padding (:149-154), i.e. every read is at (clamp(y,1,H-2), clamp(x,1,W-2)). Pixels whose window holds no
// discontinuity keep the (ring-replaced) value; the others take the t-th smallest of the n non-discontinuity values,
// t = first count whose fp32 running sum of 1/n exceeds 0.5 (np.digitize(0.5, np.cumsum(coef[order]))).
constexpr int kMedT = 16, kMedHalo = 3, kMedS = kMedT + 2 * kMedHalo;
__global__ __launch_bounds__(256) void k_img_median(const float* __restrict__ in, const unsigned char* __restrict__ disc, int H, int W,
                                                    int window, float* __restrict__ out) {
    __shared__ float sv[kMedS * kMedS];
    __shared__ unsigned char sd[kMedS * kMedS];
    const size_t plane = (size_t)blockIdx.z * H * W;
    const int bx = blockIdx.x * kMedT, by = blockIdx.y * kMedT;
    for (int i = threadIdx.x; i < kMedS * kMedS; i += 256) {
        const int ly = i / kMedS, lx = i - ly * kMedS;
        const int y = min(max(by + ly - kMedHalo, 1), H - 2), x = min(max(bx + lx - kMedHalo, 1), W - 2);
        sv[i] = in[plane + (size_t)y * W + x];
        sd[i] = disc[(size_t)y * W + x];
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = bx + tx, y = by + ty;
    if (x >= W || y >= H) return;
    const int m = window / 2;
    const int c0 = (ty + kMedHalo) * kMedS + tx + kMedHalo;
    int n = 0, nd = 0;
    for (int dy = -m; dy <= m; ++dy)
        for (int dx = -m; dx <= m; ++dx) {
            const int d = sd[c0 + dy * kMedS + dx];
            nd += d; n += 1 - d;
        }
    float res = sv[c0];
    if (nd > 0 && n > 0) {
        const float w = 1.0f / (float)n;
        float cum = 0.f;
        int t = 0;
        while (t < n) { cum = cum + w; ++t; if (cum > 0.5f) break; }   // t-th smallest (1-based)
        // rank selection among the non-discontinuity values; ties broken by window position (any order gives the same value)
        int idx_i = 0;
        for (int dy = -m; dy <= m; ++dy)
            for (int dx = -m; dx <= m; ++dx, ++idx_i) {
                const int o = c0 + dy * kMedS + dx;
                if (sd[o]) continue;
                const float vi = sv[o];
                int rank = 0, idx_k = 0;
                for (int ey = -m; ey <= m; ++ey)
                    for (int ex = -m; ex <= m; ++ex, ++idx_k) {
                        const int p = c0 + ey * kMedS + ex;
                        if (sd[p]) continue;
                        const float vk = sv[p];
                        rank += (vk < vi) | ((vk == vi) & (idx_k < idx_i));
                    }
                if (rank == t - 1) res = vi;
            }
    }
    out[plane + (size_t)y * W + x] = res;
}

// ---- DIBR forward warp ----------------------------------------------------------------------------------------------------------
struct WarpMats { double Ki[9], T[12], K2[9]; };

// Warper.py:64-95: X = depth * Ki (x, y, 1); X' = T (X, 1); p = K2 X'  ->  (u, v, z) in fp64; also max log(1 + clip(z, 0, 1000))
__global__ __launch_bounds__(256) void k_warp_points(const float* __restrict__ depth, int H, int W, const WarpMats M, double* uvz,
                                                     unsigned long long* logmax_bits) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    double lg = 0.0;
    if (t < H * W) {
        const int yi = t / W, xi = t - yi * W;
        const double x = (double)xi, y = (double)yi, d = (double)depth[t];
        double cam[3], w2[3], p[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) cam[r] = d * (M.Ki[r * 3] * x + M.Ki[r * 3 + 1] * y + M.Ki[r * 3 + 2] * 1.0);
#pragma unroll
        for (int r = 0; r < 3; ++r) w2[r] = M.T[r * 4] * cam[0] + M.T[r * 4 + 1] * cam[1] + M.T[r * 4 + 2] * cam[2] + M.T[r * 4 + 3] * 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) p[r] = M.K2[r * 3] * w2[0] + M.K2[r * 3 + 1] * w2[1] + M.K2[r * 3 + 2] * w2[2];
        uvz[(size_t)t * 3] = p[0] / p[2]; uvz[(size_t)t * 3 + 1] = p[1] / p[2]; uvz[(size_t)t * 3 + 2] = p[2];
        lg = log(1.0 + fmin(fmax(p[2], 0.0), 1000.0));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lg = fmax(lg, __shfl_xor(lg, o));
    if ((threadIdx.x & 63) == 0 && lg > 0.0) atomicMax(logmax_bits, (unsigned long long)__double_as_longlong(lg));   // lg >= 0: bit order = value order
}

// Warper.py:97-160: inverse-bilinear splat of (r, g, b, z) and the weight into the (H+2, W+2) canvas, fp64 atomics
__global__ __launch_bounds__(256) void k_warp_splat(const float* __restrict__ rgb, const unsigned char* __restrict__ mask1, int H, int W,
                                                    const double* __restrict__ uvz, const unsigned long long* logmax_bits, double* canvas) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W) return;
    const double u = uvz[(size_t)t * 3], v = uvz[(size_t)t * 3 + 1], z = uvz[(size_t)t * 3 + 2];
    const double logmax = __longlong_as_double((long long)*logmax_bits);
    double ox = u + 1.0, oy = v + 1.0;
    // floor / ceil -> integer -> clip, in the reference's order (a NaN / huge coordinate clips to the canvas ring)
    const double fxd = floor(ox), fyd = floor(oy), cxd = ceil(ox), cyd = ceil(oy);
    auto to_idx = [](double a, int hi) { if (!(a > 0.0)) return 0; if (a > (double)hi) return hi; return (int)a; };
    const int fx = to_idx(fxd, W + 1), cx = to_idx(cxd, W + 1), fy = to_idx(fyd, H + 1), cy = to_idx(cyd, H + 1);
    ox = fmin(fmax(ox, 0.0), (double)(W + 1)); oy = fmin(fmax(oy, 0.0), (double)(H + 1));
    const double logd = log(1.0 + fmin(fmax(z, 0.0), 1000.0));
    const double dw = exp(logd / logmax * 50.0);
    const double mk = mask1 ? (mask1[t] ? 1.0 : 0.0) : 1.0;
    double val[4];
#pragma unroll
    for (int k = 0; k < 3; ++k) val[k] = (double)(unsigned char)(int)(rgb[(size_t)t * 3 + k] * 255.f);   // (rgb * 255).astype(uint8)
    val[3] = z;
    const int iy[4] = {fy, cy, fy, cy}, ix[4] = {fx, fx, cx, cx};
    const double py[4] = {1.0 - (oy - fy), 1.0 - (cy - oy), 1.0 - (oy - fy), 1.0 - (cy - oy)};
    const double px[4] = {1.0 - (ox - fx), 1.0 - (ox - fx), 1.0 - (cx - ox), 1.0 - (cx - ox)};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const double wt = py[q] * px[q] * mk / dw;
        double* c = canvas + ((size_t)iy[q] * (W + 2) + ix[q]) * 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(c + k, val[k] * wt);
        atomicAdd(c + 4, wt);
    }
}

// Warper.py:162-186 + utils.py:106-113: normalise, round the image to uint8 (half-to-even), earlier views win
__global__ __launch_bounds__(256) void k_warp_resolve(const double* __restrict__ canvas, int H, int W, unsigned char* filled,
                                                      unsigned char* image, double* depth) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * W) return;
    const int y = t / W, x = t - y * W;
    const double* c = canvas + ((size_t)(y + 1) * (W + 2) + (x + 1)) * 5;
    const double w = c[4];
    if (!(w > 0.0) || filled[t]) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) image[(size_t)t * 3 + k] = (unsigned char)(int)rint(fmin(fmax(c[k] / w, 0.0), 255.0));
    depth[t] = c[3] / w;
    filled[t] = 2;   // newly filled by this view (normalised to 1 by k_warp_mark)
}
__global__ __launch_bounds__(256) void k_warp_mark(unsigned char* filled, int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < n && filled[t]) filled[t] = 1;
}
// utils.py:115-118: white where nothing landed, image / 255 -> fp32
__global__ __launch_bounds__(256) void k_warp_finish(const unsigned char* __restrict__ filled, const unsigned char* __restrict__ image, int n,
                                                     float* out_image, long long* out_mask) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const bool f = filled[t] != 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) out_image[(size_t)t * 3 + k] = (float)((double)(f ? image[(size_t)t * 3 + k] : 255) / 255.0);
    if (out_mask) out_mask[t] = f ? 1 : 0;
}

// ---- hole filling: dibr_filter_mask2 (utils.py:393-409) -------------------------------------------------------------------------
// The reference is ONE raster scan with in-place updates: a filled pixel counts as known for every later pixel. Pixel (i, j)
// reads the known-map in rows i-2..i+2 x cols j-2..j+2 (earlier rows / earlier columns of its own row must already be
// final, later ones still original) — so all pixels on a skewed front t = j + 3 i are mutually independent and the scan is
// a sequence of (W - 5) + 3 (H - 5) + 1 fronts. One workgroup walks the fronts (barrier + agent-scope loads between them:
// the values cross waves through L2); ~2000 fronts of <= H pixels for a 512^2 frame.
__device__ __forceinline__ int ld_i(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_d(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// numpy's pairwise sum of 9 values: eight accumulators combined as a tree, then the ninth
__device__ __forceinline__ double np_sum9(const double (&a)[9]) { return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + a[8]; }
__global__ __launch_bounds__(1024) void k_fill_fronts(float* image, int* known, double* depth, int H, int W, int twice_thr36) {
    // 2 x (5x5 weights): 1 -> 2, 1.5 -> 3, 3 -> 6, centre 0
    const int w5[25] = {2, 2, 3, 2, 2, 2, 3, 6, 3, 2, 3, 6, 0, 6, 3, 2, 3, 6, 3, 2, 2, 2, 3, 2, 2};
    const int t0 = 2 + 3 * 2, t1 = (W - 3) + 3 * (H - 3);
    for (int t = t0; t <= t1; ++t) {
        // rows i with 2 <= j = t - 3 i <= W - 3
        const int ilo = max(2, (t - (W - 3) + 2) / 3), ihi = min(H - 3, (t - 2) / 3);
        for (int i = ilo + (int)threadIdx.x; i <= ihi; i += blockDim.x) {
            const int j = t - 3 * i;
            if (j < 2 || j > W - 3) continue;
            if (ld_i(known + i * W + j) != 0) continue;
            int s2 = 0;
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
                for (int dx = -2; dx <= 2; ++dx) s2 += w5[(dy + 2) * 5 + dx + 2] * (ld_i(known + (i + dy) * W + j + dx) != 0 ? 1 : 0);
            if (s2 <= twice_thr36) continue;     // sum / 36 > thr  <=>  2 sum > 72 thr (sums are multiples of 0.5: exact)
            double k3[9], v[9];
            double n = 0.0;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                k3[q] = (double)ld_i(known + (i + q / 3 - 1) * W + j + q % 3 - 1);
            }
            n = np_sum9(k3);
            for (int c = 0; c < 3; ++c) {
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = (double)ld_f(image + ((size_t)(i + q / 3 - 1) * W + j + q % 3 - 1) * 3 + c) * k3[q];
                image[((size_t)i * W + j) * 3 + c] = (float)(np_sum9(v) / n);
            }
            if (depth) {
#pragma unroll
                for (int q = 0; q < 9; ++q) v[q] = ld_d(depth + (size_t)(i + q / 3 - 1) * W + j + q % 3 - 1) * k3[q];
                depth[(size_t)i * W + j] = np_sum9(v) / n;
            }
            known[i * W + j] = 1;
        }
        __threadfence();
        __syncthreads();
    }
}

// ---- hole filling: dibr_filter_mask (utils.py:345-392; no call site in the driver) ---------------------------------------------------
// After the 5x5 scan above (threshold 0.6, no depth): a 3x3 fill scan, the four border lines, a 3x3 erase scan — each raster scan in
// place. A 3x3 scan's pixel (i, j) needs rows < i and (i, j - 1) final, (i, j + 1) and row i + 1 original: fronts t = j + 2 i.
// ERASE false: unknown pixel, 3x3 weights (1 3 1 / 3 0 3 / 1 3 1) of the known map > 8 (sum / 16 > 0.5) -> mean of the known 3x3
// neighbours, known; ERASE true: known pixel (== 1), weighted sum <= 7 (sum / 16 < 0.45) -> 255, unknown.
template <bool ERASE>
__device__ __forceinline__ void fronts3(float* image, int* known, int H, int W) {
    const int w3[9] = {1, 3, 1, 3, 0, 3, 1, 3, 1};
    const int t0 = 1 + 2 * 1, t1 = (W - 2) + 2 * (H - 2);
    for (int t = t0; t <= t1; ++t) {
        const int ilo = max(1, (t - (W - 2) + 1) / 2), ihi = min(H - 2, (t - 1) / 2);
        for (int i = ilo + (int)threadIdx.x; i <= ihi; i += blockDim.x) {
            const int j = t - 2 * i;
            if (j < 1 || j > W - 2) continue;
            const int kc = ld_i(known + i * W + j);
            if (ERASE ? kc != 1 : kc != 0) continue;
            int k3i[9], s = 0;
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                k3i[q] = ld_i(known + (i + q / 3 - 1) * W + j + q % 3 - 1);
                s += w3[q] * k3i[q];
            }
            if (ERASE) {
                if (s > 7) continue;
                for (int c = 0; c < 3; ++c) image[((size_t)i * W + j) * 3 + c] = 255.f;
                known[i * W + j] = 0;
            } else {
                if (s <= 8) continue;
                double k3[9], v[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) k3[q] = (double)k3i[q];
                const double n = np_sum9(k3);
                for (int c = 0; c < 3; ++c) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) v[q] = (double)ld_f(image + ((size_t)(i + q / 3 - 1) * W + j + q % 3 - 1) * 3 + c) * k3[q];
                    image[((size_t)i * W + j) * 3 + c] = (float)(np_sum9(v) / n);
                }
                known[i * W + j] = 1;
            }
        }
        __threadfence();
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_fill_mask1_tail(float* image, int* known, int H, int W) {
    fronts3<false>(image, known, H, W);
    // border lines, in the reference's order (each line's pixels are independent: they read the neighbouring line only)
    for (int pass = 0; pass < 4; ++pass) {
        const int n = pass < 2 ? W : H;
        for (int x = (int)threadIdx.x; x < n; x += blockDim.x) {
            const int i = pass == 0 ? 0 : (pass == 1 ? H - 1 : x), j = pass < 2 ? x : (pass == 2 ? 0 : W - 1);
            const int ii = pass == 0 ? 1 : (pass == 1 ? H - 2 : x), jj = pass < 2 ? x : (pass == 2 ? 1 : W - 2);
            if (ld_i(known + i * W + j) == 0 && ld_i(known + ii * W + jj) > 0) {
                for (int c = 0; c < 3; ++c) image[((size_t)i * W + j) * 3 + c] = ld_f(image + ((size_t)ii * W + jj) * 3 + c);
                known[i * W + j] = 1;
            }
        }
        __threadfence();
        __syncthreads();
    }
    fronts3<true>(image, known, H, W);
}

}  // namespace t2n

using namespace t2n;

static size_t al256i(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t t2n_image_filter_workspace_bytes(int H, int W) {
    if (H < 3 || W < 3) return 0;
    const size_t n = (size_t)H * W;
    return al256i(n * 16) * 2 + al256i(n);
}

extern "C" int t2n_sparse_bilateral_filtering(const float* depth, const float* image, int H, int W, const int* filter_sizes_host, int num_iter,
                                              float depth_threshold, float* photo_out, float* depth_out, void* workspace,
                                              size_t workspace_bytes, t2n_stream stream) {
    if (!depth || !image || !filter_sizes_host || !photo_out || !depth_out || !workspace || H < 3 || W < 3 || num_iter < 1) {
        set_error("t2n_sparse_bilateral_filtering: bad argument");
        return T2N_ERR_INVALID;
    }
    for (int i = 0; i < num_iter; ++i)
        if (filter_sizes_host[i] < 1 || filter_sizes_host[i] > 2 * kMedHalo + 1 || !(filter_sizes_host[i] & 1)) {
            set_error("t2n_sparse_bilateral_filtering: window %d not in {1,3,5,7}", filter_sizes_host[i]);
            return T2N_ERR_UNSUPPORTED;
        }
    if (workspace_bytes < t2n_image_filter_workspace_bytes(H, W)) { set_error("t2n_sparse_bilateral_filtering: workspace too small"); return T2N_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    const int n = H * W;
    float* A = (float*)workspace;
    float* B = (float*)((char*)workspace + al256i((size_t)n * 16));
    unsigned char* disc = (unsigned char*)workspace + 2 * al256i((size_t)n * 16);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_img_pack, dim3(nb), dim3(256), 0, s, depth, image, n, A);
    for (int i = 0; i < num_iter; ++i) {
        // save_depths[i] = the depth BEFORE pass i (the image list aliases ONE array, filtered in place: only its final state exists)
        T2N_HIP(hipMemcpyAsync(depth_out + (size_t)i * n, A, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_img_disc, dim3(nb), dim3(256), 0, s, (const float*)A, depth, H, W, depth_threshold, disc);
        hipLaunchKernelGGL(k_img_median, dim3((unsigned)((W + kMedT - 1) / kMedT), (unsigned)((H + kMedT - 1) / kMedT), 4), dim3(256), 0, s,
                           (const float*)A, (const unsigned char*)disc, H, W, filter_sizes_host[i], B);
        float* tmp = A; A = B; B = tmp;
    }
    hipLaunchKernelGGL(k_img_unpack, dim3(nb), dim3(256), 0, s, (const float*)A, n, photo_out);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" size_t t2n_warp_workspace_bytes(int H, int W) {
    if (H < 1 || W < 1) return 0;
    return al256i((size_t)H * W * 24) + al256i((size_t)(H + 2) * (W + 2) * 40) + 256;
}

extern "C" int t2n_warp_view(const float* rgb, const float* depth, const uint8_t* mask1, int H, int W, const double* Ki9_host,
                             const double* T12_host, const double* K9_host, uint8_t* filled, uint8_t* image_u8, double* depth_out,
                             void* workspace, size_t workspace_bytes, t2n_stream stream) {
    if (!rgb || !depth || !Ki9_host || !T12_host || !K9_host || !filled || !image_u8 || !depth_out || !workspace || H < 1 || W < 1) {
        set_error("t2n_warp_view: bad argument");
        return T2N_ERR_INVALID;
    }
    if (workspace_bytes < t2n_warp_workspace_bytes(H, W)) { set_error("t2n_warp_view: workspace too small"); return T2N_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    WarpMats M;
    memcpy(M.Ki, Ki9_host, sizeof(M.Ki)); memcpy(M.T, T12_host, sizeof(M.T)); memcpy(M.K2, K9_host, sizeof(M.K2));
    const int n = H * W;
    double* uvz = (double*)workspace;
    double* canvas = (double*)((char*)workspace + al256i((size_t)n * 24));
    unsigned long long* logmax = (unsigned long long*)((char*)canvas + al256i((size_t)(H + 2) * (W + 2) * 40));
    T2N_HIP(hipMemsetAsync(canvas, 0, al256i((size_t)(H + 2) * (W + 2) * 40) + 256, s));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_warp_points, dim3(nb), dim3(256), 0, s, depth, H, W, M, uvz, logmax);
    hipLaunchKernelGGL(k_warp_splat, dim3(nb), dim3(256), 0, s, rgb, mask1, H, W, (const double*)uvz, (const unsigned long long*)logmax, canvas);
    hipLaunchKernelGGL(k_warp_resolve, dim3(nb), dim3(256), 0, s, (const double*)canvas, H, W, filled, image_u8, depth_out);
    hipLaunchKernelGGL(k_warp_mark, dim3(nb), dim3(256), 0, s, filled, n);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_warp_finish(const uint8_t* filled, const uint8_t* image_u8, int H, int W, float* image_out, int64_t* mask_out,
                               t2n_stream stream) {
    if (!filled || !image_u8 || !image_out || H < 1 || W < 1) { set_error("t2n_warp_finish: bad argument"); return T2N_ERR_INVALID; }
    const int n = H * W;
    hipLaunchKernelGGL(k_warp_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, filled, image_u8, n, image_out,
                       (long long*)mask_out);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_dibr_filter_mask2(float* image, int32_t* known, double* depth, int H, int W, float threshold, t2n_stream stream) {
    if (!image || !known || H < 5 || W < 5) { set_error("t2n_dibr_filter_mask2: bad argument"); return T2N_ERR_INVALID; }
    // numpy: float64(sum) / float32(36) > threshold (python float). 2 * sum is an integer: the pixel passes iff 2 sum > 72 thr
    const int twice = (int)floor(72.0 * (double)threshold + 1e-9);
    hipLaunchKernelGGL(k_fill_fronts, dim3(1), dim3(1024), 0, (hipStream_t)stream, image, known, depth, H, W, twice);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_dibr_filter_mask(float* image, int32_t* known, int H, int W, t2n_stream stream) {
    if (!image || !known || H < 5 || W < 5) { set_error("t2n_dibr_filter_mask: bad argument"); return T2N_ERR_INVALID; }
    const int rc = t2n_dibr_filter_mask2(image, known, nullptr, H, W, 0.6f, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_fill_mask1_tail, dim3(1), dim3(1024), 0, (hipStream_t)stream, image, known, H, W);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// ---- global depth alignment (text2nerf_main.py:241-270) --------------------------------------------------------------------------
// scale = mean over consecutive sampled pixel pairs of (rendered difference) / (estimated difference + 1e-8), pairs kept when the ratio
// is finite, >= 0 and within 5 |thresh - 1| of 1 (thresh = (max rendered - push) / (max estimate - push)); fallback thresh. shift = mean
// over the samples of scale * estimate - rendered within 2 |max(scale * estimate) - max rendered|; fallback that difference. Output:
// scale * estimate - shift. The pixel list is the caller's (random.sample of the filled pixels, :234-240). Everything is a reduction
// over <= 10 000 samples plus two passes over the image: one workgroup in double precision with fixed-order tree sums
// (deterministic), then an elementwise pass.
namespace t2n {
struct AlignArgs {
    const float* dr; const float* de; int H, W; const int* ps; int K; double push;
    float* out; double* ss;   // ss: [0] scale, [1] shift, [2] pairs kept, [3] samples kept
};
__device__ __forceinline__ double block_sum(double v, double* sm) {
    const int t = threadIdx.x;
    __syncthreads();
    sm[t] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) sm[t] += sm[t + o]; __syncthreads(); }
    return sm[0];
}
__device__ __forceinline__ double block_max(double v, double* sm) {
    const int t = threadIdx.x;
    __syncthreads();
    sm[t] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) sm[t] = fmax(sm[t], sm[t + o]); __syncthreads(); }
    return sm[0];
}
__global__ __launch_bounds__(1024) void k_align_stats(const AlignArgs a) {
    __shared__ double sm[1024];
    const int t = threadIdx.x;
    const long long n = (long long)a.H * a.W;
    double mr = -1e300, me = -1e300, mn = 1e300;
    for (long long i = t; i < n; i += 1024) { mr = fmax(mr, (double)a.dr[i]); me = fmax(me, (double)a.de[i]); mn = fmin(mn, (double)a.de[i]); }
    const double max_r = block_max(mr, sm), max_e = block_max(me, sm), min_e = -block_max(-mn, sm);
    const double thresh = (max_r - a.push) / (max_e - a.push);
    double s = 0.0, c = 0.0;
    for (int i = t; i < a.K - 1; i += 1024) {
        const size_t p1 = (size_t)a.ps[2 * i] * a.W + a.ps[2 * i + 1], p2 = (size_t)a.ps[2 * i + 2] * a.W + a.ps[2 * i + 3];
        const double dd1 = (double)(a.dr[p1] - a.dr[p2]);           // float32 difference, as numpy forms it
        const double dd2 = (double)a.de[p1] - (double)a.de[p2];
        const double r = dd1 / (dd2 + 1e-8);
        const bool keep = isfinite(r) && !(fabs(r - 1.0) > 5.0 * fabs(thresh - 1.0)) && !(r < 0.0);
        if (keep) { s += r; c += 1.0; }
    }
    const double ssum = block_sum(s, sm), scnt = block_sum(c, sm);
    const double scale = scnt > 0.0 ? ssum / scnt : thresh;
    const double thresh2 = (scale >= 0.0 ? max_e * scale : min_e * scale) - max_r;
    s = 0.0; c = 0.0;
    for (int i = t; i < a.K; i += 1024) {
        const size_t p1 = (size_t)a.ps[2 * i] * a.W + a.ps[2 * i + 1];
        const double d = (double)a.de[p1] * scale - (double)a.dr[p1];
        if (!(fabs(d) > 2.0 * fabs(thresh2))) { s += d; c += 1.0; }
    }
    const double hsum = block_sum(s, sm), hcnt = block_sum(c, sm);
    if (t == 0) { a.ss[0] = scale; a.ss[1] = hcnt > 0.0 ? hsum / hcnt : thresh2; a.ss[2] = scnt; a.ss[3] = hcnt; }
}
__global__ __launch_bounds__(256) void k_align_apply(const AlignArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long long)a.H * a.W) a.out[i] = (float)((double)a.de[i] * a.ss[0] - a.ss[1]);
}
}  // namespace t2n

extern "C" int t2n_depth_align_global(const float* depth_rendered, const float* depth_est, int H, int W, const int32_t* pixel_sample_yx,
                                      int n_samples, double push_depth, float* depth_shift, double* scale_shift, t2n_stream stream) {
    if (!depth_rendered || !depth_est || !pixel_sample_yx || !depth_shift || !scale_shift || H < 1 || W < 1 || n_samples < 1) {
        set_error("t2n_depth_align_global: bad argument");
        return T2N_ERR_INVALID;
    }
    AlignArgs a;
    a.dr = depth_rendered; a.de = depth_est; a.H = H; a.W = W; a.ps = pixel_sample_yx; a.K = n_samples; a.push = push_depth;
    a.out = depth_shift; a.ss = scale_shift;
    hipLaunchKernelGGL(k_align_stats, dim3(1), dim3(1024), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_align_apply, dim3((unsigned)(((long long)H * W + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
