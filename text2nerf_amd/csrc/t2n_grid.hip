// Coarse-to-fine and occupancy-mask maintenance of the VM-split field: the pieces upstream TensoRF
// drivers call between training stages. All of them reuse the density gather of the marcher.
//
// Replaces (reference): models/tensorBase.py:412-434 (compute_alpha), :328-344 (getDenseAlpha), :346-370
// (updateAlphaMask: clamp, transpose, max_pool3d(3), threshold, bounding box of the kept voxels),
// models/tensoRF.py:258-272 (up_sampling_VM: F.interpolate(bilinear, align_corners=True) of planes and lines).
#include "t2n_device.h"

namespace t2n {

// alpha = 1 - exp(-sigma * length) at world-space points; sigma = 0 where the field's AlphaGridMask (if any) samples
// <= 0. Points are explicit (xyz) or the nodes of a dense grid: node (i, j, k) = aabb0 * (1 - s) + aabb1 * s with
// s = (lin_x[i], lin_y[j], lin_z[k]) (the caller passes torch.linspace(0, 1, g) so the node positions match the
// reference bit for bit); output index ((i * gy) + j) * gz + k. 4 lanes per point, as in pass B of the marcher.
struct AlphaArgs {
    FieldDev F;
    const float* xyz; const float* lin_x; const float* lin_y; const float* lin_z;
    int gy, gz;
    long long n;
    float length;
    float* alpha;
};

__global__ __launch_bounds__(256) void k_compute_alpha(const AlphaArgs a) {
    constexpr int LPS = 4;
    const FieldDev& F = a.F;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = t / LPS;
    const int q = (int)(t % LPS);
    bool ok = p < a.n;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (ok) {
        if (a.xyz) {
            px = a.xyz[p * 3]; py = a.xyz[p * 3 + 1]; pz = a.xyz[p * 3 + 2];
        } else {
            const long long ij = p / a.gz;
            const int k = (int)(p - ij * a.gz), j = (int)(ij % a.gy), i = (int)(ij / a.gy);
            const float sx = a.lin_x[i], sy = a.lin_y[j], sz = a.lin_z[k];
            px = F.aabb0[0] * (1.f - sx) + F.aabb1[0] * sx;
            py = F.aabb0[1] * (1.f - sy) + F.aabb1[1] * sy;
            pz = F.aabb0[2] * (1.f - sz) + F.aabb1[2] * sz;
        }
    }
    const bool inb = ok;
    if (ok && F.alpha) ok = alpha_value(F, px, py, pz) > 0.f;
    float part = 0.f;
    if (ok) {
        const float xn = (px - F.aabb0[0]) * F.inv[0] - 1.f, yn = (py - F.aabb0[1]) * F.inv[1] - 1.f,
                    zn = (pz - F.aabb0[2]) * F.inv[2] - 1.f;
        QuadTaps t0, t1, t2;
        const Axes3 A = sample_axes(F.den, xn, yn, zn);
        issue_taps_ax<0>(F.den, LPS, q, A, t0);
        issue_taps_ax<1>(F.den, LPS, q, A, t1);
        issue_taps_ax<2>(F.den, LPS, q, A, t2);
        float4 pv, l;
        pv = taps_plane(t0); l = taps_line(t0);
        part = pv.x * l.x; part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
        pv = taps_plane(t1); l = taps_line(t1);
        part = fmaf(pv.x, l.x, part); part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
        pv = taps_plane(t2); l = taps_line(t2);
        part = fmaf(pv.x, l.x, part); part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
    }
    const float feat = group_sum<LPS>(part);
    if (inb && q == 0) {
        const float sg = ok ? feature2density(F, feat) : 0.f;
        a.alpha[p] = 1.f - expf((-sg) * a.length);
    }
}

// updateAlphaMask's volume: dense alpha [gx][gy][gz] -> clamp(0,1) -> transposed to [gz][gy][gx] -> 3x3x3 max pool
// (stride 1, implicit -inf padding) -> 1 where >= thres else 0. The bounding box of the kept voxels is returned as grid
// indices (min x,y,z / max x,y,z; mins start at INT_MAX, maxs at -1), which the host maps to coordinates through the
// same lerp as getDenseAlpha (the node coordinates are monotone in the index).
__global__ __launch_bounds__(256) void k_alpha_volume(const float* __restrict__ dense, int gx, int gy, int gz, float thres,
                                                      float* __restrict__ vol, int* __restrict__ bbox) {
    const long long n = (long long)gx * gy * gz;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int ix = 0, iy = 0, iz = 0;
    bool keep = false;
    if (t < n) {
        ix = (int)(t % gx);
        const long long r = t / gx;
        iy = (int)(r % gy); iz = (int)(r / gy);
        float m = -INFINITY;
        for (int dz = -1; dz <= 1; ++dz) {
            const int z = iz + dz;
            if (z < 0 || z >= gz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int y = iy + dy;
                if (y < 0 || y >= gy) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    const int x = ix + dx;
                    if (x < 0 || x >= gx) continue;
                    const float v = dense[((long long)x * gy + y) * gz + z];
                    m = fmaxf(m, fminf(fmaxf(v, 0.f), 1.f));
                }
            }
        }
        keep = m >= thres;
        vol[t] = keep ? 1.f : 0.f;
    }
    // block-level reduction of the box, then six atomics per block
    __shared__ int s_box[6];
    if (threadIdx.x < 6) s_box[threadIdx.x] = threadIdx.x < 3 ? 0x7fffffff : -1;
    __syncthreads();
    if (keep) {
        atomicMin(&s_box[0], ix); atomicMin(&s_box[1], iy); atomicMin(&s_box[2], iz);
        atomicMax(&s_box[3], ix); atomicMax(&s_box[4], iy); atomicMax(&s_box[5], iz);
    }
    __syncthreads();
    if (threadIdx.x < 3) { if (s_box[threadIdx.x] != 0x7fffffff) atomicMin(&bbox[threadIdx.x], s_box[threadIdx.x]); }
    else if (threadIdx.x < 6) { if (s_box[threadIdx.x] >= 0) atomicMax(&bbox[threadIdx.x], s_box[threadIdx.x]); }
}

// ATen upsample_bilinear2d, align_corners=True, on [C][Hin][Win] -> [C][Hout][Wout]:
// scale = (in-1)/(out-1) (0 when out == 1), src = scale * dst, i0 = (int)src, i1 = i0 + (i0 < in-1), l1 = src - i0,
// out = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).
__global__ __launch_bounds__(256) void k_upsample_bilinear(const float* __restrict__ src, int C, int Hin, int Win,
                                                           float* __restrict__ dst, int Hout, int Wout, float sh, float sw) {
    const long long n = (long long)C * Hout * Wout;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int x = (int)(t % Wout);
    const long long r = t / Wout;
    const int y = (int)(r % Hout), c = (int)(r / Hout);
    const float fy = sh * (float)y, fx = sw * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hin - 1 ? 1 : 0), x1 = x0 + (x0 < Win - 1 ? 1 : 0);
    const float h1 = fy - (float)y0, h0 = 1.f - h1, w1 = fx - (float)x0, w0 = 1.f - w1;
    const float* __restrict__ P = src + (size_t)c * Hin * Win;
    const float v00 = P[(size_t)y0 * Win + x0], v01 = P[(size_t)y0 * Win + x1];
    const float v10 = P[(size_t)y1 * Win + x0], v11 = P[(size_t)y1 * Win + x1];
    dst[t] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
}

// filtering_rays(bbox_only=False), models/tensorBase.py:393-395: keep a ray iff any of its first n_samples eval-mode
// samples (t_min + i * step, no box test) reads the occupancy volume > 0. One wave per ray.
__global__ __launch_bounds__(256) void k_filter_alpha(const FieldDev F, const float* __restrict__ rays, long long n, int stride,
                                                      int n_samples, uint8_t* mask) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const Ray ray = load_ray(F, rays + r * stride, stride);
    bool any = false;
    for (int base = 0; base < n_samples && !any; base += 64) {
        const int i = base + lane;
        bool hit = false;
        if (i < n_samples) hit = alpha_pass(F, ray, sample_z<false>(F, ray, i, 0.f));
        any = __ballot(hit) != 0ull;
    }
    if (lane == 0) mask[r] = any ? 1 : 0;
}

}  // namespace t2n

using namespace t2n;

static int launch_alpha(const t2n_field* f, const AlphaArgs& a, hipStream_t s) {
    const long long threads = a.n * 4;
    hipLaunchKernelGGL(k_compute_alpha, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_compute_alpha(const t2n_field* f, const float* xyz_world, int64_t n, float length, float* alpha,
                                 t2n_stream stream) {
    if (!f || !xyz_world || !alpha || n < 0 || n > 0x7fffffffll * 8) { set_error("t2n_compute_alpha: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_compute_alpha: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (n == 0) return T2N_OK;
    AlphaArgs a;
    memset(&a, 0, sizeof(a));
    a.F = f->dev; a.xyz = xyz_world; a.n = n; a.length = length; a.alpha = alpha; a.gy = a.gz = 1;
    return launch_alpha(f, a, (hipStream_t)stream);
}

extern "C" int t2n_dense_alpha(const t2n_field* f, const float* lin_x, const float* lin_y, const float* lin_z, int gx, int gy,
                               int gz, float length, float* alpha, t2n_stream stream) {
    if (!f || !lin_x || !lin_y || !lin_z || !alpha || gx < 1 || gy < 1 || gz < 1) { set_error("t2n_dense_alpha: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_dense_alpha: field has no uploaded parameters"); return T2N_ERR_STATE; }
    AlphaArgs a;
    memset(&a, 0, sizeof(a));
    a.F = f->dev; a.lin_x = lin_x; a.lin_y = lin_y; a.lin_z = lin_z; a.gy = gy; a.gz = gz;
    a.n = (long long)gx * gy * gz; a.length = length; a.alpha = alpha;
    return launch_alpha(f, a, (hipStream_t)stream);
}

extern "C" int t2n_alpha_volume(const float* dense_alpha, int gx, int gy, int gz, float thres, float* volume, int* bbox_idx,
                                t2n_stream stream) {
    if (!dense_alpha || !volume || !bbox_idx || gx < 1 || gy < 1 || gz < 1) { set_error("t2n_alpha_volume: bad argument"); return T2N_ERR_INVALID; }
    const int init[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1, -1, -1};
    T2N_HIP(hipMemcpyAsync(bbox_idx, init, sizeof(init), hipMemcpyHostToDevice, (hipStream_t)stream));
    const long long n = (long long)gx * gy * gz;
    hipLaunchKernelGGL(k_alpha_volume, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dense_alpha, gx,
                       gy, gz, thres, volume, bbox_idx);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_upsample_bilinear(const float* src, int C, int Hin, int Win, float* dst, int Hout, int Wout,
                                     t2n_stream stream) {
    if (!src || !dst || C < 1 || Hin < 1 || Win < 1 || Hout < 1 || Wout < 1) { set_error("t2n_upsample_bilinear: bad argument"); return T2N_ERR_INVALID; }
    const float sh = Hout > 1 ? (float)(Hin - 1) / (float)(Hout - 1) : 0.f;
    const float sw = Wout > 1 ? (float)(Win - 1) / (float)(Wout - 1) : 0.f;
    const long long n = (long long)C * Hout * Wout;
    hipLaunchKernelGGL(k_upsample_bilinear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, C,
                       Hin, Win, dst, Hout, Wout, sh, sw);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_filter_rays_alpha(const t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples,
                                     uint8_t* mask, t2n_stream stream) {
    if (!f || !rays || !mask || n_rays < 0 || ray_stride < 6 || n_samples < 1) { set_error("t2n_filter_rays_alpha: bad argument"); return T2N_ERR_INVALID; }
    if (!f->dev.alpha) { set_error("t2n_filter_rays_alpha: the field has no alpha mask"); return T2N_ERR_STATE; }
    if (n_rays == 0) return T2N_OK;
    hipLaunchKernelGGL(k_filter_alpha, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, f->dev, rays,
                       (long long)n_rays, ray_stride, n_samples, mask);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
