// extern "C" surface of libt2n_hip.so: field lifetime, parameter upload (reference layout -> channel-last), ray
// generation, the render call orchestration (march -> shade -> composite per sub-launch) and the timing hooks.
#include <stdarg.h>

#include "t2n_device.h"

namespace t2n {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return T2N_ERR_HIP;
}

void timing_begin(t2n_field* f, int k, hipStream_t s) {
    if (!((f->timing >> k) & 1)) return;
    TimingSlot& t = f->slots[k];
    if (t.used >= kTimingEvents) return;
    if (!t.start[t.used]) {
        (void)hipEventCreate(&t.start[t.used]);
        (void)hipEventCreate(&t.stop[t.used]);
    }
    (void)hipEventRecord(t.start[t.used], s);
}
void timing_end(t2n_field* f, int k, hipStream_t s) {
    if (!((f->timing >> k) & 1)) return;
    TimingSlot& t = f->slots[k];
    if (t.used >= kTimingEvents) { t.untimed++; return; }
    (void)hipEventRecord(t.stop[t.used], s);
    t.used++;
}

static void timing_flush(t2n_field* f) {
    for (int k = 0; k < T2N_K_COUNT; ++k) {
        TimingSlot& t = f->slots[k];
        double sum = 0.0;
        int n = 0;
        for (int i = 0; i < t.used; ++i) {
            (void)hipEventSynchronize(t.stop[i]);
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, t.start[i], t.stop[i]) == hipSuccess) { sum += ms; n++; }
        }
        t.ms += sum; t.launches += n;
        if (n && t.untimed) { t.ms += sum / n * (double)t.untimed; t.launches += t.untimed; }   // beyond the event pool: the timed average
        t.used = 0; t.untimed = 0;
    }
}

// [1,C,H,W] -> [H][W][C]   (lines: W == 1)
// Reference layout [C][HW] -> channel-last [HW][C] through an LDS tile of 64 texels (coalesced on both sides).
// dsth != NULL (bf16 factor storage): round to nearest-even bf16; dst gets the rounded value as fp32, dsth the 2-byte texel.
// The 12 factor tensors of an upload go through ONE launch (twelve launches of mostly tiny grids cost ~7 us each).
struct RelayoutMulti { const float* src[12]; float* dst[12]; unsigned short* dsth[12]; int C[12]; long long HW[12]; unsigned block0[13]; int count; };
__global__ __launch_bounds__(256) void k_relayout(const RelayoutMulti a) {
    __shared__ float tile[64 * 49];                      // [64 texels][C + 1], C <= 48
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < a.count; ++q) t += (a.block0[q] <= blockIdx.x) ? 1 : 0;
    const float* __restrict__ src = a.src[t];
    float* __restrict__ dst = a.dst[t];
    unsigned short* __restrict__ dsth = a.dsth[t];
    const int C = a.C[t];
    const long long HW = a.HW[t];
    const long long pix0 = (long long)(blockIdx.x - a.block0[t]) * 64;
    const int lp = threadIdx.x & 63, cs = threadIdx.x >> 6;
    const int ld = C + 1;
    if (pix0 + lp < HW)
        for (int c = cs; c < C; c += 4) tile[lp * ld + c] = src[(long long)c * HW + pix0 + lp];
    __syncthreads();
    const long long n = (HW - pix0 < 64 ? HW - pix0 : 64) * C;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int px = i / C, c = i - px * C;
        float v = tile[px * ld + c];
        const long long t = pix0 * C + i;
        if (dsth) {
            unsigned b = __float_as_uint(v);
            if ((b & 0x7fffffffu) <= 0x7f800000u) b += 0x7fffu + ((b >> 16) & 1u);   // RNE (NaN payloads pass through truncated)
            b &= 0xffff0000u;
            dsth[t] = (unsigned short)(b >> 16);
            v = __uint_as_float(b);
        }
        dst[t] = v;
    }
}

static int relayout_one(RelayoutMulti& m, unsigned& blocks, const float* src, float** dst, void** dsth, bool half, int C, long long HW) {
    if (!src) { set_error("t2n_field_upload: NULL factor tensor"); return T2N_ERR_INVALID; }
    if (!*dst) T2N_HIP(hipMalloc((void**)dst, (size_t)HW * C * sizeof(float)));
    if (half && !*dsth) T2N_HIP(hipMalloc(dsth, (size_t)HW * C * 2 + 16));
    if (C > 48) { set_error("t2n_field_upload: %d channels > 48", C); return T2N_ERR_UNSUPPORTED; }
    const int i = m.count++;
    m.src[i] = src; m.dst[i] = *dst; m.dsth[i] = half ? (unsigned short*)*dsth : nullptr; m.C[i] = C; m.HW[i] = HW; m.block0[i] = blocks;
    blocks += (unsigned)((HW + 63) / 64);
    return T2N_OK;
}

int launch_relayout(t2n_field* f, const t2n_field_params* p, hipStream_t s) {
    const int* g = f->desc.grid;
    RelayoutMulti m;
    memset(&m, 0, sizeof(m));
    unsigned blocks = 0;
    for (int k = 0; k < 3; ++k) {
        const long long HW = (long long)g[mat1(k)] * g[mat0(k)];
        const long long L = g[vecm(k)];
        int rc;
        const bool hf = f->factor_bf16 != 0;
        if ((rc = relayout_one(m, blocks, p->density_plane[k], &f->buf_den_plane[k], &f->hbuf_den_plane[k], hf, f->desc.density_n_comp, HW))) return rc;
        if ((rc = relayout_one(m, blocks, p->density_line[k], &f->buf_den_line[k], &f->hbuf_den_line[k], hf, f->desc.density_n_comp, L))) return rc;
        if ((rc = relayout_one(m, blocks, p->app_plane[k], &f->buf_app_plane[k], &f->hbuf_app_plane[k], hf, f->desc.app_n_comp, HW))) return rc;
        if ((rc = relayout_one(m, blocks, p->app_line[k], &f->buf_app_line[k], &f->hbuf_app_line[k], hf, f->desc.app_n_comp, L))) return rc;
        f->dev.den.plane[k] = f->buf_den_plane[k]; f->dev.den.line[k] = f->buf_den_line[k];
        f->dev.app.plane[k] = f->buf_app_plane[k]; f->dev.app.line[k] = f->buf_app_line[k];
        f->dev.den.plane_h[k] = hf ? f->hbuf_den_plane[k] : nullptr; f->dev.den.line_h[k] = hf ? f->hbuf_den_line[k] : nullptr;
        f->dev.app.plane_h[k] = hf ? f->hbuf_app_plane[k] : nullptr; f->dev.app.line_h[k] = hf ? f->hbuf_app_line[k] : nullptr;
    }
    m.block0[m.count] = blocks;
    if (blocks) hipLaunchKernelGGL(k_relayout, dim3(blocks), dim3(256), 0, s, m);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

static int apply_desc(t2n_field* f, const t2n_field_desc* d) {
    if (d->density_n_comp != 16 || d->app_n_comp != 48) {
        set_error("unsupported n_comp: density %d (need 16), appearance %d (need 48)", d->density_n_comp, d->app_n_comp);
        return T2N_ERR_UNSUPPORTED;
    }
    if (d->shading == T2N_SHADE_MLP_FEA_NOVIEW) {
        if (d->app_dim != 27 || d->fea_pe != 6 || d->feature_c != 128) {
            set_error("unsupported MLP_Fea_noview shape: app_dim %d fea_pe %d featureC %d (need 27/6/128)", d->app_dim, d->fea_pe, d->feature_c);
            return T2N_ERR_UNSUPPORTED;
        }
    } else if (d->shading == T2N_SHADE_SH) {
        if (d->app_dim != 27) { set_error("SH head needs app_dim 27"); return T2N_ERR_UNSUPPORTED; }
    } else if (d->shading == T2N_SHADE_RGB) {
        if (d->app_dim != 3) { set_error("RGB head needs app_dim 3"); return T2N_ERR_UNSUPPORTED; }
    } else if (head_is_generic(d->shading)) {
        const HeadDims H = head_dims(*d);
        if (d->app_dim < 1 || d->app_dim > 32 || d->feature_c != 128 || H.K0 > 512 || d->view_pe < 0 || d->pos_pe < 0 || d->fea_pe < 0) {
            set_error("unsupported view-dependent head shape: app_dim %d featureC %d inputs %d (need app_dim <= 32, featureC 128, inputs <= 512)",
                      d->app_dim, d->feature_c, H.K0);
            return T2N_ERR_UNSUPPORTED;
        }
    } else {
        set_error("unsupported shading head %d", d->shading);
        return T2N_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < 3; ++k)
        if (d->grid[k] < 2 || d->grid[k] > 4096) { set_error("grid[%d]=%d out of range [2,4096]", k, d->grid[k]); return T2N_ERR_INVALID; }
    f->desc = *d;
    if (f->term_eps > d->weight_thres) f->term_eps = d->weight_thres;   // the early-termination bound follows a changed appearance threshold
    FieldDev& D = f->dev;
    for (int k = 0; k < 3; ++k) {
        D.aabb0[k] = d->aabb_min[k]; D.aabb1[k] = d->aabb_max[k]; D.inv[k] = d->inv_aabb_size[k];
        D.den.W[k] = D.app.W[k] = d->grid[mat0(k)];
        D.den.H[k] = D.app.H[k] = d->grid[mat1(k)];
        D.den.L[k] = D.app.L[k] = d->grid[vecm(k)];
    }
    D.den.C = d->density_n_comp; D.app.C = d->app_n_comp;
    D.shift = d->density_shift; D.dscale = d->distance_scale; D.thres = d->weight_thres; D.step = d->step_size;
    D.near = d->near; D.far = d->far; D.zgate = d->z_gate; D.act = d->act; D.shading = d->shading; D.app_dim = d->app_dim;
    D.term_eps = 0.f;   // the marchers' launchers set it per launch (eval, no weights / z_vals / context)
    return T2N_OK;
}

// ---- ray generation ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ray_dirs(int H, int W, float fx, float fy, float cx, float cy, int normalize, float* dirs) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)H * W) return;
    const int y = (int)(t / W), x = (int)(t - (long long)y * W);
    const float i = (float)x + 0.5f, j = (float)y + 0.5f;
    float dx = (i - cx) / fx, dy = (j - cy) / fy, dz = 1.f;
    if (normalize) {
        const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
        dx = dx / n; dy = dy / n; dz = dz / n;
    }
    dirs[t * 3] = dx; dirs[t * 3 + 1] = dy; dirs[t * 3 + 2] = dz;
}

struct Pose { float m[12]; };

__device__ __forceinline__ void rotate(const Pose& P, float dx, float dy, float dz, float& rx, float& ry, float& rz) {
    // rays_d = directions @ c2w[:3,:3].T  (dataLoader/ray_utils.py:79)
    rx = fmaf(dz, P.m[2], fmaf(dy, P.m[1], dx * P.m[0]));
    ry = fmaf(dz, P.m[6], fmaf(dy, P.m[5], dx * P.m[4]));
    rz = fmaf(dz, P.m[10], fmaf(dy, P.m[9], dx * P.m[8]));
}

__global__ __launch_bounds__(256) void k_get_rays(const float* __restrict__ dirs, long long n, const Pose P, float* ro, float* rd, float* r6) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float rx, ry, rz;
    rotate(P, dirs[t * 3], dirs[t * 3 + 1], dirs[t * 3 + 2], rx, ry, rz);
    if (ro) { ro[t * 3] = P.m[3]; ro[t * 3 + 1] = P.m[7]; ro[t * 3 + 2] = P.m[11]; }
    if (rd) { rd[t * 3] = rx; rd[t * 3 + 1] = ry; rd[t * 3 + 2] = rz; }
    if (r6) { r6[t * 6] = P.m[3]; r6[t * 6 + 1] = P.m[7]; r6[t * 6 + 2] = P.m[11]; r6[t * 6 + 3] = rx; r6[t * 6 + 4] = ry; r6[t * 6 + 5] = rz; }
}

__global__ __launch_bounds__(256) void k_generate_rays(int H, int W, float fx, float fy, float cx, float cy, const Pose P, float* r6) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)H * W) return;
    const int y = (int)(t / W), x = (int)(t - (long long)y * W);
    const float i = (float)x + 0.5f, j = (float)y + 0.5f;
    float dx = (i - cx) / fx, dy = (j - cy) / fy, dz = 1.f;
    const float n = sqrtf((dx * dx + dy * dy) + dz * dz);
    dx = dx / n; dy = dy / n; dz = dz / n;
    float rx, ry, rz;
    rotate(P, dx, dy, dz, rx, ry, rz);
    r6[t * 6] = P.m[3]; r6[t * 6 + 1] = P.m[7]; r6[t * 6 + 2] = P.m[11]; r6[t * 6 + 3] = rx; r6[t * 6 + 4] = ry; r6[t * 6 + 5] = rz;
}


// ---- f-2: per-frame post-processing of `evaluation` / `evaluation_path` on the device (renderer.py:91-113,168-176,
// utils.py:241-257): rgb -> uint8 by truncation of 255 * clamp(rgb), depth -> (d + offset [, max 0]) -> nan_to_num ->
// (x - mi) / (ma - mi + 1e-8) -> max 0 -> uint8 by truncation (wrapping modulo 256 above 1.0 like the x86 float->uint8
// cast numpy performs) -> JET colour table in OpenCV's BGR channel order, optional sum of squared errors against a
// ground-truth frame for the PSNR.
__device__ __forceinline__ unsigned char jet_channel(int i, float centre) {
    // OpenCV's COLORMAP_JET table is the piecewise-linear clamp(1.5 - |4 x - centre|, 0, 1) sampled at x = i / 255
    // = clamp(382.5 - |4 i - 255 centre|, 0, 255) in exact integer halves, rounded half-to-even like cv::saturate_cast
    int twice = 765 - 2 * abs(4 * i - 255 * (int)centre);       // 2 * value, always odd before clamping
    twice = min(max(twice, 0), 510);
    int r = twice >> 1;                                          // floor
    if ((twice & 1) && (r & 1)) r += 1;                          // tie -> even
    return (unsigned char)r;
}
__global__ __launch_bounds__(256) void k_frame_post(const float* __restrict__ rgb, const float* __restrict__ depth, long long n,
                                                    float sub, float add, int clamp0, float mi, float den,
                                                    unsigned char* rgb8, unsigned char* depth8, const float* __restrict__ gt,
                                                    double* sq_sum) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float se = 0.f;
    if (t < n) {
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c[k] = fminf(fmaxf(rgb[t * 3 + k], 0.f), 1.f);
            if (rgb8) rgb8[t * 3 + k] = (unsigned char)(int)(c[k] * 255.f);
            if (gt) { const float d = c[k] - gt[t * 3 + k]; se += d * d; }
        }
        if (depth8) {
            float d = depth[t];
            if (clamp0) { d = d - sub; d = d + add; d = fmaxf(d, 0.f); }   // renderer.py:94-95 (two fp32 roundings)
            if (d != d) d = 0.f;
            else if (isinf(d)) d = d > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
            float x = (d - mi) / den;
            x = fmaxf(x, 0.f);
            const float y = 255.f * x;
            const int i = y < 2147483648.f ? ((int)y & 255) : 0;   // cvttss2si: out-of-range -> 0x80000000 -> low byte 0
            depth8[t * 3 + 0] = jet_channel(i, 1.f);   // B
            depth8[t * 3 + 1] = jet_channel(i, 2.f);   // G
            depth8[t * 3 + 2] = jet_channel(i, 3.f);   // R
        }
    }
    if (sq_sum) {
        float s = se;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        __shared__ float part[4];
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sq_sum, (double)part[0] + (double)part[1] + (double)part[2] + (double)part[3]);
    }
}

// ---- dda + ray_marcher (dataLoader/ray_utils.py:174-228): the AABB-clipped linspace sampler — no call sites in this driver
// (the live path is TensorBase.sample_ray), provided for the file's completeness. Thread per (ray, sample).
// dda: inv = 1 / (d + 1e-6); t0 = (bbox_min - o) * inv, t1 = (bbox_max - o) * inv; near = max_xyz(min(t0,t1)), far = min_xyz(max(t0,t1)).
// ray_marcher: z = near (1 - s) + far s  (or 1 / (1/near (1 - s) + 1/far s) with lindisp), s = torch.linspace(0,1,S) handed in;
// optional perturbation draws u [n,S]: mid-point intervals (:211-218); xyz = o + d z.
struct MarcherArgs {
    const float* rays; long long n; int stride; int S; int lindisp; int use_bbox; float bmin[3], bmax[3];
    const float* steps; const float* perturb; float* xyz; float* z_vals; float* near_far;
};
__device__ __forceinline__ void dda_ray(const MarcherArgs& a, const float* r, float& tn, float& tf) {
    float lo = -3.402823466e38f, hi = 3.402823466e38f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float inv = 1.0f / (r[3 + k] + 1e-6f);
        const float t0 = (a.bmin[k] - r[k]) * inv, t1 = (a.bmax[k] - r[k]) * inv;
        lo = fmaxf(lo, fminf(t0, t1)); hi = fminf(hi, fmaxf(t0, t1));
    }
    tn = lo; tf = hi;
}
__device__ __forceinline__ float marcher_z(const MarcherArgs& a, float tn, float tf, int j) {
    const float s = a.steps[j];
    if (!a.lindisp) return tn * (1.f - s) + tf * s;
    return 1.f / (1.f / tn * (1.f - s) + 1.f / tf * s);
}
__global__ __launch_bounds__(256) void k_ray_marcher(const MarcherArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n * a.S) return;
    const long long i = t / a.S;
    const int j = (int)(t - i * a.S);
    const float* r = a.rays + i * a.stride;
    float tn, tf;
    if (a.use_bbox) dda_ray(a, r, tn, tf); else { tn = r[6]; tf = r[7]; }
    if (a.near_far && j == 0) { a.near_far[i * 2] = tn; a.near_far[i * 2 + 1] = tf; }
    float z = marcher_z(a, tn, tf, j);
    if (a.perturb) {
        const float zl = j > 0 ? marcher_z(a, tn, tf, j - 1) : z, zu = j < a.S - 1 ? marcher_z(a, tn, tf, j + 1) : z;
        const float lower = j > 0 ? 0.5f * (zl + z) : z, upper = j < a.S - 1 ? 0.5f * (z + zu) : z;
        z = lower + (upper - lower) * a.perturb[t];
    }
    if (a.z_vals) a.z_vals[t] = z;
    if (a.xyz) { a.xyz[t * 3] = r[0] + r[3] * z; a.xyz[t * 3 + 1] = r[1] + r[4] * z; a.xyz[t * 3 + 2] = r[2] + r[5] * z; }
}

// ---- eval_sh_bases (models/sh.py:87-133): real SH basis polynomials of degree 0..4 at unit directions, (deg+1)^2 values per
// direction; products associated left to right like the reference's expressions (fp32, python-float constants rounded to
// fp32 at their first use). SHRender (models/tensorBase.py:29-33) uses degree 2; the fused head lives in k_shade.
__global__ __launch_bounds__(256) void k_sh_bases(int deg, const float* __restrict__ dirs, long long n, float* out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int nb = (deg + 1) * (deg + 1);
    float* o = out + t * nb;
    const float x = dirs[t * 3], y = dirs[t * 3 + 1], z = dirs[t * 3 + 2];
    o[0] = 0.28209479177387814f;
    if (deg < 1) return;
    const float C1 = 0.4886025119029199f;
    o[1] = -C1 * y; o[2] = C1 * z; o[3] = -C1 * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.31539156525252005f * ((2.0f * zz - xx) - yy);
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.5462742152960396f * (xx - yy);
    if (deg < 3) return;
    o[9] = (-0.5900435899266435f * y) * (3.f * xx - yy);
    o[10] = (2.890611442640554f * xy) * z;
    o[11] = (-0.4570457994644658f * y) * ((4.f * zz - xx) - yy);
    o[12] = (0.3731763325901154f * z) * ((2.f * zz - 3.f * xx) - 3.f * yy);
    o[13] = (-0.4570457994644658f * x) * ((4.f * zz - xx) - yy);
    o[14] = (1.445305721320277f * z) * (xx - yy);
    o[15] = (-0.5900435899266435f * x) * (xx - 3.f * yy);
    if (deg < 4) return;
    o[16] = (2.5033429417967046f * xy) * (xx - yy);
    o[17] = (-1.7701307697799304f * yz) * (3.f * xx - yy);
    o[18] = (0.9461746957575601f * xy) * (7.f * zz - 1.f);
    o[19] = (-0.6690465435572892f * yz) * (7.f * zz - 3.f);
    o[20] = 0.10578554691520431f * (zz * (35.f * zz - 30.f) + 3.f);
    o[21] = (-0.6690465435572892f * xz) * (7.f * zz - 3.f);
    o[22] = (0.47308734787878004f * (xx - yy)) * (7.f * zz - 1.f);
    o[23] = (-1.7701307697799304f * xz) * (xx - 3.f * yy);
    o[24] = 0.6258357354491761f * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
}

// ---- ndc_rays_blender / ndc_rays (dataLoader/ray_utils.py:88-124): rays -> normalised device coordinates; sign = -1 for the
// Blender (-z forward) form, +1 for the OpenCV form. Operation order of the reference's tensor expressions.
__global__ __launch_bounds__(256) void k_ndc_rays(int H, int W, float focal, float near, float sign, const float* __restrict__ ro,
                                                  const float* __restrict__ rd, long long n, float* o_out, float* d_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float ox = ro[i * 3], oy = ro[i * 3 + 1], oz = ro[i * 3 + 2];
    const float dx = rd[i * 3], dy = rd[i * 3 + 1], dz = rd[i * 3 + 2];
    const float t = sign < 0.f ? -(near + oz) / dz : (near - oz) / dz;
    ox = ox + t * dx; oy = oy + t * dy; oz = oz + t * dz;
    const float kx = sign * 1.f / ((float)W / (2.f * focal)), ky = sign * 1.f / ((float)H / (2.f * focal));
    const float o0 = kx * ox / oz, o1 = ky * oy / oz;
    const float o2 = sign < 0.f ? 1.f + 2.f * near / oz : 1.f - 2.f * near / oz;
    const float d0 = kx * (dx / dz - ox / oz), d1 = ky * (dy / dz - oy / oz);
    const float d2 = sign < 0.f ? -2.f * near / oz : 2.f * near / oz;
    o_out[i * 3] = o0; o_out[i * 3 + 1] = o1; o_out[i * 3 + 2] = o2;
    d_out[i * 3] = d0; d_out[i * 3 + 1] = d1; d_out[i * 3 + 2] = d2;
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

Carve carve_workspace(int64_t rays, int n_samples, bool ctx, bool feat, unsigned budget) {
    Carve c;
    c.list_cap = list_capacity_budget(rays, n_samples, budget);
    const size_t cap = (size_t)c.list_cap * kLists;
    size_t o = 0;
    c.counters = o; o = align_up(o + (size_t)kLists * kCounterStride * 4, 256);
    c.acc = o; o = align_up(o + (size_t)rays * 4, 256);
    c.ray_app = o; o = align_up(o + (size_t)rays * 16, 256);
    c.app_pos = o; o = align_up(o + cap * 16, 256);
    c.app_rgb = o; o = align_up(o + cap * 16, 256);
    c.app_ray = o; o = align_up(o + cap * 4, 256);
    c.sigma = o; if (ctx) o = align_up(o + (size_t)rays * n_samples * 4, 256);
    c.rgb_raw = o; if (ctx) o = align_up(o + (size_t)rays * 16, 256);
    // tile marcher: per-ray staging of up to n_samples / 4 appearance entries (+ one 256-B line: overflow counter) and an
    // overflow ray list
    c.scratch = o; if (ctx) o = align_up(o + (size_t)rays * (size_t)(n_samples / 4 > 0 ? n_samples / 4 : 1) * 16 + 256 + (size_t)rays * 4, 256);
    // appearance feature rows between the gather + basis kernel and the sample-stationary head: 32 rows per ray (the bench
    // scene needs ~7, BASELINE configs[0] ~44), never more than the worst case; appearance tiles past the capacity take the
    // one-kernel path
    const size_t worst_rows = cap + (size_t)kLists * 32, want_rows = (size_t)rays * (budget && budget < 64 ? (budget + 1) / 2 : 32) + 1024;
    c.feat_rows = (unsigned)(((worst_rows < want_rows ? worst_rows : want_rows) + 127) / 128 * 128);
    if (!feat) c.feat_rows = 0;
    c.feat = o; o = align_up(o + (size_t)c.feat_rows * 32 * sizeof(float), 256);
    c.total = o;
    return c;
}
static Carve carve(int64_t rays, int n_samples) { return carve_workspace(rays, n_samples, true); }
constexpr unsigned kBudgetMin = 16;          // smallest per-ray entry budget the hint asks for
constexpr unsigned kBudgetFloor = 2;         // smallest budget a launch is tried with (a caller's choice of workspace)
constexpr int64_t kBudgetMinRays = 65536;   // small calls keep worst-case lists (no host wait on the counters)

}  // namespace t2n

using namespace t2n;

extern "C" const char* t2n_last_error(void) { return g_err; }
extern "C" int t2n_version(void) { return 100; }

extern "C" int t2n_field_create(const t2n_field_desc* desc, t2n_field** out) {
    if (!desc || !out) { set_error("t2n_field_create: NULL argument"); return T2N_ERR_INVALID; }
    t2n_field* f = new t2n_field();
    memset(&f->dev, 0, sizeof(f->dev));
    for (int k = 0; k < T2N_K_COUNT; ++k) { memset(f->slots[k].start, 0, sizeof(f->slots[k].start)); memset(f->slots[k].stop, 0, sizeof(f->slots[k].stop)); }
    const int rc = apply_desc(f, desc);
    if (rc) { delete f; return rc; }
    *out = f;
    return T2N_OK;
}

extern "C" int t2n_field_set_desc(t2n_field* f, const t2n_field_desc* d) {
    if (!f || !d) { set_error("t2n_field_set_desc: NULL argument"); return T2N_ERR_INVALID; }
    if (f->uploaded && (d->grid[0] != f->desc.grid[0] || d->grid[1] != f->desc.grid[1] || d->grid[2] != f->desc.grid[2] ||
                        d->density_n_comp != f->desc.density_n_comp || d->app_n_comp != f->desc.app_n_comp)) {
        set_error("t2n_field_set_desc: grid/n_comp change needs a new field handle");
        return T2N_ERR_STATE;
    }
    return apply_desc(f, d);
}

extern "C" int t2n_field_destroy(t2n_field* f) {
    if (!f) return T2N_OK;
    for (int k = 0; k < 3; ++k) {
        if (f->buf_den_plane[k]) (void)hipFree(f->buf_den_plane[k]);
        if (f->buf_den_line[k]) (void)hipFree(f->buf_den_line[k]);
        if (f->buf_app_plane[k]) (void)hipFree(f->buf_app_plane[k]);
        if (f->buf_app_line[k]) (void)hipFree(f->buf_app_line[k]);
        if (f->hbuf_den_plane[k]) (void)hipFree(f->hbuf_den_plane[k]);
        if (f->hbuf_den_line[k]) (void)hipFree(f->hbuf_den_line[k]);
        if (f->hbuf_app_plane[k]) (void)hipFree(f->hbuf_app_plane[k]);
        if (f->hbuf_app_line[k]) (void)hipFree(f->hbuf_app_line[k]);
    }
    if (f->gbuf_all && !f->gbuf_external) (void)hipFree(f->gbuf_all);
    if (f->buf_mlp) (void)hipFree(f->buf_mlp);
    if (f->buf_mlp_h) (void)hipFree(f->buf_mlp_h);
    if (f->buf_ss) (void)hipFree(f->buf_ss);
    if (f->ss_event) (void)hipEventDestroy((hipEvent_t)f->ss_event);
    if (f->ss_ok_event) (void)hipEventDestroy((hipEvent_t)f->ss_ok_event);
    if (f->ss_ok_host) (void)hipHostFree(f->ss_ok_host);
    for (auto& c : f->count_slots) {
        if (c.host) (void)hipHostFree(c.host);
        if (c.ev) (void)hipEventDestroy((hipEvent_t)c.ev);
    }
    if (f->ev_fork) (void)hipEventDestroy((hipEvent_t)f->ev_fork);
    if (f->ev_join) (void)hipEventDestroy((hipEvent_t)f->ev_join);
    if (f->ev_den) (void)hipEventDestroy((hipEvent_t)f->ev_den);
    if (f->plan_host) (void)hipHostFree(f->plan_host);
    if (f->ev_pack) (void)hipEventDestroy((hipEvent_t)f->ev_pack);
    if (f->ev_fork2) (void)hipEventDestroy((hipEvent_t)f->ev_fork2);
    if (f->ev_join2) (void)hipEventDestroy((hipEvent_t)f->ev_join2);
    if (f->buf_alpha) (void)hipFree(f->buf_alpha);
    if (f->train_dev) (void)hipFree(f->train_dev);
    if (f->train_host) (void)hipHostFree(f->train_host);
    for (auto& e : f->train_ev) if (e) (void)hipEventDestroy((hipEvent_t)e);
    for (int k = 0; k < T2N_K_COUNT; ++k)
        for (int i = 0; i < kTimingEvents; ++i) {
            if (f->slots[k].start[i]) (void)hipEventDestroy(f->slots[k].start[i]);
            if (f->slots[k].stop[i]) (void)hipEventDestroy(f->slots[k].stop[i]);
        }
    delete f;
    return T2N_OK;
}

extern "C" int t2n_field_upload(t2n_field* f, const t2n_field_params* p, t2n_stream stream) {
    if (!f || !p) { set_error("t2n_field_upload: NULL argument"); return T2N_ERR_INVALID; }
    if (!p->basis_weight) { set_error("t2n_field_upload: NULL basis_weight"); return T2N_ERR_INVALID; }
    if ((f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW || head_is_generic(f->desc.shading)) && (!p->mlp_w0 || !p->mlp_b0 || !p->mlp_w1 || !p->mlp_b1 || !p->mlp_w2 || !p->mlp_b2)) {
        set_error("t2n_field_upload: MLP head needs all six renderModule tensors");
        return T2N_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    timing_begin(f, T2N_K_UPLOAD, s);
    int rc = launch_relayout(f, p, s);
    if (!rc) rc = launch_pack_mlp(f, p, s);
    timing_end(f, T2N_K_UPLOAD, s);
    if (rc) return rc;
    f->params_ref = *p;
    f->ss_dirty = true;
    f->train_packed = false;
    f->train_chain = false;
    f->uploaded = true;
    return T2N_OK;
}

extern "C" int t2n_field_upload_head(t2n_field* f, const t2n_field_params* p, t2n_stream stream) {
    if (!f || !p) { set_error("t2n_field_upload_head: NULL argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_field_upload_head: the field has no uploaded factors yet (t2n_field_upload first)"); return T2N_ERR_INVALID; }
    if (!p->basis_weight) { set_error("t2n_field_upload_head: NULL basis_weight"); return T2N_ERR_INVALID; }
    if ((f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW || head_is_generic(f->desc.shading)) && (!p->mlp_w0 || !p->mlp_b0 || !p->mlp_w1 || !p->mlp_b1 || !p->mlp_w2 || !p->mlp_b2)) {
        set_error("t2n_field_upload_head: MLP head needs all six renderModule tensors");
        return T2N_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    timing_begin(f, T2N_K_UPLOAD, s);
    const int rc = launch_pack_mlp(f, p, s);
    timing_end(f, T2N_K_UPLOAD, s);
    if (rc) return rc;
    f->params_ref.basis_weight = p->basis_weight;
    f->params_ref.mlp_w0 = p->mlp_w0; f->params_ref.mlp_b0 = p->mlp_b0; f->params_ref.mlp_w1 = p->mlp_w1; f->params_ref.mlp_b1 = p->mlp_b1;
    f->params_ref.mlp_w2 = p->mlp_w2; f->params_ref.mlp_b2 = p->mlp_b2;
    f->ss_dirty = true;
    f->train_packed = false;
    f->train_chain = false;
    return T2N_OK;
}

extern "C" int t2n_ray_directions(int H, int W, float fx, float fy, float cx, float cy, int normalize, float* dirs, t2n_stream stream) {
    if (H <= 0 || W <= 0 || !dirs) { set_error("t2n_ray_directions: bad argument"); return T2N_ERR_INVALID; }
    const long long n = (long long)H * W;
    hipLaunchKernelGGL(k_ray_dirs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, W, fx, fy, cx, cy, normalize, dirs);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_get_rays(const float* dirs, int64_t n, const float* c2w_host, float* rays_o, float* rays_d, float* rays6, t2n_stream stream) {
    if (!dirs || !c2w_host || n < 0) { set_error("t2n_get_rays: bad argument"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    Pose P;
    memcpy(P.m, c2w_host, sizeof(P.m));
    hipLaunchKernelGGL(k_get_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dirs, (long long)n, P, rays_o, rays_d, rays6);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_generate_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host, float* rays6, t2n_stream stream) {
    if (H <= 0 || W <= 0 || !c2w_host || !rays6) { set_error("t2n_generate_rays: bad argument"); return T2N_ERR_INVALID; }
    Pose P;
    memcpy(P.m, c2w_host, sizeof(P.m));
    const long long n = (long long)H * W;
    hipLaunchKernelGGL(k_generate_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, W, fx, fy, cx, cy, P, rays6);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_frame_postprocess(const float* rgb, const float* depth, int64_t n, float depth_sub, float depth_add, int shift_clamp,
                                     float mi, float ma, uint8_t* rgb8, uint8_t* depth8, const float* gt_rgb, double* sq_err_sum,
                                     t2n_stream stream) {
    if (!rgb || n < 0 || (depth8 && !depth) || (gt_rgb && !sq_err_sum)) { set_error("t2n_frame_postprocess: bad argument"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    const float den = (float)((double)ma - (double)mi + 1e-8);   // numpy: python-float scalar rounded to the array's fp32
    hipLaunchKernelGGL(k_frame_post, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rgb, depth, (long long)n,
                       depth_sub, depth_add, shift_clamp, mi, den, rgb8, depth8, gt_rgb, gt_rgb ? sq_err_sum : nullptr);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_ray_marcher(const float* rays, int64_t n, int ray_stride, int n_samples, int lindisp, const float* bbox_host,
                               const float* steps, const float* perturb, float* xyz, float* z_vals, float* near_far, t2n_stream stream) {
    if (!rays || n < 0 || n_samples < 1 || !steps || ray_stride < (bbox_host ? 6 : 8)) { set_error("t2n_ray_marcher: bad argument"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    MarcherArgs a;
    a.rays = rays; a.n = n; a.stride = ray_stride; a.S = n_samples; a.lindisp = lindisp ? 1 : 0; a.use_bbox = bbox_host ? 1 : 0;
    for (int k = 0; k < 3; ++k) { a.bmin[k] = bbox_host ? bbox_host[k] : 0.f; a.bmax[k] = bbox_host ? bbox_host[3 + k] : 0.f; }
    a.steps = steps; a.perturb = perturb; a.xyz = xyz; a.z_vals = z_vals; a.near_far = near_far;
    const long long tot = (long long)n * n_samples;
    hipLaunchKernelGGL(k_ray_marcher, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_eval_sh_bases(int deg, const float* dirs, int64_t n, float* out, t2n_stream stream) {
    if (deg < 0 || deg > 4 || n < 0 || (n && (!dirs || !out))) { set_error("t2n_eval_sh_bases: deg must be 0..4"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    hipLaunchKernelGGL(k_sh_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, deg, dirs, (long long)n, out);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_ndc_rays(int H, int W, float focal, float near, int blender, const float* rays_o, const float* rays_d, int64_t n,
                            float* o_out, float* d_out, t2n_stream stream) {
    if (n < 0 || (n && (!rays_o || !rays_d || !o_out || !d_out)) || H < 1 || W < 1) { set_error("t2n_ndc_rays: bad argument"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    hipLaunchKernelGGL(k_ndc_rays, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, H, W, focal, near,
                       blender ? -1.f : 1.f, rays_o, rays_d, (long long)n, o_out, d_out);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" size_t t2n_render_workspace_bytes(int64_t rays_per_launch, int n_samples) {
    if (rays_per_launch <= 0 || n_samples <= 0) return 0;
    return carve(rays_per_launch, n_samples).total;
}

// The general view-dependent heads (MLP_Fea / MLP_PE / MLP) evaluate their unfused MLP through activation scratch BEHIND the workspace
// sizes the functions above return: this many bytes more let one pass cover 32 rows per ray (rows beyond that are taken in further
// passes over the same scratch; any amount >= one 32-row tile works). 0 for the other heads.
extern "C" size_t t2n_render_head_scratch_bytes(const t2n_field* f, int64_t n_rays) {
    if (!f || n_rays <= 0 || !head_is_generic(f->desc.shading)) return 0;
    const size_t row = (size_t)(32 + head_dims(f->desc).K0pad + 128 + 128) * sizeof(float);
    return ((size_t)n_rays * (size_t)f->head_rows_per_ray + 32 * kLists) / 32 * 32 * row + 512;
}

extern "C" size_t t2n_render_workspace_bytes_ctx(int64_t n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    return carve_workspace(n_rays, n_samples, true, false).total;
}

namespace t2n {
__global__ __launch_bounds__(256) void k_setup(const SetupOps o) {
    const int r = blockIdx.y;
    if (r < o.nz) {
        unsigned* __restrict__ p = r == 0 ? o.zero_ptr[0] : (r == 1 ? o.zero_ptr[1] : (r == 2 ? o.zero_ptr[2] : (r == 3 ? o.zero_ptr[3] : (r == 4 ? o.zero_ptr[4] : o.zero_ptr[5]))));
        const unsigned long long n = r == 0 ? o.zero_words[0] : (r == 1 ? o.zero_words[1] : (r == 2 ? o.zero_words[2] : (r == 3 ? o.zero_words[3] : (r == 4 ? o.zero_words[4] : o.zero_words[5]))));
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) p[i] = 0u;
    }
    if (blockIdx.x == 0 && r == 0 && (int)threadIdx.x < o.ns) {
        const int t = threadIdx.x;
        unsigned* q = t == 0 ? o.set_ptr[0] : (t == 1 ? o.set_ptr[1] : (t == 2 ? o.set_ptr[2] : o.set_ptr[3]));
        *q = t == 0 ? o.set_val[0] : (t == 1 ? o.set_val[1] : (t == 2 ? o.set_val[2] : o.set_val[3]));
    }
}
// (a word both zeroed and set must not occur: the two parts of the kernel are unordered)
int launch_setup(const SetupOps& o, hipStream_t s) {
    if (o.overflow) { set_error("launch_setup: more initialisations than the kernel has slots"); return T2N_ERR_INVALID; }
    if (o.nz == 0 && o.ns == 0) return T2N_OK;
    unsigned long long mx = 1;
    for (int r = 0; r < o.nz; ++r) mx = o.zero_words[r] > mx ? o.zero_words[r] : mx;
    unsigned bx = (unsigned)((mx + 1023) / 1024);   // four words per thread
    bx = bx > 512 ? 512 : (bx < 1 ? 1 : bx);
    hipLaunchKernelGGL(k_setup, dim3(bx, o.nz > 0 ? (unsigned)o.nz : 1u), dim3(256), 0, s, o);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
}  // namespace t2n

namespace t2n {
// The counter block (2 KB) to pinned, device-visible host memory by ONE wave's stores: hipMemcpyAsync device -> host runs a blit kernel
// that costs the stream 11 us (+ ~6 us of dispatch gap) per frame / train step; this kernel costs its launch
__global__ __launch_bounds__(64) void k_post_counts(const unsigned* __restrict__ src, unsigned* __restrict__ host_dst, int words) {
    for (int i = threadIdx.x; i < words; i += 64) host_dst[i] = src[i];
    __threadfence_system();
}
int post_counts(const unsigned* counters_dev, unsigned* host_dst, hipStream_t s) {
    hipLaunchKernelGGL(k_post_counts, dim3(1), dim3(64), 0, s, counters_dev, host_dst, kLists * kCounterStride);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
// Budgeted launches post their counter block (sub-list fills, overflow word, entries of the rays the finisher took) to pinned host
// memory behind the march kernels, with an event; nobody waits for it. A later call that finds the event complete turns the
// counters into the next budget (list_hint) and the retry count. kCountSlots launches may be in flight; one more skips its post.
static void counts_consume(t2n_field* f, t2n_field::CountSlot& c) {
    const unsigned cap = list_capacity_budget(c.n_rays, c.n_samples, c.budget);
    unsigned long long used = c.host[kFailEntriesWord];
    for (int l = 0; l < kLists; ++l) {
        const unsigned n = c.host[l * kCounterStride];
        used += n < cap ? n : cap;
    }
    if (c.host[kOverflowWord]) f->list_retries++;
    f->list_hint = (unsigned)((used + (unsigned long long)c.n_rays - 1) / (unsigned long long)c.n_rays);
    if (!f->list_hint) f->list_hint = 1;
    c.pending = false;
}
void counts_poll(t2n_field* f, bool wait) {
    for (int i = 0; i < t2n_field::kCountSlots; ++i) {   // oldest first
        t2n_field::CountSlot& c = f->count_slots[(f->count_next + i) % t2n_field::kCountSlots];
        if (!c.pending) continue;
        if (wait) { if (hipEventSynchronize((hipEvent_t)c.ev) != hipSuccess) { (void)hipGetLastError(); break; } }
        else if (hipEventQuery((hipEvent_t)c.ev) != hipSuccess) { (void)hipGetLastError(); break; }
        counts_consume(f, c);
    }
}
static void counts_post(t2n_field* f, const unsigned* counters_dev, int64_t n_rays, int n_samples, unsigned budget, hipStream_t s) {
    t2n_field::CountSlot& c = f->count_slots[f->count_next];
    if (c.pending) return;   // the host is kCountSlots launches ahead of the device: this launch's counters are not looked at
    if (!c.host) {
        if (hipHostMalloc((void**)&c.host, sizeof(unsigned) * kLists * kCounterStride, hipHostMallocDefault) != hipSuccess) { c.host = nullptr; (void)hipGetLastError(); return; }
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {   // (a slot with a buffer but no event would never post again)
            (void)hipGetLastError(); (void)hipHostFree(c.host); c.host = nullptr; return;
        }
        c.ev = (void*)ev;
    }
    if (post_counts(counters_dev, c.host, s) != T2N_OK || hipEventRecord((hipEvent_t)c.ev, s) != hipSuccess) { (void)hipGetLastError(); return; }
    c.pending = true; c.n_rays = n_rays; c.n_samples = n_samples; c.budget = budget;
    f->count_next = (f->count_next + 1) % t2n_field::kCountSlots;
}
}  // namespace t2n

extern "C" int t2n_render_forward(t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags,
                                  const float* jitter, float* rgb, float* depth, float* weights, float* z_vals, uint64_t* stats,
                                  void* workspace, size_t workspace_bytes, t2n_stream stream) {
    if (!f || !rays || !rgb || !depth || !workspace || n_rays < 0 || ray_stride < 6) { set_error("t2n_render_forward: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_render_forward: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (n_samples <= 0 || n_samples > 2048) { set_error("t2n_render_forward: n_samples %d outside [1,2048]", n_samples); return T2N_ERR_UNSUPPORTED; }
    if ((flags & (T2N_FLAG_TRAIN | T2N_FLAG_NDC)) && !jitter) { set_error("t2n_render_forward: train / NDC mode needs the jitter draws / depth table"); return T2N_ERR_INVALID; }
    // NDC: the `jitter` argument carries the [n_samples] depth table; the kernels see it as FieldDev::ztab for this call
    struct ZtabScope { t2n_field* f; ~ZtabScope() { f->dev.ztab = nullptr; } } ztab_scope{f};
    f->dev.ztab = (flags & T2N_FLAG_NDC) ? jitter : nullptr;
    const bool ndc = (flags & T2N_FLAG_NDC) != 0;
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) { if (stats) T2N_HIP(hipMemsetAsync(stats, 0, sizeof(uint64_t) * T2N_STAT_COUNT, s)); return T2N_OK; }
    bool stats_pending = stats != nullptr;   // zeroed by the first sub-launch's setup kernel
    const bool keep = (flags & T2N_FLAG_KEEP_CTX) != 0;
    // tile marcher: image-ordered eval rays with a known width (hint), whole rows per sub-launch
    const bool tiles = (flags & T2N_FLAG_COHERENT) != 0 && !ndc && !keep && !(flags & T2N_FLAG_TRAIN) && f->frame_w >= 8 &&
                       n_rays % f->frame_w == 0 && n_rays / f->frame_w >= 8;
    if (keep) {
        if (carve_workspace(n_rays, n_samples, true, false).total > workspace_bytes || (uint64_t)list_capacity(n_rays, n_samples) * kLists > 0x7fffffffull) {
            set_error("t2n_render_forward: KEEP_CTX needs the whole call in one launch (workspace %zu B < %zu B)", workspace_bytes,
                      carve_workspace(n_rays, n_samples, true, false).total);
            return T2N_ERR_WORKSPACE;
        }
        if (!weights || !z_vals) { set_error("t2n_render_forward: KEEP_CTX needs weights and z_vals materialised"); return T2N_ERR_INVALID; }
    }
    // largest sub-launch whose worst case (every sample an appearance sample) fits the workspace
    int64_t per = n_rays;
    unsigned budget = 0;   // appearance entries per ray the lists are sized for (0: worst case)
    counts_poll(f, false);   // counters of earlier budgeted launches that have landed meanwhile: the hint, list_retries (never waits)
    // A frame for the tile marcher whose worst case does not fit is first tried as ONE launch with budgeted lists (the largest
    // budget the workspace holds): a C2 frame needs ~7 entries per ray where the worst case reserves 518. The counters reach
    // pinned host memory behind the march kernels; a launch that overflowed is redone below in worst-case sub-launches.
    const bool generic_head = head_is_generic(f->desc.shading);
    if (tiles && !generic_head && carve(n_rays, n_samples).total > workspace_bytes && n_rays >= kBudgetMinRays && !getenv("T2N_NO_BUDGET")) {
        unsigned lo = 0, hi = (unsigned)n_samples;   // largest budget in [kBudgetFloor, n_samples) that fits
        if (carve_workspace(n_rays, n_samples, true, true, kBudgetFloor).total <= workspace_bytes) {
            lo = kBudgetFloor;
            while (hi - lo > 1) {
                const unsigned mid = lo + (hi - lo) / 2;
                if (carve_workspace(n_rays, n_samples, true, true, mid).total <= workspace_bytes) lo = mid; else hi = mid;
            }
        }
        if (lo >= kBudgetFloor && (uint64_t)list_capacity_budget(n_rays, n_samples, lo) * kLists <= 0x7fffffffull) budget = lo;
    }

    // general view-dependent heads: activation scratch of the unfused MLP for kHeadRowsPerRay rows per ray behind the sub-launch's carve
    // (feat32 | X0 | h0 | h1 per row); rows beyond it are taken in further passes over the same scratch (see the loop below)
    const size_t head_row_bytes = generic_head ? (size_t)(32 + head_dims(f->desc).K0pad + 128 + 128) * sizeof(float) : 0;
    auto head_bytes = [&](int64_t rays) { return generic_head ? ((size_t)rays * (size_t)f->head_rows_per_ray + 32 * kLists) / 32 * 32 * head_row_bytes + 256 : (size_t)0; };
    if (!budget) {
        while (!keep && per > 1 && carve(per, n_samples).total + head_bytes(per) > workspace_bytes) per = (per + 1) / 2;
        if (!keep && carve(per, n_samples).total + head_bytes(per) > workspace_bytes) { set_error("t2n_render_forward: workspace %zu B too small", workspace_bytes); return T2N_ERR_WORKSPACE; }
        while ((uint64_t)list_capacity(per, n_samples) * kLists > 0x7fffffffull) per = (per + 1) / 2;   // int slots
        if (tiles && per < n_rays) {   // sub-launches must cover whole 8-row bands of the image
            const int64_t band = (int64_t)8 * f->frame_w;
            per = per / band * band;
            if (per <= 0) { set_error("t2n_render_forward: workspace too small for one 8-row band"); return T2N_ERR_WORKSPACE; }
        }
    }
    char* ws = (char*)workspace;
    for (int64_t off = 0; off < n_rays; off += per) {
        const int64_t cnt = (n_rays - off) < per ? (n_rays - off) : per;
        const Carve c = carve_workspace(per, n_samples, true, !keep, budget);
        RenderLaunch L;
        L.rays = rays + off * ray_stride; L.n_rays = cnt; L.ray_stride = ray_stride; L.n_samples = n_samples; L.flags = flags;
        L.jitter = (jitter && !ndc) ? jitter + off : jitter;     // NDC: one table for every sub-launch
        L.rgb = rgb + off * 3; L.depth = depth + off;
        L.weights = weights ? weights + off * n_samples : nullptr;
        L.z_vals = z_vals ? z_vals + off * n_samples : nullptr;
        L.stats = stats;
        L.counters = (unsigned*)(ws + c.counters); L.acc = (float*)(ws + c.acc); L.ray_app = (int4*)(ws + c.ray_app);
        L.app_pos = (float4*)(ws + c.app_pos); L.app_rgb = (float4*)(ws + c.app_rgb); L.app_ray = (int*)(ws + c.app_ray);
        L.list_cap = c.list_cap;
        L.feat = (float*)(ws + c.feat); L.feat_rows = c.feat_rows;
        L.sigma_ctx = keep ? (float*)(ws + c.sigma) : nullptr;
        L.rgb_raw = keep ? (float4*)(ws + c.rgb_raw) : nullptr;
        int rc;
        const KeptRows kr = keep && f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW ? kept_rows(c.total, workspace_bytes) : KeptRows{0, 0, 0, 0, 0};
        {   // the launch's counters (and, once per call, the statistics) zeroed, the kept-rows marker set: one kernel
            SetupOps so;
            const bool mark = kr.rows >= 32;   // recorded in the workspace itself (and in the counts' host copy): the backward checks it
            // the marker words lie inside the counter block: zero around them when they are set (the kernel's two parts are unordered)
            if (!mark) so.zero(L.counters, (size_t)kLists * kCounterStride * 4);
            else {
                const unsigned lo = kKeptMagicWord < kKeptRowsWord ? kKeptMagicWord : kKeptRowsWord, hi = kKeptMagicWord < kKeptRowsWord ? kKeptRowsWord : kKeptMagicWord;
                so.zero(L.counters, (size_t)lo * 4);
                if (hi > lo + 1) so.zero(L.counters + lo + 1, (size_t)(hi - lo - 1) * 4);
                if ((unsigned)(kLists * kCounterStride) > hi + 1) so.zero(L.counters + hi + 1, (size_t)(kLists * kCounterStride - hi - 1) * 4);
                so.set(L.counters + kKeptMagicWord, kKeptMagic);
                so.set(L.counters + kKeptRowsWord, (unsigned)kr.rows);
            }
            if (stats_pending) { so.zero(stats, sizeof(uint64_t) * T2N_STAT_COUNT); stats_pending = false; }
            if (tiles) {   // the tile marcher's overflow-ray counter (behind the per-ray staging slices: launch_march_tiles)
                const int cap_e = n_samples / 4 > 0 ? n_samples / 4 : 1;
                so.zero(ws + c.scratch + (size_t)cnt * cap_e * 16, 4);
            }
            if ((rc = launch_setup(so, s))) return rc;
        }
        if (tiles) {
            if ((rc = launch_march_tiles(f, L, f->frame_w, (int)(cnt / f->frame_w), (float*)(ws + c.sigma), (float4*)(ws + c.scratch), s))) return rc;
        } else if ((rc = launch_march(f, L, s))) return rc;
        if (budget) counts_post(f, L.counters, n_rays, n_samples, budget, s);   // -> pinned host memory behind the march; read by a LATER call
        if (keep && (rc = ctx_counts_post(workspace, L.counters, s))) return rc;   // the backward sizes itself from these without draining the stream
        if (generic_head) {
            // general head path, no host read and no allocation: one wave derives tile prefix + row count from the
            // counters (k_head_plan, in the counter block's spare words), the activation scratch is the workspace behind the carve, and
            // the rows are taken in PASSES of the scratch's capacity — gather + basis (feat32) -> head input rows -> two dense layers ->
            // layer 2 + sigmoid into the list — every kernel clipping to the device-side count. The host issues the passes the WORST
            // case needs (every sample an appearance sample); a pass past the count is six empty launches.
            const HeadDims H = head_dims(f->desc);
            HeadPlanDev* plan = (HeadPlanDev*)(L.counters + kHeadPlanWord);
            if ((rc = launch_head_plan(L.counters, L.list_cap, plan, s))) return rc;
            const size_t off = align_up(c.total, 256);
            size_t avail = workspace_bytes > off ? workspace_bytes - off : 0;
            long long cap = (long long)(avail / head_row_bytes) / 32 * 32;
            if (cap > 0x7fffffe0ll) cap = 0x7fffffe0ll;
            if (cap < 32) {
                set_error("t2n_render_forward: the workspace holds no activation row of the general head (add t2n_render_head_scratch_bytes)");
                return T2N_ERR_WORKSPACE;
            }
            {
                float* feat32 = (float*)(ws + off); float* x0 = feat32 + cap * 32; float* h0 = x0 + cap * H.K0pad; float* h1 = h0 + cap * 128;
                const long long worst = ((long long)L.list_cap + 31) / 32 * 32 * kLists;    // padded rows if every slot of every sub-list were used
                const long long worst_rows = worst < (cnt * (long long)n_samples + 32ll * kLists) ? worst : (cnt * (long long)n_samples + 32ll * kLists);
                for (long long row0 = 0; row0 < worst_rows; row0 += cap) {
                    // the kernels index activation rows by their number in the call: base pointers shifted so that row row0 is scratch row 0
                    ShadeCtx ctx{nullptr, feat32 - row0 * 32, nullptr, nullptr};
                    if ((rc = launch_shade_list(f, L.app_pos, L.app_ray, L.rays, ray_stride, L.counters, L.list_cap, L.app_rgb, &ctx, s, true,
                                                (unsigned)(row0 + cap), nullptr, 0, nullptr, (unsigned)(row0 / 32), (unsigned)((row0 + cap) / 32)))) return rc;
                    if ((rc = launch_head_forward(f, nullptr, cap, feat32, L.app_pos, L.app_ray, L.rays, ray_stride, L.counters, L.list_cap, x0, h0,
                                                  h1, L.app_rgb, s, plan, row0))) return rc;
                }
            }
        } else {
            // training forward with spare workspace: run the activation-keeping kernel now, so the backward need not re-run
            // the appearance forward (rows past the capacity keep nothing; the backward then recomputes)
            if (kr.rows >= 32) {
                ShadeCtx ctx{(float*)(ws + kr.x144), (float*)(ws + kr.feat32), (float*)(ws + kr.h0), (float*)(ws + kr.h1)};
                if ((rc = launch_shade_list(f, L.app_pos, L.app_ray, L.rays, ray_stride, L.counters, L.list_cap, L.app_rgb, &ctx, s, false, kr.rows))) return rc;
            } else if ((rc = launch_shade_list(f, L.app_pos, L.app_ray, L.rays, ray_stride, L.counters, L.list_cap, L.app_rgb, nullptr, s, false,
                                               0xffffffffu, keep ? nullptr : L.feat, L.feat_rows, stats))) return rc;
        }
        if ((rc = tiles ? launch_composite(f, L, s, f->frame_w, (int)(cnt / f->frame_w)) : launch_composite(f, L, s))) return rc;
        // the rays that found no room in the lists (budgeted lists; with worst-case lists a ray cannot be left over, but the
        // compaction kernel tags by the same rule and a tagged ray is only ever coloured here) are shaded and composited from their
        // staging slices, on the device
        if (tiles && finish_supported(f) && (rc = launch_finish_rays(f, L, (const float*)(ws + c.sigma), (const float4*)(ws + c.scratch), s))) return rc;
    }
    return T2N_OK;
}

// Workspace for one budgeted launch of a frame, from what the field's last such launch needed: twice the entries per ray it
// used + 16, at least kBudgetMin; 32 entries per ray when nothing is known yet. Never more than the worst case.
extern "C" size_t t2n_render_workspace_bytes_hint(const t2n_field* f, int64_t n_rays, int n_samples) {
    if (!f || n_rays <= 0 || n_samples <= 0) return 0;
    counts_poll(const_cast<t2n_field*>(f), false);
    const size_t worst = carve(n_rays, n_samples).total;
    if (n_rays < kBudgetMinRays) return worst;
    // nothing known yet: 32 entries per ray (the driver's scenes need ~7; a scene that needs more has its surplus rays finished on the
    // device by k_finish_rays for the frame or two until the counters of the first launches have landed)
    unsigned b = f->list_hint ? 2u * f->list_hint + 16u : 32u;
    if (b < kBudgetMin) b = kBudgetMin;
    if (b >= (unsigned)n_samples) return worst;
    const size_t want = carve_workspace(n_rays, n_samples, true, true, b).total;
    return want < worst ? want : worst;
}
extern "C" uint64_t t2n_field_list_retries(const t2n_field* f) {
    if (!f) return 0;
    counts_poll(const_cast<t2n_field*>(f), true);   // a query, not a render: waits for the counters still on their way
    return f->list_retries;
}
extern "C" size_t t2n_render_workspace_bytes_budget(int64_t n_rays, int n_samples, int entries_per_ray) {
    if (n_rays <= 0 || n_samples <= 0 || entries_per_ray < 0) return 0;
    return carve_workspace(n_rays, n_samples, true, true, (unsigned)entries_per_ray).total;
}

extern "C" int t2n_field_set_alpha_mask(t2n_field* f, const float* volume, int D, int H, int W, const float* aabb_min_host,
                                        const float* inv_size_host, t2n_stream stream) {
    if (!f) { set_error("t2n_field_set_alpha_mask: NULL field"); return T2N_ERR_INVALID; }
    if (f->buf_alpha) { T2N_HIP(hipStreamSynchronize((hipStream_t)stream)); (void)hipFree(f->buf_alpha); f->buf_alpha = nullptr; }
    f->dev.alpha = nullptr;
    if (!volume) return T2N_OK;   // clear
    if (D < 1 || H < 1 || W < 1 || !aabb_min_host || !inv_size_host) { set_error("t2n_field_set_alpha_mask: bad argument"); return T2N_ERR_INVALID; }
    const size_t bytes = (size_t)D * H * W * sizeof(float);
    T2N_HIP(hipMalloc((void**)&f->buf_alpha, bytes));
    T2N_HIP(hipMemcpyAsync(f->buf_alpha, volume, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    f->dev.alpha = f->buf_alpha; f->dev.aD = D; f->dev.aH = H; f->dev.aW = W;
    for (int k = 0; k < 3; ++k) { f->dev.a_min[k] = aabb_min_host[k]; f->dev.a_inv[k] = inv_size_host[k]; }
    return T2N_OK;
}

extern "C" int t2n_field_set_head_scratch_rows(t2n_field* f, int rows_per_ray) {
    if (!f || rows_per_ray < 1 || rows_per_ray > 4096) { set_error("t2n_field_set_head_scratch_rows: rows_per_ray must lie in [1, 4096]"); return T2N_ERR_INVALID; }
    f->head_rows_per_ray = rows_per_ray;
    return T2N_OK;
}

extern "C" int t2n_field_set_frame_width(t2n_field* f, int width) {
    if (!f || width < 0) { set_error("t2n_field_set_frame_width: bad argument"); return T2N_ERR_INVALID; }
    f->frame_w = width;
    return T2N_OK;
}

extern "C" int t2n_field_set_factor_storage(t2n_field* f, int bf16) {
    if (!f) { set_error("t2n_field_set_factor_storage: NULL field"); return T2N_ERR_INVALID; }
    if ((bf16 != 0) != (f->factor_bf16 != 0)) f->uploaded = false;   // the next render needs a fresh t2n_field_upload
    f->factor_bf16 = bf16 ? 1 : 0;
    if (!f->factor_bf16)
        for (int k = 0; k < 3; ++k) {
            f->dev.den.plane_h[k] = f->dev.den.line_h[k] = f->dev.app.plane_h[k] = f->dev.app.line_h[k] = nullptr;
        }
    return T2N_OK;
}

extern "C" int t2n_field_set_early_termination(t2n_field* f, float eps) {
    if (!f || !(eps >= 0.f) || eps > f->desc.weight_thres) {   // above the appearance threshold a skipped sample could have been a list entry
        set_error("t2n_field_set_early_termination: eps must lie in [0, weight_thres]");
        return T2N_ERR_INVALID;
    }
    f->term_eps = eps;
    return T2N_OK;
}

extern "C" int t2n_field_set_mlp_precision(t2n_field* f, int exact_fp32) {
    if (!f) { set_error("t2n_field_set_mlp_precision: NULL field"); return T2N_ERR_INVALID; }
    f->mlp_split = exact_fp32 ? 0 : 1;
    return T2N_OK;
}

extern "C" int t2n_timing_enable(t2n_field* f, int on) {
    if (!f) return T2N_ERR_INVALID;
    // 0: off; 1: every kernel; otherwise a mask, bit k + 1 = kernel id k (an event pair costs the stream ~10 us of bubble per bracketed
    // group: a caller who needs one kernel's time inside a timed region asks for that kernel only)
    f->timing = on == 0 ? 0 : (on == 1 ? (1 << T2N_K_COUNT) - 1 : (on >> 1) & ((1 << T2N_K_COUNT) - 1));
    return T2N_OK;
}

extern "C" int t2n_timing_read(t2n_field* f, double* ms, int64_t* launches, int reset) {
    if (!f || !ms || !launches) { set_error("t2n_timing_read: NULL argument"); return T2N_ERR_INVALID; }
    timing_flush(f);
    for (int k = 0; k < T2N_K_COUNT; ++k) {
        ms[k] = f->slots[k].ms; launches[k] = f->slots[k].launches;
        if (reset) { f->slots[k].ms = 0.0; f->slots[k].launches = 0; }
    }
    return T2N_OK;
}
