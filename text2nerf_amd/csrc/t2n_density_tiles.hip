// Coherent-ray density evaluation (eval frames): sigma[ray][sample] for every valid sample, computed 16 adjacent rays at
// a time so that the taps the rays share are fetched from HBM/L2 ONCE and re-read from LDS.
//
// Why: k_march's pass B is bound by the texture-addresser (PMC, profiles/: TA busy 75-84 %, ~35 cycles per scattered
// dwordx4 instruction, 18 instructions per 16 samples). Neighbouring rays of an image at the same sample index lie within
// a few texels of each other (pixel pitch t/f << texel), so the union footprint of 16 rays x 4 taps is a small rectangle.
//
// Mapping (gfx950, wave64): one wave = 16 consecutive rays x one sample index per step; 4 lanes per ray (channel quads).
//   per step and factor pair k: exact wave-wide min/max of the tap rectangle (packed u16 min/max + DPP/bpermute);
//   if every rectangle fits the wave's LDS staging area: cooperative, row-contiguous loads of the rectangle (each texel
//   once) -> LDS -> every lane reads its 4+2 taps with ds_read_b128; otherwise (incoherent rays) the step falls back to
//   the direct per-lane gathers of k_march. Arithmetic per sample is identical to k_march's pass B (same taps, same FMA
//   order), so sigma is bitwise the same on either path.
// The transmittance scan / compaction stay in k_march (SIGMA_IN variant), which reads sigma from the buffer written here.
#include "t2n_device.h"

namespace t2n {

constexpr int kPlaneTexels = 48;   // staging capacity per plane rectangle (texels of 16 channels = 64 B)
constexpr int kLineTexels = 16;
constexpr int kStageFloats = 3 * kPlaneTexels * 16 + 3 * kLineTexels * 16;   // per wave

struct TileArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples;
    const float* jitter;
    float* sigma;   // [n_rays, n_samples]; only valid samples are written
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    u16x2 r = __builtin_elementwise_min(x, y);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    u16x2 r = __builtin_elementwise_max(x, y);
    return __builtin_bit_cast(unsigned, r);
}
// reduce over the 16 quads of a wave (all 4 lanes of a quad hold the same value)
__device__ __forceinline__ unsigned wave_pk_min(unsigned v) {
    v = pk_min(v, (unsigned)__shfl_xor((int)v, 4));
    v = pk_min(v, (unsigned)__shfl_xor((int)v, 8));
    v = pk_min(v, (unsigned)__shfl_xor((int)v, 16));
    v = pk_min(v, (unsigned)__shfl_xor((int)v, 32));
    return v;
}
__device__ __forceinline__ unsigned wave_pk_max(unsigned v) {
    v = pk_max(v, (unsigned)__shfl_xor((int)v, 4));
    v = pk_max(v, (unsigned)__shfl_xor((int)v, 8));
    v = pk_max(v, (unsigned)__shfl_xor((int)v, 16));
    v = pk_max(v, (unsigned)__shfl_xor((int)v, 32));
    return v;
}

__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct PlaneTaps { int x0, x1, y0, y1, l0, l1; float wnw, wne, wsw, wse, wl0, wl1; };

template <int K>
__device__ __forceinline__ void plane_taps(const FactorSet& S, float xn, float yn, float zn, PlaneTaps& t) {
    float gx, gy, gv;
    plane_line_coords<K>(xn, yn, zn, gx, gy, gv);
    const Axis ax = axis_taps(gx, S.W[K]);
    const Axis ay = axis_taps(gy, S.H[K]);
    const Axis al = axis_taps(gv, S.L[K]);
    t.x0 = ax.i0; t.x1 = ax.i1; t.y0 = ay.i0; t.y1 = ay.i1; t.l0 = al.i0; t.l1 = al.i1;
    t.wnw = ay.w0 * ax.w0; t.wne = ay.w0 * ax.w1; t.wsw = ay.w1 * ax.w0; t.wse = ay.w1 * ax.w1;
    t.wl0 = al.w0; t.wl1 = al.w1;
}

__device__ __forceinline__ float quad_dot(const QuadTaps& t, float part, bool first) {
    const float4 p = taps_plane(t), l = taps_line(t);
    part = first ? p.x * l.x : fmaf(p.x, l.x, part);
    part = fmaf(p.y, l.y, part); part = fmaf(p.z, l.z, part); part = fmaf(p.w, l.w, part);
    return part;
}

template <bool TRAIN>
__global__ __launch_bounds__(256) void k_density_tiles(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * kStageFloats];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int q = lane & 3, rl = lane >> 2;
    float4* __restrict__ stage = reinterpret_cast<float4*>(smem + (size_t)wid * kStageFloats);
    float4* __restrict__ stP[3] = {stage, stage + kPlaneTexels * 4, stage + 2 * kPlaneTexels * 4};
    float4* __restrict__ stL[3] = {stage + 3 * kPlaneTexels * 4, stage + 3 * kPlaneTexels * 4 + kLineTexels * 4,
                                   stage + 3 * kPlaneTexels * 4 + 2 * kLineTexels * 4};
    const FieldDev& F = a.F;
    const long long g = (long long)blockIdx.x * 4 + wid;
    const long long r = g * 16 + rl;
    const bool have = r < a.n_rays;
    if (g * 16 >= a.n_rays) return;
    const int N = a.n_samples;
    Ray ray;
    if (have) ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
    else { ray.ox = ray.oy = ray.oz = 1e30f; ray.dx = ray.dy = ray.dz = 0.f; ray.tmin = 0.f; ray.last = 0.f; }
    const float u = (TRAIN && have) ? a.jitter[r] : 0.f;

    // conservative sample interval of this ray (slab test in t, +-2 samples); exact validity is re-tested per sample
    int lo = N, hi = -1;
    if (have) {
        float t0 = -3.0e38f, t1 = 3.0e38f;
        const float o[3] = {ray.ox, ray.oy, ray.oz}, d[3] = {ray.dx, ray.dy, ray.dz};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (d[k] != 0.f) {
                const float ta = (F.aabb0[k] - o[k]) / d[k], tb = (F.aabb1[k] - o[k]) / d[k];
                t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
            } else if (o[k] < F.aabb0[k] || o[k] > F.aabb1[k]) { t1 = -3.0e38f; }
        }
        if (!TRAIN && ray.dz != 0.f) {   // eval gate: world z > zgate
            const float tg = (F.zgate - ray.oz) / ray.dz;
            if (ray.dz > 0.f) t0 = fmaxf(t0, tg); else t1 = fminf(t1, tg);
        } else if (!TRAIN && !(ray.oz > F.zgate)) { t1 = -3.0e38f; }
        if (t1 >= t0) {
            const float flo = floorf((t0 - ray.tmin) / F.step) - 3.f, fhi = ceilf((t1 - ray.tmin) / F.step) + 3.f;
            lo = (int)fminf(fmaxf(flo, 0.f), (float)N);
            hi = (int)fminf(fmaxf(fhi, -1.f), (float)(N - 1));
        }
    }
    int wlo = lo, whi = hi;
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { wlo = min(wlo, __shfl_xor(wlo, o)); whi = max(whi, __shfl_xor(whi, o)); }

    for (int i = wlo; i <= whi; ++i) {
        float xn = 0.f, yn = 0.f, zn = 0.f;
        bool ok = false;
        if (have && i >= lo && i <= hi) {
            const float z = sample_z<TRAIN>(F, ray, i, u);
            ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
        }
        if (!__any(ok)) continue;
        PlaneTaps t0, t1, t2;
        plane_taps<0>(F.den, xn, yn, zn, t0);
        plane_taps<1>(F.den, xn, yn, zn, t1);
        plane_taps<2>(F.den, xn, yn, zn, t2);
        // tap rectangles over the wave (invalid lanes are neutral)
        const unsigned big = 0xffffffffu;
        const unsigned mn0 = wave_pk_min(ok ? ((unsigned)t0.x0 | ((unsigned)t0.y0 << 16)) : big);
        const unsigned mx0 = wave_pk_max(ok ? ((unsigned)t0.x1 | ((unsigned)t0.y1 << 16)) : 0u);
        const unsigned mn1 = wave_pk_min(ok ? ((unsigned)t1.x0 | ((unsigned)t1.y0 << 16)) : big);
        const unsigned mx1 = wave_pk_max(ok ? ((unsigned)t1.x1 | ((unsigned)t1.y1 << 16)) : 0u);
        const unsigned mn2 = wave_pk_min(ok ? ((unsigned)t2.x0 | ((unsigned)t2.y0 << 16)) : big);
        const unsigned mx2 = wave_pk_max(ok ? ((unsigned)t2.x1 | ((unsigned)t2.y1 << 16)) : 0u);
        const unsigned lmnA = wave_pk_min(ok ? ((unsigned)t0.l0 | ((unsigned)t1.l0 << 16)) : big);
        const unsigned lmxA = wave_pk_max(ok ? ((unsigned)t0.l1 | ((unsigned)t1.l1 << 16)) : 0u);
        const unsigned lmnB = wave_pk_min(ok ? (unsigned)t2.l0 : big);
        const unsigned lmxB = wave_pk_max(ok ? (unsigned)t2.l1 : 0u);
        const int bx[3] = {(int)(mn0 & 0xffff), (int)(mn1 & 0xffff), (int)(mn2 & 0xffff)};
        const int by[3] = {(int)(mn0 >> 16), (int)(mn1 >> 16), (int)(mn2 >> 16)};
        const int bw[3] = {(int)(mx0 & 0xffff) - bx[0] + 1, (int)(mx1 & 0xffff) - bx[1] + 1, (int)(mx2 & 0xffff) - bx[2] + 1};
        const int bh[3] = {(int)(mx0 >> 16) - by[0] + 1, (int)(mx1 >> 16) - by[1] + 1, (int)(mx2 >> 16) - by[2] + 1};
        const int bl[3] = {(int)(lmnA & 0xffff), (int)(lmnA >> 16), (int)(lmnB & 0xffff)};
        const int ln[3] = {(int)(lmxA & 0xffff) - bl[0] + 1, (int)(lmxA >> 16) - bl[1] + 1, (int)(lmxB & 0xffff) - bl[2] + 1};
        const bool fits = bw[0] * bh[0] <= kPlaneTexels && bw[1] * bh[1] <= kPlaneTexels && bw[2] * bh[2] <= kPlaneTexels &&
                          ln[0] <= kLineTexels && ln[1] <= kLineTexels && ln[2] <= kLineTexels;
        float part = 0.f;
        if (fits) {
            // cooperative stage: rectangle rows are contiguous runs of bw*4 float4 in the channel-last plane
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4* __restrict__ P = reinterpret_cast<const float4*>(F.den.plane[k]);
                const int rowq = bw[k] * 4, total = rowq * bh[k];
                const int W = F.den.W[k];
                for (int it = lane; it < total; it += 64) {
                    const int row = it / rowq, c = it - row * rowq;
                    stP[k][it] = P[((size_t)(by[k] + row) * W + bx[k]) * 4 + c];
                }
                const float4* __restrict__ Ln = reinterpret_cast<const float4*>(F.den.line[k]);
                if (lane < ln[k] * 4) stL[k][lane] = Ln[(size_t)bl[k] * 4 + lane];
            }
            lds_fence();
            if (ok) {
                QuadTaps t;
#define T2N_LDS_TAPS(K, TT)                                                                                         \
                {                                                                                                   \
                    const int rx0 = TT.x0 - bx[K], rx1 = TT.x1 - bx[K], ry0 = (TT.y0 - by[K]) * bw[K], ry1 = (TT.y1 - by[K]) * bw[K]; \
                    t.nw = stP[K][(ry0 + rx0) * 4 + q]; t.ne = stP[K][(ry0 + rx1) * 4 + q];                          \
                    t.sw = stP[K][(ry1 + rx0) * 4 + q]; t.se = stP[K][(ry1 + rx1) * 4 + q];                          \
                    t.l0 = stL[K][(TT.l0 - bl[K]) * 4 + q]; t.l1 = stL[K][(TT.l1 - bl[K]) * 4 + q];                  \
                    t.wnw = TT.wnw; t.wne = TT.wne; t.wsw = TT.wsw; t.wse = TT.wse; t.wl0 = TT.wl0; t.wl1 = TT.wl1;  \
                }
                T2N_LDS_TAPS(0, t0) part = quad_dot(t, part, true);
                T2N_LDS_TAPS(1, t1) part = quad_dot(t, part, false);
                T2N_LDS_TAPS(2, t2) part = quad_dot(t, part, false);
#undef T2N_LDS_TAPS
            }
            lds_fence();   // reads done before the next step restages
        } else if (ok) {
            QuadTaps u0, u1, u2;
            issue_taps<0>(F.den, 4, q, xn, yn, zn, u0);
            issue_taps<1>(F.den, 4, q, xn, yn, zn, u1);
            issue_taps<2>(F.den, 4, q, xn, yn, zn, u2);
            part = quad_dot(u0, part, true);
            part = quad_dot(u1, part, false);
            part = quad_dot(u2, part, false);
        }
        const float feat = group_sum<4>(part);
        if (ok && q == 0) a.sigma[r * N + i] = feature2density(F, feat);
    }
}

int launch_density_tiles(t2n_field* f, const RenderLaunch& L, float* sigma, hipStream_t s) {
    TileArgs a;
    a.F = f->dev;
    a.rays = L.rays; a.n_rays = L.n_rays; a.ray_stride = L.ray_stride; a.n_samples = L.n_samples; a.jitter = L.jitter;
    a.sigma = sigma;
    const long long groups = (L.n_rays + 15) / 16;
    const unsigned nb = (unsigned)((groups + 3) / 4);
    timing_begin(f, T2N_K_DENSITY, s);
    if (L.flags & T2N_FLAG_TRAIN) hipLaunchKernelGGL((k_density_tiles<true>), dim3(nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((k_density_tiles<false>), dim3(nb), dim3(256), 0, s, a);
    timing_end(f, T2N_K_DENSITY, s);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n
