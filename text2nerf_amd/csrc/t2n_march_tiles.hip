// Tile marcher (eval frames in image order): 8x8-pixel tiles, ONE LANE PER RAY, all 64 rays of a tile advance through the
// same sample index together. At a given depth the 64 rays lie within a few texels of each other (pixel pitch t/f is far
// below the texel size), so per factor pair the union of their bilinear footprints is a small rectangle: the wave fetches
// that rectangle ONCE (row-contiguous, coalesced), stages it in LDS and every lane reads its 4+2 taps x 16 channels with
// ds_read_b128. Against k_march's pass B (16 samples x 4 lanes per step, 18 scattered 64-B gathers per sample; PMC: texture
// addresser 71 % busy, VALU ~70 %) this removes ~15x of the addresser work and the 4x-redundant per-sample coordinate
// math. The transmittance is a per-lane running product in sample order — exactly the reference's cumprod order
// (models/tensorBase.py:23) — so no wave scan is needed; acc/depth accumulate in registers.
//
// Outputs per ray: weights of the valid window into wbuf[ray][sample] (the caller's weights tensor or scratch), acc, depth,
// (n_app, n_valid, first | Lw << 11). The appearance list is then built by k_compact (one wave per ray: reservation atomic,
// ballot/prefix compaction), which keeps the per-ray contiguous, sample-ordered slices k_shade / k_composite rely on.
// Steps whose rectangles do not fit the staging area (incoherent rays, huge field of view) fall back to direct gathers.
// Replaces the same reference lines as k_march (see t2n_march.hip).
#include "t2n_device.h"

namespace t2n {

constexpr int kRectTexels = 32;    // staging capacity per plane rectangle (texels of 16 channels = 64 B each)
constexpr int kLineRows = 8;       // staging capacity per line segment
constexpr int kStageF4 = 3 * kRectTexels * 4 + 3 * kLineRows * 4;   // float4 per wave

struct TileArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples; int img_w, img_h;
    float* wbuf;        // [n_rays, n_samples]
    float* acc; float* depth; int4* ray_app;
    // in-kernel compaction (lazy-output eval renders): every lane stages its ray's appearance entries (<= cap) in `scratch`; at
    // the end the wave reserves ONE contiguous region of the appearance list for its 64 rays and copies the entries there,
    // ray by ray in sample order. Rays with more than cap entries go to `ovf_list` and are compacted from wbuf by
    // k_compact_list. scratch == NULL: the separate k_compact pass builds the lists (weights / z_vals requested).
    float4* scratch; int cap; unsigned* ovf_count; int* ovf_list;
    float4* app_pos; int* app_ray; unsigned* counters; unsigned list_cap; unsigned long long* stats;
};

// 16 B per lane global -> LDS without passing through registers; `lds_base` must be wave-uniform (lane l lands at base + 16 l)
__device__ __forceinline__ void glds16(const void* g, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_fence_w() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide min / max with short dependency chains: DPP row_shr scans inside each 16-lane row (lane 15 of a row ends up with
// the row result), then four v_readlane + scalar min/max. (A __shfl_xor butterfly is six dependent ds_bpermute round trips,
// ~2 us per step for the six reductions of a marching step.)
template <int CTRL>
__device__ __forceinline__ int dpp_shr(int v, int identity) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_min_i(int v) {
    const int big = 0x7fffffff;
    v = min(v, dpp_shr<0x111>(v, big));
    v = min(v, dpp_shr<0x112>(v, big));
    v = min(v, dpp_shr<0x114>(v, big));
    v = min(v, dpp_shr<0x118>(v, big));
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
    const int c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ int wave_max_i(int v) {
    const int small = -0x7fffffff;
    v = max(v, dpp_shr<0x111>(v, small));
    v = max(v, dpp_shr<0x112>(v, small));
    v = max(v, dpp_shr<0x114>(v, small));
    v = max(v, dpp_shr<0x118>(v, small));
    const int a = __builtin_amdgcn_readlane(v, 15), b = __builtin_amdgcn_readlane(v, 31);
    const int c = __builtin_amdgcn_readlane(v, 47), d = __builtin_amdgcn_readlane(v, 63);
    return max(max(a, b), max(c, d));
}

// accumulate the 16-channel dot product of (bilinear plane value) x (linear line value) for one factor pair from LDS
__device__ __forceinline__ float pair_dot_lds(const float4* __restrict__ P, const float4* __restrict__ L, int nw, int ne, int sw, int se,
                                              int l0, int l1, float wnw, float wne, float wsw, float wse, float wl0, float wl1,
                                              float part) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = f4_mul(P[nw + q], wnw);
        v = f4_fma(P[ne + q], wne, v);
        v = f4_fma(P[sw + q], wsw, v);
        v = f4_fma(P[se + q], wse, v);
        float4 l = f4_mul(L[l0 + q], wl0);
        l = f4_fma(L[l1 + q], wl1, l);
        part = fmaf(v.x, l.x, part); part = fmaf(v.y, l.y, part); part = fmaf(v.z, l.z, part); part = fmaf(v.w, l.w, part);
    }
    return part;
}

template <int K>
__device__ __forceinline__ float pair_dot_global(const FactorSet& S, const Axes3& A, float part) {
    const Axis& ax = A.a[mat0(K)];
    const Axis& ay = A.a[mat1(K)];
    const Axis& al = A.a[vecm(K)];
    const float4* __restrict__ P = reinterpret_cast<const float4*>(S.plane[K]);
    const float4* __restrict__ L = reinterpret_cast<const float4*>(S.line[K]);
    const unsigned W = (unsigned)S.W[K];
    const unsigned nw = ((unsigned)ay.i0 * W + ax.i0) * 4, ne = ((unsigned)ay.i0 * W + ax.i1) * 4;
    const unsigned sw = ((unsigned)ay.i1 * W + ax.i0) * 4, se = ((unsigned)ay.i1 * W + ax.i1) * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 v = f4_mul(P[nw + q], ay.w0 * ax.w0);
        v = f4_fma(P[ne + q], ay.w0 * ax.w1, v);
        v = f4_fma(P[sw + q], ay.w1 * ax.w0, v);
        v = f4_fma(P[se + q], ay.w1 * ax.w1, v);
        float4 l = f4_mul(L[al.i0 * 4 + q], al.w0);
        l = f4_fma(L[al.i1 * 4 + q], al.w1, l);
        part = fmaf(v.x, l.x, part); part = fmaf(v.y, l.y, part); part = fmaf(v.z, l.z, part); part = fmaf(v.w, l.w, part);
    }
    return part;
}

__global__ __launch_bounds__(256) void k_march_tiles(const TileArgs a) {
    __shared__ __attribute__((aligned(16))) float4 smem[4 * kStageF4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4* __restrict__ stage = smem + (size_t)wid * kStageF4;
    float4* __restrict__ stP[3] = {stage, stage + kRectTexels * 4, stage + 2 * kRectTexels * 4};
    float4* __restrict__ stL[3] = {stage + 3 * kRectTexels * 4, stage + 3 * kRectTexels * 4 + kLineRows * 4,
                                   stage + 3 * kRectTexels * 4 + 2 * kLineRows * 4};
    const FieldDev& F = a.F;
    const int tiles_x = (a.img_w + 7) >> 3;
    const long long tile = (long long)blockIdx.x * 4 + wid;
    const int ty = (int)(tile / tiles_x), tx = (int)(tile - (long long)ty * tiles_x);
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    if (ty * 8 >= a.img_h) return;
    const bool have = px < a.img_w && py < a.img_h;
    const long long r = have ? (long long)py * a.img_w + px : 0;
    const int N = a.n_samples;
    Ray ray;
    if (have) ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
    else { ray.ox = ray.oy = ray.oz = 1e30f; ray.dx = ray.dy = ray.dz = 0.f; ray.tmin = 0.f; ray.last = 0.f; }
    int lo = N, hi = -1;
    if (have) ray_interval<false>(F, ray, N, lo, hi);
    const int wlo = wave_min_i(lo), whi = wave_max_i(hi);

    float T = 1.f, acc = 0.f, dep = 0.f;
    int first = -1, last = -1;
    unsigned napp = 0;

    for (int i = wlo; i <= whi; ++i) {
        float xn = 0.f, yn = 0.f, zn = 0.f, z = 0.f;
        bool ok = false;
        if (have && i >= lo && i <= hi) {
            z = sample_z<false>(F, ray, i, 0.f);
            ok = sample_point<false>(F, ray, z, xn, yn, zn);
        }
        if (!__any(ok)) continue;
        const Axes3 A = sample_axes(F.den, xn, yn, zn);
        // per-axis tap ranges over the wave (the three rectangles and line segments are products of these)
        const int big = 1 << 20;
        const int mn0 = wave_min_i(ok ? A.a[0].i0 : big), mx0 = wave_max_i(ok ? A.a[0].i1 : -1);
        const int mn1 = wave_min_i(ok ? A.a[1].i0 : big), mx1 = wave_max_i(ok ? A.a[1].i1 : -1);
        const int mn2 = wave_min_i(ok ? A.a[2].i0 : big), mx2 = wave_max_i(ok ? A.a[2].i1 : -1);
        const int amn[3] = {mn0, mn1, mn2};
        const int aw[3] = {mx0 - mn0 + 1, mx1 - mn1 + 1, mx2 - mn2 + 1};
        const bool fits = aw[0] * aw[1] <= kRectTexels && aw[0] * aw[2] <= kRectTexels && aw[1] * aw[2] <= kRectTexels &&
                          aw[0] <= kLineRows && aw[1] <= kLineRows && aw[2] <= kLineRows;
        float part = 0.f;
        if (fits) {
            // LDS-DMA staging (global_load_lds_dwordx4: destination = wave-uniform LDS base + lane * 16 B, no VGPRs, no
            // ds_write pass): all nine transfers of the step are in flight together — a load -> ds_write loop chained ~9 L2
            // round trips per step, and register staging spilled (the compute phase needs the registers)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int m0 = mat0(k), m1 = mat1(k), vv = vecm(k);
                const float4* __restrict__ P = reinterpret_cast<const float4*>(F.den.plane[k]);
                const int rowq = aw[m0] * 4, total = rowq * aw[m1], W = F.den.W[k];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int it = lane + 64 * u;
                    if (it < total) {
                        const int row = it / rowq, c = it - row * rowq;
                        glds16(P + ((size_t)(amn[m1] + row) * W + amn[m0]) * 4 + c, stP[k] + 64 * u);
                    }
                }
                const float4* __restrict__ Ln = reinterpret_cast<const float4*>(F.den.line[k]);
                if (lane < aw[vv] * 4) glds16(Ln + (size_t)amn[vv] * 4 + lane, stL[k]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_fence_w();
            if (ok) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int m0 = mat0(k), m1 = mat1(k), vv = vecm(k);
                    const Axis& ax = A.a[m0];
                    const Axis& ay = A.a[m1];
                    const Axis& al = A.a[vv];
                    const int rx0 = ax.i0 - amn[m0], rx1 = ax.i1 - amn[m0];
                    const int ry0 = (ay.i0 - amn[m1]) * aw[m0], ry1 = (ay.i1 - amn[m1]) * aw[m0];
                    part = pair_dot_lds(stP[k], stL[k], (ry0 + rx0) * 4, (ry0 + rx1) * 4, (ry1 + rx0) * 4, (ry1 + rx1) * 4,
                                        (al.i0 - amn[vv]) * 4, (al.i1 - amn[vv]) * 4, ay.w0 * ax.w0, ay.w0 * ax.w1, ay.w1 * ax.w0,
                                        ay.w1 * ax.w1, al.w0, al.w1, part);
                }
            }
            lds_fence_w();
        } else if (ok) {
            part = pair_dot_global<0>(F.den, A, part);
            part = pair_dot_global<1>(F.den, A, part);
            part = pair_dot_global<2>(F.den, A, part);
        }
        if (ok) {
            const float sg = feature2density(F, part);
            const float dist = i < N - 1 ? sample_z<false>(F, ray, i + 1, 0.f) - z : 0.f;     // :448
            const float alpha = 1.f - expf((-sg) * (dist * F.dscale));                      // raw2alpha :19-26
            const float w = alpha * T;
            T = T * ((1.f - alpha) + 1e-10f);
            a.wbuf[r * N + i] = w;
            acc += w;
            dep = fmaf(w, z, dep);
            if (w > F.thres) {
                if (a.scratch && napp < (unsigned)a.cap) a.scratch[(size_t)r * a.cap + napp] = make_float4(xn, yn, zn, w);
                ++napp;
            }
            if (first < 0) first = i;
            last = i;
        }
    }
    const int Lw = last >= first && first >= 0 ? last - first + 1 : 0;
    if (have) {
        a.acc[r] = acc;
        a.depth[r] = dep + (1.f - acc) * ray.last;                                         // :504-505
    }
    if (!a.scratch) {
        if (have) a.ray_app[r] = make_int4(0, (int)napp, Lw, Lw > 0 ? (first | (Lw << 11)) : 0);
        return;
    }
    // ---- in-kernel compaction: one list reservation per wave -------------------------------------------------------------
    const bool over = have && napp > (unsigned)a.cap;
    const unsigned n = (have && !over) ? napp : 0u;
    unsigned incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned total = __shfl(incl, 63);
    const unsigned list = blockIdx.x & 7u;   // the block's XCD: the reservation atomic stays in that XCD's L2
    unsigned base = 0;
    if (lane == 0 && total) base = atomicAdd(&a.counters[list * kCounterStride], total);
    base = __shfl(base, 0);
    // a region that does not fit its sub-list (small or ragged frames put many tiles on one list): the wave's rays take the
    // per-ray route below, which tries every sub-list; the inflated counter is clamped to list_cap by its readers
    const bool fits = base + total <= a.list_cap;
    const unsigned slot0 = list * a.list_cap + base + (incl - n);
    if (have) {
        if (over || !fits) {
            const unsigned k = atomicAdd(a.ovf_count, 1u);
            a.ovf_list[k] = (int)r;
            a.ray_app[r] = make_int4(0, (int)napp, Lw, Lw > 0 ? (first | (Lw << 11)) : 0);   // k_compact_list finishes this ray
        } else {
            a.ray_app[r] = make_int4((int)slot0, fits ? (int)n : 0, Lw, Lw > 0 ? (first | (Lw << 11)) : 0);
            if (fits)
                for (unsigned k = 0; k < n; ++k) {
                    a.app_pos[slot0 + k] = a.scratch[(size_t)r * a.cap + k];
                    a.app_ray[slot0 + k] = (int)r;
                }
        }
    }
}

// Build the appearance list from the weights written by k_march_tiles: one wave per ray.
struct CompactArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples;
    float* wbuf; int zero_fill;   // zero_fill: wbuf is the caller's weights tensor: write 0 outside the valid window
    float* z_vals;                // optional [n_rays, n_samples]
    int4* ray_app; float4* app_pos; int* app_ray; unsigned* counters; unsigned list_cap;
    unsigned long long* stats; unsigned nblocks;
};
__device__ __forceinline__ void compact_ray(const CompactArgs& a, long long r, unsigned list, int lane) {
    const FieldDev& F = a.F;
    const int N = a.n_samples;
    int4 ra = a.ray_app[r];
    const int first = ra.w & 2047, Lw = ra.w >> 11;
    unsigned napp = (unsigned)ra.y;
    unsigned slot0 = 0;
    if (lane == 0) {
        bool fits = napp == 0;
        for (unsigned att = 0; att < (unsigned)kLists && !fits; ++att) {     // first sub-list with room (failed tries leave the
            const unsigned l = (list + att) & (unsigned)(kLists - 1);          // counter above list_cap: readers clamp it)
            if (a.counters[l * kCounterStride] + napp > a.list_cap) continue;
            const unsigned s0 = atomicAdd(&a.counters[l * kCounterStride], napp);
            if (s0 + napp <= a.list_cap) { slot0 = l * a.list_cap + s0; fits = true; }
        }
        a.ray_app[r] = make_int4((int)slot0, fits ? (int)napp : 0, ra.z, ra.w);
        if (!fits && a.stats) a.stats[T2N_STAT_OVERFLOW] = 1ull;
        if (!fits) napp = 0;
    }
    slot0 = __shfl(slot0, 0);
    napp = __shfl(napp, 0);
    if (a.zero_fill) {
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            if (i < N && (i < first || i >= first + Lw)) a.wbuf[r * N + i] = 0.f;
        }
    }
    if (napp || a.z_vals) {
        const Ray ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
        if (a.z_vals)
            for (int base = 0; base < N; base += 64)
                if (base + lane < N) a.z_vals[r * N + base + lane] = sample_z<false>(F, ray, base + lane, 0.f);
        unsigned run = 0;
        for (int base = 0; base < Lw; base += 64) {
            const int j = base + lane;
            const float w = j < Lw ? a.wbuf[r * N + first + j] : 0.f;
            const bool m = (j < Lw) & (w > F.thres);
            const unsigned long long bal = __ballot(m);
            if (m) {
                const unsigned pre = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                const float z = sample_z<false>(F, ray, first + j, 0.f);
                float xn, yn, zn;
                sample_point<false>(F, ray, z, xn, yn, zn);
                const unsigned s = slot0 + run + pre;
                a.app_pos[s] = make_float4(xn, yn, zn, w);
                a.app_ray[s] = (int)r;
            }
            run += (unsigned)__popcll(bal);
        }
    }
}
__global__ __launch_bounds__(256) void k_compact(const CompactArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned list = blockIdx.x & 7u;
    const long long r = (long long)xcd_tile(blockIdx.x, a.nblocks) * 4 + wid;
    if (r >= a.n_rays) return;
    compact_ray(a, r, list, lane);
}
// the rays the tile marcher could not stage (more than cap appearance samples): a small persistent grid walks the list
__global__ __launch_bounds__(256) void k_compact_list(const CompactArgs a, const unsigned* ovf_count, const int* ovf_list) {
    const int lane = threadIdx.x & 63;
    const unsigned n = *ovf_count;
    for (unsigned k = blockIdx.x * 4u + (threadIdx.x >> 6); k < n; k += gridDim.x * 4u) compact_ray(a, ovf_list[k], blockIdx.x & 7u, lane);
}

int launch_march_tiles(t2n_field* f, const RenderLaunch& L, int img_w, int img_h, float* wbuf, bool wbuf_is_output, float4* scratch,
                       hipStream_t s) {
    // in-kernel compaction when nothing but rgb / depth is wanted (the eval default): no weights zero-fill, no z_vals
    const bool inline_compact = scratch && !wbuf_is_output && !L.z_vals;
    const int cap = L.n_samples / 4 > 0 ? L.n_samples / 4 : 1;
    unsigned* ovf_count = (unsigned*)((char*)scratch + (size_t)L.n_rays * cap * 16);
    int* ovf_list = (int*)((char*)ovf_count + 256);
    TileArgs a;
    a.F = f->dev;
    a.rays = L.rays; a.n_rays = L.n_rays; a.ray_stride = L.ray_stride; a.n_samples = L.n_samples; a.img_w = img_w; a.img_h = img_h;
    a.wbuf = wbuf; a.acc = L.acc; a.depth = L.depth; a.ray_app = L.ray_app;
    a.scratch = inline_compact ? scratch : nullptr; a.cap = cap; a.ovf_count = ovf_count; a.ovf_list = ovf_list;
    a.app_pos = L.app_pos; a.app_ray = L.app_ray; a.counters = L.counters; a.list_cap = L.list_cap; a.stats = (unsigned long long*)L.stats;
    if (inline_compact) T2N_HIP(hipMemsetAsync(ovf_count, 0, 4, s));
    const long long tiles = (long long)((img_w + 7) / 8) * ((img_h + 7) / 8);
    timing_begin(f, T2N_K_MARCH, s);
    hipLaunchKernelGGL(k_march_tiles, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    CompactArgs c;
    c.F = f->dev;
    c.rays = L.rays; c.n_rays = L.n_rays; c.ray_stride = L.ray_stride; c.n_samples = L.n_samples;
    c.wbuf = wbuf; c.zero_fill = wbuf_is_output ? 1 : 0; c.z_vals = L.z_vals;
    c.ray_app = L.ray_app; c.app_pos = L.app_pos; c.app_ray = L.app_ray; c.counters = L.counters; c.list_cap = L.list_cap;
    c.stats = (unsigned long long*)L.stats; c.nblocks = (unsigned)((L.n_rays + 3) / 4);
    if (inline_compact) hipLaunchKernelGGL(k_compact_list, dim3(64), dim3(256), 0, s, c, (const unsigned*)ovf_count, (const int*)ovf_list);
    else hipLaunchKernelGGL(k_compact, dim3(c.nblocks), dim3(256), 0, s, c);
    timing_end(f, T2N_K_MARCH, s);
    T2N_HIP(hipGetLastError());
    return launch_ray_stats(L, s);
}

}  // namespace t2n
