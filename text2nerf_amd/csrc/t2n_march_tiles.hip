// Tile marcher (eval frames in image order): 8x8-pixel tiles, ONE LANE PER RAY, all 64 rays of a tile advance through the
// same sample index together. At a given depth the 64 rays lie within a few texels of each other (pixel pitch t/f is far
// below the texel size), so per factor pair the union of their bilinear footprints is a small rectangle of R texels and a
// short segment of S line rows. The density feature of a sample is
//     sum_k sum_c (sum_t w_t P_k[t][c]) * (sum_l w_l L_k[l][c])  =  sum_k sum_t sum_l w_t w_l D_k[t][l],
//     D_k[t][l] = sum_c P_k[t][c] * L_k[l][c]      (t over the rectangle, l over the segment, c over the 16 components)
// so the wave computes the small table D_k = P_k L_k^T ONCE per step on the matrix cores (v_mfma_f32_16x16x4_f32, exact
// f32: one 16-B load per lane supplies the A operand of four k-steps, lane l holding texel l&15, components 4(l>>4)..+3),
// parks it in LDS (<= 64 x 16 floats per plane) and every lane reads just its 4 x 2 table entries per plane — 96 B per sample
// instead of the 1152 B of factor data the per-lane interpolation reads. (The staged-factor variant of this kernel was LDS
// bandwidth-bound: 72 ds_read_b128 per wave step = 576 clk of the CU's 128 B/clk LDS pipe x 12 resident waves.)
// Against k_march's pass B (16 samples x 4 lanes per step, 18 scattered 64-B gathers per sample; PMC: texture addresser 71 %
// busy, VALU ~70 %) this removes ~15x of the addresser work and the 4x-redundant per-sample coordinate math. The
// transmittance is a per-lane running product in sample order — exactly the reference's cumprod order
// (models/tensorBase.py:23) — so no wave scan is needed; acc/depth accumulate in registers.
//
// Outputs per ray: acc, depth, (slot0, n_app, n_valid, first | Lw << 11) and the ray's contiguous, sample-ordered slice of
// the appearance list (in-kernel compaction, see TileArgs); with weights requested (DENSE) also the weights of the wave's sampled
// window (k_dense_fill writes the z_vals rows and the zeros of the weights rows beforehand).
// Steps whose rectangles do not fit the table (incoherent rays, huge field of view) fall back to direct gathers.
// Replaces the same reference lines as k_march (see t2n_march.hip).
#include "t2n_device.h"

namespace t2n {

// Table geometry: a plane rectangle of up to 2^LOG2W x 2^LOG2W texels lives in fixed slots (slot = row << LOG2W | col, so no
// division by the rectangle width), the line segment in up to 2^LOG2W rows. LOG2W = 2 (16 slots = ONE 16x16x4 MFMA row
// block per plane) covers nearly every step of a pinhole frame; LOG2W = 3 (64 slots, four row blocks) takes the rest.
constexpr int kMaxLog2W = 2;
constexpr int kDStrideMax = (1 << (2 * kMaxLog2W)) + 4;   // floats per line row of the transposed table D^T[line row][slot]
constexpr int kStageFloats = 3 * 16 * kDStrideMax;       // per wave
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct TileArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples; int img_w, img_h;
    float* acc; float* depth; int4* ray_app;
    // in-kernel compaction: every lane stages its ray's appearance entries (<= cap) in `scratch`; at the end the wave reserves
    // ONE contiguous region of the appearance list for its 64 rays and copies the entries there, ray by ray in sample order.
    // A ray whose slice is full writes its further weights to its row of `wbuf` (the caller's weights tensor when one was
    // requested, else a scratch matrix) and goes to `ovf_list`; k_compact_list finishes those rays.
    float* wbuf;        // [n_rays, n_samples] spill rows
    float4* scratch; int cap; unsigned* ovf_count; int* ovf_list;
    float4* app_pos; int* app_ray; unsigned* counters; unsigned list_cap; unsigned long long* stats;
    // DENSE: the caller's [n_rays, n_samples] weights tensor. k_dense_fill has written zeros (and the z_vals rows) before this kernel;
    // the marcher writes the weights of the 16-step blocks its wave's window touches, through a per-wave LDS transpose: 16 steps x
    // 64 rays, then 64-B row segments per store
    float* dense_w;
#ifdef MT_PROF
    unsigned long long* prof;   // [waves][8] per-region cycle sums (instrumented build, tools/r5_march_accounting.sh)
#endif
};

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

__device__ __forceinline__ unsigned wave_or_u(unsigned v) {
    v |= (unsigned)dpp_shr<0x111>((int)v, 0);
    v |= (unsigned)dpp_shr<0x112>((int)v, 0);
    v |= (unsigned)dpp_shr<0x114>((int)v, 0);
    v |= (unsigned)dpp_shr<0x118>((int)v, 0);
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 15), b = (unsigned)__builtin_amdgcn_readlane((int)v, 31);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 47), d = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    return (a | b) | (c | d);
}

// The dot-product tables of one wave step (see the file header). amn[a] = lowest tap index of axis a over the wave; every axis
// spans at most 2^LOG2W taps. table_build computes D_k^T[line row][slot] for the three pairs into the wave's LDS area;
// table_read returns a lane's density feature from its 4 x 2 entries per pair. Several samples per lane may share one build.
template <int LOG2W>
__device__ __forceinline__ void table_build(const FactorSet& S, const int (&amn)[3], float* __restrict__ stD, int l15, int lq) {
    constexpr int WS = 1 << LOG2W, SL = WS * WS, NB = SL / 16, ST = SL + 4;
    const int gs[3] = {S.W[0], S.H[0], S.H[1]};
    // phase 1: every operand load of the step in flight together; slots beyond the rectangle read clamped (valid, unused) texels
    float4 av[3][NB], bv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int m0 = mat0(k), m1 = mat1(k), vv = vecm(k);
        const float4* __restrict__ P = reinterpret_cast<const float4*>(S.plane[k]);
        const float4* __restrict__ Ln = reinterpret_cast<const float4*>(S.line[k]);
        const unsigned jr = (unsigned)min(amn[vv] + (l15 & (WS - 1)), gs[vv] - 1);
        bv[k] = Ln[jr * 4u + (unsigned)lq];
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            const int t = mb * 16 + l15;
            const unsigned yy = (unsigned)min(amn[m1] + (t >> LOG2W), gs[m1] - 1), xx = (unsigned)min(amn[m0] + (t & (WS - 1)), gs[m0] - 1);
            av[k][mb] = P[(__umul24(yy, (unsigned)gs[m0]) + xx) * 4u + (unsigned)lq];   // (24-bit multiply: full rate)
        }
    }
    // phase 2: D_k^T[line row][slot] = sum_c L_k[row][c] P_k[slot][c]; the lane holds slots 4 lq..+3 of line row l15
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float* __restrict__ Dk = stD + k * 16 * ST + l15 * ST + lq * 4;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][mb].x, bv[k].x, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][mb].y, bv[k].y, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][mb].z, bv[k].z, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k][mb].w, bv[k].w, d, 0, 0, 0);
            *reinterpret_cast<float4*>(Dk + mb * 16) = make_float4(d[0], d[1], d[2], d[3]);
        }
    }
}
// A lane's density feature from the tables: per pair ONE table offset (low line row, low plane cell); the other seven entries sit at
// fixed distances from it, because a high tap is the low tap + 1 wherever its weight is not zero (axis_taps: the clamp at the
// grid's last texel comes with weight 0). There the entry one step further is read instead of the clamped one - a finite table
// value (every slot of the 16 line rows is written by each build, the four pad floats of a row are zeroed once per wave) times a
// zero weight: the sum is the same. 12 address computations and 24 single reads per sample became 3 and 12 paired reads.
template <int LOG2W>
__device__ __forceinline__ float table_read(const Axes3& A, const int (&amn)[3], const float* __restrict__ stD) {
    constexpr int WS = 1 << LOG2W, SL = WS * WS, ST = SL + 4;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int m0 = mat0(k), m1 = mat1(k), vv = vecm(k);
        const Axis& ax = A.a[m0];
        const Axis& ay = A.a[m1];
        const Axis& al = A.a[vv];
        // (24-bit multiply: full rate; v_mul_lo_u32 is quarter rate)
        const int off = k * 16 * ST + (int)__umul24((unsigned)(al.i0 - amn[vv]), (unsigned)ST) + ((ay.i0 - amn[m1]) << LOG2W) + (ax.i0 - amn[m0]);
        const float* __restrict__ D = stD + off;
        const float wnw = ay.w0 * ax.w0, wne = ay.w0 * ax.w1, wsw = ay.w1 * ax.w0, wse = ay.w1 * ax.w1;
        float v0 = D[0] * wnw, v1 = D[ST] * wnw;
        v0 = fmaf(D[1], wne, v0); v1 = fmaf(D[ST + 1], wne, v1);
        v0 = fmaf(D[WS], wsw, v0); v1 = fmaf(D[ST + WS], wsw, v1);
        v0 = fmaf(D[WS + 1], wse, v0); v1 = fmaf(D[ST + WS + 1], wse, v1);
        part = fmaf(v0, al.w0, part);
        part = fmaf(v1, al.w1, part);
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
#pragma unroll 1   // the rare direct-gather step: keep its register footprint below the table path's (occupancy)
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

// A reservation that did not fit leaves its sub-list counter raised over entries nobody writes (the readers clamp the counter to
// list_cap, so the part of the region inside the list IS walked by the shading kernels): those entries get a defined content — the
// volume centre, weight 0, ray 0 — whatever the scratch held before (the SH head reads the entry's ray for its view direction).
__device__ __forceinline__ void void_entries(float4* __restrict__ app_pos, int* __restrict__ app_ray, unsigned list, unsigned list_cap,
                                             unsigned lo, unsigned hi, int lane) {
    if (hi > list_cap) hi = list_cap;
    for (unsigned e = lo + (unsigned)lane; e < hi; e += 64u) {
        app_pos[(size_t)list * list_cap + e] = make_float4(0.f, 0.f, 0.f, 0.f);
        app_ray[(size_t)list * list_cap + e] = 0;
    }
}

constexpr int kDenseLd = 20;                          // floats per ray row of the transpose tile (16 steps + pad, 16-B aligned)
constexpr int kDenseFloats = 64 * kDenseLd;           // per wave: the weights tile
struct __attribute__((aligned(4))) F4U { float x, y, z, w; };   // row segments of an [n_rays, N] tensor are only 4-B aligned for odd N

// Waves per SIMD the register allocation targets. The step loop is a load -> MFMA -> LDS -> read chain that only other waves
// hide: 3 waves 0.95 ms, 4 waves (109 VGPRs) 0.78, 5 waves (96 VGPRs; ten loop-invariant dwords go to scratch and come back
// only where an appearance entry is stored) 0.74, 6 waves (80 VGPRs, 104 B of scratch in the loop) 0.86. The weights-writing
// variant (121 VGPRs) spills inside its loop at 5 (1.46 ms against 1.37 for fill + march) and stays at 4.
#ifndef T2N_MT_WAVES
#define T2N_MT_WAVES 5
#endif
#ifndef T2N_MTD_WAVES
#define T2N_MTD_WAVES 4
#endif
// -DMT_PROF (timing-only build): s_memtime at the region boundaries of a step pair, summed per wave. With five waves per SIMD a
// region's cycles include what the SIMD gave to the other four meanwhile: the sums say where a wave's time goes, not what the
// instructions cost alone.
#ifdef MT_PROF
#define MT_T(var) do { const unsigned long long tn = __builtin_readcyclecounter(); var += tn - p_t; p_t = tn; } while (0)
#else
#define MT_T(var) do {} while (0)
#endif
// ALPHA: the field carries an AlphaGridMask; RELU: fea2denseAct = relu. Compile-time, like the absent NDC depth table (the host never
// sends NDC renders here): the step loop of the driver's configuration (no mask, softplus) carries neither the mask's eight-tap
// gather with its 64-bit address arithmetic nor a run-time test per sample and option.
template <bool DENSE, bool ALPHA = false, bool RELU = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DENSE ? T2N_MTD_WAVES : T2N_MT_WAVES, DENSE ? T2N_MTD_WAVES : T2N_MT_WAVES))) void k_march_tiles(const TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    float* __restrict__ stD = smem + (size_t)wid * kStageFloats;
    for (int i = lane; i < kStageFloats; i += 64) stD[i] = 0.f;   // the rows' pad floats stay zero (table_read may touch them with weight 0)
    lds_fence_w();
    float* __restrict__ wt = smem + 4 * kStageFloats + (size_t)wid * kDenseFloats;   // DENSE only
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
    int nev = 0;         // evaluated samples (the window [first, last] has gaps under an alpha mask)
    int ovf_from = 0;    // first sample whose weight went to wbuf instead of the (full) staging slice

    // row segments of the transpose flush: lane -> (ray rho = lane / 4 + 16 j, columns 4 (lane & 3) .. + 3), j = 0..3
    long long frow[4];
    if constexpr (DENSE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rho = (lane >> 2) + 16 * j;
            const int fx = tx * 8 + (rho & 7), fy = ty * 8 + (rho >> 3);
            frow[j] = (fx < a.img_w && fy < a.img_h) ? ((long long)fy * a.img_w + fx) * N : -1;
        }
    }
    // kSteps consecutive samples per ray share ONE table build: the tap ranges of neighbouring steps overlap (the union of two
    // steps spans 3-4 taps per axis on every pair of a pinhole frame), so the reduction, the six operand loads, the twelve
    // MFMAs and the load -> MFMA -> LDS latency chain are paid once per pair.
    constexpr int kSteps = 2;
    // DENSE: whole 16-step blocks of the weights rows (a wave without samples runs no step: wlo = N, whi = -1)
    const int i_begin = DENSE ? (wlo & ~15) : wlo, i_end = DENSE ? min(N - 1, whi | 15) : whi;
    // early termination (t2n_field_set_early_termination; never with weights rows): a ray below term_eps evaluates nothing further,
    // the wave leaves the loop when none of its rays has anything left to evaluate
    const float eps = DENSE ? 0.f : F.term_eps;
    bool dead = false;   // this ray is below term_eps (as of the last look)
#ifdef MT_PROF
    unsigned long long p_coord = 0, p_axes = 0, p_build = 0, p_read = 0, p_act = 0, p_pairs = 0;
    const unsigned long long p_start = __builtin_readcyclecounter();
    unsigned long long p_t = p_start;
#endif
    for (int i = i_begin; i <= i_end; i += kSteps) {
        // looked at every eighth step pair, and the rays' own gate is refreshed only there: a test of the CURRENT transmittance in front of
        // every step pair chains each iteration to the one before (the step pair's gathers cannot start before the previous pair's
        // compositing is done) — 3 % of a frame on a scene where nothing terminates
        if (eps > 0.f && (((i - i_begin) & 15) == 0)) {
            dead = T < eps;
            if (!__any(have && i <= hi && !dead)) break;
        }
        float xn[kSteps], yn[kSteps], zn[kSteps], z[kSteps], w_out[kSteps];
        bool ok[kSteps];
        Axes3 A[kSteps];
        bool any_ok = false;
#pragma unroll
        for (int q = 0; q < kSteps; ++q) {
            const int idx = i + q;
            xn[q] = yn[q] = zn[q] = z[q] = w_out[q] = 0.f;
            ok[q] = false;
            if (have && idx >= lo && idx <= hi && idx <= i_end && !dead) {
                z[q] = sample_z<false, true>(F, ray, idx, 0.f);
                ok[q] = sample_point<false>(F, ray, z[q], xn[q], yn[q], zn[q]);
                if constexpr (ALPHA) { if (ok[q]) ok[q] = alpha_pass(F, ray, z[q]); }      // models/tensorBase.py:451-456
            }
            any_ok |= ok[q];
        }
        const unsigned long long okm = __ballot(any_ok);
        MT_T(p_coord);
        // (a lane whose staging slice is full keeps its spill row gap-free: masked-out samples inside its window get zeros)
        if (!DENSE && !okm && !__any(have && napp >= (unsigned)a.cap)) continue;
        float part[kSteps];
#pragma unroll
        for (int q = 0; q < kSteps; ++q) part[q] = 0.f;
        if (okm) {
#pragma unroll
            for (int q = 0; q < kSteps; ++q) A[q] = sample_axes_inbox(F.den, xn[q], yn[q], zn[q]);   // used by samples inside the box only (ok[q])
            // the low-tap indices present in the wave, per axis, as ONE or-reduced bit set: 10 bits per axis around the first
            // live lane's index (bit 5); an index outside [-5, +4] of it raises bit 30 (the steps then gather directly)
            int sel0 = 0, sel1 = 0, sel2 = 0;
#pragma unroll
            for (int q = kSteps - 1; q >= 0; --q)
                if (ok[q]) { sel0 = A[q].a[0].i0; sel1 = A[q].a[1].i0; sel2 = A[q].a[2].i0; }
            const int lead = (int)__builtin_ctzll(okm);
            const int ref0 = __builtin_amdgcn_readlane(sel0, lead), ref1 = __builtin_amdgcn_readlane(sel1, lead),
                      ref2 = __builtin_amdgcn_readlane(sel2, lead);
            unsigned bits = 0u;
#pragma unroll
            for (int q = 0; q < kSteps; ++q)
                if (ok[q]) {
                    const unsigned d0 = (unsigned)(A[q].a[0].i0 - ref0 + 5), d1 = (unsigned)(A[q].a[1].i0 - ref1 + 5),
                                   d2 = (unsigned)(A[q].a[2].i0 - ref2 + 5);
                    bits |= (d0 < 10u ? 1u << d0 : 0x40000000u) | (d1 < 10u ? 1u << (10u + d1) : 0x40000000u) |
                            (d2 < 10u ? 1u << (20u + d2) : 0x40000000u);
                }
            bits = wave_or_u(bits);
            const unsigned f0 = bits & 1023u, f1 = (bits >> 10) & 1023u, f2 = (bits >> 20) & 1023u;
            const int amn[3] = {ref0 - 5 + __builtin_ctz(f0), ref1 - 5 + __builtin_ctz(f1), ref2 - 5 + __builtin_ctz(f2)};
            // taps per axis: the high tap is at most one above the highest low tap (and inside the grid)
            const int span = max(max(32 - __builtin_clz(f0) - __builtin_ctz(f0), 32 - __builtin_clz(f1) - __builtin_ctz(f1)),
                                 32 - __builtin_clz(f2) - __builtin_ctz(f2)) + 1;
            const bool ranged = !(bits & 0x40000000u);
            MT_T(p_axes);
            if (ranged && span <= 4) {
                table_build<2>(F.den, amn, stD, l15, lq);
                lds_fence_w();
                MT_T(p_build);
#pragma unroll
                for (int q = 0; q < kSteps; ++q)
                    if (ok[q]) part[q] = table_read<2>(A[q], amn, stD);
                lds_fence_w();
                MT_T(p_read);
            } else {
#pragma unroll   // (a rolled loop would index A[] dynamically and push the per-sample arrays to scratch)
                for (int q = 0; q < kSteps; ++q)
                    if (ok[q]) {
                        float pq = pair_dot_global<0>(F.den, A[q], 0.f);
                        pq = pair_dot_global<1>(F.den, A[q], pq);
                        part[q] = pair_dot_global<2>(F.den, A[q], pq);
                    }
            }
        }
#pragma unroll
        for (int q = 0; q < kSteps; ++q) {
            const int idx = i + q;
            const bool spill = have && idx <= i_end && napp >= (unsigned)a.cap && !DENSE;
            if (ok[q]) {
                const float sg = feature2density<RELU ? T2N_ACT_RELU : T2N_ACT_SOFTPLUS>(F, part[q]);
                const float dist = idx < N - 1 ? sample_z<false, true>(F, ray, idx + 1, 0.f) - z[q] : 0.f;     // :448
                const float alpha = 1.f - exp_finite((-sg) * (dist * F.dscale));                      // raw2alpha :19-26 (argument <= 0)
                const float w = alpha * T;
                T = T * ((1.f - alpha) + 1e-10f);
                acc += w;
                dep = fmaf(w, z[q], dep);
                // in-kernel compaction: (x, y, z, w) of the samples above the threshold go to the ray's staging slice
                if (w > F.thres) {
                    if (napp < (unsigned)a.cap) a.scratch[(size_t)r * a.cap + napp] = make_float4(xn[q], yn[q], zn[q], w);
                    ++napp;
                    if (napp == (unsigned)a.cap) ovf_from = idx + 1;
                }
                if (first < 0) first = idx;
                last = idx;
                ++nev;
                w_out[q] = w;
            }
            if (spill) a.wbuf[r * N + idx] = w_out[q];
            if constexpr (DENSE) {
                if (idx < N) {
                    const int c16 = idx & 15;
                    wt[lane * kDenseLd + c16] = w_out[q];
                }
                if ((idx < N && (idx & 15) == 15) || idx == N - 1) {
                    lds_fence_w();
                    const int col = (idx & ~15) + 4 * (lane & 3);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (frow[j] < 0 || col >= N) continue;
                        const int rho = (lane >> 2) + 16 * j;
                        const float4 vw = *reinterpret_cast<const float4*>(wt + rho * kDenseLd + 4 * (lane & 3));
                        if (col + 3 < N) {
                            *reinterpret_cast<F4U*>(a.dense_w + frow[j] + col) = F4U{vw.x, vw.y, vw.z, vw.w};
                        } else {
                            const float ew[4] = {vw.x, vw.y, vw.z, vw.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (col + e < N) a.dense_w[frow[j] + col + e] = ew[e];
                        }
                    }
                    lds_fence_w();
                }
            }
        }
#ifdef MT_PROF
        MT_T(p_act); ++p_pairs;
#endif
    }
#ifdef MT_PROF
    if (lane == 0 && a.prof) {
        unsigned long long* o = a.prof + (size_t)(blockIdx.x * 4 + wid) * 8;
        o[0] = p_coord; o[1] = p_axes; o[2] = p_build; o[3] = p_read; o[4] = p_act; o[5] = p_pairs; o[6] = __builtin_readcyclecounter() - p_start;
    }
#endif
    const int Lw = last >= first && first >= 0 ? last - first + 1 : 0;
    if (have) {
        a.acc[r] = acc;
        a.depth[r] = dep + (1.f - acc) * ray.last;                                         // :504-505
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
    if (!fits && total) void_entries(a.app_pos, a.app_ray, list, a.list_cap, base, base + total, lane);
    const unsigned slot0 = list * a.list_cap + base + (incl - n);
    if (have) {
        if (over || !fits) {
            const unsigned k = atomicAdd(a.ovf_count, 1u);
            a.ovf_list[k] = (int)r;
            a.ray_app[r] = make_int4(ovf_from, (int)napp, nev, Lw > 0 ? (first | (Lw << 11)) : 0);   // k_compact_list finishes this ray
        } else {
            a.ray_app[r] = make_int4((int)slot0, fits ? (int)n : 0, nev, Lw > 0 ? (first | (Lw << 11)) : 0);
            if (fits) {
                // four entries in flight per trip (a wave's tail lasts as long as its longest slice; one dependent round trip
                // per entry otherwise)
                const float4* __restrict__ src = a.scratch + (size_t)r * a.cap;
                for (unsigned k = 0; k < n; k += 4) {
                    const float4 e0 = src[k], e1 = src[k + 1 < n ? k + 1 : n - 1], e2 = src[k + 2 < n ? k + 2 : n - 1],
                                 e3 = src[k + 3 < n ? k + 3 : n - 1];
                    a.app_pos[slot0 + k] = e0; a.app_ray[slot0 + k] = (int)r;
                    if (k + 1 < n) { a.app_pos[slot0 + k + 1] = e1; a.app_ray[slot0 + k + 1] = (int)r; }
                    if (k + 2 < n) { a.app_pos[slot0 + k + 2] = e2; a.app_ray[slot0 + k + 2] = (int)r; }
                    if (k + 3 < n) { a.app_pos[slot0 + k + 3] = e3; a.app_ray[slot0 + k + 3] = (int)r; }
                }
            }
        }
    }
}

// The rays the tile marcher handed over: one wave per ray finishes their appearance slices.
struct CompactArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples;
    const float* wbuf;                // spill rows (see TileArgs)
    int4* ray_app; float4* app_pos; int* app_ray; unsigned* counters; unsigned list_cap;
    unsigned long long* stats;
    const float4* scratch; int cap;   // the marcher's per-ray staging slices
    int finisher;                     // a ray that fits no sub-list is left to k_finish_rays (t2n_shade.hip) instead of losing its samples
};
// lane 0 reserves `napp` slots on the first sub-list with room; returns (slot0, napp or 0 when nothing fits) to all lanes
// A ray that fits nowhere (budgeted lists, t2n_render_forward): with a finisher behind this kernel the ray keeps its hand-over record
// in ray_app — x = -(first spilled sample) - 1 < 0 tells k_composite that the ray owns no list entries, y stays its appearance
// count (k_ray_stats) — its slot of the overflow list is tagged (sign bit) and its entries are added to word kFailEntriesWord (the
// next frame's budget); k_finish_rays shades and composites such rays from their staging slice and spill row, no list needed.
__device__ __forceinline__ void reserve_slots(const CompactArgs& a, long long r, unsigned list, int lane, const int4 ra, unsigned& slot0,
                                              unsigned& napp, int* ovf_slot) {
    slot0 = 0;
    if (lane == 0) {
        bool fits = napp == 0;
        for (unsigned att = 0; att < (unsigned)kLists && !fits; ++att) {     // first sub-list with room (a failed try leaves the
            const unsigned l = (list + att) & (unsigned)(kLists - 1);          // counter above list_cap: readers clamp it)
            if (a.counters[l * kCounterStride] + napp > a.list_cap) continue;
            const unsigned s0 = atomicAdd(&a.counters[l * kCounterStride], napp);
            if (s0 + napp <= a.list_cap) { slot0 = l * a.list_cap + s0; fits = true; }
            else                                                              // lost a race for the list's tail: its part of the
                for (unsigned e = s0; e < a.list_cap; ++e) {                  // list gets defined entries (void_entries), by this lane
                    a.app_pos[(size_t)l * a.list_cap + e] = make_float4(0.f, 0.f, 0.f, 0.f);
                    a.app_ray[(size_t)l * a.list_cap + e] = 0;
                }
        }
        if (fits || !a.finisher) a.ray_app[r] = make_int4((int)slot0, fits ? (int)napp : 0, ra.z, ra.w);
        else {
            a.ray_app[r] = make_int4(-ra.x - 1, (int)napp, ra.z, ra.w);
            *ovf_slot = (int)((unsigned)r | 0x80000000u);
            atomicAdd(&a.counters[kFailEntriesWord], napp);
        }
        if (!fits && !a.finisher && a.stats) a.stats[T2N_STAT_OVERFLOW] = 1ull;
        if (!fits) a.counters[kOverflowWord] = 1u;   // budgeted lists: read back by the host (the next frames' budget, list_retries)
        if (!fits) napp = 0;
    }
    slot0 = __shfl(slot0, 0);
    napp = __shfl(napp, 0);
}
// append the entries of wbuf[r][from, end) above the threshold at slot0 + run.. (sample order)
__device__ __forceinline__ void compact_span(const CompactArgs& a, long long r, const Ray& ray, int from, int end, unsigned slot0, unsigned run,
                                             int lane) {
    const FieldDev& F = a.F;
    const int N = a.n_samples;
    for (int base = from; base < end; base += 64) {
        const int i = base + lane;
        const float w = i < end ? a.wbuf[r * N + i] : 0.f;
        const bool m = (i < end) & (w > F.thres);
        const unsigned long long bal = __ballot(m);
        if (m) {
            const unsigned pre = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            const float z = sample_z<false>(F, ray, i, 0.f);
            float xn, yn, zn;
            sample_point<false>(F, ray, z, xn, yn, zn);
            const unsigned s = slot0 + run + pre;
            a.app_pos[s] = make_float4(xn, yn, zn, w);
            a.app_ray[s] = (int)r;
        }
        run += (unsigned)__popcll(bal);
    }
}
// a ray the tile marcher handed over (staging slice full, or its wave's region did not fit the sub-list): the first
// min(napp, cap) entries sit in the staging slice, the rest in wbuf[r][from..]
__device__ __forceinline__ void compact_ray_staged(const CompactArgs& a, long long r, unsigned list, int lane, int* ovf_slot) {
    const int4 ra = a.ray_app[r];
    const int first = ra.w & 2047, Lw = ra.w >> 11, from = ra.x;
    unsigned napp = (unsigned)ra.y, slot0;
    reserve_slots(a, r, list, lane, ra, slot0, napp, ovf_slot);
    const unsigned staged = napp < (unsigned)a.cap ? napp : (unsigned)a.cap;
    for (unsigned k = (unsigned)lane; k < staged; k += 64u) {
        a.app_pos[slot0 + k] = a.scratch[(size_t)r * a.cap + k];
        a.app_ray[slot0 + k] = (int)r;
    }
    if (napp > staged) {
        const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
        compact_span(a, r, ray, from, first + Lw, slot0, staged, lane);
    }
}
// the rays the tile marcher could not stage (more than cap appearance samples): a small persistent grid walks the list
__global__ __launch_bounds__(256) void k_compact_list(const CompactArgs a, const unsigned* ovf_count, int* ovf_list) {
    const int lane = threadIdx.x & 63;
    const unsigned n = *ovf_count;
    for (unsigned k = blockIdx.x * 4u + (threadIdx.x >> 6); k < n; k += gridDim.x * 4u)
        compact_ray_staged(a, ovf_list[k], blockIdx.x & 7u, lane, ovf_list + k);
}

// z_vals rows (z_i of every sample index, models/tensorBase.py:313-318) and zeroed weights rows of a launch's rays: one wave per
// ray, 256-B coalesced stores (2 x 1.33 GB per C2 frame at the HBM write rate). The tile marcher then writes the weights of the
// sampled windows only.
__global__ __launch_bounds__(256) void k_dense_fill(const FieldDev F, const float* __restrict__ rays, long long n_rays, int ray_stride, int N,
                                                    float* __restrict__ w, float* __restrict__ z) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rays) return;
    Ray ray;
    if (z) ray = load_ray(F, rays + r * ray_stride, ray_stride);
    for (int i = lane; i < N; i += 64) {
        if (z) z[r * N + i] = sample_z<false>(F, ray, i, 0.f);
        if (w) w[r * N + i] = 0.f;
    }
}

int launch_march_tiles(t2n_field* f, const RenderLaunch& L, int img_w, int img_h, float* spill, float4* scratch, hipStream_t s) {
    const int cap = L.n_samples / 4 > 0 ? L.n_samples / 4 : 1;
    unsigned* ovf_count = (unsigned*)((char*)scratch + (size_t)L.n_rays * cap * 16);
    int* ovf_list = (int*)((char*)ovf_count + 256);
    const bool dense = L.weights != nullptr;
    TileArgs a;
    a.F = f->dev;
    a.rays = L.rays; a.n_rays = L.n_rays; a.ray_stride = L.ray_stride; a.n_samples = L.n_samples; a.img_w = img_w; a.img_h = img_h;
    a.acc = L.acc; a.depth = L.depth; a.ray_app = L.ray_app;
    a.wbuf = L.weights ? L.weights : spill;
    a.scratch = scratch; a.cap = cap; a.ovf_count = ovf_count; a.ovf_list = ovf_list;
    a.app_pos = L.app_pos; a.app_ray = L.app_ray; a.counters = L.counters; a.list_cap = L.list_cap; a.stats = (unsigned long long*)L.stats;
    a.dense_w = L.weights;
    a.F.term_eps = (!L.weights && !L.z_vals && !L.sigma_ctx) ? f->term_eps : 0.f;   // (eval only: this marcher has no train form)
#ifdef MT_PROF
    static unsigned long long* prof = nullptr;
    static int prof_calls = 0;
    const long long prof_waves = ((long long)((img_w + 7) / 8) * ((img_h + 7) / 8) + 3) / 4 * 4;
    if (!prof) T2N_HIP(hipMalloc((void**)&prof, (size_t)65536 * 8 * 8));
    T2N_HIP(hipMemsetAsync(prof, 0, (size_t)65536 * 8 * 8, s));
    a.prof = prof_waves <= 65536 ? prof : nullptr;
#endif
    // (*ovf_count was zeroed by the launch's setup kernel, t2n_render_forward)
    const long long tiles = (long long)((img_w + 7) / 8) * ((img_h + 7) / 8);
    const dim3 grid((unsigned)((tiles + 3) / 4));
    timing_begin(f, T2N_K_MARCH, s);
    if (L.weights || L.z_vals)
        hipLaunchKernelGGL(k_dense_fill, dim3((unsigned)((L.n_rays + 3) / 4)), dim3(256), 0, s, f->dev, L.rays, (long long)L.n_rays, L.ray_stride,
                           L.n_samples, L.weights, L.z_vals);
    const size_t lds = (size_t)4 * (kStageFloats + (dense ? kDenseFloats : 0)) * sizeof(float);
    const bool mask = f->dev.alpha != nullptr, relu = f->dev.act == T2N_ACT_RELU;
#define T2N_MT_LAUNCH(D, A, R) hipLaunchKernelGGL((k_march_tiles<D, A, R>), grid, dim3(256), lds, s, a)
    if (dense) { if (mask) { if (relu) T2N_MT_LAUNCH(true, true, true); else T2N_MT_LAUNCH(true, true, false); }
                 else { if (relu) T2N_MT_LAUNCH(true, false, true); else T2N_MT_LAUNCH(true, false, false); } }
    else { if (mask) { if (relu) T2N_MT_LAUNCH(false, true, true); else T2N_MT_LAUNCH(false, true, false); }
           else { if (relu) T2N_MT_LAUNCH(false, false, true); else T2N_MT_LAUNCH(false, false, false); } }
#undef T2N_MT_LAUNCH
    T2N_HIP(hipGetLastError());
#ifdef MT_PROF
    if (a.prof && ++prof_calls == 20) {   // one report per process: per-wave cycle sums, averaged over the waves that ran steps
        static unsigned long long h[65536 * 8];
        T2N_HIP(hipStreamSynchronize(s));
        T2N_HIP(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long n = 0;
        for (long long i = 0; i < prof_waves; ++i) if (h[i * 8 + 5]) { ++n; for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k]; }
        if (n) fprintf(stderr, "[mt prof] %lld waves, per wave: step pairs %.1f, cycles in the step loop %.0f = coordinates + validity %.0f, axis taps + union bit set %.0f, "
                               "table build (loads -> MFMA -> LDS) %.0f, table reads %.0f, activation + compositing + staging %.0f; per step pair %.0f\n",
                       n, sum[5] / n, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / n, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n,
                       (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / (sum[5] > 0 ? sum[5] : 1));
    }
#endif
    CompactArgs c;
    c.F = f->dev;
    c.rays = L.rays; c.n_rays = L.n_rays; c.ray_stride = L.ray_stride; c.n_samples = L.n_samples;
    c.wbuf = a.wbuf;
    c.ray_app = L.ray_app; c.app_pos = L.app_pos; c.app_ray = L.app_ray; c.counters = L.counters; c.list_cap = L.list_cap;
    c.stats = (unsigned long long*)L.stats;
    c.scratch = scratch; c.cap = cap;
    c.finisher = finish_supported(f) ? 1 : 0;
    hipLaunchKernelGGL(k_compact_list, dim3(64), dim3(256), 0, s, c, (const unsigned*)ovf_count, ovf_list);
    timing_end(f, T2N_K_MARCH, s);
    T2N_HIP(hipGetLastError());
    return launch_ray_stats(L, s);
}

}  // namespace t2n

