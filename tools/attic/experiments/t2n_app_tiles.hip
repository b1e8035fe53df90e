// Appearance features of image-ordered frames (the tile marcher's lists): the gather + basis_mat stage as per-tile TABLES on the
// matrix cores, the appearance-side counterpart of the tile marcher's density tables (t2n_march_tiles.hip).
//
// Replaces (reference): models/tensoRF.py:223-239 (compute_appfeature: three plane / line pairs, bilinear x linear interpolation,
// product, basis_mat). Same outputs as k_app_features_p (t2n_shade.hip): fp32 feature rows [tile * 32 + sample][32] for the
// sample-stationary head — 27 features, the entry's compositing weight in column 27.
//
// Why tables. Per appearance sample the direct form reads 18 taps x 192 B (216 16-B lane loads: the feature kernel sits on the L1's
// 64 B/clk and on the texture addresser) and runs 144 x 27 basis MACs. But the 64 rays of an 8x8-pixel tile meet a surface within a
// texel or two of each other, so their taps fall into a small box of the grid, and
//     feat[j] = sum_k sum_c B[j][48 k + c] (sum_t w_t P_k[t][c]) (sum_l w_l L_k[l][c])  =  sum_k sum_t sum_l w_t w_l T_k[t][l][j],
//     T_k[t][l][j] = sum_c B[j][48 k + c] P_k[t][c] L_k[l][c]          (t over a 4 x 4 texel box, l over 4 line rows),
// is ONE small contraction per box and factor pair — [27 x 48] basis slice times the 48 x 64 matrix of plane x line products — which
// the wave computes on v_mfma_f32_32x32x16_f16 (split-f16 products, fp32 accumulate, the same operand packing and range guard as
// the per-sample basis stage) and parks in LDS. A lane then reads its sample's 8 (tap, row) entries per pair and adds them up with
// its interpolation weights: 648 FMAs and 168 16-B LDS reads per sample instead of 216 global gathers, ~1000 VALU operations and
// the per-sample share of the basis MFMAs. The sums are re-associated (table first, interpolation second), like the density
// tables: features agree with the per-sample kernels to a few 1e-7 relative, inside every tolerance of the parity suite.
//
// Mapping (gfx950, wave64): one wave per 8x8-pixel tile, lane = ray (the marcher's own map). Every ray's appearance entries are a
// contiguous, sample-ordered slice of the list (ray_app); each lane walks its slice. A ROUND serves the next entry of every lane
// whose taps fit one box: the box is anchored one texel below the taps of the first lane that still has entries (so that lane
// always fits: progress), lanes within [anchor, anchor + 2] on all three axes take part, the others wait. Rounds per tile on the
// bench frame: ~10 for ~430 entries (walls seen obliquely stagger the rays' depths). Per round and pair: 18 16-B loads per lane
// (the box's texels and rows, channel octets), 8 products + one hi / lo split per (K-step, column block), 18 MFMAs, 8 ds_write_b128;
// then the reads. Everything depends only on the tile's own rays: a frame rendered in 8-row bands or with budgeted lists walks the
// same rounds and produces the same bits.
#include "t2n_device.h"

namespace t2n {
namespace apt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kTJ = 28;                       // floats per table row: 27 features + one pad (16-B aligned rows, banks spread)
constexpr int kTRows = 64;                    // 4 line rows x 16 plane slots
constexpr int kTFloats = kTRows * kTJ;        // 7 KB per wave
constexpr int kBasisVec = 9 * 2 * 64;         // uint4: the nine real basis chunks [chunk][part][lane] of FieldDev::basisH
constexpr float kWUnscale = 1.f / 256.f;      // basisH holds the weights x 2^8 (t2n_shade.hip: kWScale)
constexpr unsigned kUnsafeBasis = 1u;

struct Args {
    FieldDev F;
    const float4* app_pos; const int4* ray_app; const unsigned* counters; unsigned list_cap; int nlists;
    int img_w, img_h;
    float* feat; unsigned feat_rows;
    unsigned* range_flag; const unsigned* split_unsafe;
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x16 mfma16(uint4 a, h8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), b, c, 0, 0, 0);
}

// hi = RTZ_f16(x), lo = RTZ_f16(x - hi) of eight values; amax tracks the largest magnitude handed to the packed convert
__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo, float& amax) {
    u4 uh, ul;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        amax = fmaxf(fmaxf(amax, fabsf(x[2 * e])), fabsf(x[2 * e + 1]));
        const hh2 p = __builtin_amdgcn_cvt_pkrtz(x[2 * e], x[2 * e + 1]);
        const h2 ph = __builtin_bit_cast(h2, p);
        const float r0 = x[2 * e] - (float)ph[0], r1 = x[2 * e + 1] - (float)ph[1];
        uh[e] = __builtin_bit_cast(unsigned, p);
        ul[e] = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
    hi = __builtin_bit_cast(h8, uh);
    lo = __builtin_bit_cast(h8, ul);
}

__device__ __forceinline__ int pick3(const int (&v)[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
__device__ __forceinline__ float pick3f(const float (&v)[3], int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }

// T_K for the box at amn: rows (l * 16 + t) x 27 features into the wave's LDS table. K is a run-time value: the three pairs run as a
// ROLLED loop (unrolled, hipcc hoists the 54 operand loads of a round to its top and spills 368 registers).
__device__ __forceinline__ void table_build(const FactorSet& S, int K, const int (&amn)[3], const int (&gs)[3], const uint4* __restrict__ Wl,
                                            float* __restrict__ T, int lane, float& amax) {
    const int m0 = mat0(K), m1 = mat1(K), vv = vecm(K);
    const int n = lane & 31, kh = lane >> 5;
    const int t = n & 15;
    const int a0 = pick3(amn, m0), a1 = pick3(amn, m1), av = pick3(amn, vv), g0 = pick3(gs, m0), g1 = pick3(gs, m1), gv = pick3(gs, vv);
    const unsigned yy = (unsigned)min(a1 + (t >> 2), g1 - 1), xx = (unsigned)min(a0 + (t & 3), g0 - 1);
    const float4* __restrict__ P = reinterpret_cast<const float4*>(S.plane[K]) + (size_t)(__umul24(yy, (unsigned)g0) + xx) * 12u + 2u * (unsigned)kh;
    const float4* __restrict__ Ln = reinterpret_cast<const float4*>(S.line[K]);
    const unsigned r0 = (unsigned)min(av + (n >> 4), gv - 1), r1 = (unsigned)min(av + 2 + (n >> 4), gv - 1);
    const float4* __restrict__ L0 = Ln + (size_t)r0 * 12u + 2u * (unsigned)kh;
    const float4* __restrict__ L1 = Ln + (size_t)r1 * 12u + 2u * (unsigned)kh;
    // every operand load of the pair in flight together: 6 + 6 + 6 x 16 B per lane
    float4 pv[3][2], l0v[3][2], l1v[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pv[c][0] = P[4 * c]; pv[c][1] = P[4 * c + 1];
        l0v[c][0] = L0[4 * c]; l0v[c][1] = L0[4 * c + 1];
        l1v[c][0] = L1[4 * c]; l1v[c][1] = L1[4 * c + 1];
    }
    f32x16 acc0 = {0}, acc1 = {0};
    const uint4* __restrict__ ap = Wl + lane + (3 * K) * 2 * 64;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint4 ahi = ap[(c * 2) * 64], alo = ap[(c * 2 + 1) * 64];
        const float p8[8] = {pv[c][0].x, pv[c][0].y, pv[c][0].z, pv[c][0].w, pv[c][1].x, pv[c][1].y, pv[c][1].z, pv[c][1].w};
        const float a8[8] = {l0v[c][0].x, l0v[c][0].y, l0v[c][0].z, l0v[c][0].w, l0v[c][1].x, l0v[c][1].y, l0v[c][1].z, l0v[c][1].w};
        const float b8[8] = {l1v[c][0].x, l1v[c][0].y, l1v[c][0].z, l1v[c][0].w, l1v[c][1].x, l1v[c][1].y, l1v[c][1].z, l1v[c][1].w};
        float x0[8], x1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { x0[e] = p8[e] * a8[e]; x1[e] = p8[e] * b8[e]; }
        h8 h0, q0, h1, q1;
        split8(x0, h0, q0, amax);
        split8(x1, h1, q1, amax);
        acc0 = mfma16(ahi, h0, acc0); acc1 = mfma16(ahi, h1, acc1);
        acc0 = mfma16(ahi, q0, acc0); acc1 = mfma16(ahi, q1, acc1);
        acc0 = mfma16(alo, h0, acc0); acc1 = mfma16(alo, h1, acc1);
    }
    // lane (n, kh) holds features 8 b + 4 kh + (0..3) of column n in registers 4 b .. 4 b + 3; table row = 32 nb + n = 16 l + t
    float* __restrict__ d0 = T + n * kTJ + 4 * kh;
    float* __restrict__ d1 = d0 + 32 * kTJ;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        if (b == 3 && kh) continue;   // features 28..31 do not exist
        *reinterpret_cast<float4*>(d0 + 8 * b) = make_float4(acc0[4 * b] * kWUnscale, acc0[4 * b + 1] * kWUnscale, acc0[4 * b + 2] * kWUnscale, acc0[4 * b + 3] * kWUnscale);
        *reinterpret_cast<float4*>(d1 + 8 * b) = make_float4(acc1[4 * b] * kWUnscale, acc1[4 * b + 1] * kWUnscale, acc1[4 * b + 2] * kWUnscale, acc1[4 * b + 3] * kWUnscale);
    }
}

// feat += sum over the sample's 4 plane taps x 2 line rows of pair K, from the table
__device__ __forceinline__ void table_read(const float* __restrict__ T, int K, const int (&o)[3], const float (&w1)[3], float (&feat)[28]) {
    const int m0 = mat0(K), m1 = mat1(K), vv = vecm(K);
    const float wx1 = pick3f(w1, m0), wx0 = 1.f - wx1, wy1 = pick3f(w1, m1), wy0 = 1.f - wy1, wl1 = pick3f(w1, vv), wl0 = 1.f - wl1;
    const float wp[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
    const float* __restrict__ base = T + ((pick3(o, vv) << 4) + (pick3(o, m1) << 2) + pick3(o, m0)) * kTJ;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const float wl = l ? wl1 : wl0;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const float ww = wp[tp] * wl;
            const float4* __restrict__ row = reinterpret_cast<const float4*>(base + (l * 16 + (tp >> 1) * 4 + (tp & 1)) * kTJ);
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const float4 v = row[g];
                feat[4 * g] = fmaf(v.x, ww, feat[4 * g]); feat[4 * g + 1] = fmaf(v.y, ww, feat[4 * g + 1]);
                feat[4 * g + 2] = fmaf(v.z, ww, feat[4 * g + 2]); feat[4 * g + 3] = fmaf(v.w, ww, feat[4 * g + 3]);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_app_features_tiles(const Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint4* __restrict__ Wl = reinterpret_cast<uint4*>(smem);
    float* __restrict__ T = smem + kBasisVec * 4 + (size_t)wid * kTFloats;
    const FieldDev& F = a.F;
    for (int i = threadIdx.x; i < kBasisVec; i += 256) Wl[i] = F.basisH[i];
    __syncthreads();   // the only workgroup barrier
    if (a.split_unsafe && (*a.split_unsafe & kUnsafeBasis)) {   // basis_mat weights beyond the fixed pre-scale's range: exact redo
        if (a.range_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.range_flag, 1u);
        return;
    }
    // sub-list geometry: tiles (32 rows) before each sub-list, as the head enumerates them
    unsigned cnt_l = 0;
    if (lane < a.nlists) {
        cnt_l = a.counters[lane * kCounterStride];
        if (cnt_l > a.list_cap) cnt_l = a.list_cap;
    }
    unsigned incl = (cnt_l + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    // rows between a sub-list's end and the end of its last tile: the head reads them (never stores their colours): zeros
    if (blockIdx.x == 0 && wid == 0) {
        for (int l = 0; l < a.nlists; ++l) {
            const unsigned c = __shfl(cnt_l, l), before = l ? __shfl(incl, l - 1) : 0u, end = __shfl(incl, l);
            const unsigned row = before * 32u + c + (unsigned)lane;
            if (row < end * 32u && row < a.feat_rows && (unsigned)lane < 32u) {
                float4* __restrict__ d = reinterpret_cast<float4*>(a.feat + (size_t)row * 32);
#pragma unroll
                for (int g = 0; g < 8; ++g) d[g] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
    const int tiles_x = (a.img_w + 7) >> 3;
    const long long tile = (long long)blockIdx.x * 4 + wid;
    const int ty = (int)(tile / tiles_x), tx = (int)(tile - (long long)ty * tiles_x);
    if (ty * 8 >= a.img_h) return;
    const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    const bool have = px < a.img_w && py < a.img_h;
    int4 ra = make_int4(0, 0, 0, 0);
    if (have) ra = a.ray_app[(long long)py * a.img_w + px];
    const unsigned n = have ? (unsigned)ra.y : 0u;
    const unsigned li = n ? (unsigned)ra.x / a.list_cap : 0u;
    const unsigned bsel = __shfl(incl, li ? (int)li - 1 : 0);            // (every lane takes part in the shuffle)
    const unsigned before = li ? bsel : 0u;
    const unsigned row0 = before * 32u + ((unsigned)ra.x - li * a.list_cap);
    const float4* __restrict__ ent = a.app_pos + (unsigned)ra.x;

    const int gs[3] = {F.app.W[0], F.app.H[0], F.app.H[1]};      // grid size along coordinate 0, 1, 2
    unsigned k = 0;
    float amax = 0.f;
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    int i0[3] = {0, 0, 0};
    float w1[3] = {0.f, 0.f, 0.f};
    auto fetch = [&]() {
        if (k < n) {
            e = ent[k];
            const Axes3 A = sample_axes_inbox(F.app, e.x, e.y, e.z);
            i0[0] = min(max(A.a[0].i0, 0), F.app.W[0] - 1); i0[1] = min(max(A.a[1].i0, 0), F.app.H[0] - 1); i0[2] = min(max(A.a[2].i0, 0), F.app.H[1] - 1);
            w1[0] = A.a[0].w1; w1[1] = A.a[1].w1; w1[2] = A.a[2].w1;
        }
    };
    fetch();
    for (;;) {
        const unsigned long long pend = __ballot(k < n);
        if (!pend) break;
        const int lead = (int)__builtin_ctzll(pend);
        const int amn[3] = {max(__builtin_amdgcn_readlane(i0[0], lead) - 1, 0), max(__builtin_amdgcn_readlane(i0[1], lead) - 1, 0),
                            max(__builtin_amdgcn_readlane(i0[2], lead) - 1, 0)};
        const int o[3] = {i0[0] - amn[0], i0[1] - amn[1], i0[2] - amn[2]};
        const bool fit = k < n && (unsigned)o[0] <= 2u && (unsigned)o[1] <= 2u && (unsigned)o[2] <= 2u;
        const int oc[3] = {fit ? o[0] : 0, fit ? o[1] : 0, fit ? o[2] : 0};     // lanes that wait read (and drop) the box's first entries
        float feat[28];
#pragma unroll
        for (int j = 0; j < 28; ++j) feat[j] = 0.f;
#pragma unroll 1
        for (int K = 0; K < 3; ++K) {
            table_build(F.app, K, amn, gs, Wl, T, lane, amax);
            wave_sync();
            table_read(T, K, oc, w1, feat);
            wave_sync();
        }
        if (fit) {
            const unsigned row = row0 + k;
            if (row < a.feat_rows) {
                float4* __restrict__ d = reinterpret_cast<float4*>(a.feat + (size_t)row * 32);
#pragma unroll
                for (int g = 0; g < 6; ++g) d[g] = make_float4(feat[4 * g], feat[4 * g + 1], feat[4 * g + 2], feat[4 * g + 3]);
                d[6] = make_float4(feat[24], feat[25], feat[26], e.w);      // column 27: the entry's compositing weight
            }
            ++k;
            fetch();
        }
    }
    if (a.range_flag && __any(!(amax <= 60000.f)) && lane == 0) atomicOr(a.range_flag, 1u);
}

}  // namespace apt

// gather + basis stage of a frame in image order (the tile marcher's lists): feature rows for tiles [0, feat_rows / 32)
int launch_app_features_tiles(t2n_field* f, const float4* app_pos, const int4* ray_app, const unsigned* counters, unsigned list_cap,
                              int img_w, int img_h, float* feat, unsigned feat_rows, unsigned* range_flag, hipStream_t s) {
    using namespace apt;
    Args a;
    a.F = f->dev;
    a.app_pos = app_pos; a.ray_app = ray_app; a.counters = counters; a.list_cap = list_cap; a.nlists = kLists;
    a.img_w = img_w; a.img_h = img_h; a.feat = feat; a.feat_rows = feat_rows; a.range_flag = range_flag; a.split_unsafe = f->split_unsafe;
    const size_t lds = (size_t)kBasisVec * 16 + (size_t)4 * kTFloats * sizeof(float);
    const long long tiles = (long long)((img_w + 7) / 8) * ((img_h + 7) / 8);
    hipLaunchKernelGGL(k_app_features_tiles, dim3((unsigned)((tiles + 3) / 4)), dim3(256), lds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n
