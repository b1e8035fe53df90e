// Appearance: VM-split appearance-factor gather, basis_mat, positional encoding, the 351->128->128->3 MLP (or the SH /
// RGB heads) for a compacted list of samples (K2).
//
// Replaces (reference): models/tensoRF.py:223-239 (compute_appfeature), models/tensorBase.py:11-17 (positional_encoding),
// :88-109 (MLPRender_Fea_noview), :29-33 + models/sh.py:87-112 (SHRender), :36-39 (RGBRender).
//
// Mapping (gfx950, wave64): one wave owns a tile of 32 samples; the dense contractions run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so results do not depend on how samples are grouped into tiles).
// Samples sit on the MFMA N axis (lane & 31), output units on M: lane (s, h = lane>>5) ends a layer holding, for
// sample s, the units u = (v&3) + 8*(v>>2) + 4*h of each 32-row block in accumulator register v. Activations go through
// the wave's private LDS tile ([unit][33], k-major, +1 pad: conflict-free) to become the next layer's B operand
// (K-step t pairs units 2t and 2t+1 on the two half-waves). The weight matrices are re-packed once per upload
// (k_pack_mlp) into that K order, one dword per lane per (K-step, M-block), so every A fetch is a coalesced 256-B read.
//
//   gather   384 (sample, channel-quad) items per plane over 64 lanes: each tap is a contiguous 192-B read;
//            plane x line products go to the LDS tile X[144][33]
//   basis    72 K-steps,  A = basisA[t][lane],  B = X[2t+h][s]            -> 27(32) features per sample -> LDS Fe[32][33]
//   layer 0  14 feature pairs x (raw, sin/cos x 6 octaves) = 182 K-steps + 1 bias step, 4 M-blocks each
//   layer 1  64 K-steps + bias, B = relu(h0) from LDS;  layer 2: 64 K-steps + bias, 1 M-block (3 live rows)
//   sigmoid, store rgb for the wave's 32 samples.
#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXld = 33;                 // padded row length of the LDS tiles
constexpr int kAppK = 144;               // 3 * 48
constexpr int kTileFloats = kAppK * kXld;  // per wave
constexpr int kPE = 6;
constexpr int kL0Pairs = 14;
// K-step counts of the four contractions (2 input units per step) and their software-pipeline stages.
// Layer 0 order: 14 raw-feature steps, then per feature pair 6 x (sin, cos) steps, then the bias step.
constexpr int kBasisReal = kAppK / 2;                       // 72
constexpr int kL0Real = kL0Pairs * (1 + 2 * kPE) + 1;        // 183 (incl. bias)
constexpr int kL1Real = 65, kL2Real = 65;                    // 64 + bias
constexpr int kKS1 = 8, kKS4 = 2;                            // K-steps per stage for 1-block / 4-block contractions
constexpr int kBasisStages = 9, kL0Stages = 93, kL1Stages = 33, kL2Stages = 9;   // multiples of 3 (triple-buffered)
constexpr int kPadStages = 2;                                // zero stages the prefetcher may over-read
constexpr int kBasisSteps = (kBasisStages + kPadStages) * kKS1;
constexpr int kL0Steps = (kL0Stages + kPadStages) * kKS4;
constexpr int kL1Steps = (kL1Stages + kPadStages) * kKS4;
constexpr int kL2Steps = (kL2Stages + kPadStages) * kKS1;
static_assert(kBasisStages * kKS1 >= kBasisReal && kL0Stages * kKS4 >= kL0Real && kL1Stages * kKS4 >= kL1Real && kL2Stages * kKS1 >= kL2Real, "stages");

__host__ __device__ constexpr int unit_of(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Software-pipelined contraction: acc[m] += A_t,m x B_t over nstages*KS K-steps. The A operands (one dword per lane per
// (step, block), contiguous) are prefetched two stages ahead into a rotating set of three register groups, so their L2
// latency hides under the MFMAs of the stages in between (hipcc issues a load where it is written and waits at first use;
// it does not pipeline across loop iterations by itself). `bf(t)` supplies the B operand of K-step t, called in order.
template <int NMB, int KS, class BF>
__device__ __forceinline__ void mfma_stream(f32x16 (&acc)[NMB], const float* __restrict__ ap, int nstages, BF& bf) {
    constexpr int NR = NMB * KS;
    float A0[NR], A1[NR], A2[NR];
    auto fetch = [&](float (&A)[NR], int st) {
        const float* p = ap + (size_t)st * NR * 64;
#pragma unroll
        for (int i = 0; i < NR; ++i) A[i] = p[i * 64];
    };
    auto run = [&](const float (&A)[NR], int st) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const float b = bf(st * KS + k);
#pragma unroll
            for (int m = 0; m < NMB; ++m) acc[m] = mfma(A[k * NMB + m], b, acc[m]);
        }
    };
    fetch(A0, 0);
    fetch(A1, 1);
#pragma unroll 1
    for (int st = 0; st < nstages; st += 3) {
        fetch(A2, st + 2); run(A0, st);
        fetch(A0, st + 3); run(A1, st + 1);
        fetch(A1, st + 4); run(A2, st + 2);
    }
}

// B operand from the wave's LDS tile T[unit][kXld]: K-step t pairs units 2t (low half-wave) and 2t+1 (high half-wave);
// step `nreal` is the bias step (B = 1 on the low half), later steps are zero padding.
struct LdsB {
    const float* base;   // T + h * kXld + s
    int nreal, h;
    __device__ __forceinline__ float operator()(int t) const {
        if (t < nreal) return base[(size_t)2 * t * kXld];
        return (t == nreal && h == 0) ? 1.f : 0.f;
    }
};
struct LdsBNoBias {
    const float* base;
    int nreal;
    __device__ __forceinline__ float operator()(int t) const { return t < nreal ? base[(size_t)2 * t * kXld] : 0.f; }
};

// B operand of layer 0: positional encoding generated on the fly from the feature tile Fe[32][kXld].
struct PeB {
    const float* fe;   // Fe + h * kXld + s  (feature 2r+h of this lane's sample at fe[2r * kXld])
    int h;
    float fv, cs, scale;
    __device__ __forceinline__ float operator()(int t) {
        if (t < kL0Pairs) return fe[(size_t)2 * t * kXld];
        const int u = t - kL0Pairs;
        if (u < kL0Pairs * 2 * kPE) {
            const int j = u % (2 * kPE);
            if (j == 0) { fv = fe[(size_t)2 * (u / (2 * kPE)) * kXld]; scale = 1.f; }
            if ((j & 1) == 0) {
                float sn;
                sincosf(fv * scale, &sn, &cs);
                scale *= 2.f;
                return sn;
            }
            return cs;
        }
        return (u == kL0Pairs * 2 * kPE && h == 0) ? 1.f : 0.f;   // bias step, then zero padding
    }
};

// ---- f16 two-way split contraction ------------------------------------------------------------------------------------
// x = hi + lo with hi = f16(x), lo = f16(x - hi): 22 effective mantissa bits. a*b ~ ahi*bhi + ahi*blo + alo*bhi on
// v_mfma_f32_32x32x16_f16 (fp32 accumulate): 3 x 32 cycles per 16 K-values against 8 x 64 for the fp32 MFMA, with a
// relative product error ~2^-21 (the dropped lo*lo term). Weights are pre-split (and pre-scaled by 2^8 so their lo halves
// stay in the f16 normal range) at upload; activations are split in registers. Lane (s, h) feeds K-values 8h..8h+7 of
// every 16-value chunk.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
constexpr float kWScale = 256.f, kWUnscale = 1.f / 256.f;
constexpr int kBasisChunks = 10, kL0Chunks = 24, kL1Chunks = 8, kL2Chunks = 8;   // even (double-buffered), zero-padded
constexpr int kBasisChunksReal = 9, kL0ChunksReal = 23;

__device__ __forceinline__ f32x16 mfma16(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// -1.0f the optimiser cannot see through: x - hi then stays an FMA whose f16 operand is read in place (v_fma_mix_f32), no
// convert instruction
__device__ __forceinline__ float opaque_neg1() {
    float x;
    asm("s_mov_b32 %0, 0xbf800000" : "=s"(x));
    return x;
}
template <bool TRACK = true>
__device__ __forceinline__ void split8(const float (&x)[8], h8& hi, h8& lo, float& amax) {
    // hi = RTZ_f16(x) (packed convert), lo = RTZ_f16(x - hi): the residual is exact in fp32 (hi is a truncation of x) and is
    // formed by v_fma_mix_f32 straight from the packed halves: two instructions per value
    typedef __fp16 hh2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    // amax: largest magnitude handed to the packed convert (it saturates silently, RTZ): the callers raise the launch's range
    // flag when it passes the f16 range and the launch is redone on the exact path (v_max3_f32, one instruction per pair)
    const float n1 = opaque_neg1();
    u4 uh, ul;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if constexpr (TRACK) amax = fmaxf(fmaxf(amax, fabsf(x[2 * e])), fabsf(x[2 * e + 1]));
        const hh2 p = __builtin_amdgcn_cvt_pkrtz(x[2 * e], x[2 * e + 1]);
        const h2 ph = __builtin_bit_cast(h2, p);
        const float r0 = fmaf((float)ph[0], n1, x[2 * e]), r1 = fmaf((float)ph[1], n1, x[2 * e + 1]);
        uh[e] = __builtin_bit_cast(unsigned, p);
        ul[e] = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
    hi = __builtin_bit_cast(h8, uh);
    lo = __builtin_bit_cast(h8, ul);
}

// acc[m] += W_chunk,m (split) x B_chunk (split on the fly); A operands double-buffered one chunk ahead.
template <int NMB, bool TRACK = true, class BF>
__device__ __forceinline__ void f16_stream(f32x16 (&acc)[NMB], const uint4* __restrict__ ap, int lane, int nchunks, BF& bf, float& amax) {
    constexpr int NR = NMB * 2;
    uint4 A0[NR], A1[NR];
    auto fetch = [&](uint4 (&A)[NR], int c) {
        const uint4* p = ap + lane + (size_t)c * NR * 64;
#pragma unroll
        for (int i = 0; i < NR; ++i) A[i] = p[i * 64];
    };
    auto run = [&](const uint4 (&A)[NR], int c) {
        float x[8];
        bf(c, x);
        h8 bhi, blo;
        split8<TRACK>(x, bhi, blo, amax);
#pragma unroll
        for (int m = 0; m < NMB; ++m) {
            const h8 ahi = __builtin_bit_cast(h8, A[2 * m]), alo = __builtin_bit_cast(h8, A[2 * m + 1]);
            acc[m] = mfma16(ahi, bhi, acc[m]);
            acc[m] = mfma16(ahi, blo, acc[m]);
            acc[m] = mfma16(alo, bhi, acc[m]);
        }
    };
    fetch(A0, 0);
#pragma unroll 1
    for (int c = 0; c < nchunks; c += 2) {
        fetch(A1, c + 1); run(A0, c);
        fetch(A0, c + 2); run(A1, c + 1);   // the last prefetch reads the zero pad chunk
    }
}

// Branch-free sincos for the positional encoding: two-constant Cody-Waite reduction by pi/2 with FMAs, cephes minimax
// polynomials on [-pi/4, pi/4], quadrant fix-up by selects. Max abs error 9.4e-8 for |x| <= 5000 (libm-class); no control
// flow, so the compiler can interleave it with the MFMAs of the neighbouring chunk (ocml's sincosf carries a
// large-argument branch that splits the basic block). Callers route |x| > 8192 to sincosf.
__device__ __forceinline__ void fast_sincosf(float x, float* sn, float* cs) {
    const float k = rintf(x * 0.6366197723675814f);
    float r = fmaf(-k, 1.5707963705062866f, x);
    r = fmaf(-k, -4.371138828673793e-8f, r);
    const float z = r * r;
    const float s = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
    const float c = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                         fmaf(z, -0.5f, 1.f));
    const int q = (int)k & 3;
    const float s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// B chunk from the wave's LDS tile T[unit][LD]
template <int LD>
struct LdsChunkT {
    const float* base;   // T + s
    int h, nreal;
    __device__ __forceinline__ void operator()(int c, float (&x)[8]) const {
        if (c < nreal) {
            const float* p = base + (size_t)(16 * c + 8 * h) * LD;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = p[e * LD];
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
        }
    }
};
typedef LdsChunkT<kXld> LdsChunk;
// B chunk of layer 0: chunks 0-1 raw features (32 rows of Fe, rows >= 27 are zero); chunks 2..22 carry the positional
// encoding as (sin, cos) pairs. Half-wave h owns features 14h .. 14h+13 and walks them in order, six octaves each, four
// pairs per chunk: pair j = 4(c-2) + p of half h is (feature 14h + j/6, octave j%6). An accurate sincos is taken at octaves
// 0 and 3, the octaves in between by the double-angle identities sin 2a = 2 sin a cos a, cos 2a = (cos a - sin a)(cos a +
// sin a): at most two doublings (error x4 of ~1e-7).
__device__ __forceinline__ void pe_double(float& sn, float& cs) {
    const float s2 = 2.f * sn * cs, c2 = (cs - sn) * (cs + sn);
    sn = s2; cs = c2;
}
struct PeChunk {
    const float* fe;   // Fe + s
    int h;
    float sn, cs;      // running pair, carried from chunk to chunk
    __device__ __forceinline__ void operator()(int c, float (&x)[8]) {
        if (c < 2) {
            const float* p = fe + (size_t)(16 * c + 8 * h) * kXld;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = p[e * kXld];
        } else if (c < kL0ChunksReal) {
            const int j0 = 4 * (c - 2);
            int f = (j0 * 171) >> 10;                    // j0 / 6 for j0 < 504
            int q = j0 - 6 * f;
            f += 14 * h;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (q == 0 || q == 3) {                  // wave-uniform
                    const float arg = fe[(size_t)f * kXld] * (float)(1 << q);
                    if (__builtin_expect(__any(fabsf(arg) > 8192.f), 0)) sincosf(arg, &sn, &cs);
                    else fast_sincosf(arg, &sn, &cs);
                } else {
                    pe_double(sn, cs);
                }
                x[2 * p] = sn; x[2 * p + 1] = cs;
                if (++q == kPE) { q = 0; ++f; }
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
        }
    }
};

// accumulator init with the (pre-scaled) bias of block m: lane (s, h) register v <- bias[m*32 + unit_of(v, h)]
__device__ __forceinline__ f32x16 bias_init(const float* __restrict__ bp, int m, int h) {
    const float4* p = reinterpret_cast<const float4*>(bp + (size_t)(m * 2 + h) * 16);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}

constexpr unsigned kUnsafeBasis = 1u, kUnsafeHead = 2u;   // bits of the split_unsafe word: which weights left the pre-scale's range
struct ShadeArgs {
    FieldDev F;
    const float4* app_pos; const int* app_ray; const float* rays; int ray_stride;
    const float* xyz; const float* viewdirs;      // explicit-point mode (t2n_shade_at): xyz [n,3], viewdirs [n,3] or null
    const unsigned* counters; unsigned list_cap; int nlists; unsigned count_max;   // list mode: counters[nlists]; point mode: count_max
    float4* app_rgb; float* feat_out; float* rgb_out;   // rgb_out: packed [n,3] (explicit-point mode)
    ShadeCtx ctx;   // activation rows for the backward pass (all NULL in the normal forward)
    unsigned ctx_rows;   // rows the ctx buffers hold: tiles past it keep nothing (forward-kept activations with a capacity guess)
    unsigned tile_lo, tile_hi;        // this launch covers tiles [tile_lo, min(ntiles, tile_hi)) of the sub-lists
    const unsigned* run_if_nonzero;   // when set: the launch is a no-op unless *run_if_nonzero != 0 (exact-path redo gate)
    unsigned* range_flag;             // when set: raised by the split path when a value left the f16 range
    const unsigned* split_unsafe;     // device word set at upload when a weight x 2^8 (the one-kernel split path's fixed pre-scale) leaves
                                      // the f16 range: split launches then do nothing (and raise range_flag) and the exact launch behind them runs
    unsigned long long* stats;        // redo launches count themselves in stats[T2N_STAT_F16_REDO]
};

// Gather: 384 (sample, channel-quad) items over 64 lanes, 6 per lane; an item computes its sample's three axis taps once
// and fetches quad q of the 4 plane taps + 2 line taps of all three factor pairs (each tap = 192 contiguous bytes across
// the 12 lanes of a sample); plane x line products go to X[k*48 + 4q + c][s] (and to the ctx rows in backward mode).
// The 18 loads of an item are issued together, branch-free (dead lanes fetch the volume centre and store zeros), and two
// items are kept in flight, so a tile costs ~3 memory round trips instead of 18 (appearance planes mostly miss the L2).
struct GatherItem {
    QuadTaps t[3];
    int s, q;
    bool live;
};
template <bool HALF = false>
__device__ __forceinline__ void gather_issue(const FactorSet& S, int it, int lane, const float4* pos_l, const float* xyz,
                                             unsigned base, unsigned count, GatherItem& g) {
    constexpr int CQ = 12;   // 48 channels / 4
    const int item = it * 64 + lane;
    g.s = item / CQ; g.q = item - g.s * CQ;
    const unsigned idx = base + (unsigned)g.s;
    g.live = idx < count;
    float xn = 0.f, yn = 0.f, zn = 0.f;
    if (g.live) {
        if (pos_l) { const float4 p = pos_l[idx]; xn = p.x; yn = p.y; zn = p.z; }
        else { xn = xyz[(size_t)idx * 3]; yn = xyz[(size_t)idx * 3 + 1]; zn = xyz[(size_t)idx * 3 + 2]; }
    }
    const Axes3 A = sample_axes(S, xn, yn, zn);
    issue_taps_ax<0, HALF>(S, CQ, g.q, A, g.t[0]);
    issue_taps_ax<1, HALF>(S, CQ, g.q, A, g.t[1]);
    issue_taps_ax<2, HALF>(S, CQ, g.q, A, g.t[2]);
}
__device__ __forceinline__ void gather_consume(const GatherItem& g, float* __restrict__ X, float* ctx_x, unsigned row0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float4 p = taps_plane(g.t[k]), l = taps_line(g.t[k]);
        float4 v = make_float4(p.x * l.x, p.y * l.y, p.z * l.z, p.w * l.w);
        if (!g.live) v = make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst = X + (size_t)(k * 48 + g.q * 4) * kXld + g.s;
        dst[0] = v.x; dst[kXld] = v.y; dst[2 * kXld] = v.z; dst[3 * kXld] = v.w;
        if (ctx_x) *reinterpret_cast<float4*>(ctx_x + (size_t)(row0 + g.s) * kAppK + k * 48 + g.q * 4) = v;
    }
}

// bf16 factor storage: a tap is 96 contiguous bytes, fetched as six 16-B octets (8 channels each) - 192 (sample, octet) items
// over 64 lanes, 3 per lane, half the load instructions and half the L1 bytes of the fp32 gather. Taps stay packed until they
// are consumed; the arithmetic per channel is the fp32 gather's (bf16 -> fp32 is exact), so the products are bit-identical.
struct OctTaps {
    uint4 nw, ne, sw, se, l0, l1;
    float wnw, wne, wsw, wse, wl0, wl1;
};
struct GatherItemH {
    OctTaps t[3];
    int s, o;
    bool live;
};
template <int K>
__device__ __forceinline__ void issue_octets_ax(const FactorSet& S, int o, const Axes3& A, OctTaps& t) {
    const Axis& ax = A.a[mat0(K)];
    const Axis& ay = A.a[mat1(K)];
    const Axis& al = A.a[vecm(K)];
    const unsigned W = (unsigned)S.W[K];
    constexpr unsigned tb = 96u;   // bytes per texel: 48 channels x 2
    const unsigned ob = (unsigned)o * 16u;
    const unsigned r0 = __umul24((unsigned)ay.i0, W), r1 = __umul24((unsigned)ay.i1, W);
    const char* __restrict__ P = reinterpret_cast<const char*>(S.plane_h[K]);
    const char* __restrict__ Ln = reinterpret_cast<const char*>(S.line_h[K]);
    t.nw = *reinterpret_cast<const uint4*>(P + ((r0 + (unsigned)ax.i0) * tb + ob));
    t.ne = *reinterpret_cast<const uint4*>(P + ((r0 + (unsigned)ax.i1) * tb + ob));
    t.sw = *reinterpret_cast<const uint4*>(P + ((r1 + (unsigned)ax.i0) * tb + ob));
    t.se = *reinterpret_cast<const uint4*>(P + ((r1 + (unsigned)ax.i1) * tb + ob));
    t.l0 = *reinterpret_cast<const uint4*>(Ln + ((unsigned)al.i0 * tb + ob));
    t.l1 = *reinterpret_cast<const uint4*>(Ln + ((unsigned)al.i1 * tb + ob));
    t.wnw = ay.w0 * ax.w0; t.wne = ay.w0 * ax.w1; t.wsw = ay.w1 * ax.w0; t.wse = ay.w1 * ax.w1;
    t.wl0 = al.w0; t.wl1 = al.w1;
}
__device__ __forceinline__ void gather_issue_h(const FactorSet& S, int it, int lane, const float4* pos_l, const float* xyz,
                                               unsigned base, unsigned count, GatherItemH& g) {
    const int item = it * 64 + lane;
    g.s = item / 6; g.o = item - g.s * 6;
    const unsigned idx = base + (unsigned)g.s;
    g.live = idx < count;
    float xn = 0.f, yn = 0.f, zn = 0.f;
    if (g.live) {
        if (pos_l) { const float4 p = pos_l[idx]; xn = p.x; yn = p.y; zn = p.z; }
        else { xn = xyz[(size_t)idx * 3]; yn = xyz[(size_t)idx * 3 + 1]; zn = xyz[(size_t)idx * 3 + 2]; }
    }
    const Axes3 A = sample_axes(S, xn, yn, zn);
    issue_octets_ax<0>(S, g.o, A, g.t[0]);
    issue_octets_ax<1>(S, g.o, A, g.t[1]);
    issue_octets_ax<2>(S, g.o, A, g.t[2]);
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ void gather_consume_h(const GatherItemH& g, float* __restrict__ X, float* ctx_x, unsigned row0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const OctTaps& t = g.t[k];
        const unsigned nw[4] = {t.nw.x, t.nw.y, t.nw.z, t.nw.w}, ne[4] = {t.ne.x, t.ne.y, t.ne.z, t.ne.w};
        const unsigned sw[4] = {t.sw.x, t.sw.y, t.sw.z, t.sw.w}, se[4] = {t.se.x, t.se.y, t.se.z, t.se.w};
        const unsigned l0[4] = {t.l0.x, t.l0.y, t.l0.z, t.l0.w}, l1[4] = {t.l1.x, t.l1.y, t.l1.z, t.l1.w};
        float v[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            // the fp32 gather's order: plane = ((nw wnw + ne wne) + sw wsw) + se wse by fma, line = l0 wl0 + l1 wl1, product
            float pa = bf16_lo(nw[d]) * t.wnw, pb = bf16_hi(nw[d]) * t.wnw;
            pa = fmaf(bf16_lo(ne[d]), t.wne, pa); pb = fmaf(bf16_hi(ne[d]), t.wne, pb);
            pa = fmaf(bf16_lo(sw[d]), t.wsw, pa); pb = fmaf(bf16_hi(sw[d]), t.wsw, pb);
            pa = fmaf(bf16_lo(se[d]), t.wse, pa); pb = fmaf(bf16_hi(se[d]), t.wse, pb);
            float la = bf16_lo(l0[d]) * t.wl0, lb = bf16_hi(l0[d]) * t.wl0;
            la = fmaf(bf16_lo(l1[d]), t.wl1, la); lb = fmaf(bf16_hi(l1[d]), t.wl1, lb);
            v[2 * d] = g.live ? pa * la : 0.f;
            v[2 * d + 1] = g.live ? pb * lb : 0.f;
        }
        float* dst = X + (size_t)(k * 48 + g.o * 8) * kXld + g.s;
#pragma unroll
        for (int c = 0; c < 8; ++c) dst[c * kXld] = v[c];
        if (ctx_x) {
            float4* cp = reinterpret_cast<float4*>(ctx_x + (size_t)(row0 + g.s) * kAppK + k * 48 + g.o * 8);
            cp[0] = make_float4(v[0], v[1], v[2], v[3]);
            cp[1] = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
}

template <bool HALF = false>
__device__ __forceinline__ void gather_all(const FactorSet& S, float* __restrict__ X, int lane, const float4* pos_l,
                                           const float* xyz, unsigned base, unsigned count, float* ctx_x, unsigned row0) {
    if constexpr (HALF) {
        GatherItemH h0, h1;
        gather_issue_h(S, 0, lane, pos_l, xyz, base, count, h0);
        gather_issue_h(S, 1, lane, pos_l, xyz, base, count, h1);
        gather_consume_h(h0, X, ctx_x, row0);
        gather_issue_h(S, 2, lane, pos_l, xyz, base, count, h0);
        gather_consume_h(h1, X, ctx_x, row0);
        gather_consume_h(h0, X, ctx_x, row0);
        return;
    }
    GatherItem g0, g1;
    gather_issue<HALF>(S, 0, lane, pos_l, xyz, base, count, g0);
    gather_issue<HALF>(S, 1, lane, pos_l, xyz, base, count, g1);
    gather_consume(g0, X, ctx_x, row0);
    gather_issue<HALF>(S, 2, lane, pos_l, xyz, base, count, g0);
    gather_consume(g1, X, ctx_x, row0);
    gather_issue<HALF>(S, 3, lane, pos_l, xyz, base, count, g1);
    gather_consume(g0, X, ctx_x, row0);
    gather_issue<HALF>(S, 4, lane, pos_l, xyz, base, count, g0);
    gather_consume(g1, X, ctx_x, row0);
    gather_issue<HALF>(S, 5, lane, pos_l, xyz, base, count, g1);
    gather_consume(g0, X, ctx_x, row0);
    gather_consume(g1, X, ctx_x, row0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

template <bool SPLIT, bool HALF = false>
__global__ __launch_bounds__(256, 2) void k_shade(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    float* __restrict__ X = smem + (size_t)wid * kTileFloats;
    float* __restrict__ Fe = X;   // reused after the basis contraction
    const FieldDev& F = a.F;
    // tile enumeration over the appearance sub-lists (list l occupies [l*list_cap, l*list_cap + counters[l]))
    unsigned cnt_l = 0;
    if (lane < a.nlists) {
        cnt_l = a.counters ? a.counters[lane * kCounterStride] : a.count_max;
        if (a.counters && cnt_l > a.list_cap) cnt_l = a.list_cap;
    }
    unsigned incl = (cnt_l + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    unsigned ntiles = __shfl(incl, a.nlists - 1);
    if (ntiles > a.tile_hi) ntiles = a.tile_hi;
    if (a.run_if_nonzero) {
        if (*a.run_if_nonzero == 0u) return;
        if (a.stats && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.stats[T2N_STAT_F16_REDO], 1ull);
    }
    if (SPLIT && a.split_unsafe && *a.split_unsafe) {   // weights beyond the fixed pre-scale's range: the exact launch behind this one does the work
        if (a.range_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.range_flag, 1u);
        return;
    }
    const unsigned wave_stride = gridDim.x * 4u;
    float amax = 0.f;   // largest magnitude handed to an f16 convert (split path): past the f16 range the launch is redone exactly

    for (unsigned tile = a.tile_lo + blockIdx.x * 4u + wid; tile < ntiles; tile += wave_stride) {
        const int li = (int)__popcll(__ballot((lane < a.nlists) & (incl <= tile)));
        const unsigned before = li ? __shfl(incl, li - 1) : 0u;
        const unsigned lbase = (unsigned)li * a.list_cap;
        const unsigned base = lbase + (tile - before) * 32u;
        const unsigned count = lbase + __shfl(cnt_l, li);      // one past the last live entry of this sub-list
        // ---- gather: plane x line products for 32 samples x 144 channels -> X ---------------------------------------
        const unsigned row0 = tile * 32u;   // activation row of lane sample 0 (ctx mode)
        const bool keep_rows = row0 + 32u <= a.ctx_rows;
        const ShadeCtx cx = keep_rows ? a.ctx : ShadeCtx{nullptr, nullptr, nullptr, nullptr};
        gather_all<HALF>(F.app, X, lane, a.app_pos, a.xyz, base, count, cx.x144, row0);
        wave_lds_sync();

        // ---- basis_mat: feat[i][s] = sum_k Wb[i][k] X[k][s] ----------------------------------------------------------
        f32x16 accb1[1] = {{0}};
        if constexpr (SPLIT) {
            LdsChunk bf{X + s, h, kBasisChunksReal};
            f16_stream<1>(accb1, F.basisH, lane, kBasisChunks, bf, amax);
            accb1[0] *= kWUnscale;
        } else {
            LdsBNoBias bf{X + (size_t)h * kXld + s, kBasisReal};
            mfma_stream<1, kKS1>(accb1, F.basisA + lane, kBasisStages, bf);
        }
        const f32x16 accb = accb1[0];
        if constexpr (SPLIT) {   // the features are layer 0's raw operand chunks
#pragma unroll
            for (int v = 0; v < 16; v += 2) amax = fmaxf(fmaxf(amax, fabsf(accb[v])), fabsf(accb[v + 1]));
        }
        wave_lds_sync();
#pragma unroll
        for (int v = 0; v < 16; ++v) Fe[unit_of(v, h) * kXld + s] = accb[v];
        wave_lds_sync();

        const unsigned idx = base + (unsigned)s;
        const bool live = idx < count;
        if (cx.feat32) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(cx.feat32 + (size_t)(row0 + s) * 32 + 8 * g + 4 * h) =
                    make_float4(accb[4 * g], accb[4 * g + 1], accb[4 * g + 2], accb[4 * g + 3]);
        }
        if (a.feat_out && live) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int fidx = unit_of(v, h);
                if (fidx < F.app_dim) a.feat_out[(size_t)idx * F.app_dim + fidx] = accb[v];
            }
        }

        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (F.shading == T2N_SHADE_MLP_FEA_NOVIEW) {
            // ---- layer 0: PE on the fly, 4 M-blocks -------------------------------------------------------------------
            f32x16 acc0[4] = {{0}, {0}, {0}, {0}};
            if constexpr (SPLIT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc0[m] = bias_init(F.biasH, m, h);
                PeChunk bf{Fe + s, h, 0.f, 1.f};
                f16_stream<4, false>(acc0, F.w0H, lane, kL0Chunks, bf, amax);   // range: features tracked below, the encoding is bounded by 1
#pragma unroll
                for (int m = 0; m < 4; ++m) acc0[m] *= kWUnscale;
            } else {
                PeB bf{Fe + (size_t)h * kXld + s, h, 0.f, 0.f, 1.f};
                mfma_stream<4, kKS4>(acc0, F.w0A + lane, kL0Stages, bf);
            }
            // ---- layer 1: relu(h0) through LDS, 64 K-steps --------------------------------------------------------------
            float* __restrict__ Hs = X;
            wave_lds_sync();   // Fe reads done
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
                for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc0[ms][v], 0.f);
            }
            if (cx.h0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(cx.h0 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc0[ms][4 * g], 0.f), fmaxf(acc0[ms][4 * g + 1], 0.f),
                                        fmaxf(acc0[ms][4 * g + 2], 0.f), fmaxf(acc0[ms][4 * g + 3], 0.f));
            }
            wave_lds_sync();
            f32x16 acc1[4] = {{0}, {0}, {0}, {0}};
            if constexpr (SPLIT) {
#pragma unroll
                for (int m = 0; m < 4; ++m) acc1[m] = bias_init(F.biasH + 128, m, h);
                LdsChunk bf{Hs + s, h, kL1Chunks};
                f16_stream<4>(acc1, F.w1H, lane, kL1Chunks, bf, amax);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc1[m] *= kWUnscale;
            } else {
                LdsB bf{Hs + (size_t)h * kXld + s, 64, h};
                mfma_stream<4, kKS4>(acc1, F.w1A + lane, kL1Stages, bf);
            }
            // ---- layer 2 (3 live rows) ---------------------------------------------------------------------------------
            wave_lds_sync();   // h0 reads done
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
                for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc1[ms][v], 0.f);
            }
            if (cx.h1) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(cx.h1 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc1[ms][4 * g], 0.f), fmaxf(acc1[ms][4 * g + 1], 0.f),
                                        fmaxf(acc1[ms][4 * g + 2], 0.f), fmaxf(acc1[ms][4 * g + 3], 0.f));
            }
            wave_lds_sync();
            f32x16 acc2a[1] = {{0}};
            if constexpr (SPLIT) {
                acc2a[0] = bias_init(F.biasH + 256, 0, h);
                LdsChunk bf{Hs + s, h, kL2Chunks};
                f16_stream<1>(acc2a, F.w2H, lane, kL2Chunks, bf, amax);
                acc2a[0] *= kWUnscale;
            } else {
                LdsB bf{Hs + (size_t)h * kXld + s, 64, h};
                mfma_stream<1, kKS1>(acc2a, F.w2A + lane, kL2Stages, bf);
            }
            const f32x16 acc2 = acc2a[0];
            cr = sigmoidf_(acc2[0]); cg = sigmoidf_(acc2[1]); cb = sigmoidf_(acc2[2]);   // rows 0..2 live on h == 0
        } else if (F.shading == T2N_SHADE_SH) {
            if (h == 0 && live) {
                float dx, dy, dz;
                if (a.viewdirs) { dx = a.viewdirs[(size_t)idx * 3]; dy = a.viewdirs[(size_t)idx * 3 + 1]; dz = a.viewdirs[(size_t)idx * 3 + 2]; }
                else {
                    const float* rp = a.rays + (size_t)a.app_ray[idx] * a.ray_stride; dx = rp[3]; dy = rp[4]; dz = rp[5];
                    if (F.ztab) { const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz); dx = dx / nrm; dy = dy / nrm; dz = dz / nrm; }   // :445
                }
                // models/sh.py:4-14,87-112 (degree 2)
                const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
                const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                            C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                float sh[9];
                sh[0] = C0; sh[1] = -C1 * dy; sh[2] = C1 * dz; sh[3] = -C1 * dx;
                sh[4] = C20 * xy; sh[5] = C21 * yz; sh[6] = C22 * (2.0f * zz - xx - yy); sh[7] = C23 * xz; sh[8] = C24 * (xx - yy);
                float o[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int b = 0; b < 9; ++b) acc = fmaf(sh[b], Fe[(c * 9 + b) * kXld + s], acc);
                    o[c] = fmaxf(acc + 0.5f, 0.f);
                }
                cr = o[0]; cg = o[1]; cb = o[2];
            }
        } else {   // T2N_SHADE_RGB: features are the colour
            if (h == 0) { cr = Fe[0 * kXld + s]; cg = Fe[1 * kXld + s]; cb = Fe[2 * kXld + s]; }
        }
        if (h == 0 && live) {
            if (a.app_rgb) a.app_rgb[idx] = make_float4(cr, cg, cb, a.app_pos[idx].w);   // .w: the compositing weight rides along (k_composite reads one array)
            if (a.rgb_out) { a.rgb_out[(size_t)idx * 3] = cr; a.rgb_out[(size_t)idx * 3 + 1] = cg; a.rgb_out[(size_t)idx * 3 + 2] = cb; }
        }
        wave_lds_sync();   // Fe reads done before the next tile's gather overwrites X
    }
    if (SPLIT && a.range_flag && __any(!(amax <= 60000.f)) && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- finisher of tile-marcher launches (budgeted appearance lists) --------------------------------------------------------------
// The rays k_compact_list tagged in the overflow list own no list entries: their appearance samples sit in the marcher's per-ray
// staging slice (the first `cap`) and, past it, in the ray's spill row of weights. One wave per such ray walks them in sample
// order through a 32-entry LDS queue and evaluates gather -> basis_mat -> head on the exact fp32 matrix path (the arithmetic of
// k_shade<false>, the reference's own: models/tensoRF.py:223-239, models/tensorBase.py:88-109), sums w * rgb in the entries' order
// like k_composite and writes the ray's colour (models/tensorBase.py:494-501). No list memory is needed whatever the number of such
// rays, so t2n_render_forward never has to wait for the march's counters: a launch without tagged rays costs this kernel's
// empty dispatch (every workgroup reads one word).
struct FinishArgs {
    FieldDev F;
    const float* rays; int ray_stride; int n_samples;
    const int4* ray_app; const float* acc; float* rgb; int add_bg;
    const float* wbuf; const float4* scratch; int cap;
    const unsigned* ovf_count; const int* ovf_list;
    unsigned long long* stats;
};
constexpr int kFinQueue = 96;                                  // float4 entries: 31 left over + 64 new candidates
constexpr int kFinFloats = kTileFloats + kFinQueue * 4;        // per wave: the shade tile + the entry queue

__global__ __launch_bounds__(256) void k_finish_rays(const FinishArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned n_ovf = *a.ovf_count;
    if (n_ovf == 0u) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    float* __restrict__ X = smem + (size_t)wid * kFinFloats;
    float* __restrict__ Fe = X;
    float4* __restrict__ Q = reinterpret_cast<float4*>(X + kTileFloats);
    const FieldDev& F = a.F;
    const int N = a.n_samples;
    bool any = false;
    for (unsigned k = blockIdx.x * 4u + wid; k < n_ovf; k += gridDim.x * 4u) {
        const int tag = a.ovf_list[k];
        if (tag >= 0) continue;                    // found room in a sub-list: an ordinary ray
        any = true;
        const long long r = (long long)(tag & 0x7fffffff);
        const int4 ra = a.ray_app[r];
        const int from = -ra.x - 1, napp = ra.y, first = ra.w & 2047, Lw = ra.w >> 11;
        const int staged = napp < a.cap ? napp : a.cap;
        const float* __restrict__ rp = a.rays + r * a.ray_stride;
        float cr = 0.f, cg = 0.f, cb = 0.f;
        int q = 0;                                 // queued entries (wave-uniform)
        // shade the first `cnt` queued entries, add w * rgb in their order, drop them from the queue
        auto flush = [&](int cnt) {
            gather_all<false>(F.app, X, lane, Q, nullptr, 0u, (unsigned)cnt, nullptr, 0u);
            const float wq = s < cnt ? Q[s].w : 0.f;
            wave_lds_sync();
            f32x16 accb1[1] = {{0}};
            {
                LdsBNoBias bf{X + (size_t)h * kXld + s, kBasisReal};
                mfma_stream<1, kKS1>(accb1, F.basisA + lane, kBasisStages, bf);
            }
            const f32x16 accb = accb1[0];
            wave_lds_sync();
#pragma unroll
            for (int v = 0; v < 16; ++v) Fe[unit_of(v, h) * kXld + s] = accb[v];
            wave_lds_sync();
            float tr = 0.f, tg = 0.f, tb = 0.f;
            if (F.shading == T2N_SHADE_MLP_FEA_NOVIEW) {
                f32x16 acc0[4] = {{0}, {0}, {0}, {0}};
                {
                    PeB bf{Fe + (size_t)h * kXld + s, h, 0.f, 0.f, 1.f};
                    mfma_stream<4, kKS4>(acc0, F.w0A + lane, kL0Stages, bf);
                }
                float* __restrict__ Hs = X;
                wave_lds_sync();
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc0[ms][v], 0.f);
                wave_lds_sync();
                f32x16 acc1[4] = {{0}, {0}, {0}, {0}};
                {
                    LdsB bf{Hs + (size_t)h * kXld + s, 64, h};
                    mfma_stream<4, kKS4>(acc1, F.w1A + lane, kL1Stages, bf);
                }
                wave_lds_sync();
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc1[ms][v], 0.f);
                wave_lds_sync();
                f32x16 acc2a[1] = {{0}};
                {
                    LdsB bf{Hs + (size_t)h * kXld + s, 64, h};
                    mfma_stream<1, kKS1>(acc2a, F.w2A + lane, kL2Stages, bf);
                }
                tr = sigmoidf_(acc2a[0][0]); tg = sigmoidf_(acc2a[0][1]); tb = sigmoidf_(acc2a[0][2]);   // rows 0..2 live on h == 0
            } else if (F.shading == T2N_SHADE_SH) {
                if (h == 0) {   // models/sh.py:4-14,87-112 (degree 2), the ray's own direction
                    const float dx = rp[3], dy = rp[4], dz = rp[5];
                    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
                    const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                                C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
                    const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                    float sh[9];
                    sh[0] = C0; sh[1] = -C1 * dy; sh[2] = C1 * dz; sh[3] = -C1 * dx;
                    sh[4] = C20 * xy; sh[5] = C21 * yz; sh[6] = C22 * (2.0f * zz - xx - yy); sh[7] = C23 * xz; sh[8] = C24 * (xx - yy);
                    float o[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        float acc = 0.f;
#pragma unroll
                        for (int b = 0; b < 9; ++b) acc = fmaf(sh[b], Fe[(c * 9 + b) * kXld + s], acc);
                        o[c] = fmaxf(acc + 0.5f, 0.f);
                    }
                    tr = o[0]; tg = o[1]; tb = o[2];
                }
            } else if (h == 0) { tr = Fe[0 * kXld + s]; tg = Fe[1 * kXld + s]; tb = Fe[2 * kXld + s]; }   // T2N_SHADE_RGB
            // the entries' own order, one fused multiply-add per entry and channel: k_composite's sum
            for (int j = 0; j < cnt; ++j) {
                const float wj = __shfl(wq, j), rj = __shfl(tr, j), gj = __shfl(tg, j), bj = __shfl(tb, j);
                cr = fmaf(wj, rj, cr); cg = fmaf(wj, gj, cg); cb = fmaf(wj, bj, cb);
            }
            wave_lds_sync();   // Fe / Hs reads done
            // queue: entries [cnt, q) move to the front
            const int rest = q - cnt;
            float4 mv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) mv[t] = (lane + 64 * t < rest) ? Q[cnt + lane + 64 * t] : make_float4(0.f, 0.f, 0.f, 0.f);
            wave_lds_sync();
#pragma unroll
            for (int t = 0; t < 2; ++t) if (lane + 64 * t < rest) Q[lane + 64 * t] = mv[t];
            wave_lds_sync();
            q = rest;
        };
        for (int k0 = 0; k0 < staged; k0 += 32) {
            const int cnt = staged - k0 < 32 ? staged - k0 : 32;
            if (lane < cnt) Q[q + lane] = a.scratch[(size_t)r * a.cap + k0 + lane];
            q += cnt;
            wave_lds_sync();
            if (q >= 32) flush(32);
        }
        if (napp > staged) {
            const Ray ray = load_ray(F, rp, a.ray_stride);
            const int end = first + Lw;
            for (int base = from; base < end; base += 64) {
                const int i = base + lane;
                const float w = i < end ? a.wbuf[r * N + i] : 0.f;
                const bool m = (i < end) & (w > F.thres);
                const unsigned long long bal = __ballot(m);
                if (m) {
                    const int pre = (int)__popcll(bal & ((1ull << lane) - 1ull));
                    const float z = sample_z<false>(F, ray, i, 0.f);
                    float xn, yn, zn;
                    sample_point<false>(F, ray, z, xn, yn, zn);
                    Q[q + pre] = make_float4(xn, yn, zn, w);
                }
                q += (int)__popcll(bal);
                wave_lds_sync();
                while (q >= 32) flush(32);
            }
        }
        if (q > 0) flush(q);
        if (lane == 0) {
            if (a.add_bg) { const float bg = 1.f - a.acc[r]; cr += bg; cg += bg; cb += bg; }
            a.rgb[r * 3 + 0] = fminf(fmaxf(cr, 0.f), 1.f);
            a.rgb[r * 3 + 1] = fminf(fmaxf(cg, 0.f), 1.f);
            a.rgb[r * 3 + 2] = fminf(fmaxf(cb, 0.f), 1.f);
        }
    }
    if (any && lane == 0 && a.stats) a.stats[T2N_STAT_LIST_RETRY] = 1ull;
}

int launch_finish_rays(t2n_field* f, const RenderLaunch& L, const float* spill, const float4* scratch, hipStream_t s) {
    const int cap = L.n_samples / 4 > 0 ? L.n_samples / 4 : 1;     // (launch_march_tiles' staging geometry)
    const unsigned* ovf_count = (const unsigned*)((const char*)scratch + (size_t)L.n_rays * cap * 16);
    FinishArgs a;
    a.F = f->dev;
    a.rays = L.rays; a.ray_stride = L.ray_stride; a.n_samples = L.n_samples;
    a.ray_app = L.ray_app; a.acc = L.acc; a.rgb = L.rgb; a.add_bg = (L.flags & T2N_FLAG_ADD_BG) ? 1 : 0;
    a.wbuf = L.weights ? L.weights : spill; a.scratch = scratch; a.cap = cap;
    a.ovf_count = ovf_count; a.ovf_list = (const int*)((const char*)ovf_count + 256);
    a.stats = (unsigned long long*)L.stats;
    const size_t lds = (size_t)4 * kFinFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_finish_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_finish_rays, dim3(128), dim3(256), lds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// ---- features only (K2a of the default render path): gather + basis_mat -> fp32 feature rows [tile * 32 + sample][32] for the
// sample-stationary head (t2n_mlp_ss.hip), one factor pair at a time. (The first form of this stage, a 144-row X tile per wave
// with two waves per SIMD, ran 0.64 ms per C2 frame against 0.58: DESIGN.md, v23 / v25; it lives in the history, not here.)
// k_app_features holds a 144-row X tile per wave (19 KB: eight waves per CU) and two 18-load items in flight (180 VGPRs): two
// waves per SIMD to hide a gather that sits on the L1's 64 B/clk. Here a wave walks the three plane/line pairs in turn: gather
// pair k (6 taps per item, 48 rows of X: 6.3 KB), three basis chunks on the matrix cores, next pair - 128 VGPRs, four waves
// per SIMD. The sample positions are parked in LDS once per tile (the per-pair items re-derive their taps from them). Same
// per-channel arithmetic, same chunk order, same three products per chunk: the feature rows are bit-identical.
constexpr int kPairRows = 48;
constexpr int kPairFloats = kPairRows * kXld + 32 * 8;   // X[48][33] + 32 x (axis taps, weight), per wave

// The three axis taps of a tile's samples are computed ONCE per tile (lane = sample) and parked in LDS as (low tap index, high tap
// weight) per axis: the 12 (or 6) lanes that share a sample re-derive the full taps from them in six instructions instead of
// ~45 each, 18 times per lane and tile (the positions are samples that passed the box test: high tap = min(low + 1, size - 1),
// low weight = 1 - high weight, exactly what axis_taps computes there).
__device__ __forceinline__ Axes3 parked_axes(const FactorSet& S, const float4* __restrict__ P, int s_) {
    const float4 a = P[2 * s_], b = P[2 * s_ + 1];
    Axes3 A;
    A.a[0].i0 = __float_as_int(a.x); A.a[1].i0 = __float_as_int(a.y); A.a[2].i0 = __float_as_int(a.z);
    A.a[0].i1 = min(A.a[0].i0 + 1, S.W[0] - 1); A.a[1].i1 = min(A.a[1].i0 + 1, S.H[0] - 1); A.a[2].i1 = min(A.a[2].i0 + 1, S.H[1] - 1);
    A.a[0].w1 = b.x; A.a[1].w1 = b.y; A.a[2].w1 = b.z;
    A.a[0].w0 = 1.f - b.x; A.a[1].w0 = 1.f - b.y; A.a[2].w0 = 1.f - b.z;
    return A;
}

#ifndef T2N_SHARE_TAPS
#define T2N_SHARE_TAPS 1
#endif
constexpr bool kShareTaps = T2N_SHARE_TAPS != 0;
template <int K, bool HALF>
__device__ __forceinline__ void gather_pair(const FactorSet& S, float* __restrict__ X, const float4* __restrict__ P, int lane, unsigned nlive) {
    // the item -> (sample, quad) maps and their LDS addresses are loop-invariant per lane: left visible, hipcc hoists ~40 of them
    // out of the tile loop and spills them; behind an opaque lane index they are recomputed (a few VALU ops per item)
    asm volatile("" : "+v"(lane));
    if constexpr (HALF) {
        auto issue = [&](int it, OctTaps& t, int& s_, int& o_) {
            const int item = it * 64 + lane;
            s_ = item / 6; o_ = item - s_ * 6;
            const Axes3 A = parked_axes(S, P, s_);
            issue_octets_ax<K>(S, o_, A, t);
        };
        auto consume = [&](const OctTaps& t, int s_, int o_) {
            const bool live = (unsigned)s_ < nlive;
            const unsigned nw[4] = {t.nw.x, t.nw.y, t.nw.z, t.nw.w}, ne[4] = {t.ne.x, t.ne.y, t.ne.z, t.ne.w};
            const unsigned sw[4] = {t.sw.x, t.sw.y, t.sw.z, t.sw.w}, se[4] = {t.se.x, t.se.y, t.se.z, t.se.w};
            const unsigned l0[4] = {t.l0.x, t.l0.y, t.l0.z, t.l0.w}, l1[4] = {t.l1.x, t.l1.y, t.l1.z, t.l1.w};
            float* dst = X + (size_t)(o_ * 8) * kXld + s_;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                float pa = bf16_lo(nw[d]) * t.wnw, pb = bf16_hi(nw[d]) * t.wnw;
                pa = fmaf(bf16_lo(ne[d]), t.wne, pa); pb = fmaf(bf16_hi(ne[d]), t.wne, pb);
                pa = fmaf(bf16_lo(sw[d]), t.wsw, pa); pb = fmaf(bf16_hi(sw[d]), t.wsw, pb);
                pa = fmaf(bf16_lo(se[d]), t.wse, pa); pb = fmaf(bf16_hi(se[d]), t.wse, pb);
                float la = bf16_lo(l0[d]) * t.wl0, lb = bf16_hi(l0[d]) * t.wl0;
                la = fmaf(bf16_lo(l1[d]), t.wl1, la); lb = fmaf(bf16_hi(l1[d]), t.wl1, lb);
                dst[(2 * d) * kXld] = live ? pa * la : 0.f;
                dst[(2 * d + 1) * kXld] = live ? pb * lb : 0.f;
            }
        };
        OctTaps t0, t1;
        int s0, o0, s1, o1;
        issue(0, t0, s0, o0);
        issue(1, t1, s1, o1);
        __builtin_amdgcn_sched_barrier(0);
        consume(t0, s0, o0);
        __builtin_amdgcn_sched_barrier(0);
        issue(2, t0, s0, o0);
        __builtin_amdgcn_sched_barrier(0);
        consume(t1, s1, o1);
        consume(t0, s0, o0);
    } else if constexpr (kShareTaps) {
        // items in PAIRS: lane -> (entries 2j and 2j+1, quad q), three pairs per lane; the second entry loads only what changed
        int ca, cb, cl;     // the first entry's cell (plane column, plane row, line row)
        auto first = [&](int p, QuadTaps& t, int& s_, int& q_, int& ca_, int& cb_, int& cl_) {
            const int item = p * 64 + lane;
            const int j = item / 12;
            q_ = item - (int)__umul24((unsigned)j, 12u); s_ = 2 * j;
            const Axes3 A = parked_axes(S, P, s_);
            ca_ = A.a[mat0(K)].i0; cb_ = A.a[mat1(K)].i0; cl_ = A.a[vecm(K)].i0;
            issue_taps_ax<K, false>(S, 12, q_, A, t);
        };
        auto second = [&](QuadTaps& t, int& s_, int q_, int ca_, int cb_, int cl_) {
            s_ += 1;
            const Axes3 A = parked_axes(S, P, s_);
            const bool pc = (A.a[mat0(K)].i0 != ca_) || (A.a[mat1(K)].i0 != cb_), lc = A.a[vecm(K)].i0 != cl_;
            issue_taps_ax_changed<K>(S, 12, q_, A, pc, lc, t);
        };
        auto consume = [&](const QuadTaps& t, int s_, int q_) {
            // (round 6: the same 28 multiply / fma per item as 14 packed-fp32 operations — v_pk_mul_f32 / v_pk_fma_f32 — took 15.6 % of the
            // tile loop's VALU instructions away and NOTHING off the kernel's time: 0.572 against 0.574 ms, it waits for the texture
            // addresser; that build's rows also differed from run to run at the 1e-6 level for a reason not found, so it is not kept:
            // profiles/round6_features_count_table.txt)
            const float4 pv = taps_plane(t), l = taps_line(t);
            float4 v = make_float4(pv.x * l.x, pv.y * l.y, pv.z * l.z, pv.w * l.w);
            if (!((unsigned)s_ < nlive)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float* dst = X + (size_t)(q_ * 4) * kXld + s_;
            dst[0] = v.x; dst[kXld] = v.y; dst[2 * kXld] = v.z; dst[3 * kXld] = v.w;
        };
        QuadTaps t0, t1;
        int s0, q0, s1, q1, ca1, cb1, cl1;
#define T2N_FENCE __builtin_amdgcn_sched_barrier(0)
        first(0, t0, s0, q0, ca, cb, cl);
        first(1, t1, s1, q1, ca1, cb1, cl1);
        T2N_FENCE;
        consume(t0, s0, q0); T2N_FENCE; second(t0, s0, q0, ca, cb, cl); T2N_FENCE;
        consume(t1, s1, q1); T2N_FENCE; second(t1, s1, q1, ca1, cb1, cl1); T2N_FENCE;
        consume(t0, s0, q0); T2N_FENCE; first(2, t0, s0, q0, ca, cb, cl); T2N_FENCE;
        consume(t1, s1, q1); T2N_FENCE;
        consume(t0, s0, q0); T2N_FENCE; second(t0, s0, q0, ca, cb, cl); T2N_FENCE;
        consume(t0, s0, q0);
#undef T2N_FENCE
    } else {
        auto issue = [&](int it, QuadTaps& t, int& s_, int& q_) {
            const int item = it * 64 + lane;
            s_ = item / 12; q_ = item - s_ * 12;
            const Axes3 A = parked_axes(S, P, s_);
            issue_taps_ax<K, false>(S, 12, q_, A, t);
        };
        auto consume = [&](const QuadTaps& t, int s_, int q_) {
            const float4 pv = taps_plane(t), l = taps_line(t);
            float4 v = make_float4(pv.x * l.x, pv.y * l.y, pv.z * l.z, pv.w * l.w);
            if (!((unsigned)s_ < nlive)) v = make_float4(0.f, 0.f, 0.f, 0.f);
            float* dst = X + (size_t)(q_ * 4) * kXld + s_;
            dst[0] = v.x; dst[kXld] = v.y; dst[2 * kXld] = v.z; dst[3 * kXld] = v.w;
        };
        QuadTaps t0, t1;
        int s0, q0, s1, q1;
        issue(0, t0, s0, q0);
        issue(1, t1, s1, q1);
        // (fenced: unfenced, hipcc hoists all six items' loads to the top and needs 227 VGPRs)
        __builtin_amdgcn_sched_barrier(0);
#define T2N_FENCE __builtin_amdgcn_sched_barrier(0)
        consume(t0, s0, q0); T2N_FENCE; issue(2, t0, s0, q0); T2N_FENCE;
        consume(t1, s1, q1); T2N_FENCE; issue(3, t1, s1, q1); T2N_FENCE;
        consume(t0, s0, q0); T2N_FENCE; issue(4, t0, s0, q0); T2N_FENCE;
        consume(t1, s1, q1); T2N_FENCE; issue(5, t1, s1, q1); T2N_FENCE;
        consume(t0, s0, q0);
        consume(t1, s1, q1);
#undef T2N_FENCE
    }
}

// basis chunks 3K .. 3K+2 (16 K-values each) of f16_stream's operand layout, from X[48][kXld]; the A operands (basis_mat as hi / lo f16
// halves, 18 KB) sit in the workgroup's LDS: streamed from L1 they were 18 of a tile's 137 load instructions, on a kernel that is bound
// by the L1's 64 B/clk
template <int K>
__device__ __forceinline__ void pair_basis(f32x16& acc, const uint4* __restrict__ Wl, const float* __restrict__ Xs, int h, int lane, float& amax) {
    const uint4* __restrict__ ap = Wl + lane + (3 * K) * 2 * 64;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float x[8];
        const float* p = Xs + (size_t)(16 * c + 8 * h) * kXld;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e * kXld];
        h8 bhi, blo;
        split8(x, bhi, blo, amax);
        const h8 ahi = __builtin_bit_cast(h8, ap[(c * 2) * 64]), alo = __builtin_bit_cast(h8, ap[(c * 2 + 1) * 64]);
        acc = mfma16(ahi, bhi, acc);
        acc = mfma16(ahi, blo, acc);
        acc = mfma16(alo, bhi, acc);
        asm volatile("" : "+v"(amax));   // the range tracking of this chunk is due here (deferred to the loop's end, its 8 values are spilled)
    }
}

constexpr int kPairWaves = 8;                               // waves per workgroup (one copy of the basis operands per workgroup)
constexpr int kPairBasisVec = kBasisChunksReal * 2 * 64;    // uint4 elements of the nine real chunks

template <bool HALF>
__global__ __launch_bounds__(64 * kPairWaves) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_app_features_p(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    uint4* __restrict__ Wl = reinterpret_cast<uint4*>(smem);
    float* __restrict__ X = smem + kPairBasisVec * 4 + (size_t)wid * kPairFloats;
    float4* __restrict__ P = reinterpret_cast<float4*>(X + kPairRows * kXld);
    const FieldDev& F = a.F;
    for (int i = threadIdx.x; i < kPairBasisVec; i += 64 * kPairWaves) Wl[i] = F.basisH[i];
    __syncthreads();   // the only workgroup barrier: from here on the waves run on their own
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
    unsigned ntiles = __shfl(incl, a.nlists - 1);
    if (ntiles > a.tile_hi) ntiles = a.tile_hi;
    if (a.split_unsafe && (*a.split_unsafe & kUnsafeBasis)) {   // basis_mat weights beyond the fixed pre-scale's range: the launch is redone on the exact path
        if (a.range_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.range_flag, 1u);
        return;
    }
    const unsigned wave_stride = gridDim.x * (unsigned)kPairWaves;
    float amax = 0.f;
    // tile -> (first entry, live entries) and the lane's position entry (lanes 0..31; dead entries sit at the volume centre and
    // store zeros); the next tile's positions are fetched while this tile is processed
    auto locate = [&](unsigned tile, unsigned& base, unsigned& nlive, float4& mine) {
        mine = make_float4(0.f, 0.f, 0.f, 0.f);
        base = 0u; nlive = 0u;
        if (tile >= ntiles) return;   // wave-uniform
        const int li = (int)__popcll(__ballot((lane < a.nlists) & (incl <= tile)));
        const unsigned before = li ? __shfl(incl, li - 1) : 0u;
        const unsigned lbase = (unsigned)li * a.list_cap;
        base = lbase + (tile - before) * 32u;
        const unsigned count = lbase + __shfl(cnt_l, li);
        nlive = count - base < 32u ? count - base : 32u;
        if (lane < 32 && (unsigned)lane < nlive) mine = a.app_pos[base + (unsigned)lane];
    };
    unsigned base, nlive, nbase, nnlive;
    float4 mine, nmine;
    locate(blockIdx.x * (unsigned)kPairWaves + wid, base, nlive, mine);
    for (unsigned tile = blockIdx.x * (unsigned)kPairWaves + wid; tile < ntiles; tile += wave_stride, base = nbase, nlive = nnlive, mine = nmine) {
        asm volatile("; FEATP_MARK tile_begin");   // (tools/featp_count_table.py: the tile loop's instructions by mnemonic)
        if (lane < 32) {   // (dead entries sit at the volume centre: in the box like every list entry)
            const Axes3 A = sample_axes_inbox(F.app, mine.x, mine.y, mine.z);
            // list slots that were reserved but never written (a budgeted launch that overflowed: this kernel is already queued when
            // the host learns of it) hold stale workspace bytes: whatever they decode to, the taps stay inside the planes
            // (parked_axes: high tap = min(low + 1, size - 1)); such rows belong to no ray's slice and are never composited
            const int ix = min(max(A.a[0].i0, 0), F.app.W[0] - 1), iy = min(max(A.a[1].i0, 0), F.app.H[0] - 1),
                      iz = min(max(A.a[2].i0, 0), F.app.H[1] - 1);
            P[2 * lane] = make_float4(__int_as_float(ix), __int_as_float(iy), __int_as_float(iz), mine.w);
            P[2 * lane + 1] = make_float4(A.a[0].w1, A.a[1].w1, A.a[2].w1, 0.f);
        }
        locate(tile + wave_stride, nbase, nnlive, nmine);
        wave_lds_sync();
        f32x16 acc = {0};
        gather_pair<0, HALF>(F.app, X, P, lane, nlive);
        wave_lds_sync();
        pair_basis<0>(acc, Wl, X + s, h, lane, amax);
        wave_lds_sync();
        gather_pair<1, HALF>(F.app, X, P, lane, nlive);
        wave_lds_sync();
        pair_basis<1>(acc, Wl, X + s, h, lane, amax);
        wave_lds_sync();
        gather_pair<2, HALF>(F.app, X, P, lane, nlive);
        wave_lds_sync();
        pair_basis<2>(acc, Wl, X + s, h, lane, amax);
        const f32x16 accb = acc * kWUnscale;
        // lane (s, h) register v holds feature (v & 3) + 8 (v >> 2) + 4 h: four float4 stores per lane. Column 27 (a zero of the
        // padded basis) carries the entry's compositing weight to the head, which hands it on in app_rgb.w
        float* __restrict__ row = a.ctx.feat32 + ((size_t)tile * 32 + s) * 32 + 4 * h;
        const float wgt = (h == 0 && (unsigned)s < nlive) ? P[2 * s].w : 0.f;   // lane (s, 0) holds columns 24..27 in registers 12..15
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(row + 8 * g) = make_float4(accb[4 * g], accb[4 * g + 1], accb[4 * g + 2], (g == 3 && h == 0) ? wgt : accb[4 * g + 3]);
        wave_lds_sync();   // X and P reads done before the next tile overwrites them
        asm volatile("; FEATP_MARK tile_end");
    }
    if (a.range_flag && __any(!(amax <= 60000.f)) && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- block-cooperative variant (default render path, split-f16 MLP head) ---------------------------------------------
// k_shade streams every weight chunk from L2 once per wave: 292 KB per 32-sample tile, ~39 GB per 800x800 frame, which is
// the cache fabric's whole budget for the kernel. Here the four waves of a block walk their four tiles in lockstep and
// share the layer-0/1 weight stream through a two-deep LDS ring: each thread fetches 32 B of the next-but-two chunk into
// registers, drops it into the ring one chunk ahead, and every wave reads its A operands from LDS (ds_read_b128, lane-linear,
// conflict-free). Weight traffic from L2 drops 4x for the two big layers; one workgroup barrier per chunk.
// LDS map (80 KB per block, two blocks per CU): wave w owns floats [w*5120, (w+1)*5120): [0, 4096) activation tile
// (X[144][33] spills into the tail during gather/basis only), [4096, 5120) slice w of the ring (ring buffer b = slices 2b, 2b+1).
constexpr int kWaveFloats = 5120, kRingOff = 4096, kHld = 32;

// Weight ring: at chunk c every wave reads its A operands from ring buffer c&1, the block pushes chunk c+1 (fetched two
// steps earlier into registers) into the other buffer and fetches chunk c+3. One __syncthreads() per chunk, by the caller.
struct CoopRing {
    float* smem; const uint4* wp; int tid, lane, last;   // last: highest chunk index that may be fetched
    uint4 n0, n1, nn0, nn1;                              // chunks c+1 and c+2 in flight
    __device__ __forceinline__ uint4* slot(int b, int i) const {
        return reinterpret_cast<uint4*>(smem + (size_t)(2 * b + (i >> 8)) * kWaveFloats + kRingOff) + (i & 255);
    }
    // buffer loads: descriptor in SGPRs, the chunk as a SCALAR byte offset, the thread as one 32-bit VGPR offset — flat
    // loads made the compiler keep a dozen 64-bit per-lane chunk pointers alive across the whole kernel (and spill them)
    __device__ __forceinline__ void gload(uint4& ga, uint4& gb, int c) const {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        c = c < last ? c : last;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wp), 0, 0x7fffffff, 0x00020000);
        const int soff = c * 8192;
        const u4v a = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, soff, 0);
        const u4v b = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, soff + 4096, 0);
        ga = make_uint4(a[0], a[1], a[2], a[3]); gb = make_uint4(b[0], b[1], b[2], b[3]);
    }
    __device__ __forceinline__ void start() {   // callers guarantee nobody still reads the ring
        gload(n0, n1, 0);
        gload(nn0, nn1, 1);
        *slot(0, tid) = n0; *slot(0, 256 + tid) = n1;
        n0 = nn0; n1 = nn1;
        gload(nn0, nn1, 2);
        __syncthreads();
    }
    __device__ __forceinline__ void advance(int c, uint4 (&A)[8]) {
        const int b = c & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) A[i] = *slot(b, i * 64 + lane);
        *slot(b ^ 1, tid) = n0; *slot(b ^ 1, 256 + tid) = n1;
        n0 = nn0; n1 = nn1;
        gload(nn0, nn1, c + 3);
    }
};

// One pipelined chunk: the 12 MFMAs of chunk c on the B operand prepared during the previous step, with the VALU work that
// builds chunk c+1's B operand (`prep`) in the same basic block so the two interleave; the four accumulators are visited
// round-robin so consecutive MFMAs never chain on one accumulator.
template <bool TRACK, class Prep>
__device__ __forceinline__ void coop_step(CoopRing& R, int c, f32x16 (&acc)[4], h8& bhi, h8& blo, float& amax, Prep&& prep) {
    uint4 A[8];
    R.advance(c, A);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = mfma16(__builtin_bit_cast(h8, A[2 * m]), bhi, acc[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = mfma16(__builtin_bit_cast(h8, A[2 * m]), blo, acc[m]);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = mfma16(__builtin_bit_cast(h8, A[2 * m + 1]), bhi, acc[m]);
    float x[8];
    prep(x);
    h8 nhi, nlo;
    split8<TRACK>(x, nhi, nlo, amax);
    bhi = nhi; blo = nlo;
    __syncthreads();
}

// Layer-0 B operands in PeChunk's order, unrolled over the three-chunk period of (four pairs per chunk, six octaves per
// feature): chunk types A = octaves 0-3 of feature fa, B = octaves 4-5 of fa + 0-1 of fb = fa+1, C = octaves 2-5 of fb.
struct PePipe {
    const float* feh;   // Fe + s + 14 h kXld
    float sn, cs, v;
    // sin / cos by v_sin_f32 / v_cos_f32 (revolutions) on the fraction of arg / (2 pi), 1 / (2 pi) as a two-constant product: max
    // abs error 4.2e-7 (tools/experiments/hw_sincos.hip), ~9 instructions against ~30 for the polynomial
    __device__ __forceinline__ void fresh(float arg) {
        const float C1 = 0.15915494309189535f, C2 = (float)(0.15915494309189533576888 - (double)C1);
        const float th = arg * C1;
        const float tl = fmaf(arg, C1, -th) + arg * C2;
        const float t = __builtin_amdgcn_fractf(th) + tl;
        sn = __builtin_amdgcn_sinf(t);
        cs = __builtin_amdgcn_cosf(t);
    }
    __device__ __forceinline__ void put(float (&x)[8], int p) { x[2 * p] = sn; x[2 * p + 1] = cs; }
    __device__ __forceinline__ void type_a(int fa, float (&x)[8]) {
        v = feh[(size_t)fa * kXld];
        fresh(v); put(x, 0);
        pe_double(sn, cs); put(x, 1);
        pe_double(sn, cs); put(x, 2);
        fresh(v * 8.f); put(x, 3);
    }
    __device__ __forceinline__ void type_b(int fa, float (&x)[8]) {
        pe_double(sn, cs); put(x, 0);
        pe_double(sn, cs); put(x, 1);
        v = feh[(size_t)(fa + 1) * kXld];
        fresh(v); put(x, 2);
        pe_double(sn, cs); put(x, 3);
    }
    __device__ __forceinline__ void type_c(float (&x)[8]) {
        pe_double(sn, cs); put(x, 0);
        fresh(v * 8.f); put(x, 1);
        pe_double(sn, cs); put(x, 2);
        pe_double(sn, cs); put(x, 3);
    }
};

__device__ __forceinline__ void coop_layer0(f32x16 (&acc)[4], const uint4* __restrict__ w0H, const float* __restrict__ fe_s, int h,
                                            float* __restrict__ smem, int tid, int lane) {
    CoopRing R{smem, w0H, tid, lane, kL0ChunksReal};   // chunk 23 is the zero pad
    R.start();
    auto raw = [&](int c, float (&x)[8]) {
        const float* p = fe_s + (size_t)(16 * c + 8 * h) * kXld;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e * kXld];
    };
    PePipe P{fe_s + (size_t)14 * h * kXld, 0.f, 1.f, 0.f};
    h8 bhi, blo;
    float unused = 0.f;   // range: the raw features are checked by the caller, the encoding is bounded by 1
    {
        float x[8];
        raw(0, x);
        split8<false>(x, bhi, blo, unused);
    }
    coop_step<false>(R, 0, acc, bhi, blo, unused, [&](float (&x)[8]) { raw(1, x); });
    coop_step<false>(R, 1, acc, bhi, blo, unused, [&](float (&x)[8]) { P.type_a(0, x); });
#pragma unroll 1
    for (int it = 0; it < 7; ++it) {   // chunks 2+3it .. 4+3it = features 2it, 2it+1 of each half
        const int c = 2 + 3 * it;
        coop_step<false>(R, c, acc, bhi, blo, unused, [&](float (&x)[8]) { P.type_b(2 * it, x); });
        coop_step<false>(R, c + 1, acc, bhi, blo, unused, [&](float (&x)[8]) { P.type_c(x); });
        // the last trip prepares a chunk past the end from zero feature rows (<= row 28 + 14 < 32 rows + pad): unused
        coop_step<false>(R, c + 2, acc, bhi, blo, unused, [&](float (&x)[8]) { P.type_a(it < 6 ? 2 * it + 2 : 0, x); });
    }
}

// Same contraction without the software pipeline, PE through PeChunk (libm sincos wherever an argument leaves the fast
// routine's range): the path for tiles with |feature| > 1024. Same number of barriers as coop_layer0.
__device__ __forceinline__ void coop_layer0_plain(f32x16 (&acc)[4], const uint4* __restrict__ w0H, const float* __restrict__ fe_s, int h,
                                                  float* __restrict__ smem, int tid, int lane) {
    CoopRing R{smem, w0H, tid, lane, kL0ChunksReal};
    R.start();
    PeChunk bf{fe_s, h, 0.f, 1.f};
#pragma unroll 1
    for (int c = 0; c < kL0ChunksReal; ++c) {
        uint4 A[8];
        R.advance(c, A);
        float x[8];
        bf(c, x);
        h8 bhi, blo;
        float unused = 0.f;
        split8<false>(x, bhi, blo, unused);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const h8 ahi = __builtin_bit_cast(h8, A[2 * m]), alo = __builtin_bit_cast(h8, A[2 * m + 1]);
            acc[m] = mfma16(ahi, bhi, acc[m]);
            acc[m] = mfma16(ahi, blo, acc[m]);
            acc[m] = mfma16(alo, bhi, acc[m]);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void coop_layer1(f32x16 (&acc)[4], const uint4* __restrict__ w1H, const float* __restrict__ hs_s, int h,
                                            float* __restrict__ smem, int tid, int lane, float& amax) {
    CoopRing R{smem, w1H, tid, lane, kL1Chunks};
    R.start();
    auto rows = [&](int c, float (&x)[8]) {
        const float* p = hs_s + (size_t)(16 * c + 8 * h) * kHld;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = p[e * kHld];
    };
    h8 bhi, blo;
    {
        float x[8];
        rows(0, x);
        split8(x, bhi, blo, amax);
    }
#pragma unroll 1
    for (int c = 0; c < kL1Chunks; ++c)
        coop_step<true>(R, c, acc, bhi, blo, amax, [&](float (&x)[8]) { rows(c < kL1Chunks - 1 ? c + 1 : c, x); });
}


// CTX: the activation rows of tiles below ctx_rows are kept (training forward / backward recompute), as k_shade does
template <bool HALF, bool CTX = false>
__global__ __launch_bounds__(256, 2) void k_shade_coop(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int s = lane & 31, h = lane >> 5;
    float* __restrict__ X = smem + (size_t)wid * kWaveFloats;
    float* __restrict__ Fe = X;
    float* __restrict__ Hs = X;
    const FieldDev& F = a.F;
    unsigned cnt_l = 0;
    if (lane < a.nlists) {
        cnt_l = a.counters ? a.counters[lane * kCounterStride] : a.count_max;
        if (a.counters && cnt_l > a.list_cap) cnt_l = a.list_cap;
    }
    unsigned incl = (cnt_l + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    unsigned ntiles = __shfl(incl, a.nlists - 1);
    if (ntiles > a.tile_hi) ntiles = a.tile_hi;
    if (a.split_unsafe && *a.split_unsafe) {   // weights beyond the fixed pre-scale's range: the exact launch behind this one does the work
        if (a.range_flag && a.tile_lo < ntiles && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.range_flag, 1u);
        return;
    }
    const unsigned block_stride = gridDim.x * 4u;
    float amax = 0.f;   // largest magnitude handed to an f16 convert: past the f16 range the launch is redone exactly

    // the block's four waves take tiles tile0 .. tile0+3 together; a wave past the end runs an all-dead tile
    for (unsigned tile0 = a.tile_lo + blockIdx.x * 4u; tile0 < ntiles; tile0 += block_stride) {
        const unsigned tile = tile0 + wid;
        unsigned base = 0, count = 0;
        if (tile < ntiles) {
            const int li = (int)__popcll(__ballot((lane < a.nlists) & (incl <= tile)));
            const unsigned before = li ? __shfl(incl, li - 1) : 0u;
            const unsigned lbase = (unsigned)li * a.list_cap;
            base = lbase + (tile - before) * 32u;
            count = lbase + __shfl(cnt_l, li);
        }
        [[maybe_unused]] const unsigned row0 = tile * 32u;
        [[maybe_unused]] const bool keep_rows = CTX && tile < ntiles && row0 + 32u <= a.ctx_rows;
        [[maybe_unused]] const ShadeCtx cx = keep_rows ? a.ctx : ShadeCtx{nullptr, nullptr, nullptr, nullptr};
        if constexpr (CTX) gather_all<HALF>(F.app, X, lane, a.app_pos, a.xyz, base, count, cx.x144, row0);
        else gather_all<HALF>(F.app, X, lane, a.app_pos, a.xyz, base, count, nullptr, 0);
        wave_lds_sync();

        f32x16 accb1[1] = {{0}};
        {
            LdsChunk bf{X + s, h, kBasisChunksReal};
            f16_stream<1>(accb1, F.basisH, lane, kBasisChunks, bf, amax);
            accb1[0] *= kWUnscale;
        }
        const f32x16 accb = accb1[0];
        __syncthreads();   // every wave is done with X (whose tail overlaps the ring slices)
#pragma unroll
        for (int v = 0; v < 16; ++v) Fe[unit_of(v, h) * kXld + s] = accb[v];
        wave_lds_sync();

        const unsigned idx = base + (unsigned)s;
        const bool live = idx < count;
        if constexpr (CTX) {
            if (cx.feat32) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(cx.feat32 + (size_t)(row0 + s) * 32 + 8 * g + 4 * h) =
                        make_float4(accb[4 * g], accb[4 * g + 1], accb[4 * g + 2], accb[4 * g + 3]);
            }
        }
        if (a.feat_out && live) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int fidx = unit_of(v, h);
                if (fidx < F.app_dim) a.feat_out[(size_t)idx * F.app_dim + fidx] = accb[v];
            }
        }

        // ---- layer 0 ---------------------------------------------------------------------------------------------------
        f32x16 acc0[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc0[m] = bias_init(F.biasH, m, h);
        {
            float fmax_ = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) fmax_ = fmaxf(fmax_, fabsf(accb[v]));
            amax = fmaxf(amax, fmax_);   // layer 0's raw-feature chunks (its encoding chunks are bounded by 1)
            // |feature| * 2^3 beyond the fast sincos' range (never, for a trained field): unpipelined libm path
            if (__builtin_expect(__any(fmax_ * 8.f > 8192.f) != 0, 0)) coop_layer0_plain(acc0, F.w0H, Fe + s, h, smem, tid, lane);
            else coop_layer0(acc0, F.w0H, Fe + s, h, smem, tid, lane);
        }
        // the stream's last barrier also orders every wave's Fe reads before the H writes below
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
            for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kHld + s] = fmaxf(acc0[ms][v] * kWUnscale, 0.f);
        }
        if constexpr (CTX) {
            if (cx.h0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(cx.h0 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc0[ms][4 * g] * kWUnscale, 0.f), fmaxf(acc0[ms][4 * g + 1] * kWUnscale, 0.f),
                                        fmaxf(acc0[ms][4 * g + 2] * kWUnscale, 0.f), fmaxf(acc0[ms][4 * g + 3] * kWUnscale, 0.f));
            }
        }
        wave_lds_sync();
        // ---- layer 1 ---------------------------------------------------------------------------------------------------
        f32x16 acc1[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc1[m] = bias_init(F.biasH + 128, m, h);
        coop_layer1(acc1, F.w1H, Hs + s, h, smem, tid, lane, amax);
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
            for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kHld + s] = fmaxf(acc1[ms][v] * kWUnscale, 0.f);
        }
        if constexpr (CTX) {
            if (cx.h1) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(cx.h1 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc1[ms][4 * g] * kWUnscale, 0.f), fmaxf(acc1[ms][4 * g + 1] * kWUnscale, 0.f),
                                        fmaxf(acc1[ms][4 * g + 2] * kWUnscale, 0.f), fmaxf(acc1[ms][4 * g + 3] * kWUnscale, 0.f));
            }
        }
        wave_lds_sync();
        // ---- layer 2 (3 live rows), wave-private weight stream -----------------------------------------------------------
        f32x16 acc2a[1];
        acc2a[0] = bias_init(F.biasH + 256, 0, h);
        {
            LdsChunkT<kHld> bf{Hs + s, h, kL2Chunks};
            f16_stream<1>(acc2a, F.w2H, lane, kL2Chunks, bf, amax);
        }
        const f32x16 acc2 = acc2a[0] * kWUnscale;
        if (h == 0 && live) {
            const float cr = sigmoidf_(acc2[0]), cg = sigmoidf_(acc2[1]), cb = sigmoidf_(acc2[2]);
            if (a.app_rgb) a.app_rgb[idx] = make_float4(cr, cg, cb, a.app_pos[idx].w);
            if (a.rgb_out) { a.rgb_out[(size_t)idx * 3] = cr; a.rgb_out[(size_t)idx * 3 + 1] = cg; a.rgb_out[(size_t)idx * 3 + 2] = cb; }
        }
        wave_lds_sync();   // H reads done before the next tile's gather overwrites the tile
    }
    if (a.range_flag && __any(!(amax <= 60000.f)) && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- parameter packing ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* basis; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    float* basisA; float* w0A; float* w1A; float* w2A;
    int app_dim, has_mlp;
    unsigned* clear_word;   // the split packing's range flag, cleared here (k_pack_mlp_h, the next launch, ORs into it)
    // optional (fused training step): max |w| of w2 / w1 / w0 / basis_mat as float bit patterns, atomicMax'ed into absmax[0..3] (zeroed
    // by the launch in front of this one): what the backward chain's operand pack scales by — this kernel reads every weight anyway
    unsigned* absmax;
};
__device__ __forceinline__ void pack_absmax(const PackArgs& a, int idx, float v) {
    if (!a.absmax) return;
    unsigned m = __float_as_uint(v) & 0x7fffffffu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&a.absmax[idx], m);
}

__global__ __launch_bounds__(256) void k_pack_mlp(const PackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid == 0 && a.clear_word) *a.clear_word = 0u;
    const int nb = kBasisSteps * 64, n0 = kL0Steps * 256, n1 = kL1Steps * 256, n2 = kL2Steps * 64;
    if (gid < nb) {
        const int t = gid / 64, l = gid % 64, i = l & 31, h = l >> 5;
        const float v = (t < kBasisReal && i < a.app_dim) ? a.basis[i * kAppK + 2 * t + h] : 0.f;
        a.basisA[gid] = v;
        pack_absmax(a, 3, v);
        return;
    }
    if (!a.has_mlp) return;
    int g = gid - nb;
    if (g < n0) {
        const int t = g / 256, mb = (g / 64) % 4, l = g % 64, i = l & 31, h = l >> 5;
        const int out = mb * 32 + i;
        float v = 0.f;
        if (t < kL0Pairs) {                                   // raw feature steps
            const int f = 2 * t + h;
            if (f < 27) v = a.w0[out * 351 + f];
        } else if (t < kL0Real - 1) {                          // (sin, cos) steps: reference columns 27 + f*6 + q / 189 + f*6 + q
            const int u = t - kL0Pairs, r = u / (2 * kPE), j = u % (2 * kPE), q = j >> 1;
            const int f = 2 * r + h;
            if (f < 27) v = a.w0[out * 351 + ((j & 1) ? 27 + 27 * kPE : 27) + f * kPE + q];
        } else if (t == kL0Real - 1) {
            v = h == 0 ? a.b0[out] : 0.f;
        }
        a.w0A[g] = v;
        pack_absmax(a, 2, t == kL0Real - 1 ? 0.f : v);    // (the bias step is not part of the matrix)
        return;
    }
    g -= n0;
    if (g < n1) {
        const int t = g / 256, mb = (g / 64) % 4, l = g % 64, i = l & 31, h = l >> 5;
        const int out = mb * 32 + i;
        float v = 0.f;
        if (t < 64) v = a.w1[out * 128 + 2 * t + h];
        else if (t == 64) v = h == 0 ? a.b1[out] : 0.f;
        a.w1A[g] = v;
        pack_absmax(a, 1, t == 64 ? 0.f : v);
        return;
    }
    g -= n1;
    if (g < n2) {
        const int t = g / 64, l = g % 64, i = l & 31, h = l >> 5;
        float v = 0.f;
        if (i < 3) {
            if (t < 64) v = a.w2[i * 128 + 2 * t + h];
            else if (t == 64) v = h == 0 ? a.b2[i] : 0.f;
        }
        a.w2A[g] = v;
        pack_absmax(a, 0, t == 64 ? 0.f : v);
    }
}

// Split-f16 operand packing: for layer L, chunk c, block m, part p (0 hi, 1 lo), lane l: 8 halves = W[m*32 + (l&31)][k],
// k = 16c + 8(l>>5) + e, scaled by 2^8. K orders: basis/layer1/layer2 natural; layer 0: 32 raw (27 real), then
// (sin, cos) pairs in the per-half-wave order of PeChunk. Biases are stored as fp32 * 2^8 in accumulator order.
__device__ __forceinline__ float l0_weight(const float* w0, int out, int c, int h, int e) {
    if (c < 2) { const int k = 16 * c + 8 * h + e; return k < 27 ? w0[out * 351 + k] : 0.f; }
    const int j = 4 * (c - 2) + (e >> 1), sc = e & 1;
    const int f = 14 * h + j / 6, q = j % 6;
    if (f >= 27) return 0.f;
    return w0[out * 351 + (sc ? 189 : 27) + f * 6 + q];
}
struct PackHArgs {
    const float* basis; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    _Float16* basisH; _Float16* w0H; _Float16* w1H; _Float16* w2H; float* biasH;
    int app_dim, has_mlp;
    unsigned* unsafe;   // raised when a pre-scaled weight leaves the f16 range (the split one-kernel launches then defer to the exact path)
};
__device__ __forceinline__ void store_split(_Float16* dst, float w, int part, unsigned* unsafe, unsigned bit) {
    const float x = w * kWScale;
    if (!(fabsf(x) < 60000.f)) atomicOr(unsafe, bit);
    const _Float16 hi = (_Float16)x;
    *dst = part == 0 ? hi : (_Float16)(x - (float)hi);
}
__global__ __launch_bounds__(256) void k_pack_mlp_h(const PackHArgs a) {
    // one thread per half element; element index -> (layer, chunk, block, part, lane, e)
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nb = (long long)(kBasisChunks + 1) * 1 * 2 * 64 * 8, n0 = (long long)(kL0Chunks + 1) * 4 * 2 * 64 * 8,
                    n1 = (long long)(kL1Chunks + 1) * 4 * 2 * 64 * 8, n2 = (long long)(kL2Chunks + 1) * 1 * 2 * 64 * 8;
    long long g = gid;
    auto decode = [](long long g, int nmb, int& c, int& m, int& part, int& i, int& h, int& e) {
        e = (int)(g & 7); g >>= 3;
        const int l = (int)(g & 63); g >>= 6;
        part = (int)(g & 1); g >>= 1;
        m = (int)(g % nmb); c = (int)(g / nmb);
        i = l & 31; h = l >> 5;
    };
    int c, m, part, i, h, e;
    if (g < nb) {
        decode(g, 1, c, m, part, i, h, e);
        const int k = 16 * c + 8 * h + e;
        store_split(a.basisH + g, (c < kBasisChunksReal && i < a.app_dim) ? a.basis[i * kAppK + k] : 0.f, part, a.unsafe, kUnsafeBasis);
        return;
    }
    g -= nb;
    if (!a.has_mlp) return;
    if (g < n0) {
        decode(g, 4, c, m, part, i, h, e);
        store_split(a.w0H + g, c < kL0ChunksReal ? l0_weight(a.w0, m * 32 + i, c, h, e) : 0.f, part, a.unsafe, kUnsafeHead);
        return;
    }
    g -= n0;
    if (g < n1) {
        decode(g, 4, c, m, part, i, h, e);
        store_split(a.w1H + g, c < kL1Chunks ? a.w1[(m * 32 + i) * 128 + 16 * c + 8 * h + e] : 0.f, part, a.unsafe, kUnsafeHead);
        return;
    }
    g -= n1;
    if (g < n2) {
        decode(g, 1, c, m, part, i, h, e);
        store_split(a.w2H + g, (c < kL2Chunks && i < 3) ? a.w2[i * 128 + 16 * c + 8 * h + e] : 0.f, part, a.unsafe, kUnsafeHead);
        return;
    }
    g -= n2;
    if (g < 128 + 128 + 32) {   // biases in accumulator order: [layer][m][h][v]
        const int idx = (int)g;
        const int layer = idx < 128 ? 0 : (idx < 256 ? 1 : 2);
        const int r = idx - (layer == 0 ? 0 : (layer == 1 ? 128 : 256));
        const int mm = r / 32, hh = (r / 16) & 1, v = r & 15;
        const int u = mm * 32 + unit_of(v, hh);
        float b = 0.f;
        if (layer == 0) b = a.b0[u];
        else if (layer == 1) b = a.b1[u];
        else if (u < 3) b = a.b2[u];
        a.biasH[idx] = b * kWScale;
    }
}

int launch_pack_mlp(t2n_field* f, const t2n_field_params* p, hipStream_t s, unsigned* absmax_out) {
    const size_t nb = (size_t)kBasisSteps * 64, n0 = (size_t)kL0Steps * 256, n1 = (size_t)kL1Steps * 256, n2 = (size_t)kL2Steps * 64;
    if (!f->buf_mlp) {
        T2N_HIP(hipMalloc((void**)&f->buf_mlp, (nb + n0 + n1 + n2) * sizeof(float)));
        f->dev.basisA = f->buf_mlp;
        f->dev.w0A = f->buf_mlp + nb;
        f->dev.w1A = f->buf_mlp + nb + n0;
        f->dev.w2A = f->buf_mlp + nb + n0 + n1;
    }
    PackArgs a;
    a.basis = p->basis_weight; a.w0 = p->mlp_w0; a.b0 = p->mlp_b0; a.w1 = p->mlp_w1; a.b1 = p->mlp_b1; a.w2 = p->mlp_w2; a.b2 = p->mlp_b2;
    a.basisA = f->buf_mlp; a.w0A = f->buf_mlp + nb; a.w1A = f->buf_mlp + nb + n0; a.w2A = f->buf_mlp + nb + n0 + n1;
    a.app_dim = f->desc.app_dim;
    a.has_mlp = f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW;
    const size_t total = nb + n0 + n1 + n2;
    // split-f16 operands
    const size_t hb = (size_t)(kBasisChunks + 1) * 1 * 2 * 64 * 8, h0 = (size_t)(kL0Chunks + 1) * 4 * 2 * 64 * 8,
                 h1 = (size_t)(kL1Chunks + 1) * 4 * 2 * 64 * 8, h2 = (size_t)(kL2Chunks + 1) * 1 * 2 * 64 * 8;
    if (!f->buf_mlp_h) {
        T2N_HIP(hipMalloc((void**)&f->buf_mlp_h, (hb + h0 + h1 + h2) * sizeof(_Float16) + (288 + 4) * sizeof(float)));
    }
    _Float16* hbase = (_Float16*)f->buf_mlp_h;
    a.clear_word = (unsigned*)((float*)(hbase + hb + h0 + h1 + h2) + 288);
    a.absmax = (a.has_mlp && f->desc.app_dim == 27) ? absmax_out : nullptr;
    hipLaunchKernelGGL(k_pack_mlp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    PackHArgs ha;
    ha.basis = p->basis_weight; ha.w0 = p->mlp_w0; ha.b0 = p->mlp_b0; ha.w1 = p->mlp_w1; ha.b1 = p->mlp_b1; ha.w2 = p->mlp_w2; ha.b2 = p->mlp_b2;
    ha.basisH = hbase; ha.w0H = hbase + hb; ha.w1H = hbase + hb + h0; ha.w2H = hbase + hb + h0 + h1;
    ha.biasH = (float*)(hbase + hb + h0 + h1 + h2);
    ha.app_dim = f->desc.app_dim; ha.has_mlp = a.has_mlp;
    ha.unsafe = (unsigned*)(ha.biasH + 288);
    f->split_unsafe = ha.unsafe;
    f->dev.basisH = (const uint4*)ha.basisH; f->dev.w0H = (const uint4*)ha.w0H; f->dev.w1H = (const uint4*)ha.w1H;
    f->dev.w2H = (const uint4*)ha.w2H; f->dev.biasH = ha.biasH;
    const size_t htotal = hb + h0 + h1 + h2 + 288;
    hipLaunchKernelGGL(k_pack_mlp_h, dim3((unsigned)((htotal + 255) / 256)), dim3(256), 0, s, ha);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

static int shade_grid(unsigned long long count_max) {
    const unsigned long long tiles = (count_max + 31u) / 32u;
    unsigned long long blocks = (tiles + 3u) / 4u;
    const unsigned cap = 256u * 2u;   // 2 workgroups per CU (LDS: 4 x 19 KB each)
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (int)blocks;
}

// The exact-path redo launch behind every split launch does nothing unless the range flag is up: 256 workgroups (one per CU; the kernel
// walks its tiles with a grid stride) instead of 512 halve what the empty launch costs the stream (13 us per train step, 5 per frame).
static dim3 grid_redo(const dim3& g) { return dim3(g.x < 256u ? g.x : 256u); }
constexpr size_t kCoopLds = (size_t)4 * kWaveFloats * sizeof(float);
static bool use_coop(const t2n_field* f) {
    return f->mlp_split && f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW;
}

static bool use_ws(const t2n_field* f) {
    return use_coop(f);
}

int launch_shade_list(t2n_field* f, const float4* app_pos, const int* app_ray, const float* rays, int ray_stride,
                      const unsigned* counters_dev, unsigned list_cap, float4* app_rgb, const ShadeCtx* ctx, hipStream_t s,
                      bool features_only, unsigned ctx_rows, float* feat, unsigned feat_rows, uint64_t* stats, unsigned tile_lo, unsigned tile_hi) {
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    if (ctx) a.ctx = *ctx;
    a.ctx_rows = ctx_rows;
    a.tile_lo = tile_lo; a.tile_hi = tile_hi;
    a.F = f->dev;
    if (features_only) a.F.shading = T2N_SHADE_RGB;   // gather + basis only; the rgb slots get placeholder values the head overwrites
    a.app_pos = app_pos; a.app_ray = app_ray; a.rays = rays; a.ray_stride = ray_stride;
    a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.app_rgb = app_rgb;
    const size_t lds = (size_t)4 * kTileFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade_coop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade_coop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds));
        attr_set = true;
    }
    const dim3 grid(shade_grid((unsigned long long)list_cap * kLists));
    const bool half = f->factor_bf16 && f->dev.app.plane_h[0];
    const unsigned ws_tiles = feat ? feat_rows / 128u * 4u : 0u;
    if (use_ws(f) && !ctx && !features_only && ws_tiles >= 4u) {
        // default render path: K2a gather + basis -> feature rows (k_app_features_p), K2b sample-stationary head (t2n_mlp_ss.hip);
        // tiles past the row capacity take the one-kernel path; a launch that met a value outside the f16 range is redone on the
        // exact fp32 path
        unsigned* flag = const_cast<unsigned*>(counters_dev) + kRangeFlagWord;
        ShadeArgs fa = a;
        fa.F.shading = T2N_SHADE_RGB;
        fa.app_rgb = nullptr;
        fa.ctx = ShadeCtx{nullptr, feat, nullptr, nullptr};
        fa.ctx_rows = ws_tiles * 32u; fa.tile_hi = ws_tiles;
        fa.range_flag = flag; fa.split_unsafe = f->split_unsafe;
        timing_begin(f, T2N_K_APPFEAT, s);
        {
            const size_t lds_p = (size_t)kPairBasisVec * 16 + (size_t)kPairWaves * kPairFloats * sizeof(float);
            static bool attr_p = false;
            if (!attr_p) {
                T2N_HIP(hipFuncSetAttribute((const void*)k_app_features_p<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
                T2N_HIP(hipFuncSetAttribute((const void*)k_app_features_p<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_p));
                attr_p = true;
            }
            const unsigned long long wg = ((unsigned long long)ws_tiles + kPairWaves - 1) / kPairWaves;
            const dim3 grid_p((unsigned)(wg < 512u ? (wg ? wg : 1u) : 512u));   // two workgroups (16 waves) per CU
            if (half) hipLaunchKernelGGL(k_app_features_p<true>, grid_p, dim3(64 * kPairWaves), lds_p, s, fa);
            else hipLaunchKernelGGL(k_app_features_p<false>, grid_p, dim3(64 * kPairWaves), lds_p, s, fa);
        }
        timing_end(f, T2N_K_APPFEAT, s);
        timing_begin(f, T2N_K_SHADE, s);
        const int rc = launch_mlp_ss(f, feat, counters_dev, list_cap, ws_tiles, app_rgb, flag, s);
        if (rc) return rc;
        ShadeArgs oa = a;
        oa.tile_lo = ws_tiles; oa.range_flag = flag; oa.split_unsafe = f->split_unsafe;
        // tiles past the feature-row capacity (none in steady state: the rows are sized from the list budget): a small persistent
        // grid — 512 workgroups that find nothing cost the frame 8 us, 64 cost it the launch
        const dim3 grid_o(grid.x < 64u ? grid.x : 64u);
        if (half) hipLaunchKernelGGL(k_shade_coop<true>, grid_o, dim3(256), kCoopLds, s, oa);
        else hipLaunchKernelGGL(k_shade_coop<false>, grid_o, dim3(256), kCoopLds, s, oa);
        ShadeArgs ra = a;   // every tile of the launch, the one-kernel path's included
        ra.run_if_nonzero = counters_dev + kRangeFlagWord; ra.stats = (unsigned long long*)stats;
        hipLaunchKernelGGL(k_shade<false>, grid_redo(grid), dim3(256), lds, s, ra);
        timing_end(f, T2N_K_SHADE, s);
        T2N_HIP(hipGetLastError());
        return T2N_OK;
    }
    // one-kernel paths (training forward / backward recompute with kept activations, overflow-free A/B builds): the split launch
    // tracks the range of everything it converts to f16; behind it an exact launch that runs only when the flag (or the
    // upload-time weight flag) is up and rewrites every entry and activation row of the launch
    unsigned* flag = const_cast<unsigned*>(counters_dev) + kRangeFlagWord;
    a.range_flag = flag; a.split_unsafe = f->split_unsafe;
    timing_begin(f, T2N_K_SHADE, s);
    if (use_coop(f) && !ctx && half) hipLaunchKernelGGL(k_shade_coop<true>, grid, dim3(256), kCoopLds, s, a);
    else if (use_coop(f) && !ctx) hipLaunchKernelGGL(k_shade_coop<false>, grid, dim3(256), kCoopLds, s, a);
    else if (use_coop(f) && ctx && !half && !features_only) {   // kept activation rows on the cooperative kernel (weights shared through LDS: 106 against 118 us per C3 iteration)
        static bool attr_c = false;
        if (!attr_c) { T2N_HIP(hipFuncSetAttribute((const void*)k_shade_coop<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds)); attr_c = true; }
        hipLaunchKernelGGL((k_shade_coop<false, true>), grid, dim3(256), kCoopLds, s, a);
    }
    else if (f->mlp_split)   /* ctx (backward recompute) too: activations at ~1e-7 relative error */ hipLaunchKernelGGL(k_shade<true>, grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL(k_shade<false>, grid, dim3(256), lds, s, a);
    if (f->mlp_split) {
        ShadeArgs ra = a;
        ra.range_flag = nullptr; ra.split_unsafe = nullptr;
        ra.run_if_nonzero = flag; ra.stats = (unsigned long long*)stats;
        hipLaunchKernelGGL(k_shade<false>, grid_redo(grid), dim3(256), lds, s, ra);
    }
    timing_end(f, T2N_K_SHADE, s);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

using namespace t2n;


extern "C" size_t t2n_shade_workspace_bytes(int64_t n) { (void)n; return 256; }

extern "C" int t2n_shade_at(const t2n_field* fc, const float* xyz_norm, const float* viewdirs, int64_t n, float* app_feat,
                            float* rgb, void* workspace, size_t workspace_bytes, t2n_stream stream) {
    t2n_field* f = const_cast<t2n_field*>(fc);
    if (!f || !xyz_norm || n < 0 || n > 0x7fffffff) { set_error("t2n_shade_at: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_shade_at: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (f->desc.shading == T2N_SHADE_SH && rgb && !viewdirs) { set_error("t2n_shade_at: SH head needs viewdirs"); return T2N_ERR_INVALID; }
    if (head_is_generic(f->desc.shading) && rgb) { set_error("t2n_shade_at: the view-dependent MLP heads are evaluated by the render call only (features are available)"); return T2N_ERR_UNSUPPORTED; }
    if (n == 0) return T2N_OK;
    if (f->mlp_split && (!workspace || workspace_bytes < 4)) { set_error("t2n_shade_at: workspace smaller than t2n_shade_workspace_bytes()"); return T2N_ERR_INVALID; }
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    a.F = f->dev;
    if (head_is_generic(f->desc.shading)) a.F.shading = T2N_SHADE_RGB;   // features only
    a.xyz = xyz_norm; a.viewdirs = viewdirs; a.count_max = (unsigned)n; a.nlists = 1; a.feat_out = app_feat; a.rgb_out = rgb;
    a.tile_lo = 0; a.tile_hi = 0xffffffffu;
    const size_t lds = (size_t)4 * kTileFloats * sizeof(float);
    T2N_HIP(hipFuncSetAttribute((const void*)k_shade<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2N_HIP(hipFuncSetAttribute((const void*)k_shade<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    T2N_HIP(hipFuncSetAttribute((const void*)k_shade_coop<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade_coop<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kCoopLds));
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(shade_grid((unsigned)n));
    if (f->mlp_split) {   // workspace word 0: the launch's f16-range flag (the exact launch behind the split one runs when it is up)
        unsigned* flag = (unsigned*)workspace;
        T2N_HIP(hipMemsetAsync(flag, 0, 4, s));
        a.range_flag = flag; a.split_unsafe = f->split_unsafe;
        if (use_coop(f)) hipLaunchKernelGGL(k_shade_coop<false>, grid, dim3(256), kCoopLds, s, a);
        else hipLaunchKernelGGL(k_shade<true>, grid, dim3(256), lds, s, a);
        ShadeArgs ra = a;
        ra.range_flag = nullptr; ra.split_unsafe = nullptr; ra.run_if_nonzero = flag;
        hipLaunchKernelGGL(k_shade<false>, grid_redo(grid), dim3(256), lds, s, ra);
    } else hipLaunchKernelGGL(k_shade<false>, grid, dim3(256), lds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
