// Weight-stationary appearance head: positional encoding + the 351 -> 128 -> 128 -> 3 MLP + sigmoid for rows of 27 appearance
// features (K2b). The features come from the gather + basis kernel (k_shade in features-only mode, K2a); this kernel is the
// matrix-core part of the appearance stage on the default (split-f16) render path.
//
// Replaces (reference): models/tensorBase.py:11-17 (positional_encoding), :88-109 (MLPRender_Fea_noview.forward).
//
// Mapping (gfx950, wave64): ONE 256-thread workgroup per CU, persistent over groups of 128 samples (4 N-tiles of 32). Wave w
// owns output units 32w .. 32w+31 of both hidden layers and keeps its slice of the layer-0 weights in registers for the whole
// kernel (24 K-chunks x (hi, lo) x 16 B per lane = 192 VGPRs: the A operands of v_mfma_f32_32x32x16_f16 never move again;
// the layer-1 / layer-2 slices sit in LDS). fp32 products are three f16 products of hi/lo splits (x = hi + lo, hi = RTZ_f16(x),
// lo = RTZ_f16(x - hi); the lo*lo term is dropped: ~2^-21 relative), fp32 accumulate. Weights are pre-split and scaled by a
// per-layer power of two chosen from max|W| at upload (k_ws_scales), activations are split where they are produced, and any
// activation beyond the f16 range raises a flag that makes the caller's exact-fp32 kernel redo the launch.
//
// The B operands (samples on N) go through LDS once per sample: the four waves share the encoding work — wave w' encodes
// features 7w' .. 7w'+6 of every sample: sin / cos by v_sin_f32 / v_cos_f32 on the fraction of f * 2^q / (2 pi), 1 / (2 pi) as
// a two-constant product (max abs error 4.2e-7 over |f| <= 3e4, measured: tools/experiments/hw_sincos.hip) — as six K-chunks
// of 16 values, one chunk per pipeline step into a two-slot ring; a step's 48 MFMAs (4 chunks x 4 N-tiles x 3 products) run
// on the chunks encoded during the previous step. Hidden activations return to LDS in the same B layout (relu, unscale,
// split), so layers 1 and 2 are plain LDS -> MFMA streams. ~10 workgroup barriers per 128 samples.
//
// K order of layer 0 (what k_pack_ws mirrors): chunk c = 6 w' + j holds values v = 16 j .. 16 j + 15 of wave w's 96-value
// sequence: v < 84: feature 7w' + v / 12, octave (v % 12) / 2, sin (even v) / cos (odd v); 84 <= v < 91: raw feature
// 7w' + v - 84; the rest zero.
#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

constexpr int kWsNT = 4;                        // N-tiles (32 samples) per group
constexpr int kWsC0 = 24, kWsC1 = 8, kWsC2 = 8;   // K-chunks (16 values) of layers 0 / 1 / 2
constexpr int kWsSlot = 4 * kWsNT * 2 * 64;     // uint4 per ring slot: [producer wave][N-tile][part][lane]
constexpr int kWsH = kWsNT * kWsC1 * 2 * 64;    // uint4 of the hidden-activation tile: [N-tile][chunk][part][lane] (aliases the ring)
constexpr int kWsW1 = 4 * kWsC1 * 2 * 64;       // [wave][chunk][part][lane]
constexpr int kWsW2 = kWsC2 * 2 * 64;
constexpr int kWsBias = 288;                    // floats: [layer][block][h][v] in accumulator order, scaled
constexpr size_t kWsLds = (size_t)(kWsH + kWsW1 + kWsW2) * 16 + kWsBias * 4;
static_assert(2 * kWsSlot == kWsH, "ring and hidden tile share one region");
constexpr float kWsRange = 60000.f;             // |activation| beyond this (f16 max 65504) -> exact-path redo

struct WsArgs {
    const uint4* w0; const uint4* w1; const uint4* w2; const float* bias; const float* inv_scale;   // packed by k_pack_ws
    const float* feat;            // [rows][32] fp32 feature rows (row = tile * 32 + sample)
    const unsigned* counters; unsigned list_cap; int nlists;
    unsigned tile_hi;             // tiles [0, min(ntiles, tile_hi)) are this kernel's; tile_hi is a multiple of 4
    float4* app_rgb;
    unsigned* range_flag;
    float neg1;                   // -1.0f at run time: x - hi stays an FMA with an f16 operand (v_fma_mix_f32, no convert)
};

__device__ __forceinline__ f32x16 ws_mfma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// (x0, x1) -> packed hi halves and packed lo halves
__device__ __forceinline__ void ws_split2(float x0, float x1, float neg1, unsigned& hi, unsigned& lo) {
    const hh2 p = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h2v ph = __builtin_bit_cast(h2v, p);
    const float r0 = fmaf((float)ph[0], neg1, x0), r1 = fmaf((float)ph[1], neg1, x1);
    hi = __builtin_bit_cast(unsigned, p);
    lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
}

__device__ __forceinline__ f32x16 ws_bias(const float* lb, int m, int h) {
    const float4* p = reinterpret_cast<const float4*>(lb + (m * 2 + h) * 16);
    const float4 a = p[0], b = p[1], c = p[2], d = p[3];
    f32x16 r = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    return r;
}

// hipcc schedules a region MFMAs first, dependent-free VALU after (measured on the first version of this kernel: 48 MFMAs
// back to back, then ~200 encoding instructions with the matrix pipe idle: 1.85 ms per frame). The instruction stream is
// therefore laid out by hand: the work of a pipeline step is cut into SLOTS of one MFMA plus the VALU / LDS work that should
// issue in its 32-cycle shadow, and a full scheduling fence after every slot keeps the order as written.
#define WS_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef T2N_PHASE_TIMING
__device__ unsigned long long g_ws_phase[16];
#define WS_PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); phacc[i] += t_ - tph; tph = t_; } while (0)
#else
#define WS_PHASE(i) do {} while (0)
#endif

// 1 / (2 pi) as a two-constant product: t = th + tl carries f / (2 pi) to ~2^-48 relative
__device__ __forceinline__ void ws_turns(float f, float& th, float& tl) {
    const float C1 = 0.15915494309189535f;                                   // float(1 / (2 pi))
    const float C2 = (float)(0.15915494309189533576888 - (double)C1);
    th = f * C1;
    tl = fmaf(f, C1, -th) + f * C2;
}

// Values V, V+1 of the lane's 96-value sequence (see the K order above) as packed hi / lo halves, in three phases of about
// equal issue time (a transcendental costs ~4 plain issues), one phase per MFMA slot.
struct WsUnit { float x0, x1; unsigned hi; };
template <int V, int PH>
__device__ __forceinline__ void ws_unit_phase(WsUnit& U, const float (&f)[7], const float (&th)[7], const float (&tl)[7], float neg1, unsigned& hi, unsigned& lo) {
    constexpr int q = V < 84 ? (V % 12) / 2 : 0;
    constexpr bool fresh = V < 84 && (q == 0 || q == 3);   // octaves 1, 2, 4, 5 by the double-angle identities from the one before
    if constexpr (PH == 0) {
        if constexpr (fresh) {
            constexpr int fi = V / 12;
            constexpr float sc = (float)(1 << q);
            U.x1 = fmaf(tl[fi], sc, __builtin_amdgcn_fractf(th[fi] * sc));   // revolutions: sin / cos have period 1
            U.x0 = __builtin_amdgcn_sinf(U.x1);
        } else if constexpr (V < 84) {
            // U still holds (sin, cos) of the previous octave of this sample (units of a pass run in sequence order)
            const float sn = U.x0, cs = U.x1;
            U.x0 = 2.f * (sn * cs);
            U.x1 = (cs - sn) * (cs + sn);
        } else {
            U.x0 = V < 91 ? f[V < 91 ? V - 84 : 0] : 0.f;
            U.x1 = V + 1 < 91 ? f[V + 1 < 91 ? V + 1 - 84 : 0] : 0.f;
        }
    } else if constexpr (PH == 1) {
        if constexpr (fresh) U.x1 = __builtin_amdgcn_cosf(U.x1);
        U.hi = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(U.x0, U.x1));
    } else {
        const h2v ph = __builtin_bit_cast(h2v, U.hi);
        const float r0 = fmaf((float)ph[0], neg1, U.x0), r1 = fmaf((float)ph[1], neg1, U.x1);
        hi = U.hi;
        lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
}

struct WsEnc {   // per-lane encoder state of one group: features of the lane's two samples (passes 0 / 1) and their turns
    float f[2][7], th[2][7], tl[2][7];
    unsigned ph[8], pl[8];   // packed halves of the chunk being encoded
    WsUnit U[2];             // per pass: the unit in flight; between units (sin, cos) of the pass's previous octave
};

// unit I (0..15) of chunk J: pass I / 8, pair I % 8
template <int J, int I, int PH>
__device__ __forceinline__ void ws_enc_phase(WsEnc& E, float neg1) {
    constexpr int p = I / 8, u = I % 8;
    ws_unit_phase<16 * J + 2 * u, PH>(E.U[p], E.f[p], E.th[p], E.tl[p], neg1, E.ph[u], E.pl[u]);
}
template <int J, int I>
__device__ __forceinline__ void ws_enc_unit(WsEnc& E, float neg1) {
    ws_enc_phase<J, I, 0>(E, neg1); ws_enc_phase<J, I, 1>(E, neg1); ws_enc_phase<J, I, 2>(E, neg1);
}
// the four 16-B stores of one pass: [producer wave][N-tile 2p + (lane >> 5)][part][column (lane & 31) + 32 * K-half]
template <int P>
__device__ __forceinline__ void ws_enc_store(const WsEnc& E, uint4* __restrict__ dst /* slot + ((w*NT + lane>>5) * 2) * 64 + (lane & 31) */) {
    uint4* d = dst + P * 2 * 128;
    d[0] = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
    d[32] = make_uint4(E.ph[4], E.ph[5], E.ph[6], E.ph[7]);
    d[64] = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
    d[96] = make_uint4(E.pl[4], E.pl[5], E.pl[6], E.pl[7]);
}

template <int J, int I0, int I1>
__device__ __forceinline__ void ws_enc_range(WsEnc& E, float neg1) {
    if constexpr (I0 < I1) { ws_enc_unit<J, I0>(E, neg1); ws_enc_range<J, I0 + 1, I1>(E, neg1); }
}

struct WsB { uint4 h[2][kWsNT], l[2][kWsNT]; };   // B operands of two chunks in flight

// slot M (0..47) of layer-0 step J: MFMA M (chunk M / 12 of the step = producer wave w', product (M % 12) / 4, N-tile M % 4), one
// B-operand read of the next chunk, and — every third slot — one encoding unit of chunk J + 1
template <int J, int M>
__device__ __forceinline__ void ws_l0_slots(f32x16 (&acc)[kWsNT], const uint4 (&A0)[kWsC0][2], WsB& B, const uint4* __restrict__ cur,
                                            uint4* __restrict__ dst, WsEnc& E, float neg1) {
    if constexpr (M < 48) {
        constexpr int k = M / 12, m = M % 12, t = m & 3, prod = m >> 2;
        acc[t] = ws_mfma(A0[6 * k + J][prod == 2 ? 1 : 0], prod == 1 ? B.l[k & 1][t] : B.h[k & 1][t], acc[t]);
        if constexpr (k < 3 && m < 8) {
            if constexpr (m < 4) B.h[(k + 1) & 1][m] = cur[(k + 1) * kWsNT * 128 + m * 128];
            else B.l[(k + 1) & 1][m - 4] = cur[(k + 1) * kWsNT * 128 + (m - 4) * 128 + 64];
        }
        if constexpr (J < 5) {
            // unit 0 ran before slot 0; unit i = 1..15 takes slots 3(i-1) .. 3(i-1)+2, one phase each; a pass is stored right after
            // its last unit (pass 0: units 0..7 -> slot 21, pass 1: slot 45)
            if constexpr (M / 3 + 1 < 16) ws_enc_phase<J + 1, M / 3 + 1, M % 3>(E, neg1);
            if constexpr (M == 21) ws_enc_store<0>(E, dst);
            if constexpr (M == 45) ws_enc_store<1>(E, dst);
        }
        WS_FENCE();
        ws_l0_slots<J, M + 1>(acc, A0, B, cur, dst, E, neg1);
    }
}

template <int J>
__device__ __forceinline__ void ws_l0_step(f32x16 (&acc)[kWsNT], const uint4 (&A0)[kWsC0][2], uint4* __restrict__ ring, int w, int lane,
                                           WsEnc& E, float neg1) {
    const uint4* __restrict__ cur = ring + (J & 1) * kWsSlot + lane;
    uint4* __restrict__ dst = ring + ((J + 1) & 1) * kWsSlot + ((w * kWsNT + (lane >> 5)) * 2) * 64 + (lane & 31);
    WsB B;
#pragma unroll
    for (int t = 0; t < kWsNT; ++t) { B.h[0][t] = cur[t * 128]; B.l[0][t] = cur[t * 128 + 64]; }
    WS_FENCE();
    if constexpr (J < 5) { ws_enc_unit<J + 1, 0>(E, neg1); WS_FENCE(); }   // covers the first reads' LDS latency
    ws_l0_slots<J, 0>(acc, A0, B, cur, dst, E, neg1);
    __syncthreads();
}

// relu(acc * inv) of units 32w + 8g + 4h .. +3 of one N-tile -> H in B layout (chunk 2w + (g >> 1), K-half g & 1, elements
// 4h .. 4h+3): one 8-B store per part. Three phases (pair 0-1, pair 2-3, range check + stores) so that a stage can spread it.
struct WsQuad { float m0, m1; uint2 hi, lo; };
template <int G, int PH>
__device__ __forceinline__ void ws_quad_phase(WsQuad& Q, const f32x16& acc, float inv, uint2* __restrict__ H2t /* tile base + lane part */, int w, float neg1, float& amax) {
    if constexpr (PH < 2) {
        const float v0 = fmaxf(acc[4 * G + 2 * PH] * inv, 0.f), v1 = fmaxf(acc[4 * G + 2 * PH + 1] * inv, 0.f);
        if constexpr (PH == 0) { Q.m0 = fmaxf(v0, v1); ws_split2(v0, v1, neg1, Q.hi.x, Q.lo.x); }
        else { Q.m1 = fmaxf(v0, v1); ws_split2(v0, v1, neg1, Q.hi.y, Q.lo.y); }
    } else {
        amax = fmaxf(amax, fmaxf(Q.m0, Q.m1));
        const int u4 = ((2 * w + (G >> 1)) * 2) * 64 + 32 * (G & 1);
        H2t[(size_t)u4 * 2] = Q.hi;
        H2t[(size_t)(u4 + 64) * 2] = Q.lo;
    }
}
template <int G>
__device__ __forceinline__ float ws_store_quad(const f32x16& acc, float inv, uint2* __restrict__ H2t, int w, float neg1, float amax) {
    WsQuad Q;
    ws_quad_phase<G, 0>(Q, acc, inv, H2t, w, neg1, amax);
    ws_quad_phase<G, 1>(Q, acc, inv, H2t, w, neg1, amax);
    ws_quad_phase<G, 2>(Q, acc, inv, H2t, w, neg1, amax);
    return amax;
}

struct WsL1 { uint4 ah[2], al[2], bh[2], bl[2]; };

// slot M (0..23) of the layer-1 stage of N-tile T: MFMA (chunk M / 3, product M % 3) on acc[T], operand reads of the next chunk,
// and in its shadow the hidden-activation stores of the neighbouring tiles: h0 of tile T + 1 (still in acc[T + 1]) and h1 of
// tile T - 1 (now in acc[T - 1])
template <int T, int M>
__device__ __forceinline__ void ws_l1_slots(f32x16 (&acc)[kWsNT], WsL1& O, WsQuad& Q, const uint4* __restrict__ w1 /* W1 + w slice + lane */,
                                            const uint4* __restrict__ hb /* H tile T + lane */, uint2* __restrict__ H2 /* H as uint2 + lane part */,
                                            int w, float inv0, float inv1, float neg1, float& amax) {
    if constexpr (M < 24) {
        constexpr int c = M / 3, prod = M % 3;
        acc[T] = ws_mfma(prod == 2 ? O.al[c & 1] : O.ah[c & 1], prod == 1 ? O.bl[c & 1] : O.bh[c & 1], acc[T]);
        if constexpr (c < 7) {
            if constexpr (prod == 0) { O.ah[(c + 1) & 1] = w1[(c + 1) * 128]; O.bh[(c + 1) & 1] = hb[(c + 1) * 128]; }
            if constexpr (prod == 1) { O.al[(c + 1) & 1] = w1[(c + 1) * 128 + 64]; O.bl[(c + 1) & 1] = hb[(c + 1) * 128 + 64]; }
        }
        // quad g of the h0 store in slots 6g .. 6g+2, of the h1 store in slots 6g+3 .. 6g+5
        if constexpr (T + 1 < kWsNT && M % 6 < 3) ws_quad_phase<M / 6, M % 6>(Q, acc[T + 1 < kWsNT ? T + 1 : T], inv0, H2 + (size_t)(T + 1) * kWsC1 * 256, w, neg1, amax);
        if constexpr (T >= 1 && M % 6 >= 3) ws_quad_phase<M / 6, M % 6 - 3>(Q, acc[T >= 1 ? T - 1 : T], inv1, H2 + (size_t)(T - 1) * kWsC1 * 256, w, neg1, amax);
        WS_FENCE();
        ws_l1_slots<T, M + 1>(acc, O, Q, w1, hb, H2, w, inv0, inv1, neg1, amax);
    }
}

template <int T>
__device__ __forceinline__ void ws_l1_stage(f32x16 (&acc)[kWsNT], const uint4* __restrict__ W1, uint4* __restrict__ H, const float* __restrict__ LB,
                                            int w, int lane, float inv0, float inv1, float neg1, float& amax) {
    const uint4* __restrict__ w1 = W1 + (w * kWsC1 * 2) * 64 + lane;
    const uint4* __restrict__ hb = H + (T * kWsC1 * 2) * 64 + lane;
    uint2* __restrict__ H2 = reinterpret_cast<uint2*>(H) + (size_t)(lane & 31) * 2 + (lane >> 5);
    WsL1 O;
    WsQuad Q;
    O.ah[0] = w1[0]; O.al[0] = w1[64]; O.bh[0] = hb[0]; O.bl[0] = hb[64];
    acc[T] = ws_bias(LB + 128, w, lane >> 5);   // acc[T] held layer 0 until its h0 store one stage ago
    WS_FENCE();
    ws_l1_slots<T, 0>(acc, O, Q, w1, hb, H2, w, inv0, inv1, neg1, amax);
    __syncthreads();
}

__global__ __launch_bounds__(256, 1) void k_mlp_ws(const WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    uint4* __restrict__ RH = lds;                       // ring slots 0 / 1 during layer 0, hidden tile afterwards
    uint4* __restrict__ W1 = lds + kWsH;
    uint4* __restrict__ W2 = W1 + kWsW1;
    float* __restrict__ LB = reinterpret_cast<float*>(W2 + kWsW2);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = lane & 31, h = lane >> 5;

    // tile enumeration over the appearance sub-lists (as k_shade): tile -> (list, offset)
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
    const unsigned ngroups = (ntiles + kWsNT - 1) / kWsNT;
    if (blockIdx.x >= ngroups) return;

    // stationary operands: layer 0 in registers, layers 1 / 2 and the biases in LDS
    uint4 A0[kWsC0][2];
#pragma unroll
    for (int c = 0; c < kWsC0; ++c) {
        A0[c][0] = a.w0[((size_t)(w * kWsC0 + c) * 2) * 64 + lane];
        A0[c][1] = a.w0[((size_t)(w * kWsC0 + c) * 2 + 1) * 64 + lane];
    }
    for (int i = tid; i < kWsW1; i += 256) W1[i] = a.w1[i];
    for (int i = tid; i < kWsW2; i += 256) W2[i] = a.w2[i];
    for (int i = tid; i < kWsBias; i += 256) LB[i] = a.bias[i];
    const float inv0 = a.inv_scale[0], inv1 = a.inv_scale[1], inv2 = a.inv_scale[2];
    const float neg1 = a.neg1;
    float amax = 0.f;

    auto load_feat = [&](unsigned g, float (&f0)[7], float (&f1)[7]) {
        // row = tile * 32 + sample = 128 g + 64 p + lane; tiles past the end re-read the last tile's rows (their results are
        // never stored): no predication, 14 plain loads
        const unsigned last = ntiles * 32u - 1u;
        unsigned r0 = g * 128u + (unsigned)lane, r1 = r0 + 64u;
        r0 = r0 < last ? r0 : last; r1 = r1 < last ? r1 : last;
        const float* p0 = a.feat + (size_t)r0 * 32 + 7 * w;
        const float* p1 = a.feat + (size_t)r1 * 32 + 7 * w;
#pragma unroll
        for (int i = 0; i < 7; ++i) { f0[i] = p0[i]; f1[i] = p1[i]; }
    };
    float nf0[7], nf1[7];   // the next group's feature rows, loaded one group ahead
    load_feat(blockIdx.x, nf0, nf1);
    __syncthreads();   // W1 / W2 / bias visible
    uint2* __restrict__ H2l = reinterpret_cast<uint2*>(RH) + (size_t)n * 2 + h;

#ifdef T2N_PHASE_TIMING
    unsigned long long phacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tph = __builtin_amdgcn_s_memtime();
#endif
    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x) {
        WS_PHASE(15);
        WsEnc E;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            E.f[0][i] = nf0[i]; E.f[1][i] = nf1[i];
            amax = fmaxf(amax, fmaxf(fabsf(nf0[i]), fabsf(nf1[i])));
            ws_turns(nf0[i], E.th[0][i], E.tl[0][i]);
            ws_turns(nf1[i], E.th[1][i], E.tl[1][i]);
        }
        // ---- layer 0: chunk 0 encoded up front, then 6 steps (step j multiplies chunks {6w' + j}, encodes chunk j + 1) ---------
        {
            uint4* __restrict__ dst = RH + ((w * kWsNT + h) * 2) * 64 + n;
            ws_enc_range<0, 0, 8>(E, neg1); ws_enc_store<0>(E, dst);
            ws_enc_range<0, 8, 16>(E, neg1); ws_enc_store<1>(E, dst);
        }
        f32x16 acc[kWsNT];
#pragma unroll
        for (int t = 0; t < kWsNT; ++t) acc[t] = ws_bias(LB, w, h);
        __syncthreads();
        WS_PHASE(0);
        ws_l0_step<0>(acc, A0, RH, w, lane, E, neg1);
        WS_PHASE(1);
        ws_l0_step<1>(acc, A0, RH, w, lane, E, neg1);
        ws_l0_step<2>(acc, A0, RH, w, lane, E, neg1);
        ws_l0_step<3>(acc, A0, RH, w, lane, E, neg1);
        ws_l0_step<4>(acc, A0, RH, w, lane, E, neg1);
        WS_PHASE(2);
        ws_l0_step<5>(acc, A0, RH, w, lane, E, neg1);
        WS_PHASE(3);
        // the next group's feature rows: in flight under layers 1 and 2
        const unsigned gn = g + gridDim.x;
        load_feat(gn < ngroups ? gn : g, nf0, nf1);
        // ---- layers 1 and the hidden-activation stores, one N-tile per stage ------------------------------------------------------
        // (every wave passed the last ring read at the barrier that ended step 5; h1 of tile t overwrites h0 of tile t one stage
        // after every wave finished reading it)
        amax = ws_store_quad<0>(acc[0], inv0, H2l, w, neg1, amax);
        amax = ws_store_quad<1>(acc[0], inv0, H2l, w, neg1, amax);
        amax = ws_store_quad<2>(acc[0], inv0, H2l, w, neg1, amax);
        amax = ws_store_quad<3>(acc[0], inv0, H2l, w, neg1, amax);
        __syncthreads();
        WS_PHASE(4);
        ws_l1_stage<0>(acc, W1, RH, LB, w, lane, inv0, inv1, neg1, amax);
        WS_PHASE(5);
        ws_l1_stage<1>(acc, W1, RH, LB, w, lane, inv0, inv1, neg1, amax);
        ws_l1_stage<2>(acc, W1, RH, LB, w, lane, inv0, inv1, neg1, amax);
        WS_PHASE(6);
        ws_l1_stage<3>(acc, W1, RH, LB, w, lane, inv0, inv1, neg1, amax);
        WS_PHASE(7);
        {
            uint2* __restrict__ H2t = H2l + (size_t)3 * kWsC1 * 256;
            amax = ws_store_quad<0>(acc[3], inv1, H2t, w, neg1, amax);
            amax = ws_store_quad<1>(acc[3], inv1, H2t, w, neg1, amax);
            amax = ws_store_quad<2>(acc[3], inv1, H2t, w, neg1, amax);
            amax = ws_store_quad<3>(acc[3], inv1, H2t, w, neg1, amax);
        }
        __syncthreads();
        WS_PHASE(8);
        // ---- layer 2: wave w finishes N-tile w (three independent product chains) -----------------------------------------
        {
            f32x16 c0 = ws_bias(LB + 256, 0, h), c1 = {0}, c2 = {0};
            const uint4* __restrict__ w2 = W2 + lane;
            const uint4* __restrict__ hb = RH + (w * kWsC1 * 2) * 64 + lane;
            uint4 ah[3], al[3], bh[3], bl[3];   // operands of three chunks in flight
#pragma unroll
            for (int c = 0; c < 2; ++c) { ah[c] = w2[c * 128]; al[c] = w2[c * 128 + 64]; bh[c] = hb[c * 128]; bl[c] = hb[c * 128 + 64]; }
            WS_FENCE();
#pragma unroll
            for (int c = 0; c < kWsC2; ++c) {
                if (c + 2 < kWsC2) { ah[(c + 2) % 3] = w2[(c + 2) * 128]; al[(c + 2) % 3] = w2[(c + 2) * 128 + 64]; bh[(c + 2) % 3] = hb[(c + 2) * 128]; bl[(c + 2) % 3] = hb[(c + 2) * 128 + 64]; }
                c0 = ws_mfma(ah[c % 3], bh[c % 3], c0);
                c1 = ws_mfma(ah[c % 3], bl[c % 3], c1);
                c2 = ws_mfma(al[c % 3], bh[c % 3], c2);
                WS_FENCE();
            }
            const unsigned tile = g * kWsNT + (unsigned)w;
            if (tile < ntiles) {
                const int li = (int)__popcll(__ballot((lane < a.nlists) & (incl <= tile)));
                const unsigned before = li ? __shfl(incl, li - 1) : 0u;
                const unsigned lbase = (unsigned)li * a.list_cap;
                const unsigned idx = lbase + (tile - before) * 32u + (unsigned)n;
                const unsigned count = lbase + __shfl(cnt_l, li);
                if (h == 0 && idx < count) {
                    const float r = ((c0[0] + c1[0]) + c2[0]) * inv2, gg = ((c0[1] + c1[1]) + c2[1]) * inv2, b = ((c0[2] + c1[2]) + c2[2]) * inv2;
                    // sigmoid by v_exp_f32 / v_rcp_f32 (1 ulp each: ~2e-7 absolute on a value in (0, 1))
                    a.app_rgb[idx] = make_float4(__builtin_amdgcn_rcpf(1.f + __expf(-r)), __builtin_amdgcn_rcpf(1.f + __expf(-gg)),
                                                 __builtin_amdgcn_rcpf(1.f + __expf(-b)), 0.f);
                }
            }
        }
        WS_PHASE(9);
        __syncthreads();   // hidden-tile reads done before the next group's ring writes
        WS_PHASE(10);
    }
#ifdef T2N_PHASE_TIMING
    if (tid == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_ws_phase[i], phacc[i]);
#endif
    if (__any(!(amax <= kWsRange)) && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- operand packing -----------------------------------------------------------------------------------------------------
// scales[l] = 2^k with max|W_l| * 2^k in [2^12, 2^13): hi halves use the top of the f16 range, lo halves stay normal for
// every weight within 2^-11 of the largest one. inv_scale = 1 / scale. [0..2] scale, [4..6] inverse.
struct WsPackArgs {
    const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    uint4* w0p; uint4* w1p; uint4* w2p; float* biasp; float* scales;
};

__global__ __launch_bounds__(256) void k_ws_scales(const WsPackArgs a) {
    __shared__ float red[3][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float m[3] = {0.f, 0.f, 0.f};
    for (int i = tid; i < 128 * 351; i += 256) m[0] = fmaxf(m[0], fabsf(a.w0[i]));
    for (int i = tid; i < 128 * 128; i += 256) m[1] = fmaxf(m[1], fabsf(a.w1[i]));
    for (int i = tid; i < 3 * 128; i += 256) m[2] = fmaxf(m[2], fabsf(a.w2[i]));
#pragma unroll
    for (int l = 0; l < 3; ++l) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m[l] = fmaxf(m[l], __shfl_xor(m[l], o));
        if (lane == 0) red[l][wv] = m[l];
    }
    __syncthreads();
    if (tid < 3) {
        const float mx = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
        float s = 1.f;
        if (mx > 0.f && mx < 3e38f) {
            int k;
            (void)frexpf(mx, &k);            // mx = m * 2^k, m in [0.5, 1)
            s = ldexpf(1.f, 13 - k);
        }
        a.scales[tid] = s;
        a.scales[4 + tid] = 1.f / s;
    }
}

__device__ __forceinline__ unsigned ws_pack2(float x0, float x1, int part) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    h2v r;
    if (part == 0) { r[0] = h0; r[1] = h1; }
    else { r[0] = (_Float16)(x0 - (float)h0); r[1] = (_Float16)(x1 - (float)h1); }
    return __builtin_bit_cast(unsigned, r);
}

// reference column of layer-0 K index (chunk c, half hh, element e); -1: zero padding
__host__ __device__ inline int ws_l0_col(int c, int hh, int e) {
    const int wp = c / 6, j = c % 6, v = 16 * j + 8 * hh + e;
    if (v < 84) {
        const int F = 7 * wp + v / 12, r = v % 12, q = r >> 1, sc = r & 1;
        return F < 27 ? (sc ? 189 : 27) + F * 6 + q : -1;
    }
    if (v < 91) { const int F = 7 * wp + v - 84; return F < 27 ? F : -1; }
    return -1;
}

__global__ __launch_bounds__(256) void k_pack_ws(const WsPackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = 4 * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    const float s0 = a.scales[0], s1 = a.scales[1], s2 = a.scales[2];
    int g = gid;
    if (g < n0) {
        const int lane = g & 63, part = (g >> 6) & 1, c = (g >> 7) % kWsC0, w = (g >> 7) / kWsC0;
        const int unit = 32 * w + (lane & 31), hh = lane >> 5;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int col = ws_l0_col(c, hh, e); x[e] = col >= 0 ? a.w0[unit * 351 + col] * s0 : 0.f; }
        a.w0p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n0;
    if (g < n1) {
        const int lane = g & 63, part = (g >> 6) & 1, c = (g >> 7) % kWsC1, w = (g >> 7) / kWsC1;
        const int unit = 32 * w + (lane & 31), hh = lane >> 5;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a.w1[unit * 128 + 16 * c + 8 * hh + e] * s1;
        a.w1p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n1;
    if (g < n2) {
        const int lane = g & 63, part = (g >> 6) & 1, c = g >> 7;
        const int row = lane & 31, hh = lane >> 5;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = row < 3 ? a.w2[row * 128 + 16 * c + 8 * hh + e] * s2 : 0.f;
        a.w2p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n2;
    if (g < kWsBias) {   // [layer][block][h][v]: unit = block*32 + (v&3) + 8*(v>>2) + 4*h
        const int layer = g < 128 ? 0 : (g < 256 ? 1 : 2);
        const int r = g - (layer == 0 ? 0 : (layer == 1 ? 128 : 256));
        const int mm = r / 32, hh = (r / 16) & 1, v = r & 15;
        const int u = mm * 32 + (v & 3) + 8 * (v >> 2) + 4 * hh;
        float b = 0.f;
        if (layer == 0) b = a.b0[u] * s0;
        else if (layer == 1) b = a.b1[u] * s1;
        else if (u < 3) b = a.b2[u] * s2;
        a.biasp[g] = b;
    }
}

int ws_pack(t2n_field* f, hipStream_t s) {
    const size_t n0 = (size_t)4 * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    if (!f->buf_ws) {
        T2N_HIP(hipMalloc((void**)&f->buf_ws, (n0 + n1 + n2) * 16 + (kWsBias + 8 + 2) * 4));
    }
    uint4* base = (uint4*)f->buf_ws;
    WsPackArgs a;
    const t2n_field_params& p = f->params_ref;
    a.w0 = p.mlp_w0; a.b0 = p.mlp_b0; a.w1 = p.mlp_w1; a.b1 = p.mlp_b1; a.w2 = p.mlp_w2; a.b2 = p.mlp_b2;
    a.w0p = base; a.w1p = base + n0; a.w2p = base + n0 + n1;
    a.biasp = (float*)(base + n0 + n1 + n2); a.scales = a.biasp + kWsBias;
    hipLaunchKernelGGL(k_ws_scales, dim3(1), dim3(256), 0, s, a);
    const size_t total = n0 + n1 + n2 + kWsBias;
    hipLaunchKernelGGL(k_pack_ws, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    f->ws_dirty = false;
    return T2N_OK;
}

// the head of the appearance stage for tiles [0, tile_hi) of the sub-lists, from feature rows; *range_flag is set when an
// activation left the f16 range (the caller then re-runs the launch on the exact path)
int launch_mlp_ws(t2n_field* f, const float* feat, const unsigned* counters_dev, unsigned list_cap, unsigned tile_hi, float4* app_rgb,
                  unsigned* range_flag, hipStream_t s) {
    if (f->ws_dirty || !f->buf_ws) { const int rc = ws_pack(f, s); if (rc) return rc; }
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ws, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWsLds));
        attr_set = true;
    }
    const size_t n0 = (size_t)4 * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    uint4* base = (uint4*)f->buf_ws;
    WsArgs a;
    a.w0 = base; a.w1 = base + n0; a.w2 = base + n0 + n1;
    a.bias = (const float*)(base + n0 + n1 + n2); a.inv_scale = a.bias + kWsBias + 4;
    a.feat = feat; a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.tile_hi = tile_hi; a.app_rgb = app_rgb;
    a.range_flag = range_flag; a.neg1 = -1.f;
    hipLaunchKernelGGL(k_mlp_ws, dim3(256), dim3(256), kWsLds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

#ifdef T2N_PHASE_TIMING
extern "C" int t2n_debug_ws_phase_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(t2n::g_ws_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(t2n::g_ws_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
