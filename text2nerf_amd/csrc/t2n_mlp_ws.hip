// Weight-stationary appearance head: positional encoding + the 351 -> 128 -> 128 -> 3 MLP + sigmoid for rows of 27 appearance
// features (K2b). The features come from the gather + basis kernel (k_shade in features-only mode, K2a); this kernel is the
// matrix-core part of the appearance stage on the default (split-f16) render path.
//
// Replaces (reference): models/tensorBase.py:11-17 (positional_encoding), :88-109 (MLPRender_Fea_noview.forward).
//
// Mapping (gfx950, wave64): ONE 512-thread workgroup per CU (two waves per SIMD), persistent over groups of 128 samples
// (8 N-tiles of 16). Wave w owns output units 16w .. 16w+15 of both hidden layers and keeps its slice of the layer-0 weights in
// registers for the whole kernel (12 K-chunks of 32 x (hi, lo) x 16 B per lane = 96 VGPRs: the A operands of
// v_mfma_f32_16x16x32_f16 never move again; the layer-1 / layer-2 slices sit in LDS). fp32 products are three f16 products of
// hi/lo splits (x = hi + lo, hi = RTZ_f16(x), lo = RTZ_f16(x - hi); the lo*lo term is dropped: ~2^-21 relative), fp32
// accumulate. Weights are pre-split and scaled by a per-layer power of two chosen from max|W| at upload (k_ws_scales),
// activations are split where they are produced, and any activation beyond the f16 range raises a flag that makes the caller's
// exact-fp32 kernel redo the launch.
//
// (A first version ran four waves with 32 units each on 32x32x16 tiles — 192 weight registers, one wave per SIMD: with a single
// in-order wave per SIMD the ~5 non-MFMA issues per MFMA of the encoding did not fit the MFMA's shadow even hand-interleaved:
// 1.49 ms per frame, 48 cycles per MFMA in the encoding steps against 37 in a pure-MFMA step. Two waves per SIMD need <= 256
// registers each, hence the 16-unit slices.)
//
// The B operands (samples on N) go through LDS once per sample: the waves share the encoding work — wave w' < 7 encodes features
// 4w' .. 4w'+3 of every sample (sin / cos of octaves 0 and 3 by v_sin_f32 / v_cos_f32 on the fraction of f 2^q / (2 pi), 1 / (2 pi)
// as a two-constant product: max abs error 4.2e-7 over |f| <= 3e4, tools/experiments/hw_sincos.hip; octaves 1, 2, 4, 5 by the
// double-angle identities), wave 7 passes the raw features — as six K-octets of 8 values, one octet per pipeline step into a
// two-slot ring; a step's 48 MFMAs per wave (2 chunks x 8 N-tiles x 3 products) run on the octets encoded during the previous
// step. Hidden activations return to LDS in the same B layout (relu, unscale, split), so layers 1 and 2 are plain LDS -> MFMA
// streams. 12 workgroup barriers per 128 samples.
//
// K order of layer 0 (what k_pack_ws mirrors): K-octet o = 8 j + w' (step j, producer w') = chunk 2j + w'/4, K-quarter w' % 4, holds
// values v = 8j .. 8j+7 of producer w's 48-value sequence; w' < 7: feature 4w' + v / 12, octave (v % 12) / 2, sin (even v) / cos
// (odd v); w' = 7: raw feature v (zero from 27 on).
//
// hipcc schedules a region MFMAs first, VALU after; two waves in lockstep would then fight for the matrix pipe and leave it idle
// together. The instruction stream is therefore laid out by hand: a step is cut into SLOTS of one MFMA plus the VALU / LDS work
// that should issue in its shadow, with a full scheduling fence after every slot.
#include "t2n_device.h"

namespace t2n {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

constexpr int kWsW = 8;                          // waves per workgroup = 16-unit slices of the hidden layers
constexpr int kWsT = 8;                          // N-tiles (16 samples) per group
constexpr int kWsC0 = 12, kWsC1 = 4, kWsC2 = 4;   // K-chunks (32 values) of layers 0 / 1 / 2
constexpr int kWsSlot = 2 * kWsT * 2 * 64;       // uint4 per ring slot: [chunk of the step][N-tile][part][lane]
constexpr int kWsH = kWsT * kWsC1 * 2 * 64;      // uint4 of the hidden-activation tile: [N-tile][chunk][part][lane] (aliases the ring)
constexpr int kWsW1 = kWsW * kWsC1 * 2 * 64;     // [wave][chunk][part][lane]
constexpr int kWsW2 = kWsC2 * 2 * 64;
constexpr int kWsBias = 288;                     // floats: layer 0 [128], layer 1 [128], layer 2 [32], scaled
constexpr size_t kWsLds = (size_t)(kWsH + kWsW1 + kWsW2) * 16 + kWsBias * 4 + 16 * 4;   // + the sub-list table
static_assert(2 * kWsSlot == kWsH, "ring and hidden tile share one region");
constexpr float kWsRange = 60000.f;              // |activation| beyond this (f16 max 65504) -> exact-path redo

struct WsArgs {
    const uint4* w0; const uint4* w1; const uint4* w2; const float* bias; const float* inv_scale;   // packed by k_pack_ws
    const float* feat;            // [rows][32] fp32 feature rows (row = tile * 32 + sample), columns >= 27 zero
    const unsigned* counters; unsigned list_cap; int nlists;
    unsigned tile_hi;             // 32-sample tiles [0, min(ntiles, tile_hi)) are this kernel's; tile_hi is a multiple of 4
    float4* app_rgb;
    unsigned* range_flag;
    float neg1;                   // -1.0f at run time: x - hi stays an FMA with an f16 operand (v_fma_mix_f32, no convert)
};

#define WS_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef T2N_PHASE_TIMING
__device__ unsigned long long g_ws_phase[16];
#define WS_PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); phacc[i] += t_ - tph; tph = t_; } while (0)
#else
#define WS_PHASE(i) do {} while (0)
#endif

__device__ __forceinline__ f32x4 ws_mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// (x0, x1) -> packed hi halves and packed lo halves
__device__ __forceinline__ void ws_split2(float x0, float x1, float neg1, unsigned& hi, unsigned& lo) {
    const hh2 p = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h2v ph = __builtin_bit_cast(h2v, p);
    const float r0 = fmaf((float)ph[0], neg1, x0), r1 = fmaf((float)ph[1], neg1, x1);
    hi = __builtin_bit_cast(unsigned, p);
    lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// ---- encoding -------------------------------------------------------------------------------------------------------------------
// Values V, V+1 of the lane's 48-value sequence as packed hi / lo halves, in three phases of about equal issue time, one phase
// per slot. RAW (wave 7): the values are the step's eight raw features.
struct WsUnit { float x0, x1; unsigned hi; };
struct WsEnc {
    float f[2][4];           // PE waves: features 4w' .. 4w'+3 of the lane's two samples (passes 0 / 1)
    float4 r[2][2];          // raw wave: the current octet's eight feature values per pass
    WsUnit U[2];             // per pass: the unit in flight; between units (sin, cos) of the pass's previous octave
    unsigned ph[4], pl[4];   // packed halves of the octet being encoded
    unsigned amax_u;         // raw wave: running max of the features' |bit patterns| (orders like |x| and ranks inf / NaN on top)
};

template <int V, int PH, bool RAW>
__device__ __forceinline__ void ws_unit_phase(WsUnit& U, const float (&f)[4], const float4 (&r)[2], float neg1, unsigned& hi, unsigned& lo, unsigned& amax_u) {
    constexpr int q = (V % 12) / 2;
    constexpr bool fresh = q == 0 || q == 3;   // octaves 1, 2, 4, 5 from the one before
    if constexpr (PH == 0) {
        if constexpr (RAW) {
            constexpr int e = V % 8;
            const float4 s = r[e / 4];
            U.x0 = (e % 4) == 0 ? s.x : s.z;
            U.x1 = (e % 4) == 0 ? s.y : s.w;
            amax_u = max(amax_u, max(__float_as_uint(U.x0) & 0x7fffffffu, __float_as_uint(U.x1) & 0x7fffffffu));
        } else if constexpr (fresh) {
            // f / (2 pi) as th + tl (two-constant product, ~2^-48 relative), then the fraction of its 2^q multiple: sin / cos take
            // revolutions and have period 1
            const float C1 = 0.15915494309189535f;                                   // float(1 / (2 pi))
            const float C2 = (float)(0.15915494309189533576888 - (double)C1);
            constexpr float sc = (float)(1 << q);
            const float x = f[V / 12];
            const float th = x * C1;
            const float tl = fmaf(x, C1, -th) + x * C2;
            U.x1 = fmaf(tl, sc, __builtin_amdgcn_fractf(th * sc));
        } else {
            const float sn = U.x0, cs = U.x1;   // (sin, cos) of the previous octave of this sample
            U.x0 = 2.f * (sn * cs);
            U.x1 = (cs - sn) * (cs + sn);
        }
    } else if constexpr (PH == 1) {
        if constexpr (!RAW && fresh) {
            U.x0 = __builtin_amdgcn_sinf(U.x1);
            U.x1 = __builtin_amdgcn_cosf(U.x1);
        }
        U.hi = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(U.x0, U.x1));
    } else {
        const h2v ph = __builtin_bit_cast(h2v, U.hi);
        const float r0 = fmaf((float)ph[0], neg1, U.x0), r1 = fmaf((float)ph[1], neg1, U.x1);
        hi = U.hi;
        lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
    }
}

// unit I (0..7) of octet J: pass I / 4, pair I % 4
template <int J, int I, int PH, bool RAW>
__device__ __forceinline__ void ws_enc_phase(WsEnc& E, float neg1) {
    constexpr int p = I / 4, u = I % 4;
    ws_unit_phase<8 * J + 2 * u, PH, RAW>(E.U[p], E.f[p], E.r[p], neg1, E.ph[u], E.pl[u], E.amax_u);
}
// the two 16-B stores of one pass: ring [chunk w' / 4][N-tile 4P + (lane >> 4)][part][column (lane & 15) + 16 (w' % 4)]
template <int P>
__device__ __forceinline__ void ws_enc_store(const WsEnc& E, uint4* __restrict__ dst /* slot + (((w/4)*T + lane>>4) * 2) * 64 + (lane & 15) + 16 (w%4) */) {
    uint4* d = dst + P * 4 * 128;
    d[0] = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
    d[64] = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
}
template <int J, bool RAW>
__device__ __forceinline__ void ws_enc_octet(WsEnc& E, uint4* __restrict__ dst, float neg1) {   // unscheduled form (prologue)
#define WS_U(I) ws_enc_phase<J, I, 0, RAW>(E, neg1); ws_enc_phase<J, I, 1, RAW>(E, neg1); ws_enc_phase<J, I, 2, RAW>(E, neg1)
    WS_U(0); WS_U(1); WS_U(2); WS_U(3); ws_enc_store<0>(E, dst);
    WS_U(4); WS_U(5); WS_U(6); WS_U(7); ws_enc_store<1>(E, dst);
#undef WS_U
}
// raw wave: the eight raw features of octet J for both passes (columns 8J .. 8J+7 of the feature rows; zero from column 27 on)
template <int J>
__device__ __forceinline__ void ws_raw_load(float4 (&d)[2][2], const float* __restrict__ row0, const float* __restrict__ row1) {
    if constexpr (J < 4) {
        const float4* p0 = reinterpret_cast<const float4*>(row0 + 8 * J);
        const float4* p1 = reinterpret_cast<const float4*>(row1 + 8 * J);
        d[0][0] = p0[0]; d[0][1] = p0[1]; d[1][0] = p1[0]; d[1][1] = p1[1];
    } else {
        d[0][0] = d[0][1] = d[1][0] = d[1][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ---- MFMA streams ---------------------------------------------------------------------------------------------------------------
struct WsB { uint4 h[2][4], l[4]; };   // B operands: hi parts of two half-chunks (4 N-tiles each) in flight, lo parts of one

// Slot M (0..47) of a two-chunk step: chunk M / 24, N-tile half (M % 24) / 12, product (M % 12) / 4 (hi*hi, lo*hi, hi*lo), tile
// 4 half + M % 4. B operands come from LDS at b + chunk * SC + tile * ST (+ 64: lo part): the hi parts one half-chunk ahead, the
// lo parts during the half's first slots (their product comes last) — 48 operand registers, not 64: a spill anywhere between
// the issue of the next group's global loads and their use makes the wave wait for those loads (vmcnt retires in order). A
// operands from the register file (layer 0) or from LDS one chunk ahead (layer 1). ENC: every second slot carries one encoding
// phase of octet JN.
template <int M, int SC, int ST, bool A_LDS, bool ENC, int JN, bool RAW, class AOps>
__device__ __forceinline__ void ws_slots(f32x4 (&acc)[kWsT], AOps& A, WsB& B, const uint4* __restrict__ b, WsEnc& E, uint4* __restrict__ dst, float neg1) {
    if constexpr (M < 48) {
        constexpr int c = M / 24, hf = (M % 24) / 12, prod = (M % 12) / 4, t = M % 4, hb = (M / 12) & 1, m = M % 12;
        acc[4 * hf + t] = ws_mfma(prod == 1 ? A.lo(c) : A.hi(c), prod == 2 ? B.l[t] : B.h[hb][t], acc[4 * hf + t]);
        if constexpr (m < 4) {
            B.l[m] = b[c * SC + (4 * hf + m) * ST + 64];   // this half's lo part (the previous half's last product issued before slot 0)
            if constexpr (M < 36) {
                constexpr int nh = M / 12 + 1, nc = nh / 2, nhf = nh % 2;
                B.h[nh & 1][m] = b[nc * SC + (4 * nhf + m) * ST];
            }
        }
        if constexpr (A_LDS && M == 8) A.fetch(1);
        if constexpr (ENC) {
            if constexpr (M % 2 == 0) ws_enc_phase<JN, (M / 2) / 3, (M / 2) % 3, RAW>(E, neg1);
            if constexpr (M == 23) ws_enc_store<0>(E, dst);
            if constexpr (M == 47) ws_enc_store<1>(E, dst);
        }
        WS_FENCE();
        ws_slots<M + 1, SC, ST, A_LDS, ENC, JN, RAW>(acc, A, B, b, E, dst, neg1);
    }
}
template <int SC, int ST>
__device__ __forceinline__ void ws_first_half(WsB& B, const uint4* __restrict__ b) {
#pragma unroll
    for (int t = 0; t < 4; ++t) B.h[0][t] = b[t * ST];
}

template <int J2>
struct WsA0 {   // layer-0 A operands of one step: chunks J2, J2+1 of the wave's register-resident slice
    const uint4 (&a)[kWsC0][2];
    __device__ __forceinline__ uint4 hi(int c) const { return a[J2 + c][0]; }
    __device__ __forceinline__ uint4 lo(int c) const { return a[J2 + c][1]; }
    __device__ __forceinline__ void fetch(int) {}
};
struct WsA1 {   // layer-1 / layer-2 A operands from LDS: two chunks per step
    const uint4* __restrict__ p;   // slice + first chunk of the step + lane; chunk stride 128
    uint4 h_[2], l_[2];
    __device__ __forceinline__ uint4 hi(int c) const { return h_[c]; }
    __device__ __forceinline__ uint4 lo(int c) const { return l_[c]; }
    __device__ __forceinline__ void fetch(int c) { h_[c] = p[c * 128]; l_[c] = p[c * 128 + 64]; }
};

template <int J, bool RAW>
__device__ __forceinline__ void ws_l0_step(f32x4 (&acc)[kWsT], const uint4 (&A0)[kWsC0][2], uint4* __restrict__ ring, int w, int lane, WsEnc& E,
                                           const float* __restrict__ row0, const float* __restrict__ row1, float neg1) {
    const uint4* __restrict__ cur = ring + (J & 1) * kWsSlot + lane;
    uint4* __restrict__ dst = ring + ((J + 1) & 1) * kWsSlot + (((w >> 2) * kWsT + (lane >> 4)) * 2) * 64 + (lane & 15) + 16 * (w & 3);
    WsB B;
    ws_first_half<kWsT * 128, 128>(B, cur);
    if constexpr (RAW && J < 5) ws_raw_load<J + 1>(E.r, row0, row1);   // same cache lines as octet 0: vector-L1 hits
    WS_FENCE();
    WsA0<2 * J> A{A0};
    ws_slots<0, kWsT * 128, 128, false, (J < 5), J + 1, RAW>(acc, A, B, cur, E, dst, neg1);
    __syncthreads();
}

// relu(acc * inv) of the lane's four units 16w + 4 (lane >> 4) .. +3 of one N-tile -> H in B layout: chunk w / 2, K-quarter
// 2 (w & 1) + (lane >> 5), elements 4 ((lane >> 4) & 1) .. +3: one 8-B store per part
__device__ __forceinline__ float ws_store_quad(const f32x4& acc, float inv, uint2* __restrict__ H2t /* H as uint2, tile + lane part */, float neg1, float amax) {
    const float v0 = fmaxf(acc[0] * inv, 0.f), v1 = fmaxf(acc[1] * inv, 0.f), v2 = fmaxf(acc[2] * inv, 0.f), v3 = fmaxf(acc[3] * inv, 0.f);
    amax = fmaxf(fmaxf(v0, v1), fmaxf(amax, fmaxf(v2, v3)));
    uint2 hi, lo;
    ws_split2(v0, v1, neg1, hi.x, lo.x);
    ws_split2(v2, v3, neg1, hi.y, lo.y);
    H2t[0] = hi;
    H2t[128] = lo;   // part 1: + 64 uint4
    return amax;
}

__global__ __launch_bounds__(512) void k_mlp_ws(const WsArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    uint4* __restrict__ RH = lds;                       // ring slots 0 / 1 during layer 0, hidden tile afterwards
    uint4* __restrict__ W1 = lds + kWsH;
    uint4* __restrict__ W2 = W1 + kWsW1;
    float* __restrict__ LB = reinterpret_cast<float*>(W2 + kWsW2);
    unsigned* __restrict__ LT = reinterpret_cast<unsigned*>(LB + kWsBias);   // [0..7] inclusive tile prefix of the sub-lists, [8..15] their counts
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, rq = lane >> 4;

    // tile enumeration over the appearance sub-lists (as k_shade): 32-sample tile -> (list, offset)
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
    const unsigned ngroups = (ntiles + 3u) / 4u;
    if (blockIdx.x >= ngroups) return;
    if (tid < 8) { LT[tid] = tid < a.nlists ? incl : 0xffffffffu; LT[8 + tid] = cnt_l; }   // kept in LDS: no live registers across the loop

    // stationary operands: layer 0 in registers, layers 1 / 2 and the biases in LDS
    uint4 A0[kWsC0][2];
#pragma unroll
    for (int c = 0; c < kWsC0; ++c) {
        A0[c][0] = a.w0[((size_t)(w * kWsC0 + c) * 2) * 64 + lane];
        A0[c][1] = a.w0[((size_t)(w * kWsC0 + c) * 2 + 1) * 64 + lane];
    }
    for (int i = tid; i < kWsW1; i += 512) W1[i] = a.w1[i];
    for (int i = tid; i < kWsW2; i += 512) W2[i] = a.w2[i];
    for (int i = tid; i < kWsBias; i += 512) LB[i] = a.bias[i];
    const float inv0 = a.inv_scale[0], inv1 = a.inv_scale[1], inv2 = a.inv_scale[2];
    const float neg1 = a.neg1;
    const bool raw = w == 7;
    float amax = 0.f;          // hidden activations (non-negative, finite inputs)
    unsigned amax_u = 0u;      // raw features

    // feature rows of the lane's two samples of group g (row = 128 g + 64 p + lane; tiles past the end re-read the last tile's rows:
    // their results are never stored)
    const unsigned last = ntiles * 32u - 1u;
    auto rows_of = [&](unsigned g, const float*& r0, const float*& r1) {
        unsigned i0 = g * 128u + (unsigned)lane, i1 = i0 + 64u;
        i0 = i0 < last ? i0 : last; i1 = i1 < last ? i1 : last;
        r0 = a.feat + (size_t)i0 * 32; r1 = a.feat + (size_t)i1 * 32;
    };
    // the next group's inputs, loaded one group ahead: PE waves: four features per pass (nx[0], nx[1]); raw wave: octet 0
    float4 nx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) nx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    // two parts (the register allocator spills a four-register-tuple prefetch that lives across layer 1 or the h1 store, and a
    // spill waits for the load): part 0 = PE features / raw pass 0, part 1 = raw pass 1
    auto prefetch = [&](unsigned g, int part) {
        const float *r0, *r1;
        rows_of(g, r0, r1);
        if (!raw) { if (part == 0) { nx[0] = *reinterpret_cast<const float4*>(r0 + 4 * w); nx[1] = *reinterpret_cast<const float4*>(r1 + 4 * w); } }
        else if (part == 0) { const float4* p0 = reinterpret_cast<const float4*>(r0); nx[0] = p0[0]; nx[2] = p0[1]; }
        else { const float4* p1 = reinterpret_cast<const float4*>(r1); nx[1] = p1[0]; nx[3] = p1[1]; }
    };
    prefetch(blockIdx.x, 0);
    prefetch(blockIdx.x, 1);
    __syncthreads();   // W1 / W2 / bias visible
    uint2* __restrict__ H2l = reinterpret_cast<uint2*>(RH) + (size_t)((((w >> 1) * 2) * 64 + n + 16 * (2 * (w & 1) + (rq >> 1))) * 2 + (rq & 1));
#ifdef T2N_PHASE_TIMING
    unsigned long long phacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tph = __builtin_amdgcn_s_memtime();
#endif

    for (unsigned g = blockIdx.x; g < ngroups; g += gridDim.x) {
        WS_PHASE(15);
        WsEnc E;
        E.amax_u = amax_u;
        E.f[0][0] = nx[0].x; E.f[0][1] = nx[0].y; E.f[0][2] = nx[0].z; E.f[0][3] = nx[0].w;
        E.f[1][0] = nx[1].x; E.f[1][1] = nx[1].y; E.f[1][2] = nx[1].z; E.f[1][3] = nx[1].w;
        E.r[0][0] = nx[0]; E.r[0][1] = nx[2]; E.r[1][0] = nx[1]; E.r[1][1] = nx[3];
        const float *row0, *row1;
        rows_of(g, row0, row1);
        // ---- layer 0: octet 0 encoded up front, then 6 steps (step j multiplies chunks 2j, 2j+1 and encodes octet j + 1) -----------
        f32x4 acc[kWsT];
        {
            const float4 b = *reinterpret_cast<const float4*>(LB + 16 * w + 4 * rq);
#pragma unroll
            for (int t = 0; t < kWsT; ++t) { acc[t][0] = b.x; acc[t][1] = b.y; acc[t][2] = b.z; acc[t][3] = b.w; }
        }
        uint4* __restrict__ dst0 = RH + (((w >> 2) * kWsT + rq) * 2) * 64 + n + 16 * (w & 3);
        if (raw) ws_enc_octet<0, true>(E, dst0, neg1);
        else ws_enc_octet<0, false>(E, dst0, neg1);
        __syncthreads();
        WS_PHASE(0);
        if (raw) {
            ws_l0_step<0, true>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<1, true>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<2, true>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<3, true>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<4, true>(acc, A0, RH, w, lane, E, row0, row1, neg1);
        } else {
            ws_l0_step<0, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<1, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<2, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<3, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
            ws_l0_step<4, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
        }
        WS_PHASE(1);
        ws_l0_step<5, false>(acc, A0, RH, w, lane, E, row0, row1, neg1);
        WS_PHASE(2);
        amax_u = E.amax_u;
        // ---- h0 -> LDS (every wave passed the last ring read at the barrier that ended step 5) ------------------------------------------
#pragma unroll
        for (int t = 0; t < kWsT; ++t) amax = ws_store_quad(acc[t], inv0, H2l + (size_t)t * kWsC1 * 256, neg1, amax);
        __syncthreads();
        WS_PHASE(3);
        // ---- layer 1: two steps of two chunks, A operands from the wave's LDS slice ------------------------------------------------------
        {
            const float4 b = *reinterpret_cast<const float4*>(LB + 128 + 16 * w + 4 * rq);
#pragma unroll
            for (int t = 0; t < kWsT; ++t) { acc[t][0] = b.x; acc[t][1] = b.y; acc[t][2] = b.z; acc[t][3] = b.w; }
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const uint4* __restrict__ hb = RH + (2 * st * 2) * 64 + lane;
            WsA1 A{W1 + ((w * kWsC1 + 2 * st) * 2) * 64 + lane};
            WsB B;
            ws_first_half<128, kWsC1 * 128>(B, hb);
            A.fetch(0);
            WS_FENCE();
            ws_slots<0, 128, kWsC1 * 128, true, false, 0, false>(acc, A, B, hb, E, dst0, neg1);
        }
        __syncthreads();   // h0 reads done
        WS_PHASE(4);
        // the next group's inputs: in flight under the h1 store and layer 2 (issued here, not before layer 1: the layer-1 stream
        // leaves no registers for them, and a spill would wait for the loads)
        const unsigned gn = g + gridDim.x < ngroups ? g + gridDim.x : g;
        prefetch(gn, 0);
#pragma unroll
        for (int t = 0; t < kWsT; ++t) amax = ws_store_quad(acc[t], inv1, H2l + (size_t)t * kWsC1 * 256, neg1, amax);
        __syncthreads();
        WS_PHASE(5);
        prefetch(gn, 1);
        // ---- layer 2: wave w finishes N-tile w (three independent product chains) -----------------------------------------------------------
        {
            f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0;
            if (rq == 0) { const float4 b = *reinterpret_cast<const float4*>(LB + 256); c0[0] = b.x; c0[1] = b.y; c0[2] = b.z; c0[3] = b.w; }
            const uint4* __restrict__ w2 = W2 + lane;
            const uint4* __restrict__ hb = RH + (w * kWsC1 * 2) * 64 + lane;
            uint4 ah[kWsC2], al[kWsC2], bh[kWsC2], bl[kWsC2];
#pragma unroll
            for (int c = 0; c < kWsC2; ++c) { ah[c] = w2[c * 128]; al[c] = w2[c * 128 + 64]; bh[c] = hb[c * 128]; bl[c] = hb[c * 128 + 64]; }
#pragma unroll
            for (int c = 0; c < kWsC2; ++c) {
                c0 = ws_mfma(ah[c], bh[c], c0);
                c1 = ws_mfma(ah[c], bl[c], c1);
                c2 = ws_mfma(al[c], bh[c], c2);
            }
            const unsigned tile = g * 4u + (unsigned)(w >> 1);
            if (tile < ntiles) {
                const uint4 i0 = *reinterpret_cast<const uint4*>(LT), i1 = *reinterpret_cast<const uint4*>(LT + 4);
                const unsigned pre[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
                int li = 0;
                unsigned before = 0u;
#pragma unroll
                for (int l = 0; l < 8; ++l) if (pre[l] <= tile) { li = l + 1; before = pre[l]; }
                const unsigned lbase = (unsigned)li * a.list_cap;
                const unsigned idx = lbase + (tile - before) * 32u + 16u * (unsigned)(w & 1) + (unsigned)n;
                const unsigned count = lbase + LT[8 + li];
                if (rq == 0 && idx < count) {   // output rows 0..2 live in registers 0..2 of lanes 0..15
                    const float r = ((c0[0] + c1[0]) + c2[0]) * inv2, gg = ((c0[1] + c1[1]) + c2[1]) * inv2, b = ((c0[2] + c1[2]) + c2[2]) * inv2;
                    // sigmoid by v_exp_f32 / v_rcp_f32 (1 ulp each: ~2e-7 absolute on a value in (0, 1))
                    a.app_rgb[idx] = make_float4(__builtin_amdgcn_rcpf(1.f + __expf(-r)), __builtin_amdgcn_rcpf(1.f + __expf(-gg)),
                                                 __builtin_amdgcn_rcpf(1.f + __expf(-b)), 0.f);
                }
            }
        }
        WS_PHASE(6);
        __syncthreads();   // hidden-tile reads done before the next group's ring writes
        WS_PHASE(7);
    }
#ifdef T2N_PHASE_TIMING
    if (tid == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_ws_phase[i], phacc[i]);
#endif
    if (__any(!(amax <= kWsRange) || amax_u > __float_as_uint(kWsRange)) && lane == 0) atomicOr(a.range_flag, 1u);
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

// reference column (models/tensorBase.py:11-17,101-104: [features | sin block | cos block], feature-major, octave-minor) of
// layer-0 K index (chunk cc, K-quarter kq, element e); -1: zero padding
__host__ __device__ inline int ws_l0_col(int cc, int kq, int e) {
    const int o = 4 * cc + kq, j = o / 8, wp = o % 8, v = 8 * j + e;
    if (wp == 7) return v < 27 ? v : -1;
    const int F = 4 * wp + v / 12, r = v % 12, q = r >> 1, sc = r & 1;
    return F < 27 ? (sc ? 189 : 27) + F * 6 + q : -1;
}

__global__ __launch_bounds__(256) void k_pack_ws(const WsPackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = kWsW * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    const float s0 = a.scales[0], s1 = a.scales[1], s2 = a.scales[2];
    int g = gid;
    if (g < n0) {   // [wave][chunk][part][lane]: unit 16 wave + (lane & 15), K-quarter lane >> 4
        const int lane = g & 63, part = (g >> 6) & 1, c = (g >> 7) % kWsC0, w = (g >> 7) / kWsC0;
        const int unit = 16 * w + (lane & 15), kq = lane >> 4;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int col = ws_l0_col(c, kq, e); x[e] = col >= 0 ? a.w0[unit * 351 + col] * s0 : 0.f; }
        a.w0p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n0;
    if (g < n1) {
        const int lane = g & 63, part = (g >> 6) & 1, c = (g >> 7) % kWsC1, w = (g >> 7) / kWsC1;
        const int unit = 16 * w + (lane & 15), kq = lane >> 4;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = a.w1[unit * 128 + 32 * c + 8 * kq + e] * s1;
        a.w1p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n1;
    if (g < n2) {
        const int lane = g & 63, part = (g >> 6) & 1, c = g >> 7;
        const int row = lane & 15, kq = lane >> 4;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = row < 3 ? a.w2[row * 128 + 32 * c + 8 * kq + e] * s2 : 0.f;
        a.w2p[g] = make_uint4(ws_pack2(x[0], x[1], part), ws_pack2(x[2], x[3], part), ws_pack2(x[4], x[5], part), ws_pack2(x[6], x[7], part));
        return;
    }
    g -= n2;
    if (g < kWsBias) {   // natural unit order (the 16x16 accumulator holds units 4 (lane >> 4) .. +3 of the wave's slice), scaled
        float b = 0.f;
        if (g < 128) b = a.b0[g] * s0;
        else if (g < 256) b = a.b1[g - 128] * s1;
        else if (g - 256 < 3) b = a.b2[g - 256] * s2;
        a.biasp[g] = b;
    }
}

int ws_pack(t2n_field* f, hipStream_t s) {
    const size_t n0 = (size_t)kWsW * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    if (!f->buf_ws) {
        T2N_HIP(hipMalloc((void**)&f->buf_ws, (n0 + n1 + n2) * 16 + (kWsBias + 8) * 4));
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
    const size_t n0 = (size_t)kWsW * kWsC0 * 2 * 64, n1 = kWsW1, n2 = kWsW2;
    uint4* base = (uint4*)f->buf_ws;
    WsArgs a;
    a.w0 = base; a.w1 = base + n0; a.w2 = base + n0 + n1;
    a.bias = (const float*)(base + n0 + n1 + n2); a.inv_scale = a.bias + kWsBias + 4;
    a.feat = feat; a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.tile_hi = tile_hi; a.app_rgb = app_rgb;
    a.range_flag = range_flag; a.neg1 = -1.f;
    hipLaunchKernelGGL(k_mlp_ws, dim3(256), dim3(512), kWsLds, s, a);
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
