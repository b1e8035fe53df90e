// Sample-stationary appearance head: positional encoding + the 351 -> 128 -> 128 -> 3 MLP + sigmoid for rows of 27 appearance
// features (K2b of the default render path). The features come from the gather + basis kernel (k_app_features, K2a).
//
// Replaces (reference): models/tensorBase.py:11-17 (positional_encoding), :88-109 (MLPRender_Fea_noview.forward).
//
// Mapping (gfx950, wave64): ONE 512-thread workgroup per CU (two waves per SIMD), persistent over rounds of 256 samples. Every
// wave owns ONE 32-sample tile of the appearance lists (two N-tiles of 16) from the feature rows to the colours: the encoded
// inputs, both hidden activations and the output never leave its registers. That works because the C/D layout of
// v_mfma_f32_16x16x32_f16 (lane (n, q): rows 4q .. 4q+3 of column n) IS a B-operand layout (lane (n, q): 8 K-values of column n)
// once the K order of the next layer is permuted accordingly — the permutation is folded into the weight packing (k_pack_ss):
//   layer 0: lane quarter q owns features 7q .. 7q+6 of its samples; its 96 K-values are, per feature, (sin, cos) of octaves 0..5
//            [octaves 0 and 3 by v_sin_f32 / v_cos_f32 on the fraction of f 2^o / (2 pi), the others by double-angle steps], then
//            the 7 raw features, then zeros; K-chunk c (32 values = one MFMA) takes values 8c .. 8c+7 of every quarter.
//   layers 1, 2: K index (chunk c, quarter q, element e) = hidden unit 16 (2c + e / 4) + 4q + e % 4 — what the lane holds of the
//            unit tiles 2c and 2c+1 of the previous layer's accumulators.
// The weights are the A operands. Layer-0 weights (196 KB as hi / lo f16 halves) stream through a three-slot LDS ring of 16-KB
// K-chunks that the eight waves share (each thread carries 32 B of the chunk after next in registers; one barrier per chunk);
// layers 1 and 2 (72 KB) stay resident in LDS. One A fetch (hi + lo, 2 KB per wave) feeds six MFMAs.
//
// fp32 products are three f16 products of hi / lo splits (x = hi + lo, hi = RTZ_f16(x), lo = RTZ_f16(x - hi); the lo*lo term is
// dropped: ~2^-21 relative), fp32 accumulate. Weights are pre-split and scaled by a per-layer power of two chosen from max|W| at
// upload (k_ss_scales); activations are split where they are produced; any activation beyond the f16 range raises a flag that
// makes the caller's exact-fp32 kernel redo the launch.
//
// Why not weight-stationary (the round-2 first form, t2n_mlp_ws.hip): there the samples went through LDS between the layers, so
// every layer boundary was a workgroup barrier around a VALU-only phase (split + store) with the matrix pipe idle: a quarter of
// the kernel. Here the conversion work rides in the MFMA slots of the wave's own stream, and the waves only meet at the ring.
//
// hipcc schedules a region MFMAs first, VALU after. The instruction stream is therefore laid out by hand: a K-chunk is cut into
// 48 SLOTS of one MFMA plus the VALU / LDS work that should issue in its shadow, with a full scheduling fence after every slot.
#include "t2n_device.h"

namespace t2n {
namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

constexpr int kC0 = 12, kC1 = 4, kC2 = 4;     // K-chunks (32 values) of layers 0 / 1 / 2
constexpr int kChunk = 8 * 2 * 64;            // uint4 per K-chunk of layers 0 / 1: [unit tile][part][lane]
constexpr int kRing = 3 * kChunk;
constexpr int kW0 = kC0 * kChunk;
constexpr int kW1 = kC1 * kChunk;
constexpr int kW2 = kC2 * 2 * 64;             // [chunk][part][lane], one unit tile (rows 0..2 live)
constexpr int kBias = 288;                    // floats: layer 0 [128], layer 1 [128], layer 2 [32], scaled
constexpr size_t kLds = (size_t)(kRing + kW1 + kW2) * 16 + kBias * 4 + 16 * 4;   // + the sub-list table
constexpr float kRange = 60000.f;             // |activation| beyond this (f16 max 65504) -> exact-path redo

struct Args {
    const uint4* w0; const uint4* w1; const uint4* w2; const float* bias; const float* inv_scale;   // packed by k_pack_ss
    const float* feat;            // [rows][32] fp32 feature rows (row = tile * 32 + sample), columns >= 27 zero
    const unsigned* counters; unsigned list_cap; int nlists;
    unsigned tile_hi;             // 32-sample tiles [0, min(ntiles, tile_hi)) are this kernel's
    float4* app_rgb;
    unsigned* range_flag;
    float neg1;                   // -1.0f at run time: x - hi stays an FMA with an f16 operand (v_fma_mix_f32, no convert)
};

#define SS_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifdef T2N_PHASE_TIMING
__device__ unsigned long long g_ss_phase[16];
#define SS_PHASE(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); phacc[i] += t_ - tph; tph = t_; } while (0)
#else
#define SS_PHASE(i) do {} while (0)
#endif

__device__ __forceinline__ f32x4 mfma(uint4 a, uint4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}

// ---- one pair of values -> packed hi / lo halves, in three phases of about equal issue time (one phase per slot) -----------
struct Unit { float x0, x1; unsigned hi; };

__device__ __forceinline__ void unit_pack(Unit& U) {
    U.hi = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(U.x0, U.x1));
}
__device__ __forceinline__ void unit_split(const Unit& U, float neg1, unsigned& hi, unsigned& lo) {
    const h2v ph = __builtin_bit_cast(h2v, U.hi);
    const float r0 = fmaf((float)ph[0], neg1, U.x0), r1 = fmaf((float)ph[1], neg1, U.x1);
    hi = U.hi;
    lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
}

// Values V, V+1 of the lane's 96-value layer-0 sequence for one of its samples (f: the sample's features 7q .. 7q+6).
template <int V, int PH>
__device__ __forceinline__ void enc_unit(Unit& U, const float (&f)[7], float neg1, unsigned& hi, unsigned& lo) {
    constexpr bool raw = V >= 84;
    constexpr int q = raw ? 0 : (V % 12) / 2;
    constexpr bool fresh = !raw && (q == 0 || q == 3);   // octaves 1, 2, 4, 5 from the one before
    if constexpr (PH == 0) {
        if constexpr (raw) {
            constexpr int r = V - 84;
            U.x0 = r < 7 ? f[r < 7 ? r : 0] : 0.f;
            U.x1 = r + 1 < 7 ? f[r + 1 < 7 ? r + 1 : 0] : 0.f;
        } else if constexpr (fresh) {
            // f / (2 pi) as th + tl (two-constant product, ~2^-48 relative), then the fraction of its 2^q multiple: sin / cos take
            // revolutions and have period 1 (max abs error 4.2e-7 over |f| <= 3e4, tools/experiments/hw_sincos.hip)
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
        if constexpr (fresh) {
            U.x0 = __builtin_amdgcn_sinf(U.x1);
            U.x1 = __builtin_amdgcn_cosf(U.x1);
        }
        unit_pack(U);
    } else {
        unit_split(U, neg1, hi, lo);
    }
}

struct Enc {
    float f[2][7];           // features 7q .. 7q+6 of the lane's two samples (N-tiles 0 / 1)
    Unit U[2];               // per sample: the unit in flight; between units (sin, cos) of the previous octave
    Unit cu;                 // the conversion fillers' unit in flight
    unsigned ph[4], pl[4];   // packed halves of the operand being built
};

// ---- fillers: the VALU work that rides in a chunk's MFMA slots; run<IDX>() for IDX = 0..23 (unit IDX / 3 of eight, phase IDX % 3:
// units 0..3 build the operand of N-tile 0, 4..7 of N-tile 1), done<P>() once N-tile P's four units are through ------------------
struct NoFill {
    template <int IDX> __device__ __forceinline__ void run() {}
    template <int P> __device__ __forceinline__ void done() {}
};
template <int C>
struct EncFill {   // layer-0 B operands of chunk C -> (Bh, Bl)
    Enc& E; uint4 (&Bh)[2]; uint4 (&Bl)[2]; float neg1;
    template <int IDX> __device__ __forceinline__ void run() {
        constexpr int I = IDX / 3, p = I / 4, u = I % 4;
        enc_unit<8 * C + 2 * u, IDX % 3>(E.U[p], E.f[p], neg1, E.ph[u], E.pl[u]);
    }
    template <int P> __device__ __forceinline__ void done() {
        Bh[P] = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
        Bl[P] = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
    }
};
template <int C>
struct ConvFill {   // relu(acc * inv) of unit tiles 2C, 2C+1 -> the next layer's B operands of chunk C
    const f32x4 (&src)[2][8]; uint4 (&Hh)[2][4]; uint4 (&Hl)[2][4]; Enc& E; float inv, neg1; float& amax;
    template <int IDX> __device__ __forceinline__ void run() {
        constexpr int I = IDX / 3, PH = IDX % 3, p = I / 4, j = I % 4;
        Unit& U = E.cu;   // not E.U: the encoder's (sin, cos) state of chunk 0 lives across layer 2
        if constexpr (PH == 0) {
            const f32x4& a = src[p][2 * C + j / 2];
            U.x0 = fmaxf(a[2 * (j % 2)] * inv, 0.f);
            U.x1 = fmaxf(a[2 * (j % 2) + 1] * inv, 0.f);
            amax = fmaxf(amax, fmaxf(U.x0, U.x1));
        } else if constexpr (PH == 1) {
            unit_pack(U);
        } else {
            unit_split(U, neg1, E.ph[j], E.pl[j]);
        }
    }
    template <int P> __device__ __forceinline__ void done() {
        Hh[P][C] = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
        Hl[P][C] = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
    }
};
template <class F>
__device__ __forceinline__ void fill_all(F& f) {   // unscheduled form (prologue, layer boundaries)
#define SS_U(I) f.template run<3 * (I)>(); f.template run<3 * (I) + 1>(); f.template run<3 * (I) + 2>()
    SS_U(0); SS_U(1); SS_U(2); SS_U(3); f.template done<0>();
    SS_U(4); SS_U(5); SS_U(6); SS_U(7); f.template done<1>();
#undef SS_U
}

// ---- MFMA streams ---------------------------------------------------------------------------------------------------------------
struct AOp { uint4 h, l; };

// Slot M (0..47) of one K-chunk of a 128-unit layer: unit tile M / 6, product (M % 6) / 2 (hi*hi, lo*hi, hi*lo), N-tile M % 2.
// A operands of unit tile u + 1 are fetched from LDS (cur + (u + 1) * 128 [+ 64: lo part]) in tile u's first slot; the last tile
// fetches the first operand pair of the next chunk (nxt). Every second slot carries one filler phase.
template <int M, class Fill>
__device__ __forceinline__ void slots(f32x4 (&acc)[2][8], AOp (&A)[2], const uint4 (&Bh)[2], const uint4 (&Bl)[2],
                                      const uint4* __restrict__ cur, const uint4* __restrict__ nxt, Fill& F) {
    if constexpr (M < 48) {
        constexpr int u = M / 6, k = M % 6, p = k / 2, t = k % 2;
        acc[t][u] = mfma(p == 1 ? A[u & 1].l : A[u & 1].h, p == 2 ? Bl[t] : Bh[t], acc[t][u]);
        if constexpr (k == 0) {
            if constexpr (u < 7) { A[(u + 1) & 1].h = cur[(u + 1) * 128]; A[(u + 1) & 1].l = cur[(u + 1) * 128 + 64]; }
            else { A[0].h = nxt[0]; A[0].l = nxt[64]; }
        }
        if constexpr (M % 2 == 0) F.template run<M / 2>();
        if constexpr (M == 23) F.template done<0>();
        if constexpr (M == 47) F.template done<1>();
        SS_FENCE();
        slots<M + 1>(acc, A, Bh, Bl, cur, nxt, F);
    }
}

// Layer 2, slot M (0..23): chunk M / 6, product (M % 6) / 2 on its own accumulator chain, N-tile M % 2; four filler phases per slot.
template <int M, class F0, class F1, class F2>
__device__ __forceinline__ void slots2(f32x4 (&ch)[3][2], AOp (&A)[2], const uint4 (&Hh)[2][4], const uint4 (&Hl)[2][4],
                                       const uint4* __restrict__ w2, const uint4* __restrict__ nxt, F0& f0, F1& f1, F2& f2) {
    if constexpr (M < 24) {
        constexpr int c = M / 6, k = M % 6, p = k / 2, t = k % 2;
        ch[p][t] = mfma(p == 1 ? A[c & 1].l : A[c & 1].h, p == 2 ? Hl[t][c] : Hh[t][c], ch[p][t]);
        if constexpr (k == 0) {
            if constexpr (c < 3) { A[(c + 1) & 1].h = w2[(c + 1) * 128]; A[(c + 1) & 1].l = w2[(c + 1) * 128 + 64]; }
            else { A[0].h = nxt[0]; A[0].l = nxt[64]; }
        }
        auto fill = [&](auto& f) {
            f.template run<4 * k>(); f.template run<4 * k + 1>(); f.template run<4 * k + 2>(); f.template run<4 * k + 3>();
            if constexpr (k == 2) f.template done<0>();
            if constexpr (k == 5) f.template done<1>();
        };
        if constexpr (c == 0) fill(f0);
        if constexpr (c == 1) fill(f1);
        if constexpr (c == 2) fill(f2);
        SS_FENCE();
        slots2<M + 1>(ch, A, Hh, Hl, w2, nxt, f0, f1, f2);
    }
}

// the layer-0 weight stream: thread tid carries uint4 [tid] and [512 + tid] of a chunk. Buffer loads (descriptor in SGPRs, the
// chunk as a scalar byte offset, the thread as one 32-bit VGPR offset): no 64-bit per-lane pointers kept alive.
struct Stream {
    const uint4* wp; int tid;
    uint4 s0, s1;
    __device__ __forceinline__ void load(int c) {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wp), 0, 0x7fffffff, 0x00020000);
        const u4v a = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, c * (kChunk * 16), 0);
        const u4v b = __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, c * (kChunk * 16) + 8192, 0);
        s0 = make_uint4(a[0], a[1], a[2], a[3]); s1 = make_uint4(b[0], b[1], b[2], b[3]);
    }
    __device__ __forceinline__ void store(uint4* __restrict__ slot) const { slot[tid] = s0; slot[512 + tid] = s1; }
};

__global__ __launch_bounds__(512) void k_mlp_ss(const Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    uint4* __restrict__ RING = lds;
    uint4* __restrict__ W1 = lds + kRing;
    uint4* __restrict__ W2 = W1 + kW1;
    float* __restrict__ LB = reinterpret_cast<float*>(W2 + kW2);
    unsigned* __restrict__ LT = reinterpret_cast<unsigned*>(LB + kBias);   // [0..7] inclusive tile prefix of the sub-lists, [8..15] their counts
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, q = lane >> 4;

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
    const unsigned nrounds = (ntiles + 7u) / 8u;
    if (blockIdx.x >= nrounds) return;
    if (tid < 8) { LT[tid] = tid < a.nlists ? incl : 0xffffffffu; LT[8 + tid] = cnt_l; }   // kept in LDS: no live registers across the loop

    // resident operands: layers 1 / 2 and the biases; ring slots 0 / 1 <- chunks 0 / 1, chunk 2 in flight in registers
    for (int i = tid; i < kW1; i += 512) W1[i] = a.w1[i];
    for (int i = tid; i < kW2; i += 512) W2[i] = a.w2[i];
    for (int i = tid; i < kBias; i += 512) LB[i] = a.bias[i];
    Stream S{a.w0, tid};
    S.load(0); S.store(RING);
    S.load(1); S.store(RING + kChunk);
    S.load(2);
    const float inv0 = a.inv_scale[0], inv1 = a.inv_scale[1], inv2 = a.inv_scale[2];
    const float neg1 = a.neg1;
    float amax = 0.f;          // hidden activations (non-negative, finite inputs)
    unsigned amax_u = 0u;      // raw features: max of the |bit patterns| (orders like |x| and ranks inf / NaN on top)

    // features 7q .. 7q+6 of the lane's two samples of round r (row = 32 (8 r + w) + 16 t + n; tiles past the end re-read the last
    // row: their results are never stored)
    const unsigned last = ntiles * 32u - 1u;
    auto load_feat = [&](unsigned r, float (&f)[2][7]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            unsigned i = (r * 8u + (unsigned)w) * 32u + 16u * (unsigned)t + (unsigned)n;
            i = i < last ? i : last;
            const float* __restrict__ row = a.feat + (size_t)i * 32 + 7 * q;
#pragma unroll
            for (int e = 0; e < 7; ++e) f[t][e] = row[e];
        }
    };
    Enc E;
    float nf[2][7];
    load_feat(blockIdx.x, E.f);
    uint4 Bh[2][2], Bl[2][2];   // [chunk parity][N-tile]
    {
        EncFill<0> f{E, Bh[0], Bl[0], neg1};
        fill_all(f);
    }
    __syncthreads();   // W1 / W2 / bias / ring slots 0, 1 visible
    AOp A[2];
    A[0].h = RING[lane]; A[0].l = RING[64 + lane];
#ifdef T2N_PHASE_TIMING
    unsigned long long phacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tph = __builtin_amdgcn_s_memtime();
#endif

    for (unsigned r = blockIdx.x; r < nrounds; r += gridDim.x) {
        SS_PHASE(15);
        // ---- layer 0: 12 chunks; chunk C multiplies while chunk C + 1 is encoded and chunk C + 2 enters the ring --------------------
        f32x4 acc0[2][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 b = *reinterpret_cast<const float4*>(LB + 16 * u + 4 * q);
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc0[t][u][0] = b.x; acc0[t][u][1] = b.y; acc0[t][u][2] = b.z; acc0[t][u][3] = b.w; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 7; ++e) amax_u = max(amax_u, __float_as_uint(E.f[t][e]) & 0x7fffffffu);
        const unsigned rn = r + gridDim.x < nrounds ? r + gridDim.x : r;
#define SS_L0(C)                                                                                                                  \
        {                                                                                                                         \
            __syncthreads();   /* chunk C + 1 written by every wave; chunk C - 1 read by every wave */                            \
            S.store(RING + ((C + 2) % 3) * kChunk);                                                                               \
            S.load((C + 3) % kC0);                                                                                                \
            if constexpr (C == 2) load_feat(rn, nf);                                                                              \
            const uint4* __restrict__ cur = RING + (C % 3) * kChunk + lane;                                                       \
            const uint4* __restrict__ nxt = C < 11 ? RING + ((C + 1) % 3) * kChunk + lane : W1 + lane;                            \
            if constexpr (C < 11) {                                                                                               \
                EncFill<(C < 11 ? C + 1 : 0)> f{E, Bh[(C + 1) & 1], Bl[(C + 1) & 1], neg1};                                       \
                slots<0>(acc0, A, Bh[C & 1], Bl[C & 1], cur, nxt, f);                                                             \
            } else {                                                                                                              \
                NoFill f;                                                                                                         \
                slots<0>(acc0, A, Bh[C & 1], Bl[C & 1], cur, nxt, f);                                                             \
            }                                                                                                                     \
        }
        SS_L0(0) SS_L0(1) SS_L0(2) SS_L0(3) SS_L0(4) SS_L0(5) SS_L0(6) SS_L0(7) SS_L0(8) SS_L0(9) SS_L0(10) SS_L0(11)
#undef SS_L0
        SS_PHASE(0);
        // ---- h0 -> layer-1 B operands (chunk 0 here, chunks 1..3 in the slots of layer 1); no barrier from here to the next round ---
        uint4 H0h[2][4], H0l[2][4];
        {
            ConvFill<0> f{acc0, H0h, H0l, E, inv0, neg1, amax};
            fill_all(f);
        }
        SS_PHASE(1);
        f32x4 acc1[2][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float4 b = *reinterpret_cast<const float4*>(LB + 128 + 16 * u + 4 * q);
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc1[t][u][0] = b.x; acc1[t][u][1] = b.y; acc1[t][u][2] = b.z; acc1[t][u][3] = b.w; }
        }
#define SS_L1(C)                                                                                                                  \
        {                                                                                                                         \
            const uint4* __restrict__ cur = W1 + C * kChunk + lane;                                                               \
            const uint4* __restrict__ nxt = C < 3 ? W1 + (C + 1) * kChunk + lane : W2 + lane;                                     \
            uint4 xh[2] = {H0h[0][C], H0h[1][C]}, xl[2] = {H0l[0][C], H0l[1][C]};                                                 \
            if constexpr (C < 3) {                                                                                                \
                ConvFill<(C < 3 ? C + 1 : 0)> f{acc0, H0h, H0l, E, inv0, neg1, amax};                                             \
                slots<0>(acc1, A, xh, xl, cur, nxt, f);                                                                           \
            } else {                                                                                                              \
                EncFill<0> f{E, Bh[0], Bl[0], neg1};   /* the next round's first operand */                                       \
                slots<0>(acc1, A, xh, xl, cur, nxt, f);                                                                           \
            }                                                                                                                     \
        }
        SS_L1(0) SS_L1(1) SS_L1(2)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 7; ++e) E.f[t][e] = nf[t][e];
        SS_L1(3)
#undef SS_L1
        SS_PHASE(2);
        // ---- h1 -> layer-2 B operands, layer 2 (three product chains per N-tile), sigmoid, store --------------------------------------
        uint4 H1h[2][4], H1l[2][4];
        {
            ConvFill<0> f{acc1, H1h, H1l, E, inv1, neg1, amax};
            fill_all(f);
        }
        f32x4 ch[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 2; ++t) ch[p][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q == 0) {
            const float4 b = *reinterpret_cast<const float4*>(LB + 256);
#pragma unroll
            for (int t = 0; t < 2; ++t) { ch[0][t][0] = b.x; ch[0][t][1] = b.y; ch[0][t][2] = b.z; ch[0][t][3] = b.w; }
        }
        {
            ConvFill<1> f0{acc1, H1h, H1l, E, inv1, neg1, amax};
            ConvFill<2> f1{acc1, H1h, H1l, E, inv1, neg1, amax};
            ConvFill<3> f2{acc1, H1h, H1l, E, inv1, neg1, amax};
            slots2<0>(ch, A, H1h, H1l, W2 + lane, RING + lane, f0, f1, f2);
        }
        SS_PHASE(3);
        const unsigned tile = r * 8u + (unsigned)w;
        if (tile < ntiles && q == 0) {   // output rows 0..2 live in registers 0..2 of lanes 0..15
            const uint4 i0 = *reinterpret_cast<const uint4*>(LT), i1 = *reinterpret_cast<const uint4*>(LT + 4);
            const unsigned pre[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            int li = 0;
            unsigned before = 0u;
#pragma unroll
            for (int l = 0; l < 8; ++l) if (pre[l] <= tile) { li = l + 1; before = pre[l]; }
            const unsigned lbase = (unsigned)li * a.list_cap;
            const unsigned count = lbase + LT[8 + li];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const unsigned idx = lbase + (tile - before) * 32u + 16u * (unsigned)t + (unsigned)n;
                if (idx < count) {
                    const float rr = ((ch[0][t][0] + ch[2][t][0]) + ch[1][t][0]) * inv2, gg = ((ch[0][t][1] + ch[2][t][1]) + ch[1][t][1]) * inv2,
                                bb = ((ch[0][t][2] + ch[2][t][2]) + ch[1][t][2]) * inv2;
                    // sigmoid by v_exp_f32 / v_rcp_f32 (1 ulp each: ~2e-7 absolute on a value in (0, 1))
                    a.app_rgb[idx] = make_float4(__builtin_amdgcn_rcpf(1.f + __expf(-rr)), __builtin_amdgcn_rcpf(1.f + __expf(-gg)),
                                                 __builtin_amdgcn_rcpf(1.f + __expf(-bb)), 0.f);
                }
            }
        }
        SS_PHASE(4);
    }
#ifdef T2N_PHASE_TIMING
    if (tid == 0) for (int i = 0; i < 16; ++i) atomicAdd(&g_ss_phase[i], phacc[i]);
#endif
    if (__any(!(amax <= kRange) || amax_u > __float_as_uint(kRange)) && lane == 0) atomicOr(a.range_flag, 1u);
}

// ---- operand packing -----------------------------------------------------------------------------------------------------
// scales[l] = 2^k with max|W_l| * 2^k in [2^12, 2^13): hi halves use the top of the f16 range, lo halves stay normal for
// every weight within 2^-11 of the largest one. inv_scale = 1 / scale. [0..2] scale, [4..6] inverse.
struct PackArgs {
    const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    uint4* w0p; uint4* w1p; uint4* w2p; float* biasp; float* scales;
};

__global__ __launch_bounds__(256) void k_ss_scales(const PackArgs a) {
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

__device__ __forceinline__ unsigned pack2(float x0, float x1, int part) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    h2v r;
    if (part == 0) { r[0] = h0; r[1] = h1; }
    else { r[0] = (_Float16)(x0 - (float)h0); r[1] = (_Float16)(x1 - (float)h1); }
    return __builtin_bit_cast(unsigned, r);
}

// reference column (models/tensorBase.py:11-17,101-104: [features | sin block | cos block], feature-major, octave-minor) of
// layer-0 K index (chunk c, K-quarter kq, element e); -1: zero padding
__host__ __device__ inline int l0_col(int c, int kq, int e) {
    const int v = 8 * c + e;
    if (v >= 84) { const int r = v - 84, F = 7 * kq + r; return (r < 7 && F < 27) ? F : -1; }
    const int F = 7 * kq + v / 12, o = (v % 12) >> 1, sc = v & 1;
    return F < 27 ? (sc ? 189 : 27) + F * 6 + o : -1;
}
// hidden unit of layer-1 / layer-2 K index (chunk c, K-quarter kq, element e)
__host__ __device__ inline int hid_unit(int c, int kq, int e) { return 16 * (2 * c + e / 4) + 4 * kq + e % 4; }

__global__ __launch_bounds__(256) void k_pack_ss(const PackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const float s0 = a.scales[0], s1 = a.scales[1], s2 = a.scales[2];
    int g = gid;
    float x[8];
    if (g < kW0 + kW1) {   // [chunk][unit tile][part][lane]: unit 16 u + (lane & 15), K-quarter lane >> 4
        const bool l1 = g >= kW0;
        if (l1) g -= kW0;
        const int lane = g & 63, part = (g >> 6) & 1, u = (g >> 7) & 7, c = g >> 10;
        const int unit = 16 * u + (lane & 15), kq = lane >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (l1) x[e] = a.w1[unit * 128 + hid_unit(c, kq, e)] * s1;
            else { const int col = l0_col(c, kq, e); x[e] = col >= 0 ? a.w0[unit * 351 + col] * s0 : 0.f; }
        }
        (l1 ? a.w1p : a.w0p)[g] = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
        return;
    }
    g -= kW0 + kW1;
    if (g < kW2) {
        const int lane = g & 63, part = (g >> 6) & 1, c = g >> 7;
        const int row = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = row < 3 ? a.w2[row * 128 + hid_unit(c, kq, e)] * s2 : 0.f;
        a.w2p[g] = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
        return;
    }
    g -= kW2;
    if (g < kBias) {   // natural unit order (the 16x16 accumulator holds units 4 (lane >> 4) .. +3 of a unit tile), scaled
        float b = 0.f;
        if (g < 128) b = a.b0[g] * s0;
        else if (g < 256) b = a.b1[g - 128] * s1;
        else if (g - 256 < 3) b = a.b2[g - 256] * s2;
        a.biasp[g] = b;
    }
}

}  // namespace ss

int ss_pack(t2n_field* f, hipStream_t s) {
    using namespace ss;
    const size_t nw = (size_t)kW0 + kW1 + kW2;
    if (!f->buf_ss) {
        T2N_HIP(hipMalloc((void**)&f->buf_ss, nw * 16 + (kBias + 8) * 4));
    }
    uint4* base = (uint4*)f->buf_ss;
    PackArgs a;
    const t2n_field_params& p = f->params_ref;
    a.w0 = p.mlp_w0; a.b0 = p.mlp_b0; a.w1 = p.mlp_w1; a.b1 = p.mlp_b1; a.w2 = p.mlp_w2; a.b2 = p.mlp_b2;
    a.w0p = base; a.w1p = base + kW0; a.w2p = base + kW0 + kW1;
    a.biasp = (float*)(base + nw); a.scales = a.biasp + kBias;
    hipLaunchKernelGGL(k_ss_scales, dim3(1), dim3(256), 0, s, a);
    const size_t total = nw + kBias;
    hipLaunchKernelGGL(k_pack_ss, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    f->ss_dirty = false;
    return T2N_OK;
}

// the head of the appearance stage for tiles [0, tile_hi) of the sub-lists, from feature rows; *range_flag is set when an
// activation left the f16 range (the caller then re-runs the launch on the exact path)
int launch_mlp_ss(t2n_field* f, const float* feat, const unsigned* counters_dev, unsigned list_cap, unsigned tile_hi, float4* app_rgb,
                  unsigned* range_flag, hipStream_t s) {
    using namespace ss;
    if (f->ss_dirty || !f->buf_ss) { const int rc = ss_pack(f, s); if (rc) return rc; }
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        attr_set = true;
    }
    const size_t nw = (size_t)kW0 + kW1 + kW2;
    uint4* base = (uint4*)f->buf_ss;
    Args a;
    a.w0 = base; a.w1 = base + kW0; a.w2 = base + kW0 + kW1;
    a.bias = (const float*)(base + nw); a.inv_scale = a.bias + kBias + 4;
    a.feat = feat; a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.tile_hi = tile_hi; a.app_rgb = app_rgb;
    a.range_flag = range_flag; a.neg1 = -1.f;
    hipLaunchKernelGGL(k_mlp_ss, dim3(256), dim3(512), kLds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

#ifdef T2N_PHASE_TIMING
extern "C" int t2n_debug_ss_phase_read(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(t2n::ss::g_ss_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(t2n::ss::g_ss_phase), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#endif
