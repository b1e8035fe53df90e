// Sample-stationary appearance head: positional encoding + the 351 -> 128 -> 128 -> 3 MLP + sigmoid for rows of 27 appearance
// features (K2b of the default render path). The features come from the gather + basis kernel (k_app_features_p in t2n_shade.hip, K2a).
//
// Replaces (reference): models/tensorBase.py:11-17 (positional_encoding), :88-109 (MLPRender_Fea_noview.forward).
//
// Mapping (gfx950, wave64): ONE 512-thread workgroup per CU (two waves per SIMD), persistent over rounds of 256 samples. Every
// wave owns ONE 32-sample tile of the appearance lists from the feature rows to the colours: the encoded inputs, both hidden
// activations and the output never leave its registers. That works because the C/D layout of v_mfma_f32_32x32x16_f16 (lane
// (j, h): rows 8b + 4h + t of column j) IS a B-operand layout (lane (j, h): 8 K-values of column j) once the K order of the next
// layer is permuted accordingly — the permutation is folded into the weight packing (k_pack_ss):
//   layer 0: lane half h owns features 14h .. 14h+12 of its sample and half of feature 13 (octaves 0..2 / 3..5); its 176 K-values are
//            (sin, cos) of those three octaves, then per own feature (sin, cos) of octaves 0..5 [octaves 0 and 3 by v_sin_f32 /
//            v_cos_f32 on the fraction of f 2^o / (2 pi), the others by double-angle steps], then the 13 raw features, then
//            (half 1) the raw feature 13: 351 values in 352 slots; K-step s (16 values = one MFMA) takes values 8s .. 8s+7 of
//            both halves: 22 K-steps.
//   layers 1, 2: K index (step s, half h, element e) = hidden unit 32 (s / 2) + 8 (2 (s % 2) + e / 4) + 4h + e % 4 — what the lane
//            holds of unit tile s / 2 of the previous layer's accumulators.
// The weights are the A operands. Layer-0 weights (196 KB as hi / lo f16 halves) stream through a three-slot LDS ring of 16-KB
// chunks (two K-steps) that the eight waves share (filled by LDS-DMA two chunks ahead; one barrier per chunk); layers 1 and 2
// (80 KB) stay resident in LDS.
//
// fp32 products are three f16 products of hi / lo splits (x = hi + lo, hi = RTZ_f16(x), lo = RTZ_f16(x - hi); the lo*lo term is
// dropped: ~2^-21 relative), fp32 accumulate. Weights are pre-split and scaled by a per-layer power of two chosen from max|W| at
// upload (k_ss_scales); activations are split where they are produced; any activation beyond the f16 range raises a flag that
// makes the caller's exact-fp32 kernel redo the launch.
//
// Why not weight-stationary (the round-2 first form, 1.51 ms per C2 frame against 1.24 ms: profiles/round2_v16_* / round2_v17_*): there the samples went through LDS between the layers, so
// every layer boundary was a workgroup barrier around a VALU-only phase (split + store) with the matrix pipe idle: a quarter of
// the kernel. Here the conversion work rides in the MFMA slots of the wave's own stream, and the waves only meet at the ring.
// Why 32x32x16 tiles: measured on the 16x16x32 form of this kernel, every non-MFMA instruction of a wave costs its ~4 issue
// cycles on top of the MFMA time (17 cycles per 16x16x32 MFMA + 2.9 other instructions = 29 cycles); the 32-cycle MFMA has room
// for about five such instructions in its shadow, and the same work needs half as many MFMA issues.
//
// hipcc schedules a region MFMAs first, VALU after. The instruction stream is therefore laid out by hand: a K-step is cut into
// 12 SLOTS of one MFMA plus the VALU / LDS work that should issue in its shadow, with a full scheduling fence after every slot.
#include "t2n_device.h"

namespace t2n {
namespace ss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

constexpr int kC0 = 12, kC1 = 4;              // chunks (two K-steps of 16 values) of layers 0 / 1
constexpr int kS2 = 8;                        // K-steps of layer 2
constexpr int kStep = 4 * 2 * 64;             // uint4 per K-step of layers 0 / 1: [unit tile][part][lane]
constexpr int kChunk = 2 * kStep;
constexpr int kRing = 3 * kChunk;
constexpr int kW0 = kC0 * kChunk;
constexpr int kW1 = kC1 * kChunk;
constexpr int kW2 = kS2 * 2 * 64;             // [step][part][lane], one unit tile (rows 0..2 live)
constexpr int kBias = 288;                    // floats: layer 0 [128], layer 1 [128], layer 2 [32], scaled
constexpr size_t kLds = (size_t)(kRing + kW1 + kW2) * 16 + kBias * 4 + 16 * 4;   // + the sub-list table
constexpr float kRange = 59968.f;             // |activation| from here on (an f16 value; f16 max 65504) -> exact-path redo

struct Args {
    const uint4* w0; const uint4* w1; const uint4* w2; const float* bias; const float* inv_scale;   // packed by k_pack_ss
    const float* feat;            // [rows][32] fp32 feature rows (row = tile * 32 + sample): 27 features, the entry's compositing weight, zeros
    const unsigned* counters; unsigned list_cap; int nlists;
    unsigned tile_hi;             // 32-sample tiles [0, min(ntiles, tile_hi)) are this kernel's
    float4* app_rgb;
    unsigned* range_flag;
    float neg1;                   // -1.0f at run time: x - hi stays an FMA with an f16 operand (v_fma_mix_f32, no convert)
};

#define SS_FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
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

// Values V, V+1 of the lane's 176-value layer-0 sequence. Both lane halves run the same code on their own data: values 0..5 the
// half's three octaves of the SHARED feature 13 (octaves 0..2 for half 0, 3..5 for half 1: `hs` = 1 or 8 scales the fresh argument),
// 6..161 the half's own 13 features x 6 octaves x (sin, cos), 162..174 their raw values, 175 the shared feature's raw value
// (half 1; zero for half 0): 27 x 13 = 351 values in 2 x 176 slots = 22 K-steps.
struct EncIn { float f[14]; float fh, extra, hs; };   // f[0..12]: own features 14h + i; f[13]: what column 14h + 13 held (see load_feat)
template <int V, int PH>
__device__ __forceinline__ void enc_unit(Unit& U, const EncIn& I, float neg1, unsigned& hi, unsigned& lo) {
    constexpr bool raw = V >= 162;
    constexpr bool head = V < 6;                          // the shared feature's three octaves
    constexpr int q = raw ? 0 : (head ? V / 2 : ((V - 6) % 12) / 2);
    constexpr bool fresh = !raw && (head ? q == 0 : (q == 0 || q == 3));   // the other octaves from the one before
    if constexpr (PH == 0) {
        if constexpr (raw) {
            constexpr int r = V - 162;
            U.x0 = I.f[r < 13 ? r : 0];
            U.x1 = r + 1 < 13 ? I.f[r + 1 < 13 ? r + 1 : 0] : I.extra;
        } else if constexpr (fresh) {
            // f / (2 pi) as th + tl (two-constant product, ~2^-48 relative), then the fraction of its 2^q multiple: sin / cos take
            // revolutions and have period 1 (max abs error 4.2e-7 over |f| <= 3e4, tools/experiments/hw_sincos.hip)
            const float C1 = 0.15915494309189535f;                                   // float(1 / (2 pi))
            const float C2 = (float)(0.15915494309189533576888 - (double)C1);
            const float sc = head ? I.hs : (float)(1 << q);
            const float x = head ? I.fh : I.f[head ? 0 : (V - 6) / 12];
            const float th = x * C1;
            const float tl = fmaf(x, C1, -th) + x * C2;
            U.x1 = fmaf(tl, sc, __builtin_amdgcn_fractf(th * sc));
        } else {
            const float sn = U.x0, cs = U.x1;   // (sin, cos) of the previous octave of this sample
            const float t = sn + sn;
            U.x0 = t * cs;
            U.x1 = fmaf(-t, sn, 1.f);           // cos 2a = 1 - 2 sin^2 a: absolute error ~6e-8 (what matters: the value multiplies a weight)
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
    EncIn in;                // the lane's features (see enc_unit)
    Unit U;                  // the encoder's unit in flight; between units (sin, cos) of the previous octave
    Unit cu;                 // the conversion fillers' unit in flight (U lives across layer 2: step 0 of the next round is encoded before it)
    unsigned ph[4], pl[4];   // packed halves of the operand being built
};

// ---- fillers: the VALU work that rides in a K-step's MFMA slots; run<IDX>() for IDX = 0..11 (unit IDX / 3 of four, phase IDX % 3),
// done() once the four units are through -----------------------------------------------------------------------------------------
struct NoFill {
    template <int IDX> __device__ __forceinline__ void run() {}
    __device__ __forceinline__ void done() {}
};
template <int S>
struct EncFill {   // layer-0 B operand of K-step S -> (Bh, Bl)
    Enc& E; uint4& Bh; uint4& Bl; float neg1;
    template <int IDX> __device__ __forceinline__ void run() {
        constexpr int u = IDX / 3;
        enc_unit<8 * S + 2 * u, IDX % 3>(E.U, E.in, neg1, E.ph[u], E.pl[u]);
    }
    __device__ __forceinline__ void done() {
        Bh = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
        Bl = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
    }
};
template <int S>
struct ConvFill {   // relu(acc * inv) of registers 8 (S % 2) .. +7 of unit tile S / 2 -> the next layer's B operand of K-step S
    const f32x16 (&src)[4]; uint4 (&Hh)[8]; uint4 (&Hl)[8]; Enc& E; float inv, neg1; h2v& amax;
    template <int IDX> __device__ __forceinline__ void run() {
        constexpr int j = IDX / 3, PH = IDX % 3;
        Unit& U = E.cu;
        if constexpr (PH == 0) {
            const f32x16& a = src[S / 2];
            U.x0 = fmaxf(a[8 * (S % 2) + 2 * j] * inv, 0.f);
            U.x1 = fmaxf(a[8 * (S % 2) + 2 * j + 1] * inv, 0.f);
        } else if constexpr (PH == 1) {
            unit_pack(U);
            amax = __builtin_elementwise_max(amax, __builtin_bit_cast(h2v, U.hi));   // RTZ halves: a value beyond the range packs to 65504
        } else {
            unit_split(U, neg1, E.ph[j], E.pl[j]);
        }
    }
    __device__ __forceinline__ void done() {
        Hh[S] = make_uint4(E.ph[0], E.ph[1], E.ph[2], E.ph[3]);
        Hl[S] = make_uint4(E.pl[0], E.pl[1], E.pl[2], E.pl[3]);
    }
};
template <class F>
__device__ __forceinline__ void fill_all(F& f) {   // unscheduled form (prologue, layer boundaries)
#define SS_U(I) f.template run<3 * (I)>(); f.template run<3 * (I) + 1>(); f.template run<3 * (I) + 2>()
    SS_U(0); SS_U(1); SS_U(2); SS_U(3); f.done();
#undef SS_U
}

// ---- MFMA streams ---------------------------------------------------------------------------------------------------------------
struct AOp { uint4 h, l; };

// Slot M (0..11) of one K-step of a 128-unit layer: unit-tile pair M / 6, product (M % 6) / 2 (hi*hi, lo*hi, hi*lo), tile of the pair
// M % 2 — an accumulator is touched every second slot. The A operands of the next pair (cur + tile * 128 [+ 64: lo part]; the second
// pair fetches the first pair of the next step, nxt) are fetched in a pair's first slot. One filler phase per slot.
struct NoRing { template <int M> __device__ __forceinline__ void run() {} };
template <int M, class Fill, class Ring = NoRing>
__device__ __forceinline__ void slots(f32x16 (&acc)[4], AOp (&A)[2][2], const uint4& Bh, const uint4& Bl,
                                      const uint4* __restrict__ cur, const uint4* __restrict__ nxt, Fill& F, Ring R = Ring()) {
    if constexpr (M < 12) {
        constexpr int g = M / 6, k = M % 6, p = k / 2, i = k % 2, u = 2 * g + i;
        acc[u] = mfma(p == 1 ? A[g][i].l : A[g][i].h, p == 2 ? Bl : Bh, acc[u]);
        if constexpr (k == 0) {
            if constexpr (g == 0) {
                A[1][0].h = cur[2 * 128]; A[1][0].l = cur[2 * 128 + 64]; A[1][1].h = cur[3 * 128]; A[1][1].l = cur[3 * 128 + 64];
            } else {
                A[0][0].h = nxt[0]; A[0][0].l = nxt[64]; A[0][1].h = nxt[128]; A[0][1].l = nxt[128 + 64];
            }
        }
        F.template run<M>();
        if constexpr (M == 11) F.done();
        R.template run<M>();
        SS_FENCE();   // (MFMAs of a tile pair or of a whole K-step issued back to back, fillers behind them: 2-3 % slower)
        slots<M + 1>(acc, A, Bh, Bl, cur, nxt, F, R);
    }
}

// Layer 2, K-step S (0..7): three products, each on its own accumulator chain; four filler phases per slot. The step's operand pair
// sits in A[0][S % 2]; once its three MFMAs are issued that slot takes the operand of step S + 2 (nxt).
template <int S, class Fill>
__device__ __forceinline__ void step2(f32x16 (&ch)[3], AOp (&A)[2][2], const uint4& Hh, const uint4& Hl, const uint4* __restrict__ nxt, Fill& F) {
    constexpr int b = S % 2;
    ch[0] = mfma(A[0][b].h, Hh, ch[0]);
    F.template run<0>(); F.template run<1>(); F.template run<2>(); F.template run<3>();
    SS_FENCE();
    ch[1] = mfma(A[0][b].l, Hh, ch[1]);
    F.template run<4>(); F.template run<5>(); F.template run<6>(); F.template run<7>();
    SS_FENCE();
    ch[2] = mfma(A[0][b].h, Hl, ch[2]);
    A[0][b].h = nxt[0]; A[0][b].l = nxt[64];
    F.template run<8>(); F.template run<9>(); F.template run<10>(); F.template run<11>();
    F.done();
    SS_FENCE();
}

// the layer-0 weight stream: LDS-DMA (buffer_load ... lds: descriptor in SGPRs, the chunk as a scalar byte offset, the thread as one
// 32-bit VGPR offset; the data never touches a VGPR). Wave w moves KB w and 8 + w of a 16-KB chunk, one
// 1-KB piece per instruction (the LDS side of the instruction is wave-linear: M0 base + lane * 16).
struct Stream {
    const uint4* wp; int tid;
    template <int HALF> __device__ __forceinline__ void dma(uint4* __restrict__ slot, int c) const {
        typedef __attribute__((address_space(3))) void* lp;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wp), 0, 0x7fffffff, 0x00020000);
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lp)(slot + HALF * 512 + w * 64), 16, tid * 16, c * (kChunk * 16) + HALF * 8192, 0, 0);
    }
};
// the ring traffic of one chunk iteration rides in the MFMA slots of its first K-step (the piece lands one iteration later:
// hipcc waits vmcnt(0) in front of the next barrier, by which time it has long arrived)
struct RingOps {
    const Stream& S; uint4* __restrict__ slot; int chunk;
    template <int M> __device__ __forceinline__ void run() {
        if constexpr (M == 1) S.template dma<0>(slot, chunk);
        if constexpr (M == 5) S.template dma<1>(slot, chunk);
    }
};

__global__ __launch_bounds__(512) void k_mlp_ss(const Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    // LDS map: W2 [0, 16 KB) | ring [16 KB, 64 KB) | W1 [64 KB, 128 KB) | biases | sub-list table. A DS instruction carries a 16-bit
    // byte offset: with the lane's operand addresses written as TWO opaque bases (lane * 16 and lane * 16 + 64 KB) plus
    // constants every fetch is base + immediate; left to itself hipcc materialises one base register per 64-KB-crossing constant
    // (nine of them) ahead of the loop and spills them.
    uint4* __restrict__ W2 = lds;
    uint4* __restrict__ RING = lds + kW2;
    uint4* __restrict__ W1 = RING + kRing;
    float* __restrict__ LB = reinterpret_cast<float*>(W1 + kW1);
    unsigned* __restrict__ LT = reinterpret_cast<unsigned*>(LB + kBias);   // [0..7] inclusive tile prefix of the sub-lists, [8..15] their counts
    static_assert((kW2 + kRing) * 16 == 65536, "W2 + ring fill the first 64 KB");
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    unsigned ob0 = (unsigned)lane * 16u, ob1 = (unsigned)lane * 16u + 65536u, obb = (unsigned)((kW2 + kRing + kW1) * 16) + 16u * (unsigned)h;
    asm volatile("" : "+v"(ob0));
    asm volatile("" : "+v"(ob1));
    asm volatile("" : "+v"(obb));
    const uint4* __restrict__ LA0 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + ob0);   // W2 / ring, + lane
    const uint4* __restrict__ LA1 = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + ob1);   // W1, + lane
    const float* __restrict__ LBh = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + obb);   // biases, + 4 h

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

    // resident operands: layers 1 / 2 and the biases; ring slots 0 / 1 <- chunks 0 / 1
    for (int i = tid; i < kW1; i += 512) W1[i] = a.w1[i];
    for (int i = tid; i < kW2; i += 512) W2[i] = a.w2[i];
    for (int i = tid; i < kBias; i += 512) LB[i] = a.bias[i];
    const Stream S{a.w0, tid};
    S.dma<0>(RING, 0); S.dma<1>(RING, 0);
    S.dma<0>(RING + kChunk, 1); S.dma<1>(RING + kChunk, 1);
    const float inv0 = a.inv_scale[0], inv1 = a.inv_scale[1], inv2 = a.inv_scale[2];
    const float neg1 = a.neg1;
    h2v amax = {(_Float16)0.f, (_Float16)0.f};   // hidden activations (non-negative, finite inputs), as packed RTZ f16 halves
    unsigned amax_u = 0u;      // raw features: max of the |bit patterns| (orders like |x| and ranks inf / NaN on top)

    // columns 14h .. 14h+13 of the lane's feature row of round r (row = 32 (8 r + w) + j; tiles past the end re-read the last row: their
    // results are never stored): the half's own 13 features + column 13 (the shared feature, half 0) / column 27 (the entry's
    // compositing weight, half 1). finish_feat hands each half what the other one loaded: one lane exchange per round.
    const unsigned last = ntiles * 32u - 1u;
    auto load_feat = [&](unsigned r, float (&f)[14]) {
        unsigned i = (r * 8u + (unsigned)w) * 32u + (unsigned)j;
        i = i < last ? i : last;
        const float2* __restrict__ row = reinterpret_cast<const float2*>(a.feat + (size_t)i * 32 + 14 * h);
#pragma unroll
        for (int e = 0; e < 7; ++e) { const float2 v = row[e]; f[2 * e] = v.x; f[2 * e + 1] = v.y; }
    };
    float wnext = 0.f;   // lanes of half 0: the compositing weight of the sample whose features were loaded last
    auto finish_feat = [&](EncIn& I) {
        const float xs = __shfl_xor(I.f[13], 32);
        I.fh = h ? xs : I.f[13];
        I.extra = h ? I.fh : 0.f;
        wnext = xs;
    };
    Enc E;
    E.in.hs = h ? 8.f : 1.f;
    load_feat(blockIdx.x, E.in.f);
    finish_feat(E.in);
    uint4 Bh[2], Bl[2];   // [K-step parity]
    {
        EncFill<0> f{E, Bh[0], Bl[0], neg1};
        fill_all(f);
    }
    __syncthreads();   // W1 / W2 / bias / ring slots 0, 1 visible
    AOp A[2][2];
    A[0][0].h = LA0[kW2]; A[0][0].l = LA0[kW2 + 64]; A[0][1].h = LA0[kW2 + 128]; A[0][1].l = LA0[kW2 + 128 + 64];

    for (unsigned r = blockIdx.x; r < nrounds; r += gridDim.x) {
        // ---- layer 0: 12 chunks of two K-steps; a step multiplies while the next one is encoded, chunk C + 2 enters the ring -------
        f32x16 acc0[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(LBh + 32 * u + 8 * b);
                acc0[u][4 * b] = v.x; acc0[u][4 * b + 1] = v.y; acc0[u][4 * b + 2] = v.z; acc0[u][4 * b + 3] = v.w;
            }
#pragma unroll
        for (int e = 0; e < 14; ++e) amax_u = max(amax_u, __float_as_uint(E.in.f[e]) & 0x7fffffffu);
        const unsigned rn = r + gridDim.x < nrounds ? r + gridDim.x : r;
        // column 27 of the feature row carries the entry's compositing weight: lane (j, 0) stores it next to the colour, k_composite
        // then reads one array
        const float wgt = wnext;
#define SS_L0(C)                                                                                                                  \
        {                                                                                                                         \
            __syncthreads();   /* chunk C + 1 written by every wave; chunk C - 1 read by every wave */                            \
            RingOps ring{S, RING + ((C + 2) % 3) * kChunk, (C + 2) % kC0};   /* lands before the next barrier */                  \
            const uint4* __restrict__ cur = LA0 + kW2 + (C % 3) * kChunk;                                                         \
            const uint4* __restrict__ nxt = C < 10 ? LA0 + kW2 + ((C + 1) % 3) * kChunk : LA1;                                    \
            /* 22 K-steps: chunks 0..10; chunk 11 of the stream is padding (its iteration only keeps the ring's rhythm) */         \
            if constexpr (C == 11) {                                                                                              \
                S.dma<0>(RING + ((C + 2) % 3) * kChunk, (C + 2) % kC0); S.dma<1>(RING + ((C + 2) % 3) * kChunk, (C + 2) % kC0);   \
            } else {                                                                                                              \
                EncFill<2 * C + 1> f0{E, Bh[1], Bl[1], neg1};                                                                     \
                slots<0>(acc0, A, Bh[0], Bl[0], cur, cur + kStep, f0, ring);                                                      \
                if constexpr (C < 10) {                                                                                           \
                    EncFill<(C < 10 ? 2 * C + 2 : 0)> f1{E, Bh[0], Bl[0], neg1};                                                  \
                    slots<0>(acc0, A, Bh[1], Bl[1], cur + kStep, nxt, f1);                                                        \
                } else {                                                                                                          \
                    NoFill f1;                                                                                                    \
                    slots<0>(acc0, A, Bh[1], Bl[1], cur + kStep, nxt, f1);                                                        \
                }                                                                                                                 \
            }                                                                                                                     \
        }
        SS_L0(0) SS_L0(1) SS_L0(2) SS_L0(3) SS_L0(4) SS_L0(5) SS_L0(6) SS_L0(7) SS_L0(8) SS_L0(9) SS_L0(10) SS_L0(11)
#undef SS_L0
        // ---- h0 -> layer-1 B operands (step 0 here, steps 1..7 in the slots of layer 1); no barrier from here to the next round ------
        uint4 H0h[8], H0l[8];
        {
            ConvFill<0> f{acc0, H0h, H0l, E, inv0, neg1, amax};
            fill_all(f);
        }
        f32x16 acc1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(LBh + 128 + 32 * u + 8 * b);
                acc1[u][4 * b] = v.x; acc1[u][4 * b + 1] = v.y; acc1[u][4 * b + 2] = v.z; acc1[u][4 * b + 3] = v.w;
            }
#define SS_L1(St)                                                                                                                 \
        {                                                                                                                         \
            const uint4* __restrict__ cur = LA1 + St * kStep;                                                                     \
            const uint4* __restrict__ nxt = St < 7 ? LA1 + (St + 1) * kStep : LA0;                                                \
            if constexpr (St < 7) {                                                                                               \
                ConvFill<(St < 7 ? St + 1 : 0)> f{acc0, H0h, H0l, E, inv0, neg1, amax};                                           \
                slots<0>(acc1, A, H0h[St], H0l[St], cur, nxt, f);                                                                 \
            } else {                                                                                                              \
                NoFill f;                                                                                                         \
                slots<0>(acc1, A, H0h[St], H0l[St], cur, nxt, f);                                                                 \
            }                                                                                                                     \
        }
        SS_L1(0) SS_L1(1) SS_L1(2) SS_L1(3) SS_L1(4) SS_L1(5) SS_L1(6) SS_L1(7)
#undef SS_L1
        // ---- h1 -> layer-2 B operands, layer 2 (three product chains), sigmoid, store ---------------------------------------------------
        // (layer 1's last pair fetched W2 steps 0 / 1 as if they were a tile pair: A[0][0] = step 0, A[0][1] = step 1)
        load_feat(rn, E.in.f);                  // the next round's features: in flight under layer 2, encoded in its last step
        uint4 H1h[8], H1l[8];
        {
            ConvFill<0> f{acc1, H1h, H1l, E, inv1, neg1, amax};
            fill_all(f);
        }
        f32x16 ch[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) ch[p][i] = 0.f;
        if (h == 0) {
            const float4 v = *reinterpret_cast<const float4*>(LB + 256);
            ch[0][0] = v.x; ch[0][1] = v.y; ch[0][2] = v.z; ch[0][3] = v.w;
        }
#define SS_L2(St)                                                                                                                 \
        {                                                                                                                         \
            const uint4* __restrict__ nxt = St < 6 ? LA0 + (St + 2) * 128 : LA0 + kW2 + (St - 6) * 128;                           \
            if constexpr (St < 7) {                                                                                               \
                ConvFill<(St < 7 ? St + 1 : 0)> f{acc1, H1h, H1l, E, inv1, neg1, amax};                                           \
                step2<St>(ch, A, H1h[St], H1l[St], nxt, f);                                                                       \
            } else {                                                                                                              \
                EncFill<0> f{E, Bh[0], Bl[0], neg1};   /* the next round's first operand */                                       \
                step2<St>(ch, A, H1h[St], H1l[St], nxt, f);                                                                       \
            }                                                                                                                     \
        }
        SS_L2(0) SS_L2(1) SS_L2(2) SS_L2(3) SS_L2(4) SS_L2(5) SS_L2(6)
        finish_feat(E.in);
        SS_L2(7)
#undef SS_L2
        const unsigned tile = r * 8u + (unsigned)w;
        if (tile < ntiles && h == 0) {   // output rows 0..2 live in registers 0..2 of lanes 0..31
            const uint4 i0 = *reinterpret_cast<const uint4*>(LT), i1 = *reinterpret_cast<const uint4*>(LT + 4);
            const unsigned pre[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
            int li = 0;
            unsigned before = 0u;
#pragma unroll
            for (int l = 0; l < 8; ++l) if (pre[l] <= tile) { li = l + 1; before = pre[l]; }
            const unsigned lbase = (unsigned)li * a.list_cap;
            const unsigned count = lbase + LT[8 + li];
            const unsigned idx = lbase + (tile - before) * 32u + (unsigned)j;
            if (idx < count) {
                const float rr = ((ch[0][0] + ch[2][0]) + ch[1][0]) * inv2, gg = ((ch[0][1] + ch[2][1]) + ch[1][1]) * inv2,
                            bb = ((ch[0][2] + ch[2][2]) + ch[1][2]) * inv2;
                // sigmoid by v_exp_f32 / v_rcp_f32 (1 ulp each: ~2e-7 absolute on a value in (0, 1))
                a.app_rgb[idx] = make_float4(__builtin_amdgcn_rcpf(1.f + __expf(-rr)), __builtin_amdgcn_rcpf(1.f + __expf(-gg)),
                                             __builtin_amdgcn_rcpf(1.f + __expf(-bb)), wgt);
            }
        }
    }
    if (__any(!((float)amax[0] < kRange) || !((float)amax[1] < kRange) || amax_u > __float_as_uint(kRange)) && lane == 0) atomicOr(a.range_flag, 1u);
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
// layer-0 K index (step s, K-half kh, element e); -1: zero padding
__host__ __device__ inline int l0_col(int s, int kh, int e) {
    const int v = 8 * s + e;
    if (v >= 176) return -1;                                                   // K-steps 22, 23 of the stream are padding
    if (v == 175) return kh ? 13 : -1;                                         // the shared feature's raw value rides with half 1
    if (v >= 162) return 14 * kh + (v - 162);                                  // raw own features
    if (v < 6) return ((v & 1) ? 189 : 27) + 13 * 6 + (kh ? 3 : 0) + v / 2;    // shared feature 13: octaves 0..2 (half 0) / 3..5 (half 1)
    const int F = 14 * kh + (v - 6) / 12, o = ((v - 6) % 12) >> 1;
    return ((v & 1) ? 189 : 27) + F * 6 + o;
}
// hidden unit of layer-1 / layer-2 K index (step s, K-half kh, element e)
__host__ __device__ inline int hid_unit(int s, int kh, int e) { return 32 * (s / 2) + 8 * (2 * (s % 2) + e / 4) + 4 * kh + e % 4; }

__global__ __launch_bounds__(256) void k_pack_ss(const PackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const float s0 = a.scales[0], s1 = a.scales[1], s2 = a.scales[2];
    int g = gid;
    float x[8];
    if (g < kW0 + kW1) {   // [step][unit tile][part][lane]: unit 32 u + (lane & 31), K-half lane >> 5
        const bool l1 = g >= kW0;
        if (l1) g -= kW0;
        const int lane = g & 63, part = (g >> 6) & 1, u = (g >> 7) & 3, st = g >> 9;
        const int unit = 32 * u + (lane & 31), kh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (l1) x[e] = a.w1[unit * 128 + hid_unit(st, kh, e)] * s1;
            else { const int col = l0_col(st, kh, e); x[e] = col >= 0 ? a.w0[unit * 351 + col] * s0 : 0.f; }
        }
        (l1 ? a.w1p : a.w0p)[g] = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
        return;
    }
    g -= kW0 + kW1;
    if (g < kW2) {
        const int lane = g & 63, part = (g >> 6) & 1, st = g >> 7;
        const int row = lane & 31, kh = lane >> 5;
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = row < 3 ? a.w2[row * 128 + hid_unit(st, kh, e)] * s2 : 0.f;
        a.w2p[g] = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
        return;
    }
    g -= kW2;
    if (g < kBias) {   // natural unit order (the 32x32 accumulator holds units 8b + 4 (lane >> 5) + t of a unit tile), scaled
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

