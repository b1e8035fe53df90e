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
#ifdef SS3_PROF
    unsigned long long* prof;     // [waves][8] cycle sums (instrumented build)
#endif
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

// =====================================================================================================================================
// Three tiles per wave, one wave per SIMD (k_mlp_ss3): the same layouts, operand packing and arithmetic as k_mlp_ss above, with
// layer 0 (69 % of the MFMAs) of THREE 32-sample tiles running on shared A operands. A 256-thread workgroup per CU; a round is
// 12 tiles (384 samples). Per MFMA that is a third of the A-operand ds_read_b128, two thirds of the ring's LDS-DMA pieces and a
// third of the barriers of the two-waves-per-SIMD form.
//
// Register plan (one wave owns its SIMD's whole 512-register file): the accumulators live in the AccVGPRs, named literally in
// inline asm — the compiler allocates only the architectural half (operands, encoders, conversion units):
//   a[64 t + 16 u .. +15]  layer-0 accumulators of tile t, unit tile u (192 registers); once tile t's layer 1 has consumed them
//                          a[64 t .. 64 t + 47] hold the three product chains of its layer 2
//   layer-1 accumulators of the tile in flight (layers 1 and 2 run tile by tile): 64 ARCHITECTURAL registers (compiler-visible MFMAs)
// Accumulators start from the constant 0 in their first MFMA (srcC = 0); the biases are added where an accumulator is converted
// (one v_fma_f32 instead of the scale multiply), from unscaled copies in LDS.
// A operands go through a ring of four register buffers in consumption order (208 per round: 22 x 4 of layer 0, then per tile
// 8 x 4 of layer 1 and 8 of layer 2); the buffer of element k is refilled with element k + 4 right behind k's last MFMA.
// What the compiler cannot see inside the asm statements, and how it is covered:
//   * MFMA result -> v_accvgpr_read: every read sits >= 9 MFMA issues (>= 288 cycles) behind the last MFMA of its accumulator,
//     except the final read of the layer-2 chains, which waits 16 states (s_nop 15; an 8-pass MFMA needs 12);
//   * a VALU-written B operand -> MFMA: the operands are finished >= 2 slots ahead; the first MFMA of a K-step carries s_nop 1;
//   * the compiler keeps nothing in the AccVGPRs: every MFMA statement clobbers all 256 of them.
#define SS3_AGPRS \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", \
    "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", \
    "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", \
    "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", \
    "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", \
    "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", \
    "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", \
    "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", \
    "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
    "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", \
    "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", \
    "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", \
    "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", \
    "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", \
    "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", \
    "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", \
    "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"

typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef short s2v __attribute__((ext_vector_type(2)));
constexpr int kT3 = 3;                                   // tiles per wave
constexpr int kE0 = 88, kEL = 40, kER = kE0 + kT3 * kEL;   // A elements per round: layer 0, per tile (32 + 8), all
static_assert(kER % 4 == 0, "the A ring keeps its phase from round to round");

template <int ACC, bool ZERO, bool NOP>
__device__ __forceinline__ void mfma_acc(const u4v& a, const u4v& b) {
    if constexpr (ZERO) {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, 0" :: "n"(ACC), "n"(ACC + 15), "v"(a), "v"(b) : SS3_AGPRS);
        else asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, 0" :: "n"(ACC), "n"(ACC + 15), "v"(a), "v"(b) : SS3_AGPRS);
    } else {
        if constexpr (NOP) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(ACC), "n"(ACC + 15), "v"(a), "v"(b) : SS3_AGPRS);
        else asm volatile("v_mfma_f32_32x32x16_f16 a[%c0:%c1], %2, %3, a[%c0:%c1]" :: "n"(ACC), "n"(ACC + 15), "v"(a), "v"(b) : SS3_AGPRS);
    }
}
template <int R>
__device__ __forceinline__ float acc_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "n"(R));
    return x;
}

struct AEl { u4v h, l; };
// element K of the round's A stream -> LDS operand address (uint4 units; base 0: W2 / ring, base 1: W1)
template <int K> struct ElAddr {
    static constexpr int k = K % kER;
    static constexpr bool l0 = k < kE0;
    static constexpr int q = l0 ? 0 : (k - kE0) % kEL;
    static constexpr bool l1 = !l0 && q < 32;
    static constexpr int s = l0 ? k / 4 : (l1 ? q / 4 : q - 32), ut = l0 ? k % 4 : (l1 ? q % 4 : 0);
    static constexpr int off = l0 ? kW2 + ((s / 2) % 3) * kChunk + (s % 2) * kStep + ut * 128 : (l1 ? s * kStep + ut * 128 : s * 128);
};
struct LdsA { const u4v* a0; const u4v* a1; };
template <int K>
__device__ __forceinline__ void fetch(AEl (&A)[4], const LdsA& L) {
    if constexpr (K >= kER) return;   // the next round fetches its first four elements itself (32 registers less across the round boundary)
#ifdef SS3_ABL_NO_AFETCH
    if (K >= 4) return;
#endif
    using E = ElAddr<K>;
    const u4v* __restrict__ p = (E::l1 ? L.a1 : L.a0) + E::off;
    A[K % 4].h = p[0]; A[K % 4].l = p[64];
}

struct Enc3 {   // a tile's encoder (see Enc)
    EncIn in; Unit U; unsigned ph[4], pl[4];
};
template <int S>
struct EncFill3 {   // layer-0 B operand of K-step S of one tile -> (Bh, Bl)
    Enc3& E; u4v& Bh; u4v& Bl; float neg1;
    template <int IDX> __device__ __forceinline__ void run() {
#ifdef SS3_ABL_NO_ENC
        if (S > 0) return;
#endif
        constexpr int u = IDX / 3;
        enc_unit<8 * S + 2 * u, IDX % 3>(E.U, E.in, neg1, E.ph[u], E.pl[u]);
    }
    __device__ __forceinline__ void done() {
#ifdef SS3_ABL_NO_ENC
        if (S > 0) return;
#endif
        Bh = u4v{E.ph[0], E.ph[1], E.ph[2], E.ph[3]};
        Bl = u4v{E.pl[0], E.pl[1], E.pl[2], E.pl[3]};
    }
};
struct Conv3 {   // the conversion fillers' state (one tile at a time)
    Unit cu; unsigned ph[4], pl[4];
    float4 bq[2][2];   // [K-step parity][half]: the biases of the eight units a step converts
};
// biases of conversion step S (units 32 (S / 2) + 8 b + 4 h + t, b = 2 (S % 2), 2 (S % 2) + 1) -> bq[S % 2]; issued a step ahead
template <int S, int BOFF>
__device__ __forceinline__ void conv_bias(Conv3& V, const float* __restrict__ LBh) {
    V.bq[S % 2][0] = *reinterpret_cast<const float4*>(LBh + BOFF + 32 * (S / 2) + 16 * (S % 2));
    V.bq[S % 2][1] = *reinterpret_cast<const float4*>(LBh + BOFF + 32 * (S / 2) + 16 * (S % 2) + 8);
}
template <int S, int BASE, int BOFF>
struct ConvFill3 {   // relu(acc * inv + bias) of registers 8 (S % 2) .. +7 of unit tile S / 2 -> B operand of K-step S; the accumulators are
                     // AccVGPRs from BASE (>= 0) or the architectural registers src[4] (BASE < 0)
    u4v (&Hh)[8]; u4v (&Hl)[8]; Conv3& V; float inv, neg1; s2v& amax; const float* __restrict__ LBh; const f32x16* src;
    template <int IDX> __device__ __forceinline__ void run() {
#ifdef SS3_ABL_NO_CONV
        if (S > 0) return;
#endif
        constexpr int j = IDX / 3, PH = IDX % 3;
        Unit& U = V.cu;
        if constexpr (PH == 0) {
            float a0, a1;
            if constexpr (BASE >= 0) {
                constexpr int reg = BASE + 16 * (S / 2) + 8 * (S % 2) + 2 * j;
                a0 = acc_read<reg>(); a1 = acc_read<reg + 1>();
            } else {
                a0 = src[S / 2][8 * (S % 2) + 2 * j]; a1 = src[S / 2][8 * (S % 2) + 2 * j + 1];
            }
            const float4& b = V.bq[S % 2][j / 2];
            U.x0 = fmaxf(fmaf(a0, inv, (j % 2) ? b.z : b.x), 0.f);
            U.x1 = fmaxf(fmaf(a1, inv, (j % 2) ? b.w : b.y), 0.f);
        } else if constexpr (PH == 1) {
            unit_pack(U);
            amax = __builtin_elementwise_max(amax, __builtin_bit_cast(s2v, U.hi));   // bit patterns of non-negative halves order like their values
            if constexpr (IDX == 1 && S < 7) conv_bias<(S < 7 ? S + 1 : 0), BOFF>(V, LBh);
        } else {
            unit_split(U, neg1, V.ph[j], V.pl[j]);
        }
    }
    __device__ __forceinline__ void done() {
#ifdef SS3_ABL_NO_CONV
        if (S > 0) { Hh[S] = Hh[0]; Hl[S] = Hl[0]; return; }
#endif
        Hh[S] = u4v{V.ph[0], V.ph[1], V.ph[2], V.ph[3]};
        Hl[S] = u4v{V.pl[0], V.pl[1], V.pl[2], V.pl[3]};
        asm volatile("" : "+v"(amax));   // pinned to its K-step: left alone, hipcc defers the maxima to the next round and spills the halves for it
    }
};

// the layer-0 weight stream for four waves: wave w moves KB 4 p + w of a 16-KB chunk in piece p
struct Stream3 {
    const uint4* wp; int tid;
    template <int P> __device__ __forceinline__ void dma(uint4* __restrict__ slot, int c) const {
        typedef __attribute__((address_space(3))) void* lp;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wp), 0, 0x7fffffff, 0x00020000);
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lp)(slot + P * 256 + w * 64), 16, tid * 16, c * (kChunk * 16) + P * 4096, 0, 0);
    }
};
struct RingOps3 {
    const Stream3& S; uint4* __restrict__ slot; int chunk;
    template <int M> __device__ __forceinline__ void run() {
#ifdef SS3_ABL_NO_DMA
        return;
#endif
        if (chunk == kC0 - 1) return;   // the stream's padding chunk (K-steps 22, 23) is never read
        if constexpr (M == 1) S.template dma<0>(slot, chunk);
        if constexpr (M == 10) S.template dma<1>(slot, chunk);
        if constexpr (M == 19) S.template dma<2>(slot, chunk);
        if constexpr (M == 28) S.template dma<3>(slot, chunk);
    }
    __device__ __forceinline__ void all() { run<1>(); run<10>(); run<19>(); run<28>(); }
};

// Slot M (0..35) of K-step S of layer 0: unit tile M / 9, product (M % 9) / 3 (hi*hi, lo*hi, hi*lo), tile M % 3 — an accumulator is
// touched every third slot; tile t's encoder runs phase M / 3 in the slots M % 3 == t.
template <int M, int S, class F, class R>
__device__ __forceinline__ void l0_slots(AEl (&A)[4], const u4v (&Bh)[kT3], const u4v (&Bl)[kT3], F (&f)[kT3], R ring, const LdsA& L) {
    if constexpr (M < 36) {
        constexpr int ut = M / 9, p = (M % 9) / 3, t = M % 3, k = 4 * S + ut;
        mfma_acc<64 * t + 16 * ut, (S == 0 && p == 0), (M == 0)>(p == 1 ? A[k % 4].l : A[k % 4].h, p == 2 ? Bl[t] : Bh[t]);
        if constexpr (M % 9 == 8) fetch<k + 4>(A, L);
        f[t].template run<M / 3>();
        if constexpr (M / 3 == 11) f[t].done();
        ring.template run<M>();
        SS_FENCE();
        l0_slots<M + 1, S>(A, Bh, Bl, f, ring, L);
    }
}
// Slot M (0..11) of a K-step of layer 1 of the tile in flight: unit tile M / 3, product M % 3; K0 = the A element of unit tile 0
template <int M, int K0, bool FIRST, class F>
__device__ __forceinline__ void l1_slots(f32x16 (&acc1)[4], AEl (&A)[4], const u4v& Hh, const u4v& Hl, F& f, const LdsA& L) {
    if constexpr (M < 12) {
        constexpr int ut = M / 3, p = M % 3, k = K0 + ut;
        // (architectural accumulators, still through asm: a compiler-visible MFMA here made hipcc stage values in AccVGPRs of its own choice)
        const u4v& av = p == 1 ? A[k % 4].l : A[k % 4].h;
        const u4v& bv = p == 2 ? Hl : Hh;
        if constexpr (FIRST && p == 0) {
            if constexpr (M == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc1[ut]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc1[ut]) : "v"(av), "v"(bv));
        } else {
            if constexpr (M == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1[ut]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc1[ut]) : "v"(av), "v"(bv));
        }
        if constexpr (p == 2) fetch<k + 4>(A, L);
        f.template run<M>();
        if constexpr (M == 11) f.done();
        SS_FENCE();
        l1_slots<M + 1, K0, FIRST>(acc1, A, Hh, Hl, f, L);
    }
}
// Layer 2, K-step of A element K: three products on their own chains a[BASE + 16 p ..]; four filler phases per slot
template <int K, int BASE, bool FIRST, class F>
__device__ __forceinline__ void l2_step(AEl (&A)[4], const u4v& Hh, const u4v& Hl, F& f, const LdsA& L) {
    mfma_acc<BASE, FIRST, true>(A[K % 4].h, Hh);
    f.template run<0>(); f.template run<1>(); f.template run<2>(); f.template run<3>();
    SS_FENCE();
    mfma_acc<BASE + 16, FIRST, false>(A[K % 4].l, Hh);
    f.template run<4>(); f.template run<5>(); f.template run<6>(); f.template run<7>();
    SS_FENCE();
    mfma_acc<BASE + 32, FIRST, false>(A[K % 4].h, Hl);
    fetch<K + 4>(A, L);
    f.template run<8>(); f.template run<9>(); f.template run<10>(); f.template run<11>();
    f.done();
    SS_FENCE();
}

#ifdef SS3_ABL_NO_BARRIER
#define SS3_BARRIER() do {} while (0)
#elif defined(SS3_PROF)
#define SS3_BARRIER() do { const unsigned long long tb = __builtin_readcyclecounter(); __syncthreads(); p_bar += __builtin_readcyclecounter() - tb; } while (0)
#else
#define SS3_BARRIER() __syncthreads()
#endif
#ifdef SS3_PROF
#define SS3_T(var) do { const unsigned long long tn = __builtin_readcyclecounter(); var += tn - p_t; p_t = tn; } while (0)
#else
#define SS3_T(var) do {} while (0)
#endif
// the round's last layer-2 step: slot p carries all twelve encoder phases of tile p's first operand of the next round
template <int K, int BASE, class F>
__device__ __forceinline__ void l2_last(AEl (&A)[4], const u4v& Hh, const u4v& Hl, F (&f)[kT3], const LdsA& L) {
    mfma_acc<BASE, false, true>(A[K % 4].h, Hh);
    fill_all(f[0]);
    SS_FENCE();
    mfma_acc<BASE + 16, false, false>(A[K % 4].l, Hh);
    fill_all(f[1]);
    SS_FENCE();
    mfma_acc<BASE + 32, false, false>(A[K % 4].h, Hl);
    fetch<K + 4>(A, L);
    fill_all(f[2]);
    SS_FENCE();
}

__global__ __launch_bounds__(256) void k_mlp_ss3(const Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    // LDS map as k_mlp_ss: W2 [0, 16 KB) | ring [16 KB, 64 KB) | W1 [64 KB, 128 KB) | biases (UNSCALED here) | sub-list table
    uint4* __restrict__ W2 = lds;
    uint4* __restrict__ RING = lds + kW2;
    uint4* __restrict__ W1 = RING + kRing;
    float* __restrict__ LB = reinterpret_cast<float*>(W1 + kW1);
    unsigned* __restrict__ LT = reinterpret_cast<unsigned*>(LB + kBias);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    unsigned ob0 = (unsigned)lane * 16u, ob1 = (unsigned)lane * 16u + 65536u, obb = (unsigned)((kW2 + kRing + kW1) * 16) + 16u * (unsigned)h;
    asm volatile("" : "+v"(ob0));
    asm volatile("" : "+v"(ob1));
    asm volatile("" : "+v"(obb));
    const LdsA L{reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(lds) + ob0), reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(lds) + ob1)};
    const float* __restrict__ LBh = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds) + obb);

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
    const unsigned nrounds = (ntiles + 11u) / 12u;
    if (blockIdx.x >= nrounds) return;
    if (tid < 8) { LT[tid] = tid < a.nlists ? incl : 0xffffffffu; LT[8 + tid] = cnt_l; }

    const float inv0 = a.inv_scale[0], inv1 = a.inv_scale[1], inv2 = a.inv_scale[2];
    for (int i = tid; i < kW1; i += 256) W1[i] = a.w1[i];
    for (int i = tid; i < kW2; i += 256) W2[i] = a.w2[i];
    for (int i = tid; i < kBias; i += 256) LB[i] = a.bias[i] * (i < 128 ? inv0 : (i < 256 ? inv1 : inv2));   // power-of-two scales: exact
    const Stream3 S{a.w0, tid};
    S.dma<0>(RING, 0); S.dma<1>(RING, 0); S.dma<2>(RING, 0); S.dma<3>(RING, 0);
    S.dma<0>(RING + kChunk, 1); S.dma<1>(RING + kChunk, 1); S.dma<2>(RING + kChunk, 1); S.dma<3>(RING + kChunk, 1);
    const float neg1 = a.neg1;
    s2v amax = {0, 0};   // hidden activations: max of the packed RTZ halves' bit patterns (non-negative values; -0 and NaN never reach here)
    unsigned amax_u = 0u;

    // rows of round r: tile 12 r + 3 w + t, row 32 tile + j (see k_mlp_ss::load_feat)
    const unsigned last = ntiles * 32u - 1u;
    auto load_feat = [&](unsigned r, int t, float (&f)[14]) {
        unsigned i = ((r * 4u + (unsigned)w) * 3u + (unsigned)t) * 32u + (unsigned)j;
        i = i < last ? i : last;
        const float2* __restrict__ row = reinterpret_cast<const float2*>(a.feat + (size_t)i * 32 + 14 * h);
#pragma unroll
        for (int e = 0; e < 7; ++e) { const float2 v = row[e]; f[2 * e] = v.x; f[2 * e + 1] = v.y; }
    };
    float wnext[kT3] = {0.f, 0.f, 0.f};
    auto finish_feat = [&](EncIn& I, float& wn) {
        const float xs = __shfl_xor(I.f[13], 32);
        I.fh = h ? xs : I.f[13];
        I.extra = h ? I.fh : 0.f;
        wn = xs;
    };
    Enc3 E[kT3];
    u4v Bh[2][kT3], Bl[2][kT3];   // [K-step parity][tile]
#pragma unroll
    for (int t = 0; t < kT3; ++t) {
        E[t].in.hs = h ? 8.f : 1.f;
        load_feat(blockIdx.x, t, E[t].in.f);
    }
#pragma unroll
    for (int t = 0; t < kT3; ++t) {
        finish_feat(E[t].in, wnext[t]);
        EncFill3<0> f{E[t], Bh[0][t], Bl[0][t], neg1};
        fill_all(f);
    }
    __syncthreads();   // W1 / W2 / bias / ring slots 0, 1 visible
    AEl A[4];
    Conv3 V;

#ifdef SS3_PROF
    unsigned long long p_bar = 0, p_l0 = 0, p_tile[3] = {0, 0, 0}, p_top = 0, p_rounds = 0;
    const unsigned long long p_start = __builtin_readcyclecounter();
    unsigned long long p_t = p_start;
#endif
    for (unsigned r = blockIdx.x; r < nrounds; r += gridDim.x) {
        float wgt[kT3];
#pragma unroll
        for (int t = 0; t < kT3; ++t) {
#pragma unroll
            for (int e = 0; e < 14; ++e) amax_u = max(amax_u, __float_as_uint(E[t].in.f[e]) & 0x7fffffffu);
            wgt[t] = wnext[t];
        }
        asm volatile("" : "+v"(amax_u));   // pinned here: left alone, hipcc sinks these maxima to the end of the round and keeps (spills) the 42 features for it
        const unsigned rn = r + gridDim.x < nrounds ? r + gridDim.x : r;
        SS3_T(p_top);
        // ---- layer 0, three tiles on shared A operands: 11 chunks of two K-steps (+ the ring's padding chunk) ----------------------------
#define SS3_ENC(S_, P_) {{E[0], Bh[P_][0], Bl[P_][0], neg1}, {E[1], Bh[P_][1], Bl[P_][1], neg1}, {E[2], Bh[P_][2], Bl[P_][2], neg1}}
#define SS3_L0(C)                                                                                                                 \
        {                                                                                                                         \
            SS3_BARRIER();   /* chunk C + 1 written by every wave; chunk C - 1 read by every wave */                              \
            if constexpr (C == 0) { fetch<0>(A, L); fetch<1>(A, L); fetch<2>(A, L); fetch<3>(A, L); }                             \
            RingOps3 ring{S, RING + ((C + 2) % 3) * kChunk, (C + 2) % kC0};                                                       \
            if constexpr (C == 11) {                                                                                              \
                ring.all();                                                                                                       \
            } else {                                                                                                              \
                {                                                                                                                 \
                    EncFill3<2 * C + 1> f0[kT3] = SS3_ENC(2 * C + 1, 1);                                                          \
                    l0_slots<0, 2 * C>(A, Bh[0], Bl[0], f0, ring, L);                                                             \
                }                                                                                                                 \
                if constexpr (C < 10) {                                                                                           \
                    EncFill3<(C < 10 ? 2 * C + 2 : 0)> f1[kT3] = SS3_ENC(2 * C + 2, 0);                                           \
                    l0_slots<0, 2 * C + 1>(A, Bh[1], Bl[1], f1, NoRing(), L);                                                     \
                } else {                                                                                                          \
                    NoFill f1[kT3];                                                                                               \
                    conv_bias<0, 0>(V, LBh);   /* tile 0's first conversion step */                                               \
                    l0_slots<0, 2 * C + 1>(A, Bh[1], Bl[1], f1, NoRing(), L);                                                     \
                }                                                                                                                 \
            }                                                                                                                     \
        }
#ifndef SS3_ABL_TAIL_ONLY
        SS3_L0(0) SS3_L0(1) SS3_L0(2) SS3_L0(3) SS3_L0(4) SS3_L0(5) SS3_L0(6) SS3_L0(7) SS3_L0(8) SS3_L0(9) SS3_L0(10) SS3_L0(11)
#endif
#undef SS3_L0
        SS3_T(p_l0);

        // ---- layers 1 and 2, tile by tile -------------------------------------------------------------------------------------------------
#define SS3_L1(T_, St)                                                                                                            \
        {                                                                                                                         \
            if constexpr (St < 7) {                                                                                               \
                ConvFill3<(St < 7 ? St + 1 : 0), 64 * T_, 0> f{H0h, H0l, V, inv0, neg1, amax, LBh, nullptr};                      \
                l1_slots<0, kE0 + kEL * T_ + 4 * St, St == 0>(acc1, A, H0h[St], H0l[St], f, L);                                   \
            } else {                                                                                                              \
                NoFill f;                                                                                                         \
                conv_bias<0, 128>(V, LBh);                                                                                        \
                l1_slots<0, kE0 + kEL * T_ + 4 * St, St == 0>(acc1, A, H0h[St], H0l[St], f, L);                                   \
            }                                                                                                                     \
        }
#define SS3_L2(T_, St)                                                                                                            \
        {                                                                                                                         \
            if constexpr (St < 7) {                                                                                               \
                ConvFill3<(St < 7 ? St + 1 : 0), -1, 128> f{H1h, H1l, V, inv1, neg1, amax, LBh, acc1};                            \
                l2_step<kE0 + kEL * T_ + 32 + St, 64 * T_, St == 0>(A, H1h[St], H1l[St], f, L);                                   \
            } else if constexpr (T_ + 1 < kT3) {                                                                                  \
                NoFill f;                                                                                                         \
                conv_bias<0, 0>(V, LBh);   /* the next tile's first conversion step */                                            \
                l2_step<kE0 + kEL * T_ + 32 + St, 64 * T_, St == 0>(A, H1h[St], H1l[St], f, L);                                   \
            } else {   /* the round's last step: the next round's first operands of the three tiles, one tile per slot */        \
                _Pragma("unroll") for (int t = 0; t < kT3; ++t) finish_feat(E[t].in, wnext[t]);                                   \
                EncFill3<0> f3[kT3] = SS3_ENC(0, 0);                                                                              \
                l2_last<kE0 + kEL * T_ + 32 + St, 64 * T_>(A, H1h[St], H1l[St], f3, L);                                           \
            }                                                                                                                     \
            /* the next round's features (needed from its first encoder phase to its last layer-0 step): loaded here, not */     \
            /* earlier — 42 registers that nothing in the tile-by-tile part needs */                                             \
            if constexpr (T_ + 1 == kT3 && St == 0) { _Pragma("unroll") for (int t = 0; t < kT3; ++t) load_feat(rn, t, E[t].in.f); } \
        }
#define SS3_TILE(T_)                                                                                                              \
        {                                                                                                                         \
            u4v H0h[8], H0l[8], H1h[8], H1l[8];                                                                                   \
            f32x16 acc1[4];                                                                                                       \
            {                                                                                                                     \
                ConvFill3<0, 64 * T_, 0> f{H0h, H0l, V, inv0, neg1, amax, LBh, nullptr};                                          \
                fill_all(f);                                                                                                      \
            }                                                                                                                     \
            SS3_L1(T_, 0) SS3_L1(T_, 1) SS3_L1(T_, 2) SS3_L1(T_, 3) SS3_L1(T_, 4) SS3_L1(T_, 5) SS3_L1(T_, 6) SS3_L1(T_, 7)       \
            {                                                                                                                     \
                ConvFill3<0, -1, 128> f{H1h, H1l, V, inv1, neg1, amax, LBh, acc1};                                                \
                fill_all(f);                                                                                                      \
            }                                                                                                                     \
            SS3_L2(T_, 0) SS3_L2(T_, 1) SS3_L2(T_, 2) SS3_L2(T_, 3) SS3_L2(T_, 4) SS3_L2(T_, 5) SS3_L2(T_, 6) SS3_L2(T_, 7)       \
            asm volatile("s_nop 15");   /* the chains' last MFMAs -> their reads */                                              \
            const float c00 = acc_read<64 * T_>(), c01 = acc_read<64 * T_ + 1>(), c02 = acc_read<64 * T_ + 2>();                  \
            const float c10 = acc_read<64 * T_ + 16>(), c11 = acc_read<64 * T_ + 17>(), c12 = acc_read<64 * T_ + 18>();           \
            const float c20 = acc_read<64 * T_ + 32>(), c21 = acc_read<64 * T_ + 33>(), c22 = acc_read<64 * T_ + 34>();           \
            const unsigned tile = (r * 4u + (unsigned)w) * 3u + T_;                                                               \
            if (tile < ntiles && h == 0) {   /* output rows 0..2 live in registers 0..2 of lanes 0..31 */                         \
                const uint4 i0 = *reinterpret_cast<const uint4*>(LT), i1 = *reinterpret_cast<const uint4*>(LT + 4);               \
                const unsigned pre[8] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};                                         \
                int li = 0;                                                                                                       \
                unsigned before = 0u;                                                                                             \
                _Pragma("unroll") for (int l = 0; l < 8; ++l) if (pre[l] <= tile) { li = l + 1; before = pre[l]; }                \
                const unsigned lbase = (unsigned)li * a.list_cap;                                                                 \
                const unsigned count = lbase + LT[8 + li];                                                                        \
                const unsigned idx = lbase + (tile - before) * 32u + (unsigned)j;                                                 \
                if (idx < count) {                                                                                                \
                    const float4 b2 = *reinterpret_cast<const float4*>(LB + 256);                                                 \
                    const float rr = fmaf((c00 + c20) + c10, inv2, b2.x), gg = fmaf((c01 + c21) + c11, inv2, b2.y),               \
                                bb = fmaf((c02 + c22) + c12, inv2, b2.z);                                                         \
                    a.app_rgb[idx] = make_float4(__builtin_amdgcn_rcpf(1.f + __expf(-rr)), __builtin_amdgcn_rcpf(1.f + __expf(-gg)), \
                                                 __builtin_amdgcn_rcpf(1.f + __expf(-bb)), wgt[T_]);                              \
                }                                                                                                                 \
            }                                                                                                                     \
        }
#ifndef SS3_ABL_L0_ONLY
#ifdef SS3_PROF
        SS3_TILE(0) SS3_T(p_tile[0]); SS3_TILE(1) SS3_T(p_tile[1]); SS3_TILE(2) SS3_T(p_tile[2]); ++p_rounds;
#else
        SS3_TILE(0) SS3_TILE(1) SS3_TILE(2)
#endif
#endif
#undef SS3_TILE
#undef SS3_L2
#undef SS3_L1
#undef SS3_ENC
    }
#ifdef SS3_PROF
    if (lane == 0 && a.prof) {
        unsigned long long* o = a.prof + (size_t)(blockIdx.x * 4 + w) * 8;
        o[0] = p_bar; o[1] = p_l0; o[2] = p_tile[0]; o[3] = p_tile[1]; o[4] = p_tile[2]; o[5] = p_top; o[6] = p_rounds;
        o[7] = __builtin_readcyclecounter() - p_start;
    }
#endif
    const h2v am = __builtin_bit_cast(h2v, amax);
#if defined(SS3_ABL_NO_DMA) || defined(SS3_ABL_NO_ENC) || defined(SS3_ABL_NO_CONV) || defined(SS3_ABL_TAIL_ONLY) || defined(SS3_ABL_L0_ONLY)
    if (am[0] == (_Float16)12345.f) atomicOr(a.range_flag, 1u);   // timing-only build: garbage values must not trigger the exact redo
#else
    if (__any(!((float)am[0] < kRange) || !((float)am[1] < kRange) || amax_u > __float_as_uint(kRange)) && lane == 0) atomicOr(a.range_flag, 1u);
#endif
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
    // frames of a trajectory run on alternating streams (renderer._FramePipe): the other streams' head launches wait for this pack
    if (!f->ss_event) { hipEvent_t e; T2N_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); f->ss_event = (void*)e; }
    T2N_HIP(hipEventRecord((hipEvent_t)f->ss_event, s));
    f->ss_stream = (void*)s;
    f->ss_dirty = false;
    return T2N_OK;
}

// the head of the appearance stage for tiles [0, tile_hi) of the sub-lists, from feature rows; *range_flag is set when an
// activation left the f16 range (the caller then re-runs the launch on the exact path)
int launch_mlp_ss(t2n_field* f, const float* feat, const unsigned* counters_dev, unsigned list_cap, unsigned tile_hi, float4* app_rgb,
                  unsigned* range_flag, hipStream_t s) {
    using namespace ss;
    if (f->ss_dirty || !f->buf_ss) { const int rc = ss_pack(f, s); if (rc) return rc; }
    else if (f->ss_event && f->ss_stream != (void*)s) T2N_HIP(hipStreamWaitEvent(s, (hipEvent_t)f->ss_event, 0));
    static bool attr_set = false;
    static bool two_wave = false;   // T2N_SS_TWO_WAVE=1: the two-waves-per-SIMD form (A/B against the three-tile kernel)
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ss3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        const char* e = getenv("T2N_SS_TWO_WAVE");
        two_wave = e && e[0] == '1';
        attr_set = true;
    }
    const size_t nw = (size_t)kW0 + kW1 + kW2;
    uint4* base = (uint4*)f->buf_ss;
    Args a;
    a.w0 = base; a.w1 = base + kW0; a.w2 = base + kW0 + kW1;
    a.bias = (const float*)(base + nw); a.inv_scale = a.bias + kBias + 4;
    a.feat = feat; a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.tile_hi = tile_hi; a.app_rgb = app_rgb;
    a.range_flag = range_flag; a.neg1 = -1.f;
#ifdef SS3_PROF
    static unsigned long long* prof = nullptr;
    static int prof_calls = 0;
    if (!prof) { T2N_HIP(hipMalloc((void**)&prof, 1024 * 8 * 8)); }
    T2N_HIP(hipMemsetAsync(prof, 0, 1024 * 8 * 8, s));
    a.prof = prof;
#endif
    if (two_wave) hipLaunchKernelGGL(k_mlp_ss, dim3(256), dim3(512), kLds, s, a);
    else hipLaunchKernelGGL(k_mlp_ss3, dim3(256), dim3(256), kLds, s, a);
#ifdef SS3_PROF
    if (++prof_calls == 20) {   // one report per process: per-wave cycle sums, averaged over the waves
        static unsigned long long h[1024 * 8];
        T2N_HIP(hipStreamSynchronize(s));
        T2N_HIP(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        double sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 1024; ++i) for (int k = 0; k < 8; ++k) sum[k] += (double)h[i * 8 + k];
        fprintf(stderr, "[ss3 prof] per wave: barrier-wait %.0f, layer0 (incl. barriers) %.0f, tile0 %.0f, tile1 %.0f, tile2 %.0f, top %.0f, rounds %.1f, total %.0f cycles\n",
                sum[0] / 1024, sum[1] / 1024, sum[2] / 1024, sum[3] / 1024, sum[4] / 1024, sum[5] / 1024, sum[6] / 1024, sum[7] / 1024);
    }
#endif
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

