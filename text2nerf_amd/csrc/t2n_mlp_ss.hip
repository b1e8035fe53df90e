// Sample-stationary appearance head: positional encoding + the 351 -> 128 -> 128 -> 3 MLP + sigmoid for rows of 27 appearance
// features (K2b of the default render path). The features come from the gather + basis kernel (k_app_features_p in t2n_shade.hip, K2a).
//
// Replaces (reference): models/tensorBase.py:11-17 (positional_encoding), :88-109 (MLPRender_Fea_noview.forward).
//
// Mapping (gfx950, wave64): ONE 256-thread workgroup per CU — ONE wave per SIMD, owning its SIMD's whole 512-entry register file —
// persistent over rounds of 12 tiles (384 samples). Every wave owns THREE 32-sample tiles of the appearance lists from the feature
// rows to the colours: the encoded inputs, both hidden activations and the output never leave its registers. That works because the
// C/D layout of v_mfma_f32_32x32x16_f16 (lane (j, h): rows 8b + 4h + t of column j) IS a B-operand layout (lane (j, h): 8 K-values of
// column j) once the K order of the next layer is permuted accordingly — the permutation is folded into the weight packing (k_pack_ss):
//   layer 0: lane half h owns features 14h .. 14h+12 of its sample and half of feature 13 (octaves 0..2 / 3..5); its 176 K-values are
//            (sin, cos) of those three octaves, then per own feature (sin, cos) of octaves 0..5 [octaves 0 and 3 by v_sin_f32 /
//            v_cos_f32 on the fraction of f 2^o / (2 pi), the others by double-angle steps], then the 13 raw features, then
//            (half 1) the raw feature 13: 351 values in 352 slots; K-step s (16 values = one MFMA) takes values 8s .. 8s+7 of
//            both halves: 22 K-steps.
//   layers 1, 2: K index (step s, half h, element e) = hidden unit 32 (s / 2) + 8 (2 (s % 2) + e / 4) + 4h + e % 4 — what the lane
//            holds of unit tile s / 2 of the previous layer's accumulators.
// The weights are the A operands. Layer-0 weights (176 KB as hi / lo f16 halves) stream through a four-slot LDS ring of 16-KB
// chunks (two K-steps) that the four waves share (filled by LDS-DMA three chunks ahead; one barrier per chunk, which waits only for
// the pieces issued two iterations ago); layers 1 and 2 (80 KB) stay resident in LDS. Layer 0 (69 % of the MFMAs) runs for the wave's three tiles at once on SHARED A operands (every
// ds_read_b128 of an operand feeds nine MFMAs); layers 1 and 2 run tile by tile, software-pipelined (see stage()).
//
// fp32 products are three f16 products of hi / lo splits (x = hi + lo, hi = RTZ_f16(x), lo = RTZ_f16(x - hi); the lo*lo term is
// dropped: ~2^-21 relative), fp32 accumulate. Weights are pre-split and scaled by a per-layer power of two chosen from max|W| at
// upload (k_ss_scales); activations are split where they are produced; any activation beyond the f16 range raises a flag that
// makes the caller's exact-fp32 kernel redo the launch.
//
// History of the shape (records under profiles/): weight-stationary with the samples through LDS between the layers, 1.51 ms per C2
// frame (round 2: every layer boundary a workgroup barrier around a VALU-only phase); sample-stationary, ONE tile per wave and two
// waves per SIMD, 1.18-1.24 ms (rounds 2-3: 47 cycles per MFMA — 295 A-operand reads, 44 LDS-DMA pieces and 12 barriers per
// 384 MFMAs); three tiles per wave in HIP source with compiler-allocated accumulators, 1.74-1.86 ms (round 3: 149 spilled
// registers); this form, 1.18-1.22 ms (round 4: same-box A/B against the two-wave kernel 1.178-1.182 vs 1.184-1.187 ms; where its
// 45 cycles per MFMA go is accounted region by region in profiles/round4_head_issue_accounting.txt).
//
// hipcc schedules a region MFMAs first, VALU after. The instruction stream is therefore laid out by hand: a K-step is cut into
// SLOTS of one MFMA plus the VALU / LDS work that should issue in its shadow, with a full scheduling fence after every slot.
// Timing-only build switches (-DSS3_PROF: per-region cycle sums through s_memtime; -DSS3_ABL_*: ablations whose pictures are wrong)
// exist for tools/r4_head_accounting.sh; the shipped build defines none of them.
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
constexpr int kRingSlots = 4;                 // chunks of the layer-0 stream in LDS: the chunk in use, the next one, two in flight
constexpr int kRing = kRingSlots * kChunk;
constexpr int kW0 = kC0 * kChunk;
constexpr int kW1 = kC1 * kChunk;
constexpr int kW2 = kS2 * 2 * 64;             // [step][part][lane], one unit tile (rows 0..2 live)
constexpr int kBias = 288;                    // floats: layer 0 [128], layer 1 [128], layer 2 [32], scaled
constexpr size_t kLds = (size_t)(kRing + kW1 + kW2) * 16 + kBias * 4 + 16 * 4;   // + the sub-list table
constexpr float kRange = 59968.f;             // |activation| from here on (an f16 value; f16 max 65504) -> exact-path redo

struct Args {
    const uint4* w0; const uint4* w1; const uint4* w2; const float* bias; const float* inv_scale;   // packed by k_pack_ss
    const float* bounds;          // [0..3] row-norm bounds of the hidden activations, [4] (as unsigned) 1: the untracked kernel may run (k_ss_scales)
    const float* feat;            // [rows][32] fp32 feature rows (row = tile * 32 + sample): 27 features, the entry's compositing weight, zeros
    const unsigned* counters; unsigned list_cap; int nlists;
    unsigned tile_hi;             // 32-sample tiles [0, min(ntiles, tile_hi)) are this kernel's
    float4* app_rgb;
    unsigned* range_flag;
    float neg1;                   // -1.0f at run time: x - hi stays an FMA with an f16 operand (v_fma_mix_f32, no convert)
#ifdef SS3_PROF
    unsigned long long* prof;     // [waves][10] cycle sums (instrumented build)
#endif
};

#define SS_FENCE() __builtin_amdgcn_sched_barrier(0)


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

// ---- fillers: the VALU work that rides in a K-step's MFMA slots; run<IDX>() for IDX = 0..11 (unit IDX / 3 of four, phase IDX % 3),
// done() once the four units are through -----------------------------------------------------------------------------------------
struct NoFill {
    template <int IDX> __device__ __forceinline__ void run() {}
    __device__ __forceinline__ void done() {}
};
template <class F>
__device__ __forceinline__ void fill_all(F& f) {   // unscheduled form (prologue, layer boundaries)
#define SS_U(I) f.template run<3 * (I)>(); f.template run<3 * (I) + 1>(); f.template run<3 * (I) + 2>()
    SS_U(0); SS_U(1); SS_U(2); SS_U(3); f.done();
#undef SS_U
}

// =====================================================================================================================================
// The kernel: three tiles per wave, one wave per SIMD.
//
// Register plan (one wave owns its SIMD's whole 512-register file): the accumulators live in the AccVGPRs, named literally in
// inline asm — the compiler allocates only the architectural half (operands, encoders, conversion units):
//   a[64 t + 16 u .. +15]  layer-0 accumulators of tile t, unit tile u (192 registers); once tile t's layer 1 has consumed them
//                          a[64 t .. 64 t + 47] hold the three product chains of its layer 2
//   a[192 + 16 u .. +15]   layer-1 accumulators of tile 1; tiles 0 and 2 keep theirs in 64 ARCHITECTURAL registers ("+v" operands of
//                          the asm MFMAs), so that a tile's layer-1 accumulators can be converted while the next tile's accumulate
// Accumulators start from the constant 0 in their first MFMA (srcC = 0); the biases are added where an accumulator is converted
// (one v_fma_f32 instead of the scale multiply), from unscaled copies in LDS.
// A operands go through a ring of four register buffers in consumption order (184 per round: 22 x 4 of layer 0, then 8 x 4 of
// layer 1 per tile); the buffer of element k is refilled with element k + 4 right behind k's last MFMA. Layer 2's eight operands
// have a double buffer of their own (A2).
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
constexpr int kE0 = 88, kEL = 32, kER = kE0 + kT3 * kEL;   // A elements per round in the ring: layer 0, layer 1 per tile (layer 2's eight go through A2)
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
    static constexpr bool l0 = k < kE0, l1 = !l0;
    static constexpr int q = l0 ? 0 : (k - kE0) % kEL;
    static constexpr int s = l0 ? k / 4 : q / 4, ut = l0 ? k % 4 : q % 4;
    static constexpr int off = l0 ? ((s / 2) % kRingSlots) * kChunk + (s % 2) * kStep + ut * 128 : s * kStep + ut * 128;
};
struct LdsA { const u4v* a0; const u4v* a1; const u4v* a2; };   // + lane: the ring, W1, W2 (a DS instruction's immediate offset reaches 64 KB)
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
template <int K, bool LO>   // one half of element K (layer 1 frees an element's lo half eight slots before its hi half)
__device__ __forceinline__ void fetch_half(AEl (&A)[4], const LdsA& L) {
    if constexpr (K >= kER) return;
#ifdef SS3_ABL_NO_AFETCH
    return;
#endif
    using E = ElAddr<K>;
    const u4v* __restrict__ p = (E::l1 ? L.a1 : L.a0) + E::off;
    if constexpr (LO) A[K % 4].l = p[64];
    else A[K % 4].h = p[0];
}

// layer 2's operand of K-step St (W2 is resident at the start of LDS) -> A2[St % 2]: a double buffer of its own, because a layer-2 step of
// the previous tile rides in a layer-1 step of the current one and its operand lives across that step's twelve slots
template <int St>
__device__ __forceinline__ void fetch2(AEl (&A2)[2], const LdsA& L) {
    if constexpr (St < kS2) {
        const u4v* __restrict__ p = L.a2 + St * 128;
        A2[St % 2].h = p[0]; A2[St % 2].l = p[64];
    }
}

struct Enc3 {   // a tile's encoder: its features, the unit in flight ((sin, cos) of the previous octave between units), the operand being built
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
struct Conv3 {   // the conversion fillers' state (one tile at a time): up to three of a step's four units are in flight
    float x[4][2]; unsigned ph[4], pl[4];
    u4v bq[2][2];      // [K-step parity][half]: the biases of the eight units a step converts (bit patterns: loaded as the A operands
                       // are — a typed vector load carries TBAA, and hipcc drains the LDS-DMA counter in front of LDS loads that carry none)
};
// biases of conversion step S (units 32 (S / 2) + 8 b + 4 h + t, b = 2 (S % 2), 2 (S % 2) + 1) -> bq[S % 2]; issued a step ahead
template <int S, int BOFF>
__device__ __forceinline__ void conv_bias(Conv3& V, const float* __restrict__ LBh) {
    const u4v* __restrict__ p = reinterpret_cast<const u4v*>(LBh + BOFF + 32 * (S / 2) + 16 * (S % 2));
    V.bq[S % 2][0] = p[0];
    V.bq[S % 2][1] = p[2];
}
// relu(acc * inv + bias) of registers 8 (S % 2) .. +7 of unit tile S / 2 -> B operand of K-step S; the accumulators are AccVGPRs from
// BASE (>= 0) or the architectural registers src[4] (BASE < 0). A unit (two values) goes through SIX micro-operations — read, scale +
// bias, ReLU, hi halves, residuals (+ range tracking), lo halves — each depending on the one before; call IDX (0..11, one per MFMA slot)
// runs micro-operation IDX - 2 j of unit j, so that a slot holds at most three instructions groups of three DIFFERENT units and no
// instruction sits next to its producer. (One wave per SIMD: an instruction issued right behind the one it depends on costs 8 cycles
// instead of 4, and v_fma_mix -> v_cvt_pkrtz needs a wait state on top: tools/experiments/issue_cost.hip, 8.0 against 5.6 cycles per
// instruction for the split sequence alone.)
template <int S, int BASE, int BOFF, bool TRACK = true>
struct ConvFill3 {
    u4v (&Hh)[8]; u4v (&Hl)[8]; Conv3& V; float inv, neg1; s2v& amax; const float* __restrict__ LBh; const f32x16* src;
    template <int IDX> __device__ __forceinline__ void run() {
#ifdef SS3_ABL_NO_CONV
        if (S > 0) return;
#endif
        micro<3, IDX - 6>(); micro<2, IDX - 4>(); micro<1, IDX - 2>(); micro<0, IDX>();
        if constexpr (IDX == 1 && S < 7) conv_bias<(S < 7 ? S + 1 : 0), BOFF>(V, LBh);
    }
    __device__ __forceinline__ void all() {   // (where no MFMA stream needs the calls spread out: a stage's first conversion step, the round's last tile)
        run<0>(); run<1>(); run<2>(); run<3>(); run<4>(); run<5>(); run<6>(); run<7>(); run<8>(); run<9>(); run<10>(); run<11>();
        done();
    }
    template <int j, int m> __device__ __forceinline__ void micro() {
        if constexpr (m == 0) {
            if constexpr (BASE >= 0) {
                constexpr int reg = BASE + 16 * (S / 2) + 8 * (S % 2) + 2 * j;
                V.x[j][0] = acc_read<reg>(); V.x[j][1] = acc_read<reg + 1>();
            } else {
                V.x[j][0] = src[S / 2][8 * (S % 2) + 2 * j]; V.x[j][1] = src[S / 2][8 * (S % 2) + 2 * j + 1];
            }
        } else if constexpr (m == 1) {
            const u4v& b = V.bq[S % 2][j / 2];
            V.x[j][0] = fmaf(V.x[j][0], inv, __uint_as_float((j % 2) ? b[2] : b[0]));
            V.x[j][1] = fmaf(V.x[j][1], inv, __uint_as_float((j % 2) ? b[3] : b[1]));
        } else if constexpr (m == 2) {
            V.x[j][0] = fmaxf(V.x[j][0], 0.f);
            V.x[j][1] = fmaxf(V.x[j][1], 0.f);
        } else if constexpr (m == 3) {
            V.ph[j] = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(V.x[j][0], V.x[j][1]));
        } else if constexpr (m == 4) {
            const h2v ph = __builtin_bit_cast(h2v, V.ph[j]);
            V.x[j][0] = fmaf((float)ph[0], neg1, V.x[j][0]);
            V.x[j][1] = fmaf((float)ph[1], neg1, V.x[j][1]);
            if constexpr (TRACK) amax = __builtin_elementwise_max(amax, __builtin_bit_cast(s2v, V.ph[j]));   // bit patterns of non-negative halves order like their values
        } else if constexpr (m == 5) {
            V.pl[j] = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(V.x[j][0], V.x[j][1]));
        }
    }
    __device__ __forceinline__ void done() {
#ifdef SS3_ABL_NO_CONV
        if (S > 0) { Hh[S] = Hh[0]; Hl[S] = Hl[0]; return; }
#endif
        Hh[S] = u4v{V.ph[0], V.ph[1], V.ph[2], V.ph[3]};
        Hl[S] = u4v{V.pl[0], V.pl[1], V.pl[2], V.pl[3]};
        if constexpr (TRACK) asm volatile("" : "+v"(amax));   // pinned to its K-step: left alone, hipcc defers the maxima to the next round and spills the halves for it
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
struct NoRing { template <int M> __device__ __forceinline__ void run() {} };
struct RingOps3 {
    const Stream3& S; uint4* __restrict__ slot; int chunk;
    template <int M> __device__ __forceinline__ void run() {
#ifdef SS3_ABL_NO_DMA
        return;
#endif
        if (chunk == kC0 - 1) return;   // the stream's padding chunk (K-steps 22, 23) is never read
        // (the four pieces of a chunk spread over the K-step: issued back to back they cost 15 k cycles per wave and frame more)
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
// ---- layers 1 and 2, software-pipelined over the three tiles ------------------------------------------------------------------------
// Stage t = layer 1 of tile t (96 MFMA slots, the conversion of ITS layer-0 accumulators one phase per slot, just in time) with, from
// the second stage on, the previous tile riding along: the conversion of its layer-1 accumulators (one more phase per slot) and its
// layer 2 (K-step St - 1 as three extra MFMAs in slots 3, 7, 11 of layer-1 step St). Alone, a tile's layer 2 is 24 MFMAs under ~340
// conversion instructions (VALU-bound, the matrix pipe idle two thirds of the time: 2.1 of a tile's 6.6 k cycles); under the next
// tile's layer 1 it costs its instructions only. Layer-1 accumulators alternate between the architectural registers (tiles 0, 2)
// and a[192..255] (tile 1), so that a tile's accumulators can be converted while the next tile's accumulate.
struct NoL2 { template <int M> __device__ __forceinline__ void run() {} };
template <int BASE, bool FIRST>
struct L2Ops {   // the three products of one layer-2 K-step, each on its own chain a[BASE + 16 p ..]
    const AEl& W; const u4v& Hh; const u4v& Hl;
    template <int M> __device__ __forceinline__ void run() {
        if constexpr (M == 3) mfma_acc<BASE, FIRST, true>(W.h, Hh);
        if constexpr (M == 7) mfma_acc<BASE + 16, FIRST, false>(W.l, Hh);
        if constexpr (M == 11) mfma_acc<BASE + 32, FIRST, false>(W.h, Hl);
    }
};
// Slot M (0..11) of a K-step of layer 1: unit tile M % 4, product by slot group M / 4 (lo*hi, hi*hi, hi*lo): an accumulator is touched
// every FOURTH slot. (Unit tile M / 3, product M % 3 — three dependent MFMAs back to back on one accumulator — ran 43 cycles per
// slot with one filler phase and 67 with two: a dependent MFMA waits ~12 cycles beyond the pipe's 32 for its predecessor, and an
// in-order wave issues nothing meanwhile.) K0 = the A element of unit tile 0; ACC >= 0: accumulators a[ACC + 16 u ..], ACC < 0: the
// architectural accv[u]. fa / fb: one filler phase each per slot; l2: see L2Ops. An element's lo half is refilled (with the next
// K-step's) behind slot u, its hi half behind slot 8 + u.
template <int M, int K0, bool FIRST, int ACC, class FA, class FB, class L2C>
__device__ __forceinline__ void l1_slots(f32x16 (&accv)[4], AEl (&A)[4], const u4v& Hh, const u4v& Hl, FA& fa, FB& fb, L2C l2, const LdsA& L) {
    if constexpr (M < 12) {
        constexpr int ut = M % 4, g = M / 4, p = g == 0 ? 1 : (g == 1 ? 0 : 2), k = K0 + ut;
        const u4v& av = p == 1 ? A[k % 4].l : A[k % 4].h;
        const u4v& bv = p == 2 ? Hl : Hh;
        if constexpr (ACC >= 0) {
            mfma_acc<ACC + 16 * ut, (FIRST && g == 0), (M == 0)>(av, bv);
        } else if constexpr (FIRST && g == 0) {   // (architectural accumulators, still through asm: a compiler-visible MFMA made hipcc stage values in AccVGPRs of its own choice)
            if constexpr (M == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(accv[ut]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(accv[ut]) : "v"(av), "v"(bv));
        } else {
            if constexpr (M == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accv[ut]) : "v"(av), "v"(bv));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(accv[ut]) : "v"(av), "v"(bv));
        }
        if constexpr (g == 0) fetch_half<k + 4, true>(A, L);
        if constexpr (g == 2) fetch_half<k + 4, false>(A, L);
        fa.template run<M>();
        if constexpr (M == 11) fa.done();
        fb.template run<M>();
        if constexpr (M == 11) fb.done();
        l2.template run<M>();
        SS_FENCE();
        l1_slots<M + 1, K0, FIRST, ACC>(accv, A, Hh, Hl, fa, fb, l2, L);
    }
}
struct TileCtx {   // what the stages share (references: everything is inlined into the kernel)
    f32x16 (&accv)[4]; AEl (&A)[4]; AEl (&A2)[2]; Conv3& V0; Conv3& V1; s2v& amax; float inv0, inv1, neg1; const float* __restrict__ LBh;
    const LdsA& L; u4v (&H0h)[8]; u4v (&H0l)[8]; u4v (&H1h)[8]; u4v (&H1l)[8];
};
constexpr int acc1_of(int t) { return t == 1 ? 192 : -1; }   // where tile t's layer-1 accumulators live (see above)
template <int T_, int St, bool TRACK>
__device__ __forceinline__ void stage_step(TileCtx& c) {
    constexpr int K0 = kE0 + kEL * T_ + 4 * St;
    if constexpr (T_ > 0) fetch2<St>(c.A2, c.L);   // consumed by the layer-2 step riding in the NEXT layer-1 step (step 7: the stage's epilogue)
    auto run = [&](auto& fa, auto& fb, auto l2) { l1_slots<0, K0, St == 0, acc1_of(T_)>(c.accv, c.A, c.H0h[St], c.H0l[St], fa, fb, l2, c.L); };
    auto with_a = [&](auto& fb, auto l2) {
#ifdef SS3_ABL_S0_NOCONV
        if constexpr (T_ == 0) { NoFill fa; run(fa, fb, l2); return; }
#endif
        if constexpr (St < 7) {
            ConvFill3<St + 1, 64 * T_, 0, TRACK> fa{c.H0h, c.H0l, c.V0, c.inv0, c.neg1, c.amax, c.LBh, nullptr};
            run(fa, fb, l2);
        } else {
            NoFill fa;
            run(fa, fb, l2);
        }
    };
    if constexpr (T_ == 0) {
        NoFill fb;
        with_a(fb, NoL2());
    } else {
        ConvFill3<St, acc1_of(T_ - 1), 128, TRACK> fb{c.H1h, c.H1l, c.V1, c.inv1, c.neg1, c.amax, c.LBh, c.accv};
        if constexpr (St == 0) with_a(fb, NoL2());
        else with_a(fb, L2Ops<64 * (T_ - 1), St == 1>{c.A2[(St - 1) % 2], c.H1h[St - 1], c.H1l[St - 1]});
    }
}
template <int T_, bool TRACK>
__device__ __forceinline__ void stage(TileCtx& c) {
    if constexpr (T_ > 0) conv_bias<0, 128>(c.V1, c.LBh);   // the previous tile's first layer-1 conversion step (in flight under the block below)
    {
        ConvFill3<0, 64 * T_, 0, TRACK> f{c.H0h, c.H0l, c.V0, c.inv0, c.neg1, c.amax, c.LBh, nullptr};
        f.all();
    }
    stage_step<T_, 0, TRACK>(c); stage_step<T_, 1, TRACK>(c); stage_step<T_, 2, TRACK>(c); stage_step<T_, 3, TRACK>(c);
    stage_step<T_, 4, TRACK>(c); stage_step<T_, 5, TRACK>(c); stage_step<T_, 6, TRACK>(c); stage_step<T_, 7, TRACK>(c);
    if constexpr (T_ + 1 < kT3) conv_bias<0, 0>(c.V0, c.LBh);   // the next tile's first layer-0 conversion step
    if constexpr (T_ > 0) {   // the previous tile's last layer-2 step
        L2Ops<64 * (T_ - 1), false> l2{c.A2[1], c.H1h[7], c.H1l[7]};
        l2.template run<3>(); l2.template run<7>(); l2.template run<11>();
        SS_FENCE();
    }
}
// The last tile's layer 2 (nothing of this round is left to ride under): K-step St on the chains a[BASE ..], four conversion phases per
// slot; W2 operands through A2
template <int St, int BASE, class F>
__device__ __forceinline__ void l2_step(AEl (&A2)[2], const u4v& Hh, const u4v& Hl, F& f, const LdsA& L) {
    // (VALU-bound: 3 MFMAs under ~100 conversion instructions. What matters is the instructions' own issue rate: the four units of the
    // step run side by side, f.all())
    mfma_acc<BASE, St == 0, true>(A2[St % 2].h, Hh);
    mfma_acc<BASE + 16, St == 0, false>(A2[St % 2].l, Hh);
    mfma_acc<BASE + 32, St == 0, false>(A2[St % 2].h, Hl);
    fetch2<St + 2>(A2, L);
    f.all();
    SS_FENCE();
}

// The chunk barrier: every wave has its OWN older LDS-DMA pieces landed (vmcnt(KEEP): the KEEP most recent ones may stay in flight — a
// __syncthreads() would drain them all and make every barrier wait for the landing of pieces nobody needs for another iteration: 3 k of
// a round's 52 k cycles with the three-slot ring this kernel started with) and its LDS reads done, then s_barrier.
// (the wait as a builtin — simm16: vmcnt[3:0], expcnt[6:4] = 7, lgkmcnt[11:8] = 0, vmcnt[5:4] in [15:14] — so that hipcc's own counter
// bookkeeping sees it: behind an opaque wait it kept re-waiting for LDS reads that were long complete)
#define SS3_BAR_ASM(KEEP) do { __builtin_amdgcn_s_waitcnt((KEEP) == 0 ? 0x0070 : 0x0074); asm volatile("s_barrier" ::: "memory"); } while (0)
#ifdef SS3_ABL_NO_BARRIER
#define SS3_BARRIER(KEEP) do {} while (0)
#elif defined(SS3_PROF)
#define SS3_BARRIER(KEEP) do { const unsigned long long tb = __builtin_readcyclecounter(); SS3_BAR_ASM(KEEP); p_bar += __builtin_readcyclecounter() - tb; } while (0)
#else
#define SS3_BARRIER(KEEP) SS3_BAR_ASM(KEEP)
#endif
#ifdef SS3_PROF
#define SS3_T(var) do { const unsigned long long tn = __builtin_readcyclecounter(); var += tn - p_t; p_t = tn; } while (0)
#else
#define SS3_T(var) do {} while (0)
#endif
// the round's last layer-2 step: slot p carries all twelve encoder phases of tile p's first operand of the next round
template <int BASE, class F>
__device__ __forceinline__ void l2_last(AEl (&A2)[2], const u4v& Hh, const u4v& Hl, F (&f)[kT3]) {
    mfma_acc<BASE, false, true>(A2[1].h, Hh);
    mfma_acc<BASE + 16, false, false>(A2[1].l, Hh);
    mfma_acc<BASE + 32, false, false>(A2[1].h, Hl);
    // the three tiles' encoders phase by phase: three independent chains side by side
#define SS3_E(I) f[0].template run<I>(); f[1].template run<I>(); f[2].template run<I>()
    SS3_E(0); SS3_E(1); SS3_E(2); SS3_E(3); SS3_E(4); SS3_E(5); SS3_E(6); SS3_E(7); SS3_E(8); SS3_E(9); SS3_E(10); SS3_E(11);
#undef SS3_E
    f[0].done(); f[1].done(); f[2].done();
    SS_FENCE();
}

// TRACK: the hidden activations' f16 range is tracked value by value (one packed maximum per pair: 1.7 % of a round's issue cycles);
// !TRACK: it is bounded from the weights instead — |h0| <= max_j (sum_c |W0[j][pe c]| + |b0[j]|) + max|feature| * max_j sum_c |W0[j][raw c]|,
// |h1| <= max_j sum_k |W1[j][k]| * that + max |b1| (row norms by k_ss_scales at upload; the features' own maximum is tracked per round
// in both forms) — and the launch raises the range flag when the bound passes the f16 range. Both instantiations are launched; the word
// bounds[4] (written at upload: bound fine for features up to 64) makes one of them return at once.
template <bool TRACK>
__global__ __launch_bounds__(256) void k_mlp_ss3(const Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    if ((__float_as_uint(a.bounds[4]) != 0u) == TRACK) return;
    // LDS map: ring [0, 64 KB) | W1 [64 KB, 128 KB) | W2 [128 KB, 144 KB) | biases (unscaled) | sub-list table. A DS instruction carries a
    // 16-bit byte offset: with the lane's operand addresses written as THREE opaque bases (lane * 16 + 0 / 64 KB / 128 KB) plus constants
    // every fetch is base + immediate; left to itself hipcc materialises one base register per 64-KB-crossing constant and spills them.
    uint4* __restrict__ RING = lds;
    uint4* __restrict__ W1 = RING + kRing;
    uint4* __restrict__ W2 = W1 + kW1;
    float* __restrict__ LB = reinterpret_cast<float*>(W2 + kW2);
    static_assert(kRing * 16 == 65536 && kW1 * 16 == 65536, "the ring and W1 each fill one 64-KB window");
    unsigned* __restrict__ LT = reinterpret_cast<unsigned*>(LB + kBias);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    unsigned ob0 = (unsigned)lane * 16u, ob1 = (unsigned)lane * 16u + 65536u, ob2 = (unsigned)lane * 16u + 131072u,
             obb = (unsigned)((kW2 + kRing + kW1) * 16) + 16u * (unsigned)h;
    asm volatile("" : "+v"(ob0));
    asm volatile("" : "+v"(ob1));
    asm volatile("" : "+v"(ob2));
    asm volatile("" : "+v"(obb));
    const LdsA L{reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(lds) + ob0), reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(lds) + ob1),
                 reinterpret_cast<const u4v*>(reinterpret_cast<const char*>(lds) + ob2)};
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
    S.dma<0>(RING + 2 * kChunk, 2); S.dma<1>(RING + 2 * kChunk, 2); S.dma<2>(RING + 2 * kChunk, 2); S.dma<3>(RING + 2 * kChunk, 2);
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
    AEl A[4], A2[2];
    Conv3 V, V1;   // conversion fillers of a stage's own tile (layer-0 accumulators) / of the tile riding along (layer-1 accumulators)

#ifdef SS3_PROF
    unsigned long long p_bar = 0, p_l0 = 0, p_tile[3] = {0, 0, 0}, p_top = 0, p_rounds = 0, p_tail = 0;
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
        asm volatile("; SS3_MARK layer0");
        // ---- layer 0, three tiles on shared A operands: 11 chunks of two K-steps (+ the ring's padding chunk) ----------------------------
#define SS3_ENC(S_, P_) {{E[0], Bh[P_][0], Bl[P_][0], neg1}, {E[1], Bh[P_][1], Bl[P_][1], neg1}, {E[2], Bh[P_][2], Bl[P_][2], neg1}}
#define SS3_L0(C)                                                                                                                 \
        {                                                                                                                         \
            /* chunk C + 1 written by every wave (issued two iterations ago: the four pieces of the LAST iteration - chunk C + 2 - may  */ \
            /* still be in flight, except behind iteration 8, which issues none: the stream's padding chunk), chunk C - 1 read by all  */ \
            SS3_BARRIER(C == 9 ? 0 : 4);                                                                                          \
            if constexpr (C == 0) { fetch<0>(A, L); fetch<1>(A, L); fetch<2>(A, L); fetch<3>(A, L); }                             \
            RingOps3 ring{S, RING + ((C + 3) % kRingSlots) * kChunk, (C + 3) % kC0};                                              \
            if constexpr (C == 11) {                                                                                              \
                /* nothing to multiply (the stream's padding chunk): this iteration is the barrier behind chunk 10's last reads. The   */ \
                /* DMA it would issue (the next round's chunk 2) goes out at the END of the round instead: see there                  */ \
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
                    l0_slots<0, 2 * C + 1>(A, Bh[1], Bl[1], f1, NoRing(), L);                                                     \
                }                                                                                                                 \
            }                                                                                                                     \
        }
        SS3_L0(0) SS3_L0(1) SS3_L0(2) SS3_L0(3) SS3_L0(4) SS3_L0(5) SS3_L0(6) SS3_L0(7) SS3_L0(8) SS3_L0(9) SS3_L0(10) SS3_L0(11)
#undef SS3_L0
        SS3_T(p_l0);

        // ---- layers 1 and 2: three pipelined stages and the last tile's layer 2 (see stage()) -------------------------------------------
#define SS3_FINAL(T_)                                                                                                             \
        {                                                                                                                         \
            asm volatile("s_nop 15");   /* the chains' last MFMAs -> their reads */                                              \
            const float c00 = acc_read<64 * T_>(), c01 = acc_read<64 * T_ + 1>(), c02 = acc_read<64 * T_ + 2>();                  \
            const float c10 = acc_read<64 * T_ + 16>(), c11 = acc_read<64 * T_ + 17>(), c12 = acc_read<64 * T_ + 18>();           \
            const float c20 = acc_read<64 * T_ + 32>(), c21 = acc_read<64 * T_ + 33>(), c22 = acc_read<64 * T_ + 34>();           \
            const unsigned tile = (r * 4u + (unsigned)w) * 3u + T_;                                                               \
            if (tile < ntiles && h == 0) {   /* output rows 0..2 live in registers 0..2 of lanes 0..31 */                         \
                const u4v i0 = *reinterpret_cast<const u4v*>(LT), i1 = *reinterpret_cast<const u4v*>(LT + 4);   /* (typed: see Conv3) */ \
                const unsigned pre[8] = {i0[0], i0[1], i0[2], i0[3], i1[0], i1[1], i1[2], i1[3]};                                 \
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
#define SS3_L2(St)                                                                                                                \
        {                                                                                                                         \
            ConvFill3<St + 1, -1, 128, TRACK> f{H1h, H1l, V1, inv1, neg1, amax, LBh, accv};                                       \
            l2_step<St, 128>(A2, H1h[St], H1l[St], f, L);                                                                         \
        }
        {
            f32x16 accv[4];
            u4v H0h[8], H0l[8], H1h[8], H1l[8];
            TileCtx ctx{accv, A, A2, V, V1, amax, inv0, inv1, neg1, LBh, L, H0h, H0l, H1h, H1l};
            asm volatile("; SS3_MARK stage0");
            // (hipcc cannot tell the bias reads from the LDS-DMA's destination and drains vmcnt in front of the first one behind a DMA issue:
            // none is issued between here and the round's last bias read — the next round's chunk 2 goes out at the end of the round)
            conv_bias<0, 0>(V, LBh);   // tile 0's first conversion step
            stage<0, TRACK>(ctx);
            SS3_T(p_tile[0]);
            asm volatile("; SS3_MARK stage1");
            // the next round's features (needed from the round's last step on): 42 registers in flight through two stages — loaded at the
            // tail instead, their HBM / MALL latency (~2 us behind the gather kernel's 548 MB of rows) stalled every round for ~2.5 k cycles
#pragma unroll
            for (int t = 0; t < kT3; ++t) load_feat(rn, t, E[t].in.f);
            stage<1, TRACK>(ctx);
            SS3_FINAL(0)
            SS3_T(p_tile[1]);
            asm volatile("; SS3_MARK stage2");
            stage<2, TRACK>(ctx);
            SS3_FINAL(1)
            SS3_T(p_tile[2]);
            asm volatile("; SS3_MARK tail");
            // the last tile: its layer-1 accumulators (architectural) -> layer 2, alone. The next round's features (needed from its
            // first encoder phase to its last layer-0 step: 42 registers that nothing above needs) are loaded here.
            fetch2<0>(A2, L); fetch2<1>(A2, L);
            conv_bias<0, 128>(V1, LBh);
            {
                ConvFill3<0, -1, 128, TRACK> f{H1h, H1l, V1, inv1, neg1, amax, LBh, accv};
                f.all();
            }
            SS3_L2(0) SS3_L2(1) SS3_L2(2) SS3_L2(3) SS3_L2(4) SS3_L2(5) SS3_L2(6)
            {   // the round's last step: the next round's first operands of the three tiles, one tile per slot
#pragma unroll
                for (int t = 0; t < kT3; ++t) finish_feat(E[t].in, wnext[t]);
                EncFill3<0> f3[kT3] = SS3_ENC(0, 0);
                l2_last<128>(A2, H1h[7], H1l[7], f3);
            }
            SS3_FINAL(2)
            // the next round's chunk 2 -> ring slot 2 (chunk 10's, free since the barrier of iteration 11): issued here, behind the round's
            // last bias read; it has the next round's whole first iteration to land
            { RingOps3 ring{S, RING + 2 * kChunk, 2}; ring.all(); }
            asm volatile("; SS3_MARK end");
            SS3_T(p_tail);
#ifdef SS3_PROF
            ++p_rounds;
#endif
        }
#undef SS3_L2
#undef SS3_FINAL
#undef SS3_ENC
    }
#ifdef SS3_PROF
    if (lane == 0 && a.prof) {
        unsigned long long* o = a.prof + (size_t)(blockIdx.x * 4 + w) * 10;
        o[0] = p_bar; o[1] = p_l0; o[2] = p_tile[0]; o[3] = p_tile[1]; o[4] = p_tile[2]; o[5] = p_top; o[6] = p_rounds;
        o[7] = __builtin_readcyclecounter() - p_start; o[8] = p_tail;
    }
#endif
    const h2v am = __builtin_bit_cast(h2v, amax);
#if defined(SS3_ABL_NO_DMA) || defined(SS3_ABL_NO_ENC) || defined(SS3_ABL_NO_CONV) || defined(SS3_ABL_S0_NOCONV) || defined(SS3_TIMING_ONLY)
    if (am[0] == (_Float16)12345.f) atomicOr(a.range_flag, 1u);   // timing-only build: garbage values must not trigger the exact redo
#else
    bool out_of_range = !((float)am[0] < kRange) || !((float)am[1] < kRange) || amax_u > __float_as_uint(kRange);
    if constexpr (!TRACK) {
        const float fmx = __uint_as_float(amax_u);                                    // largest |feature| this wave saw
        const float b0 = fmaf(fmx, a.bounds[1], a.bounds[0]), b1 = fmaf(a.bounds[2], b0, a.bounds[3]);
        out_of_range = out_of_range || !(b0 < kRange) || !(b1 < kRange);
    }
    if (__any(out_of_range) && lane == 0) atomicOr(a.range_flag, 1u);
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
    // row-norm bounds for the untracked head (see k_mlp_ss3): thread j < 128 owns hidden unit j of both layers
    __shared__ float nb[4][128];
    if (tid < 128) {
        float pe = 0.f, raw = 0.f, r1 = 0.f;
        for (int c = 0; c < 351; ++c) { const float v = fabsf(a.w0[tid * 351 + c]); if (c < 27) raw += v; else pe += v; }
        for (int c = 0; c < 128; ++c) r1 += fabsf(a.w1[tid * 128 + c]);
        nb[0][tid] = pe + fabsf(a.b0[tid]); nb[1][tid] = raw; nb[2][tid] = r1; nb[3][tid] = fabsf(a.b1[tid]);
    }
    __syncthreads();
    if (tid == 8) {
        float m4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < 4; ++q) for (int j = 0; j < 128; ++j) m4[q] = fmaxf(m4[q], nb[q][j]);
        for (int q = 0; q < 4; ++q) a.scales[8 + q] = m4[q] * 1.0001f;        // (the sums above round: a margin)
        const float b0 = m4[0] + 64.f * m4[1], b1 = m4[2] * b0 + m4[3];
        const bool ok = b0 < 0.5f * kRange && b1 < 0.5f * kRange;             // NaN / inf weights compare false: the tracked kernel runs
        reinterpret_cast<unsigned*>(a.scales)[12] = ok ? 1u : 0u;
    }
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
        T2N_HIP(hipMalloc((void**)&f->buf_ss, nw * 16 + (kBias + 16) * 4));   // + scales [0..7], activation bounds [8..11], untracked-ok word [12]
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
    // which instantiation of the head these weights allow (bounds[4]) also travels to pinned host memory behind an event: launches
    // that find the copy landed start only that one; until then both are started and the word on the device makes one return at once
    f->ss_variant = -1;
    if (!f->ss_ok_host) {
        if (hipHostMalloc((void**)&f->ss_ok_host, 4, hipHostMallocDefault) != hipSuccess) { f->ss_ok_host = nullptr; (void)hipGetLastError(); }
        hipEvent_t e;
        if (f->ss_ok_host && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess) f->ss_ok_event = (void*)e;
    }
    if (f->ss_ok_host && f->ss_ok_event) {
        *f->ss_ok_host = 0xffffffffu;
        if (hipMemcpyAsync(f->ss_ok_host, a.scales + 12, 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipEventRecord((hipEvent_t)f->ss_ok_event, s) != hipSuccess) { (void)hipGetLastError(); f->ss_variant = -2; }
    } else f->ss_variant = -2;   // no read-back: always both launches
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
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ss3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_ss3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        attr_set = true;
    }
    const size_t nw = (size_t)kW0 + kW1 + kW2;
    uint4* base = (uint4*)f->buf_ss;
    Args a;
    a.w0 = base; a.w1 = base + kW0; a.w2 = base + kW0 + kW1;
    a.bias = (const float*)(base + nw); a.inv_scale = a.bias + kBias + 4; a.bounds = a.bias + kBias + 8;
    a.feat = feat; a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.tile_hi = tile_hi; a.app_rgb = app_rgb;
    a.range_flag = range_flag; a.neg1 = -1.f;
#ifdef SS3_PROF
    static unsigned long long* prof = nullptr;
    static int prof_calls = 0;
    if (!prof) { T2N_HIP(hipMalloc((void**)&prof, 1024 * 10 * 8)); }
    T2N_HIP(hipMemsetAsync(prof, 0, 1024 * 10 * 8, s));
    a.prof = prof;
#endif
    if (f->ss_variant == -1 && hipEventQuery((hipEvent_t)f->ss_ok_event) == hipSuccess && *f->ss_ok_host != 0xffffffffu)
        f->ss_variant = *f->ss_ok_host ? 0 : 1;   // 0: untracked (bounds hold), 1: tracked
    else if (f->ss_variant == -1) (void)hipGetLastError();   // (hipErrorNotReady)
    // (a kernel started for the wrong word returns at once: bounds[4] on the device decides, the host copy only saves the empty launch)
    if (f->ss_variant != 1) hipLaunchKernelGGL(k_mlp_ss3<false>, dim3(256), dim3(256), kLds, s, a);
    if (f->ss_variant != 0) hipLaunchKernelGGL(k_mlp_ss3<true>, dim3(256), dim3(256), kLds, s, a);
#ifdef SS3_PROF
    if (++prof_calls == 20) {   // one report per process: per-wave cycle sums, averaged over the waves
        static unsigned long long h[1024 * 10];
        T2N_HIP(hipStreamSynchronize(s));
        T2N_HIP(hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost));
        double sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 1024; ++i) for (int k = 0; k < 10; ++k) sum[k] += (double)h[i * 10 + k];
        fprintf(stderr, "[ss3 prof] per wave: barrier-wait %.0f, layer0 (incl. barriers) %.0f, stage0 %.0f, stage1 %.0f, stage2 %.0f, tail %.0f, top %.0f, rounds %.1f, total %.0f cycles\n",
                sum[0] / 1024, sum[1] / 1024, sum[2] / 1024, sum[3] / 1024, sum[4] / 1024, sum[8] / 1024, sum[5] / 1024, sum[6] / 1024, sum[7] / 1024);
    }
#endif
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

