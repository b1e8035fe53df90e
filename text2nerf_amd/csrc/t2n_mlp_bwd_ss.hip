// Sample-stationary input-gradient chain of the appearance head's backward: for every appearance row
//     g1 = (W2^T go) * [h1 > 0],  g0 = (W1^T g1) * [h0 > 0],  gx = W0^T g0,  gf = PE'(feat)^T gx,  gX = Wb^T gf
// in ONE kernel, one 32-row tile per wave, the vectors never leaving the wave's registers between the stages (the C/D layout of
// v_mfma_f32_32x32x16_f16 is a B-operand layout under a K permutation folded into the weight packing — t2n_mlp_ss.hip has the
// forward form). g1, g0, gf and gX are written out once each for the weight-gradient GEMMs and the factor scatter.
//
// Replaces (reference): the autograd of MLPRender_Fea_noview.forward and basis_mat w.r.t. their inputs (models/tensorBase.py:11-17,
// 88-109, models/tensoRF.py:147,239; triggered by text2nerf_main.py:589). It stands in for five launches of the unfused form
// (k_bwd_l2's g1, three k_gemm_nn_h, k_pe_bwd: 300 us per C3 iteration, each re-reading what the previous one wrote).
//
// Arithmetic: fp32 products as three f16 MFMA products of hi / lo splits (lo*lo dropped: ~2^-21 relative), fp32 accumulate. A
// gradient vector is linear in its sample's `go`, and gradients span many orders of magnitude: before every stage the lane pair of
// a sample rescales its vector by a power of two (largest magnitude to [2^13, 2^14)) and carries the cumulative exponent; what
// is stored is scaled back exactly. Weights carry one power of two per matrix (largest magnitude in [2^12, 2^13)).
//
// Mapping: 512-thread workgroups, persistent over rounds of 8 tiles. W2^T, W1^T and Wb^T (92 KB packed) stay in LDS; W0^T (192 KB:
// 384 permuted encoding rows x 128, hi + lo) streams through a three-slot ring of 16-KB chunks filled by LDS-DMA two chunks ahead,
// one barrier per chunk, in three passes of 128 encoding rows; after each pass the lane folds its 64 encoding gradients into the
// gradients of its 14 features (sin / cos by the forward head's hardware path: octaves 0 and 3 fresh, the others by double angle).
#include "t2n_device.h"

namespace t2n {
namespace bss {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

constexpr int kA2 = 4 * 2 * 64;               // uint4: W2^T [unit tile 4][part][lane], one K-step (3 live values)
constexpr int kChunk = 2 * 4 * 2 * 64;        // uint4 per ring chunk: [step of the pair][row tile 4][part][lane]
constexpr int kRing = 3 * kChunk;
constexpr int kA1 = 8 * 4 * 2 * 64;           // W1^T [step 8][unit tile 4][part][lane]
constexpr int kAb = 2 * 5 * 2 * 64;           // Wb^T [step 2][channel tile 5][part][lane]
constexpr int kA0 = 12 * kChunk;              // W0^T [chunk 12 = pass 3 x step pair 4][step 2][row tile 4][part][lane] (global)
constexpr size_t kLds = (size_t)(kA2 + kRing + kA1 + kAb) * 16;
// LDS byte offsets (every operand fetch is one of three opaque bases + an immediate, see t2n_mlp_ss.hip)
constexpr int oRing = 0, oA2 = kRing * 16, oA1 = oA2 + kA2 * 16, oAb = oA1 + kA1 * 16;
static_assert(oAb + kAb * 16 == (int)kLds && kLds <= 160 * 1024, "LDS map");

struct Args {
    const uint4* a2; const uint4* a1; const uint4* a0; const uint4* ab; const float* inv_scale;   // [0] W2, [1] W1, [2] W0, [3] basis
    const float4* go; float* h1; const float* h0; const float* feat;   // h1 is overwritten with g1 unless g1 points elsewhere
    float* g1;
    float* g0; float* gf; float* gx;                                   // [rows,128], [rows,32], [rows,144]
    long long rows;                                                    // a multiple of 32
    const unsigned* rows_dev;                                          // optional: the row count in device memory (rows = capacity; see k_bwd_l2)
    float neg1;
};

__device__ __forceinline__ f32x16 mfma(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split2(float x0, float x1, float neg1, unsigned& hi, unsigned& lo) {
    const hh2 p = __builtin_amdgcn_cvt_pkrtz(x0, x1);
    const h2v ph = __builtin_bit_cast(h2v, p);
    const float r0 = fmaf((float)ph[0], neg1, x0), r1 = fmaf((float)ph[1], neg1, x1);
    hi = __builtin_bit_cast(unsigned, p);
    lo = __builtin_bit_cast(unsigned, (hh2)__builtin_amdgcn_cvt_pkrtz(r0, r1));
}
// 2^k as a float, k in [-126, 127]
__device__ __forceinline__ float pow2i(int k) { return __uint_as_float((unsigned)(127 + k) << 23); }
// exponent shift that takes a magnitude with bit pattern mb into [2^13, 2^14) (0 for zero / non-finite; bounded)
__device__ __forceinline__ int shift_of(unsigned mb) {
    const int eb = (int)(mb >> 23);
    int k = (mb == 0u || eb == 255) ? 0 : 127 + 13 - eb;
    return k > 60 ? 60 : (k < -60 ? -60 : k);
}

// [h > 0] of the lane's 64 units of an activation row as bits (bit 16 u + 4 b + t <-> register 4 b + t of tile u): the 64 loaded floats
// would otherwise sit in registers under a whole stage of MFMAs (and spill)
__device__ __forceinline__ unsigned long long relu_mask(const float* __restrict__ hr /* row + 4 h */) {
    unsigned long long m = 0ull;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float4 v[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) v[b] = *reinterpret_cast<const float4*>(hr + 32 * u + 8 * b);
        unsigned bits = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            bits |= (v[b].x > 0.f ? 1u << (4 * b) : 0u) | (v[b].y > 0.f ? 2u << (4 * b) : 0u) | (v[b].z > 0.f ? 4u << (4 * b) : 0u) |
                    (v[b].w > 0.f ? 8u << (4 * b) : 0u);
        m |= (unsigned long long)bits << (16 * u);
    }
    return m;
}
struct Ops { uint4 h[8], l[8]; };   // B operands of 8 K-steps: a 128-vector of the lane pair's sample

// C-layout values v[u][16] (unit 32 u + 8 b + 4 h + t at register 4 b + t) -> rescaled B operands; E += shift. The K order of the
// consuming layer is hid_unit (k_pack_bwd_ss): step s = 2 u + b / 2, element e = 4 (b % 2) + t.
__device__ __forceinline__ void renorm_split(f32x16 (&v)[4], int& E, Ops& O, float neg1) {
    unsigned mb = 0u;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) mb = max(mb, __float_as_uint(v[u][i]) & 0x7fffffffu);
    mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));
    const int k = shift_of(mb);
    E += k;
    const float sc = pow2i(k);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int s = 2 * u + hf;
            split2(v[u][8 * hf + 0] * sc, v[u][8 * hf + 1] * sc, neg1, O.h[s].x, O.l[s].x);
            split2(v[u][8 * hf + 2] * sc, v[u][8 * hf + 3] * sc, neg1, O.h[s].y, O.l[s].y);
            split2(v[u][8 * hf + 4] * sc, v[u][8 * hf + 5] * sc, neg1, O.h[s].z, O.l[s].z);
            split2(v[u][8 * hf + 6] * sc, v[u][8 * hf + 7] * sc, neg1, O.h[s].w, O.l[s].w);
        }
}

// one K-step against NT row tiles at A (+ tile * 128 [+ 64: lo part] uint4): products hi*hi, lo*hi, hi*lo per tile, two tiles at a
// time (an accumulator is touched every second MFMA; four operand registers in flight, not eight: the kernel lives at the VGPR cap)
template <int NT>
__device__ __forceinline__ void kstep(f32x16 (&acc)[NT], const uint4* __restrict__ A, const uint4& Bh, const uint4& Bl) {
#pragma unroll
    for (int u = 0; u + 1 < NT; u += 2) {
        const uint4 ah0 = A[u * 128], al0 = A[u * 128 + 64], ah1 = A[(u + 1) * 128], al1 = A[(u + 1) * 128 + 64];
        acc[u] = mfma(ah0, Bh, acc[u]); acc[u + 1] = mfma(ah1, Bh, acc[u + 1]);
        acc[u] = mfma(al0, Bh, acc[u]); acc[u + 1] = mfma(al1, Bh, acc[u + 1]);
        acc[u] = mfma(ah0, Bl, acc[u]); acc[u + 1] = mfma(ah1, Bl, acc[u + 1]);
        __builtin_amdgcn_sched_barrier(0);   // one pair's operands in flight at a time: hoisting the next fetches up spills
    }
    if constexpr (NT % 2) {
        constexpr int u = NT - 1;
        const uint4 ah0 = A[u * 128], al0 = A[u * 128 + 64];
        acc[u] = mfma(ah0, Bh, acc[u]);
        acc[u] = mfma(al0, Bh, acc[u]);
        acc[u] = mfma(ah0, Bl, acc[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// (sin, cos)(2^o f) of the forward head's encoder: o = 0, 3 by v_sin / v_cos on the reduced argument, the others by double angle
struct SinCos { float sn, cs; };
template <int O>
__device__ __forceinline__ void pe_advance(SinCos& S, float f) {
    if constexpr (O == 0 || O == 3) {
        const float C1 = 0.15915494309189535f;
        const float C2 = (float)(0.15915494309189533576888 - (double)C1);
        constexpr float sc = (float)(1 << O);
        const float th = f * C1;
        const float tl = fmaf(f, C1, -th) + f * C2;
        const float arg = fmaf(tl, sc, __builtin_amdgcn_fractf(th * sc));
        S.sn = __builtin_amdgcn_sinf(arg);
        S.cs = __builtin_amdgcn_cosf(arg);
    } else {
        const float t = S.sn + S.sn;
        const float c2 = fmaf(-t, S.sn, 1.f);
        S.sn = t * S.cs;
        S.cs = c2;
    }
}
// fold the 64 encoding gradients of pass P (value index 64 P + 16 u + r at register r of tile u: the lane half's sequence of the
// forward head: per feature (sin, cos) of octaves 0..5, then the raw features from 168) into the feature gradients
template <int P, int IDX>
__device__ __forceinline__ void pe_fold_pair(const f32x16 (&acc)[4], float inv, const float (&f)[14], SinCos& S, float (&gfe)[14]) {
    if constexpr (IDX < 32) {
        constexpr int u = IDX / 8, r = 2 * (IDX % 8), v = 64 * P + 16 * u + r;
        const float d0 = acc[u][r] * inv, d1 = acc[u][r + 1] * inv;
        if constexpr (v < 168) {
            constexpr int F = v / 12, o = (v % 12) / 2;
            pe_advance<o>(S, f[F]);
            constexpr float w = (float)(1 << o);
            gfe[F] = fmaf(w * d0, S.cs, gfe[F]);      // d sin(2^o f) / df = 2^o cos
            gfe[F] = fmaf(-(w * d1), S.sn, gfe[F]);   // d cos(2^o f) / df = -2^o sin
        } else if constexpr (v < 182) {
            gfe[v - 168] += d0;
            if constexpr (v + 1 < 182) gfe[v - 167] += d1;
        }
        if constexpr (IDX % 4 == 3) __builtin_amdgcn_sched_barrier(0);   // the (sin, cos) chain does not depend on the accumulators: unfenced,
        pe_fold_pair<P, IDX + 1>(acc, inv, f, S, gfe);                    // hipcc computes every feature's chain up front and spills it
    }
}
template <int P>
__device__ __forceinline__ void pe_fold(const f32x16 (&acc)[4], float inv, const float (&f)[14], SinCos& S, float (&gfe)[14]) {
    __builtin_amdgcn_sched_barrier(0);
    pe_fold_pair<P, 0>(acc, inv, f, S, gfe);
    __builtin_amdgcn_sched_barrier(0);
}

struct Stream {
    const uint4* wp; int tid;
    template <int HALF> __device__ __forceinline__ void dma(uint4* __restrict__ slot, int c) const {
        typedef __attribute__((address_space(3))) void* lp;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wp), 0, 0x7fffffff, 0x00020000);
        const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lp)(slot + HALF * 512 + w * 64), 16, tid * 16, c * (kChunk * 16) + HALF * 8192, 0, 0);
    }
};

__global__ __launch_bounds__(512) void k_mlp_bwd_ss(const Args a) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const long long ntiles = (a.rows_dev && (long long)*a.rows_dev < a.rows ? (long long)*a.rows_dev : a.rows) / 32;
    const long long nrounds = (ntiles + 7) / 8;
    if ((long long)blockIdx.x >= nrounds) return;
    unsigned ob0 = (unsigned)lane * 16u, ob1 = (unsigned)lane * 16u + 65536u, ob2 = (unsigned)lane * 16u + 131072u;
    asm volatile("" : "+v"(ob0));
    asm volatile("" : "+v"(ob1));
    asm volatile("" : "+v"(ob2));
    const char* __restrict__ L0 = reinterpret_cast<const char*>(lds) + ob0;
    const char* __restrict__ L1 = reinterpret_cast<const char*>(lds) + ob1;
    const char* __restrict__ L2 = reinterpret_cast<const char*>(lds) + ob2;
    // operand pointer at LDS byte offset OFF (+ lane * 16): base picked by the 64-KB window of the constant
    auto at = [&](int off) -> const uint4* {
        return reinterpret_cast<const uint4*>(off < 65536 ? L0 + off : (off < 131072 ? L1 + (off - 65536) : L2 + (off - 131072)));
    };
    uint4* __restrict__ RING = lds + oRing / 16;
    // resident operands; ring slots 0 / 1 <- chunks 0 / 1
    for (int i = tid; i < kA2; i += 512) lds[oA2 / 16 + i] = a.a2[i];
    for (int i = tid; i < kA1; i += 512) lds[oA1 / 16 + i] = a.a1[i];
    for (int i = tid; i < kAb; i += 512) lds[oAb / 16 + i] = a.ab[i];
    const Stream S{a.a0, tid};
    S.dma<0>(RING, 0); S.dma<1>(RING, 0);
    S.dma<0>(RING + kChunk, 1); S.dma<1>(RING + kChunk, 1);
    const float inv2 = a.inv_scale[0], inv1 = a.inv_scale[1], inv0 = a.inv_scale[2], invb = a.inv_scale[3];
    const float neg1 = a.neg1;
    __syncthreads();

    for (long long r = blockIdx.x; r < nrounds; r += gridDim.x) {
        long long tile = r * 8 + w;
        const bool live = tile < ntiles;
        if (!live) tile = ntiles - 1;                     // a spare wave re-does the last tile (barriers stay uniform), stores nothing
        const long long row = tile * 32 + j;
        // ---- inputs: go (both lanes of the pair), the masks' activations, the features of this lane half -------------------------
        const float4 g = a.go[row];
        unsigned long long mk = relu_mask(a.h1 + row * 128 + 4 * h);
        float f[14];
        {
            const float2* __restrict__ fr = reinterpret_cast<const float2*>(a.feat + row * 32 + 14 * h);
#pragma unroll
            for (int e = 0; e < 7; ++e) { const float2 v = fr[e]; f[2 * e] = v.x; f[2 * e + 1] = v.y; }
        }
        // ---- stage A: g1 = (W2^T go) * [h1 > 0] -----------------------------------------------------------------------------------
        int E = shift_of(max(__float_as_uint(g.x) & 0x7fffffffu, max(__float_as_uint(g.y) & 0x7fffffffu, __float_as_uint(g.z) & 0x7fffffffu)));
        uint4 Bh = make_uint4(0u, 0u, 0u, 0u), Bl = Bh;
        {
            const float sc = pow2i(E);
            const float gx0 = h == 0 ? g.x * sc : 0.f, gy0 = h == 0 ? g.y * sc : 0.f, gz0 = h == 0 ? g.z * sc : 0.f;
            split2(gx0, gy0, neg1, Bh.x, Bl.x);
            split2(gz0, 0.f, neg1, Bh.y, Bl.y);
        }
        f32x16 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x16{0};
        kstep<4>(acc, at(oA2), Bh, Bl);
        {
            const float us = pow2i(-E);
            float* __restrict__ gr = a.g1 + row * 128 + 4 * h;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned q = (unsigned)(mk >> (16 * u + 4 * b));
                    acc[u][4 * b] = (q & 1u) ? acc[u][4 * b] * inv2 : 0.f;
                    acc[u][4 * b + 1] = (q & 2u) ? acc[u][4 * b + 1] * inv2 : 0.f;
                    acc[u][4 * b + 2] = (q & 4u) ? acc[u][4 * b + 2] * inv2 : 0.f;
                    acc[u][4 * b + 3] = (q & 8u) ? acc[u][4 * b + 3] * inv2 : 0.f;
                    if (live) *reinterpret_cast<float4*>(gr + 32 * u + 8 * b) =
                        make_float4(acc[u][4 * b] * us, acc[u][4 * b + 1] * us, acc[u][4 * b + 2] * us, acc[u][4 * b + 3] * us);
                }
        }
        Ops O;
        renorm_split(acc, E, O, neg1);
        // ---- stage B: g0 = (W1^T g1) * [h0 > 0] -----------------------------------------------------------------------------------
        mk = relu_mask(a.h0 + row * 128 + 4 * h);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x16{0};
#pragma unroll
        for (int s = 0; s < 8; ++s) kstep<4>(acc, at(oA1 + s * (4 * 2 * 64 * 16)), O.h[s], O.l[s]);
        {
            const float us = pow2i(-E);
            float* __restrict__ gr = a.g0 + row * 128 + 4 * h;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const unsigned q = (unsigned)(mk >> (16 * u + 4 * b));
                    acc[u][4 * b] = (q & 1u) ? acc[u][4 * b] * inv1 : 0.f;
                    acc[u][4 * b + 1] = (q & 2u) ? acc[u][4 * b + 1] * inv1 : 0.f;
                    acc[u][4 * b + 2] = (q & 4u) ? acc[u][4 * b + 2] * inv1 : 0.f;
                    acc[u][4 * b + 3] = (q & 8u) ? acc[u][4 * b + 3] * inv1 : 0.f;
                    if (live) *reinterpret_cast<float4*>(gr + 32 * u + 8 * b) =
                        make_float4(acc[u][4 * b] * us, acc[u][4 * b + 1] * us, acc[u][4 * b + 2] * us, acc[u][4 * b + 3] * us);
                }
        }
        renorm_split(acc, E, O, neg1);
        // ---- stage C: gx = W0^T g0 in three passes of 128 encoding rows, folded into the feature gradients -------------------------------
        float gfe[14];
#pragma unroll
        for (int e = 0; e < 14; ++e) gfe[e] = 0.f;
        SinCos SC{0.f, 1.f};
#define BSS_CHUNK(C)                                                                                                              \
        {                                                                                                                         \
            __syncthreads();   /* chunk C landed (every wave waited for its DMA pieces); chunk C - 1 read by every wave */        \
            S.dma<0>(RING + ((C + 2) % 3) * kChunk, (C + 2) % 12); S.dma<1>(RING + ((C + 2) % 3) * kChunk, (C + 2) % 12);       \
            kstep<4>(acc, at(oRing + (C % 3) * kChunk * 16), O.h[2 * (C % 4)], O.l[2 * (C % 4)]);                                \
            kstep<4>(acc, at(oRing + (C % 3) * kChunk * 16 + 4 * 2 * 64 * 16), O.h[2 * (C % 4) + 1], O.l[2 * (C % 4) + 1]);      \
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x16{0};
        BSS_CHUNK(0) BSS_CHUNK(1) BSS_CHUNK(2) BSS_CHUNK(3)
        pe_fold<0>(acc, inv0, f, SC, gfe);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x16{0};
        BSS_CHUNK(4) BSS_CHUNK(5) BSS_CHUNK(6) BSS_CHUNK(7)
        pe_fold<1>(acc, inv0, f, SC, gfe);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x16{0};
        BSS_CHUNK(8) BSS_CHUNK(9) BSS_CHUNK(10) BSS_CHUNK(11)
        pe_fold<2>(acc, inv0, f, SC, gfe);
#undef BSS_CHUNK
        // ---- gf out; stage D: gX = Wb^T gf ------------------------------------------------------------------------------------------------
        if (live) {
            const float us = pow2i(-E);
            float2* __restrict__ fr = reinterpret_cast<float2*>(a.gf + row * 32 + 14 * h);
#pragma unroll
            for (int e = 0; e < 7; ++e) fr[e] = make_float2(gfe[2 * e] * us, gfe[2 * e + 1] * us);   // (half 1's last value is "feature 27": zero rows of W0^T)
            if (h == 1) { fr[7] = make_float2(0.f, 0.f); fr[8] = make_float2(0.f, 0.f); }
        }
        uint4 Gh[2], Gl[2];
        {
            unsigned mb = 0u;
#pragma unroll
            for (int e = 0; e < 14; ++e) mb = max(mb, __float_as_uint(gfe[e]) & 0x7fffffffu);
            mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));
            const int k = shift_of(mb);
            E += k;
            const float sc = pow2i(k);
            split2(gfe[0] * sc, gfe[1] * sc, neg1, Gh[0].x, Gl[0].x);
            split2(gfe[2] * sc, gfe[3] * sc, neg1, Gh[0].y, Gl[0].y);
            split2(gfe[4] * sc, gfe[5] * sc, neg1, Gh[0].z, Gl[0].z);
            split2(gfe[6] * sc, gfe[7] * sc, neg1, Gh[0].w, Gl[0].w);
            split2(gfe[8] * sc, gfe[9] * sc, neg1, Gh[1].x, Gl[1].x);
            split2(gfe[10] * sc, gfe[11] * sc, neg1, Gh[1].y, Gl[1].y);
            split2(gfe[12] * sc, gfe[13] * sc, neg1, Gh[1].z, Gl[1].z);
            Gh[1].w = 0u; Gl[1].w = 0u;
        }
        f32x16 ax[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) ax[u] = f32x16{0};
        kstep<5>(ax, at(oAb), Gh[0], Gl[0]);
        kstep<5>(ax, at(oAb + 5 * 2 * 64 * 16), Gh[1], Gl[1]);
        if (live) {
            const float us = pow2i(-E) * invb;
            float* __restrict__ xr = a.gx + row * 144 + 4 * h;
#pragma unroll
            for (int u = 0; u < 5; ++u)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (32 * u + 8 * b >= 144) continue;
                    *reinterpret_cast<float4*>(xr + 32 * u + 8 * b) =
                        make_float4(ax[u][4 * b] * us, ax[u][4 * b + 1] * us, ax[u][4 * b + 2] * us, ax[u][4 * b + 3] * us);
                }
        }
    }
}

// ---- operand packing -----------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w2; const float* w1; const float* w0; const float* wb;   // [3,128], [128,128], [128,351], [27,144]
    uint4* a2; uint4* a1; uint4* a0; uint4* ab; const unsigned* absmax; float* scales;   // scales[t] = 2^k, scales[4 + t] = 2^-k
};
__device__ __forceinline__ float scale_of(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3e38f)) return 1.f;
    int k;
    (void)frexpf(mx, &k);
    return ldexpf(1.f, 13 - k);
}
__device__ __forceinline__ unsigned pack2(float x0, float x1, int part) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    h2v r;
    if (part == 0) { r[0] = h0; r[1] = h1; }
    else { r[0] = (_Float16)(x0 - (float)h0); r[1] = (_Float16)(x1 - (float)h1); }
    return __builtin_bit_cast(unsigned, r);
}
__host__ __device__ inline int hid_unit(int s, int kh, int e) { return 32 * (s / 2) + 8 * (2 * (s % 2) + e / 4) + 4 * kh + e % 4; }
// reference column of encoding row (tile U, row i): i = 8 b + 4 hh + t is value 16 U + 4 b + t of lane half hh; -1: padding
__host__ __device__ inline int enc_col(int U, int i) {
    const int b = i >> 3, hh = (i >> 2) & 1, t = i & 3, v = 16 * U + 4 * b + t;
    if (v >= 168) { const int r = v - 168, F = 14 * hh + r; return (r < 14 && F < 27) ? F : -1; }
    const int F = 14 * hh + v / 12, o = (v % 12) >> 1, sc = v & 1;
    return F < 27 ? (sc ? 189 : 27) + F * 6 + o : -1;
}
__device__ __forceinline__ unsigned block_absmax(const float* __restrict__ p, int n, unsigned* sh) {
    unsigned m = 0u;
    for (int i = threadIdx.x; i < n; i += 256) m = max(m, __float_as_uint(p[i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    return max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}
// SELF: every workgroup derives the four matrices' largest magnitudes itself (66 000 values from L2) instead of reading them from a
// k_bss_absmax launch in front of this one: one launch per re-pack (the fused training step re-packs behind every optimiser step)
template <bool SELF>
__global__ __launch_bounds__(256) void k_pack_bwd_ss(const PackArgs a) {
    float s2, s1, s0, sb;
    if (SELF) {
        __shared__ unsigned sh[4];
        s2 = scale_of(block_absmax(a.w2, 3 * 128, sh)); s1 = scale_of(block_absmax(a.w1, 128 * 128, sh));
        s0 = scale_of(block_absmax(a.w0, 128 * 351, sh)); sb = scale_of(block_absmax(a.wb, 27 * 144, sh));
    } else {
        s2 = scale_of(a.absmax[0]); s1 = scale_of(a.absmax[1]); s0 = scale_of(a.absmax[2]); sb = scale_of(a.absmax[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const float s = threadIdx.x == 0 ? s2 : (threadIdx.x == 1 ? s1 : (threadIdx.x == 2 ? s0 : sb));
        a.scales[threadIdx.x] = s; a.scales[4 + threadIdx.x] = 1.f / s;
    }
    const int total = kA2 + kA1 + kA0 + kAb;
    for (int gid = blockIdx.x * 256 + threadIdx.x; gid < total; gid += gridDim.x * 256) {
        int g = gid;
        float x[8];
        uint4* dst;
        int part;
        if (g < kA2) {                       // [u][part][lane]: unit 32 u + i, K value (kh, e) = channel e of go for kh = 0, e < 3
            const int lane = g & 63, u = g >> 7; part = (g >> 6) & 1;
            const int unit = 32 * u + (lane & 31), kh = lane >> 5;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (kh == 0 && e < 3) ? a.w2[e * 128 + unit] * s2 : 0.f;
            dst = a.a2 + g;
        } else if ((g -= kA2) < kA1) {       // [s][u][part][lane]: in-unit 32 u + i, K = out-unit hid_unit(s, kh, e)
            const int lane = g & 63, u = (g >> 7) & 3, s = g >> 9; part = (g >> 6) & 1;
            const int in = 32 * u + (lane & 31), kh = lane >> 5;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = a.w1[hid_unit(s, kh, e) * 128 + in] * s1;
            dst = a.a1 + g;
        } else if ((g -= kA1) < kA0) {       // [chunk = 4 pass + step pair][st][ut][part][lane]: encoding row (4 pass + ut, i), K = hid_unit
            const int lane = g & 63, ut = (g >> 7) & 3, st = (g >> 9) & 1, c = g >> 10; part = (g >> 6) & 1;
            const int U = 4 * (c / 4) + ut, s = 2 * (c % 4) + st, kh = lane >> 5;
            const int col = enc_col(U, lane & 31);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = col >= 0 ? a.w0[hid_unit(s, kh, e) * 351 + col] * s0 : 0.f;
            dst = a.a0 + g;
        } else {                             // [s][u 5][part][lane]: channel 32 u + i, K = feature 14 kh + 8 s + e (8 s + e < 14)
            g -= kA0;
            const int lane = g & 63, q = g >> 7, u = q % 5, s = q / 5; part = (g >> 6) & 1;
            const int ch = 32 * u + (lane & 31), kh = lane >> 5;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int gi = 8 * s + e, F = 14 * kh + gi;
                x[e] = (gi < 14 && F < 27 && ch < 144) ? a.wb[F * 144 + ch] * sb : 0.f;
            }
            dst = a.ab + g;
        }
        *dst = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
    }
}
struct AbsMaxArgs { const float* p[4]; int n[4]; unsigned* out; };
__global__ __launch_bounds__(256) void k_bss_absmax(const AbsMaxArgs a) {
    const int t = blockIdx.y;
    unsigned m = 0u;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.n[t]; i += gridDim.x * 256) m = max(m, __float_as_uint(a.p[t][i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&a.out[t], m);
}

}  // namespace bss

size_t mlp_bwd_ss_pack_bytes() {
    using namespace bss;
    return (size_t)(kA2 + kA1 + kA0 + kAb) * 16 + 64;
}
// the fused chain for the MLP_Fea_noview head (app_dim 27, 351 encoded inputs): packs the transposed weights (once per backward), then
// g1 -> h1 (in place), g0, gf, gx for `rows` (a multiple of 32) appearance rows
// the four absmax words of the pack buffer (the caller may zero them with its own setup kernel: mlp_bwd_ss_pack(..., zeroed = true))
void* mlp_bwd_ss_absmax_words(void* packbuf) {
    using namespace bss;
    return (void*)((uint4*)packbuf + kA2 + kA1 + kA0 + kAb);
}
// The transposed split-f16 operands of the chain from the CURRENT weights: two small launches that depend on nothing but the parameters —
// the backward runs them before it forks its side streams (queued behind the density scatter's workgroups, the 6-us pack kernel sat
// 50 us on the critical path in front of k_mlp_bwd_ss).
int mlp_bwd_ss_pack(t2n_field* f, void* packbuf, hipStream_t s, bool zeroed, bool one_launch, bool absmax_done) {
    using namespace bss;
    const t2n_field_params& p = f->params_ref;
    uint4* base = (uint4*)packbuf;
    PackArgs pa;
    pa.w2 = p.mlp_w2; pa.w1 = p.mlp_w1; pa.w0 = p.mlp_w0; pa.wb = p.basis_weight;
    pa.a2 = base; pa.a1 = base + kA2; pa.a0 = base + kA2 + kA1; pa.ab = base + kA2 + kA1 + kA0;
    unsigned* am = (unsigned*)(base + kA2 + kA1 + kA0 + kAb);
    pa.absmax = am; pa.scales = (float*)(am + 4);
    if (one_launch) {
        hipLaunchKernelGGL(k_pack_bwd_ss<true>, dim3(64), dim3(256), 0, s, pa);
        T2N_HIP(hipGetLastError());
        return T2N_OK;
    }
    if (absmax_done) {   // the caller's previous launch left the absmax words (launch_pack_mlp)
        hipLaunchKernelGGL(k_pack_bwd_ss<false>, dim3(64), dim3(256), 0, s, pa);
        T2N_HIP(hipGetLastError());
        return T2N_OK;
    }
    if (!zeroed) T2N_HIP(hipMemsetAsync(am, 0, 16, s));
    AbsMaxArgs m;
    m.p[0] = p.mlp_w2; m.n[0] = 3 * 128; m.p[1] = p.mlp_w1; m.n[1] = 128 * 128; m.p[2] = p.mlp_w0; m.n[2] = 128 * 351; m.p[3] = p.basis_weight; m.n[3] = 27 * 144;
    m.out = am;
    hipLaunchKernelGGL(k_bss_absmax, dim3(16, 4), dim3(256), 0, s, m);
    hipLaunchKernelGGL(k_pack_bwd_ss<false>, dim3(64), dim3(256), 0, s, pa);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

int launch_mlp_bwd_ss(t2n_field* f, void* packbuf, const float4* go, float* h1, const float* h0, const float* feat, float* g0, float* gf,
                      float* gx, long long rows, hipStream_t s, bool packed, const unsigned* rows_dev, float* g1_out) {
    using namespace bss;
    if (!packed) { const int rc = mlp_bwd_ss_pack(f, packbuf, s, false); if (rc) return rc; }
    uint4* base = (uint4*)packbuf;
    PackArgs pa;
    pa.a2 = base; pa.a1 = base + kA2; pa.a0 = base + kA2 + kA1; pa.ab = base + kA2 + kA1 + kA0;
    pa.scales = (float*)((unsigned*)(base + kA2 + kA1 + kA0 + kAb) + 4);
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_mlp_bwd_ss, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
        attr_set = true;
    }
    Args a;
    a.a2 = pa.a2; a.a1 = pa.a1; a.a0 = pa.a0; a.ab = pa.ab; a.inv_scale = pa.scales + 4;
    a.go = go; a.h1 = h1; a.g1 = g1_out ? g1_out : h1; a.h0 = h0; a.feat = feat; a.g0 = g0; a.gf = gf; a.gx = gx; a.rows = rows; a.rows_dev = rows_dev; a.neg1 = -1.f;
    const long long nrounds = (rows / 32 + 7) / 8;
    hipLaunchKernelGGL(k_mlp_bwd_ss, dim3((unsigned)(nrounds < 256 ? nrounds : 256)), dim3(512), kLds, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n
