// Input-gradient GEMMs of the MLP backward on the f16 matrix cores: OUT[rows, N] = (IN[rows, K] W[K, N]) (* [ACT > 0]).
//
// Replaces (reference): the autograd of nn.Linear / basis_mat w.r.t. their inputs (models/tensorBase.py:94-109,
// models/tensoRF.py:147,239; triggered by text2nerf_main.py:589): g0 = (g1 W1) * [h0 > 0], gx = g0 W0, gX = gf Wb.
//
// Same arithmetic as the forward head (t2n_mlp_ss.hip): every fp32 product is three v_mfma_f32_32x32x16_f16 products of hi / lo
// f16 splits (x = hi + lo, RTZ; the lo*lo term is dropped: ~2^-21 relative), fp32 accumulate. Gradients span many orders of
// magnitude, f16 does not: every ROW of IN (one appearance sample's gradient vector) is scaled by its own power of two — the row's
// largest magnitude lands in [2^13, 2^14) — and the result row is scaled back (exact: the product is linear in the row); the
// weights carry one power of two per matrix (largest magnitude in [2^12, 2^13), k_gemm_h_absmax + k_gemm_h_pack, once per backward).
//
// Mapping: rows on the MFMA N axis (lane (s, h): row s of the wave's 32-row tile, K-half h), output columns on M. A workgroup owns one
// 128-column group of N: its slab of W^T (packed A operands, hi + lo) is copied to LDS once, then its 4 waves walk 32-row tiles
// (grid-stride). A lane loads its half of the row once (K <= 128: 64 values), finds the row's scale with its partner lane, converts
// to packed hi / lo halves in registers, and the K loop is LDS reads + MFMAs only. The fp32 form of this GEMM (k_gemm_nn,
// v_mfma_f32_32x32x2_f32) ran at 40-50 % of the fp32 matrix peak: 236 us per C3 iteration for the three calls.
#include <type_traits>

#include "t2n_device.h"

namespace t2n {
namespace gh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __fp16 hh2 __attribute__((ext_vector_type(2)));

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

// max |x| of up to three tensors as uint bit patterns (orders like |x| for finite values): out[t] = max, out must be zeroed
struct AbsMaxArgs { const float* p[3]; long long n[3]; unsigned* out; };
__global__ __launch_bounds__(256) void k_gemm_h_absmax(const AbsMaxArgs a) {
    const int t = blockIdx.y;
    unsigned m = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < a.n[t]; i += (long long)gridDim.x * 256) m = max(m, __float_as_uint(a.p[t][i]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(&a.out[t], m);
}

// W [K, N] (row stride ldw) -> packed A operands of W^T for column group ng, K-step st, column block m: uint4 [ng][st][m][part][lane],
// lane (i, kh): column ng * 128 + 32 m + i, K values 16 st + 8 kh + e; scaled by 2^k with max|W| 2^k in [2^12, 2^13)
struct PackDesc { const float* W; int ldw, K, N, steps, ngs; uint4* out; };
struct PackArgs { PackDesc d[3]; const unsigned* absmax; float* scales; };   // scales[t] = 2^k, scales[4 + t] = 2^-k
__device__ __forceinline__ float scale_of(unsigned maxbits) {
    const float mx = __uint_as_float(maxbits);
    if (!(mx > 0.f) || !(mx < 3e38f)) return 1.f;
    int k;
    (void)frexpf(mx, &k);            // mx = m * 2^k, m in [0.5, 1)
    return ldexpf(1.f, 13 - k);
}
__device__ __forceinline__ unsigned pack2(float x0, float x1, int part) {
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    h2v r;
    if (part == 0) { r[0] = h0; r[1] = h1; }
    else { r[0] = (_Float16)(x0 - (float)h0); r[1] = (_Float16)(x1 - (float)h1); }
    return __builtin_bit_cast(unsigned, r);
}
__global__ __launch_bounds__(256) void k_gemm_h_pack(const PackArgs a) {
    const int t = blockIdx.y;
    const PackDesc d = a.d[t];
    const float sc = scale_of(a.absmax[t]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.scales[t] = sc; a.scales[4 + t] = 1.f / sc; }
    const long long total = (long long)d.ngs * d.steps * 4 * 2 * 64;
    for (long long g = (long long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long long)gridDim.x * 256) {
        const int lane = (int)(g & 63), part = (int)((g >> 6) & 1), m = (int)((g >> 7) & 3);
        const long long q = g >> 9;
        const int st = (int)(q % d.steps), ng = (int)(q / d.steps);
        const int n = ng * 128 + 32 * m + (lane & 31), k0 = 16 * st + 8 * (lane >> 5);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = (k0 + e < d.K && n < d.N) ? d.W[(size_t)(k0 + e) * d.ldw + n] * sc : 0.f;
        d.out[g] = make_uint4(pack2(x[0], x[1], part), pack2(x[2], x[3], part), pack2(x[4], x[5], part), pack2(x[6], x[7], part));
    }
}

// STEPS = K-steps of 16 (K <= 16 STEPS <= ldin: columns K .. 16 STEPS of IN exist, are finite and zero-weighted)
struct NnArgs {
    const float* IN; int ldin; const uint4* Wp; const float* inv_wscale; long long rows; int N; const float* ACT; int ldact; float* OUT; int ldo;
    float neg1;
};
template <int STEPS>
__global__ __launch_bounds__(256) void k_gemm_nn_h(const NnArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint4 slab[];   // [STEPS][4][2][64]
    const int lane = threadIdx.x & 63, s = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int ng = blockIdx.x;
    constexpr int SLAB = STEPS * 4 * 2 * 64;
    {
        const uint4* __restrict__ src = a.Wp + (size_t)ng * SLAB;
        uint4 v[SLAB / 256];
#pragma unroll
        for (int u = 0; u < SLAB / 256; ++u) v[u] = src[u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < SLAB / 256; ++u) slab[u * 256 + threadIdx.x] = v[u];
    }
    __syncthreads();
    const float inv_w = *a.inv_wscale, neg1 = a.neg1;
    const uint4* __restrict__ A = slab + lane;
    const long long ntiles = (a.rows + 31) / 32;
    for (long long tile = (long long)blockIdx.y * 4 + w; tile < ntiles; tile += (long long)gridDim.y * 4) {
        const long long r = tile * 32 + s;
        const bool rok = r < a.rows;
        const float* __restrict__ inr = a.IN + (rok ? r : 0) * a.ldin + 8 * h;
        // the lane's half of the row: K values 16 st + 8 h .. + 7, all loads in flight
        float4 x[STEPS][2];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            x[st][0] = *reinterpret_cast<const float4*>(inr + 16 * st);       // ldin >= 16 STEPS (checked by the launcher): no
            x[st][1] = *reinterpret_cast<const float4*>(inr + 16 * st + 4);   // per-load conditions, every load in flight at once
        }
        unsigned mb = 0u;
#pragma unroll
        for (int st = 0; st < STEPS; ++st)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float4 v = x[st][q];
                mb = max(max(mb, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                         max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
            }
        mb = max(mb, (unsigned)__shfl_xor((int)mb, 32));
        // 2^e with (row max) 2^e in [2^13, 2^14); an all-zero or non-finite row keeps e = 0; tiny rows stop at 2^100
        int eb = (int)(mb >> 23);
        int e = (mb == 0u || eb == 255) ? 0 : 127 + 13 - eb;
        e = e > 100 ? 100 : e;
        const float sc = __uint_as_float((unsigned)(127 + e) << 23), isc = __uint_as_float((unsigned)(127 - e) << 23) * inv_w;
        uint4 Bh[STEPS], Bl[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const float4 p = x[st][0], q = x[st][1];
            split2(p.x * sc, p.y * sc, neg1, Bh[st].x, Bl[st].x);
            split2(p.z * sc, p.w * sc, neg1, Bh[st].y, Bl[st].y);
            split2(q.x * sc, q.y * sc, neg1, Bh[st].z, Bl[st].z);
            split2(q.z * sc, q.w * sc, neg1, Bh[st].w, Bl[st].w);
        }
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            uint4 ah[4], al[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) { ah[m] = A[((st * 4 + m) * 2) * 64]; al[m] = A[((st * 4 + m) * 2 + 1) * 64]; }
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = mfma(ah[m], Bh[st], acc[m]);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = mfma(al[m], Bh[st], acc[m]);
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = mfma(ah[m], Bl[st], acc[m]);
        }
        if (!rok) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = ng * 128 + m * 32 + 8 * g + 4 * h;
                if (n >= a.ldo) continue;
                float4 v = make_float4(acc[m][4 * g] * isc, acc[m][4 * g + 1] * isc, acc[m][4 * g + 2] * isc, acc[m][4 * g + 3] * isc);
                if (a.ACT) {
                    const float4 t = *reinterpret_cast<const float4*>(a.ACT + r * a.ldact + n);
                    v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(a.OUT + r * a.ldo + n) = v;
            }
    }
}


// ---- weight-gradient GEMMs: part[chunk][M, N] = A^T B over a row chunk (A [rows, lda] gradients, B [rows, ldb] activations; the chunks
// are summed by k_gemm_tn_reduce). Here the contraction runs over ROWS, so no per-row scale can be factored out and the operands span
// fp32's whole exponent range: every fp32 value is split THREE ways into bf16 parts (x = h + m + l by truncation, each residual exact;
// bf16 has fp32's exponent range, so nothing needs scaling) and a product is the six bf16 MFMA products of weight >= 2^-16
// (h h, h m, m h, h l, l h, m m: ~2^-24 relative, fp32 accumulate) on v_mfma_f32_32x32x16_bf16 — 192 matrix cycles per 16 rows of a
// 32 x 32 block against 512 for v_mfma_f32_32x32x2_f32. Lane (i, kh) of a wave loads 8 consecutive rows of ONE column straight
// from global memory (coalesced across the 32 columns of a block), which IS its MFMA operand: wave w converts A block w once per step
// and shares it through LDS (two buffers, one barrier per step); every wave keeps its own B block in registers. One step of rows is
// in flight under the current step's conversion and MFMAs (two steps ahead measured slower: 134 -> 162 us).
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_b(uint4 a, uint4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), c, 0, 0, 0);
}
// two fp32 values -> packed bf16 parts (value 0 in the low half): h / m / l words
__device__ __forceinline__ void split3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
struct Op3 { uint4 h, m, l; };
__device__ __forceinline__ Op3 split_unit(const float (&x)[8]) {
    Op3 o;
    split3(x[0], x[1], o.h.x, o.m.x, o.l.x);
    split3(x[2], x[3], o.h.y, o.m.y, o.l.y);
    split3(x[4], x[5], o.h.z, o.m.z, o.l.z);
    split3(x[6], x[7], o.h.w, o.m.w, o.l.w);
    return o;
}

// PE: B is not read but computed — column c of the positional encoding of the 27 features in `B` (= feat [rows, 32]; reference column
// order, models/tensorBase.py:11-17: [f | sin(f 2^o) feature-major | cos(f 2^o)]) by the forward head's hardware sin on the reduced
// argument (4.2e-7 absolute): the [rows, 352] encoding is never written or read (k_pe_fwd: 73 us + 2 x 161 MB per C3 iteration).
// bsum (may be NULL): += column sums of A (the bias gradient, [128]), added by the ng == 0 blocks.
template <int MB, bool PE>
__global__ __launch_bounds__(256) void k_gemm_tn_b(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                   long long rows, int N, float* __restrict__ part, int ldp, int chunk_rows,
                                                   float* __restrict__ bsum, const unsigned* __restrict__ rows_dev) {
    if (rows_dev) {   // (see k_bwd_l2) the chunks are re-cut for the actual row count: every workgroup of the capacity-sized grid gets its share
        rows = rows < (long long)*rows_dev ? rows : (long long)*rows_dev;
        const long long c = ((rows + gridDim.y - 1) / gridDim.y + 31) / 32 * 32;
        chunk_rows = c < 64 ? 64 : (int)c;
    }
    __shared__ __attribute__((aligned(16))) uint4 sA[2][MB][3][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, kh = lane >> 5;
    const int ng = blockIdx.x;
    const long long r0 = (long long)blockIdx.y * chunk_rows;
    const long long r1 = (r0 + chunk_rows < rows) ? r0 + chunk_rows : rows;
    const bool loads_a = w < MB;
    const int bcol = ng * 128 + 32 * w + i;
    // PE: the lane's column as (feature, 2^octave, phase offset in revolutions: 0 = sin, 0.25 = cos), raw features apart
    int pf = 0; float pscale = 1.f, poff = 0.f; bool praw = false;
    bool bok;
    if constexpr (PE) {
        bok = bcol < 351;
        if (bcol < 27) { pf = bcol; praw = true; }
        else if (bok) { const int q = (bcol - 27) % 162; pf = q / 6; pscale = (float)(1 << (q % 6)); poff = bcol >= 189 ? 0.25f : 0.f; }
    } else {
        bok = bcol < ldb && bcol < N + 3;                  // columns N .. ldb are padding the caller ignores
    }
    const float* __restrict__ Ab = A + (loads_a ? 32 * w + i : 0) + r0 * lda;
    const float* __restrict__ Bb = B + (PE ? pf : (bok ? bcol : 0)) + r0 * ldb;
    // rows are addressed chunk-relative with 32-bit offsets, and only a chunk's LAST step can be partial: the full steps run without the
    // per-element row clamps and masks (64-bit compares and selects on 16 elements were a third of the kernel's VALU instructions, and the
    // kernel is VALU-bound: 346 instructions per 16-row step against 24 MFMAs)
    const int nrows = (int)(r1 - r0);
    float xa[8], xb[8];
    auto fetch = [&](int kt, auto masked) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int r = kt + 8 * kh + e;
            if constexpr (decltype(masked)::value) r = r < nrows ? r : nrows - 1;   // clamped address, masked at the conversion
            xa[e] = Ab[r * lda];
            xb[e] = Bb[r * ldb];
        }
    };
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x16{0};
    float bs = 0.f;
    int buf = 0;
    auto step = [&](int kt, auto masked) {
        float a8[8], b8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            bool ok = true;
            if constexpr (decltype(masked)::value) ok = kt + 8 * kh + e < nrows;
            a8[e] = ok ? xa[e] : 0.f;
            float v = xb[e];
            if constexpr (PE) {
                // f / (2 pi) as th + tl (two-constant product), the fraction of its 2^o multiple, + a quarter turn for the cosine
                const float C1 = 0.15915494309189535f;
                const float C2 = (float)(0.15915494309189533576888 - (double)C1);
                const float th = v * C1;
                const float tl = fmaf(v, C1, -th) + v * C2;
                const float arg = fmaf(tl, pscale, __builtin_amdgcn_fractf(th * pscale)) + poff;
                v = praw ? v : __builtin_amdgcn_sinf(arg);
            }
            b8[e] = (ok && bok) ? v : 0.f;
        }
        // the next step's rows in flight under this step's conversion and MFMAs (the one after the last full step may be the partial one)
        if (kt + 32 <= nrows) fetch(kt + 16, std::false_type{});
        else if (kt + 16 < nrows) fetch(kt + 16, std::true_type{});
        if (loads_a) {
            bs += ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
            const Op3 oa = split_unit(a8);
            sA[buf][w][0][lane] = oa.h; sA[buf][w][1][lane] = oa.m; sA[buf][w][2][lane] = oa.l;
        }
        const Op3 ob = split_unit(b8);
        __syncthreads();   // this step's A blocks are in buffer `buf`; the other buffer's readers finished before the previous barrier
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint4 ah = sA[buf][m][0][lane], am = sA[buf][m][1][lane], al = sA[buf][m][2][lane];
            acc[m] = mfma_b(ah, ob.h, acc[m]);
            acc[m] = mfma_b(ah, ob.m, acc[m]);
            acc[m] = mfma_b(am, ob.h, acc[m]);
#if !defined(T2N_EXP_GEMM_3MFMA)   // (experiment: what the kernel costs with half the matrix work — results are then wrong)
            acc[m] = mfma_b(ah, ob.l, acc[m]);
            acc[m] = mfma_b(al, ob.h, acc[m]);
            acc[m] = mfma_b(am, ob.m, acc[m]);
#endif
        }
        buf ^= 1;
    };
    if (nrows >= 16) fetch(0, std::false_type{});
    else if (nrows > 0) fetch(0, std::true_type{});
    int kt = 0;
    for (; kt + 16 <= nrows; kt += 16) step(kt, std::false_type{});
    if (kt < nrows) step(kt, std::true_type{});
    float* __restrict__ P = part + (size_t)blockIdx.y * (MB * 32) * ldp + ng * 128 + w * 32 + i;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) P[(size_t)(m * 32 + (v & 3) + 8 * (v >> 2) + 4 * kh) * ldp] = acc[m][v];
    if (bsum && ng == 0 && loads_a) {   // one atomic per column and chunk (~400 per address, like k_colsum's)
        bs += __shfl_xor(bs, 32);
        if (kh == 0) atomicAdd(&bsum[32 * w + i], bs);
    }
}
}  // namespace gh

// ---- host side: one pack per backward (the weights change every optimiser step), three GEMM calls -------------------------------------
// pack buffer layout (uint4): W1^T [1][8] | W0^T [ngs0][8] | Wb^T [2][STEPSb]; then 3 absmax words + 8 scale floats
size_t gemm_h_pack_bytes(int K0) {
    const size_t ngs0 = (size_t)(K0 + 127) / 128;
    return ((size_t)8 + ngs0 * 8 + (size_t)2 * 2) * 512 * 16 + 64;
}
int gemm_h_pack(t2n_field* f, void* buf, int K0, hipStream_t s) {
    using namespace gh;
    const t2n_field_params& p = f->params_ref;
    const int ngs0 = (K0 + 127) / 128;
    uint4* base = (uint4*)buf;
    PackArgs a;
    a.d[0] = PackDesc{p.mlp_w1, 128, 128, 128, 8, 1, base};
    a.d[1] = PackDesc{p.mlp_w0, K0, 128, K0, 8, ngs0, base + (size_t)8 * 512};
    a.d[2] = PackDesc{p.basis_weight, 144, f->desc.app_dim, 144, 2, 2, base + (size_t)(8 + ngs0 * 8) * 512};
    unsigned* am = (unsigned*)(base + (size_t)(8 + ngs0 * 8 + 4) * 512);
    a.absmax = am; a.scales = (float*)(am + 4);
    T2N_HIP(hipMemsetAsync(am, 0, 16, s));
    AbsMaxArgs m;
    m.p[0] = p.mlp_w1; m.n[0] = 128 * 128; m.p[1] = p.mlp_w0; m.n[1] = (long long)128 * K0; m.p[2] = p.basis_weight; m.n[2] = (long long)f->desc.app_dim * 144;
    m.out = am;
    hipLaunchKernelGGL(k_gemm_h_absmax, dim3(16, 3), dim3(256), 0, s, m);
    hipLaunchKernelGGL(k_gemm_h_pack, dim3(24, 3), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
// which: 0 = W1 (K = N = 128), 1 = W0 (K = 128, N = K0), 2 = basis (K = app_dim <= 32, N = 144)
int launch_gemm_nn_h(void* packbuf, int which, int K0, const float* IN, int ldin, long long rows, const float* ACT, int ldact, float* OUT,
                     int ldo, hipStream_t s) {
    using namespace gh;
    const int ngs0 = (K0 + 127) / 128;
    uint4* base = (uint4*)packbuf;
    const float* scales = (const float*)((unsigned*)(base + (size_t)(8 + ngs0 * 8 + 4) * 512) + 4);
    NnArgs a;
    a.IN = IN; a.ldin = ldin; a.rows = rows; a.ACT = ACT; a.ldact = ldact; a.OUT = OUT; a.ldo = ldo; a.neg1 = -1.f;
    a.inv_wscale = scales + 4 + which;
    int ng, steps;
    if (which == 0) { a.Wp = base; a.N = 128; ng = 1; steps = 8; }
    else if (which == 1) { a.Wp = base + (size_t)8 * 512; a.N = K0; ng = ngs0; steps = 8; }
    else { a.Wp = base + (size_t)(8 + ngs0 * 8) * 512; a.N = 144; ng = 2; steps = 2; }
    if (ldin < 16 * steps || (ldin & 3) || (ldo & 3)) { set_error("launch_gemm_nn_h: ldin %d / ldo %d do not fit %d K-steps", ldin, ldo, steps); return T2N_ERR_INVALID; }
    const long long tiles4 = ((rows + 31) / 32 + 3) / 4;
    long long by = 512 / ng;                       // ~2 workgroups per CU (the LDS slab allows two), each copying its slab once
    if (by > tiles4) by = tiles4;
    if (by < 1) by = 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_gemm_nn_h<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 512 * 16);
        attr_set = true;
    }
    if (steps == 8) hipLaunchKernelGGL(k_gemm_nn_h<8>, dim3((unsigned)ng, (unsigned)by), dim3(256), (size_t)8 * 512 * 16, s, a);
    else hipLaunchKernelGGL(k_gemm_nn_h<2>, dim3((unsigned)ng, (unsigned)by), dim3(256), (size_t)2 * 512 * 16, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// part[chunks][128][ldp] = per-chunk A^T B for the 128-column gradients (the caller reduces the chunks). pe: B = feat [rows, 32] and the
// product is with its positional encoding (N = 351). db (may be NULL): += column sums of A.
int launch_gemm_tn_b(const float* A, int lda, const float* B, int ldb, long long rows, int N, float* part, int ldp, int chunk_rows,
                     int ng, int chunks, bool pe, float* db, hipStream_t s, const unsigned* rows_dev) {
    using namespace gh;
    if (pe) hipLaunchKernelGGL((k_gemm_tn_b<4, true>), dim3((unsigned)ng, (unsigned)chunks), dim3(256), 0, s, A, lda, B, ldb, rows, N, part, ldp, chunk_rows, db, rows_dev);
    else hipLaunchKernelGGL((k_gemm_tn_b<4, false>), dim3((unsigned)ng, (unsigned)chunks), dim3(256), 0, s, A, lda, B, ldb, rows, N, part, ldp, chunk_rows, db, rows_dev);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n
