// MLP part of the render backward (the driver is t2n_backward.hip): layer 2 on the VALU (3 outputs), the fp32-MFMA
// GEMMs of the exact / generic paths (dW = g^T x over row chunks + a reduce, g_in = (g W) * [act > 0]), bias column sums, and the
// positional encoding's forward / backward for heads whose GEMMs do not compute it themselves. The f16 / bf16 forms of these GEMMs live
// in t2n_gemm_h.hip, the fused input-gradient chain of the MLP_Fea_noview head in t2n_mlp_bwd_ss.hip. What autograd derives for
// models/tensorBase.py:88-109 (MLPRender_Fea_noview) and :11-17 (positional_encoding).
#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ constexpr int unit_of(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }   // row of accumulator register v, lane half h
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// ---- layer 2 (3 outputs): VALU ------------------------------------------------------------------------------------------
// go [rows,4], h1 [rows,128] -> g1 [rows,128] = (go W2) * [h1>0]; dW2[3,128] += go^T h1; db2[3] += colsum(go)
// rows_dev (all row-streaming kernels of the backward, optional): the row count in DEVICE memory — the backward of a speculative
// train step (T2N_FLAG_DEVICE_ROWS) sizes its grids and buffers from a row CAPACITY and never reads the count on the host; `rows` is
// then the capacity and the kernels clip it (k_bwd_plan, t2n_backward.hip)
__global__ __launch_bounds__(256) void k_bwd_l2(const float4* __restrict__ go, const float* __restrict__ h1, long long rows,
                                                const float* __restrict__ w2, float* g1, float* __restrict__ part, const unsigned* __restrict__ rows_dev) {
    if (rows_dev) rows = rows < (long long)*rows_dev ? rows : (long long)*rows_dev;
    // The kernel is a [rows, 128] stream with 3 x 128 running sums: memory-latency bound unless enough of it is in flight. Thread
    // (c, q) = (tid & 31, tid >> 5) owns units 4c .. 4c+3 (one 16-byte load per row) of rows base + 8k + q, k = 0..7: eight rows per
    // thread and 64 rows per workgroup in flight, 1024 workgroups (grid-stride). The one-dword-per-thread, four-rows-in-flight,
    // one-workgroup-per-CU form of this kernel took 183 us per C3 iteration for 150 MB.
    const int tid = threadIdx.x, c = tid & 31, q = tid >> 5;
    const float4 w0 = *reinterpret_cast<const float4*>(w2 + 4 * c), w1 = *reinterpret_cast<const float4*>(w2 + 128 + 4 * c),
                 w2v = *reinterpret_cast<const float4*>(w2 + 256 + 4 * c);
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (long long base = (long long)blockIdx.x * 64; base < rows; base += (long long)gridDim.x * 64) {
        float4 g[8], h[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long r = base + 8 * k + q;
            const long long rc = r < rows ? r : rows - 1;   // clamped address + select: a conditional load is a branch and a full wait per element
            g[k] = go[rc];
            h[k] = *reinterpret_cast<const float4*>(h1 + rc * 128 + 4 * c);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const long long r = base + 8 * k + q;
            const bool ok = r < rows;
            const float gx = ok ? g[k].x : 0.f, gy = ok ? g[k].y : 0.f, gz = ok ? g[k].z : 0.f;
            const float4 hv = h[k];
            a0.x = fmaf(gx, hv.x, a0.x); a0.y = fmaf(gx, hv.y, a0.y); a0.z = fmaf(gx, hv.z, a0.z); a0.w = fmaf(gx, hv.w, a0.w);
            a1.x = fmaf(gy, hv.x, a1.x); a1.y = fmaf(gy, hv.y, a1.y); a1.z = fmaf(gy, hv.z, a1.z); a1.w = fmaf(gy, hv.w, a1.w);
            a2.x = fmaf(gz, hv.x, a2.x); a2.y = fmaf(gz, hv.y, a2.y); a2.z = fmaf(gz, hv.z, a2.z); a2.w = fmaf(gz, hv.w, a2.w);
            s0 += gx; s1 += gy; s2 += gz;
            if (g1 && ok) {   // (NULL: the fused input-gradient chain, t2n_mlp_bwd_ss.hip, makes g1 itself and needs h1 intact)
                float4 v;
                v.x = hv.x > 0.f ? fmaf(gz, w2v.x, fmaf(gy, w1.x, gx * w0.x)) : 0.f;
                v.y = hv.y > 0.f ? fmaf(gz, w2v.y, fmaf(gy, w1.y, gx * w0.y)) : 0.f;
                v.z = hv.z > 0.f ? fmaf(gz, w2v.z, fmaf(gy, w1.z, gx * w0.z)) : 0.f;
                v.w = hv.w > 0.f ? fmaf(gz, w2v.w, fmaf(gy, w1.w, gx * w0.w)) : 0.f;
                *reinterpret_cast<float4*>(g1 + r * 128 + 4 * c) = v;   // in place over h1: this thread read the element above
            }
        }
    }
    // per-workgroup partial sums [block][3 x 128 + 3]: the eight row slots meet in LDS, k_bwd_l2_reduce adds the workgroups up in a
    // fixed order (the same-address atomics of ~800 000 threads on 12 cache lines were most of the first form: 105 us per C3 iteration)
    __shared__ float red[8][388];
    *reinterpret_cast<float4*>(&red[q][4 * c]) = a0;
    *reinterpret_cast<float4*>(&red[q][128 + 4 * c]) = a1;
    *reinterpret_cast<float4*>(&red[q][256 + 4 * c]) = a2;
    if (c == 0) { red[q][384] = s0; red[q][385] = s1; red[q][386] = s2; }
    __syncthreads();
    float* __restrict__ P = part + (size_t)blockIdx.x * 388;
    for (int i = tid; i < 387; i += 256)
        P[i] = ((red[0][i] + red[1][i]) + (red[2][i] + red[3][i])) + ((red[4][i] + red[5][i]) + (red[6][i] + red[7][i]));
}
// dw2 [3,128] += sum of the workgroups' partials, db2 [3] likewise: one workgroup per output, thread b sums partials b, b + 256, ...
// in order, then a fixed-order tree (deterministic)
__device__ __forceinline__ void bwd_l2_reduce_body(const float* __restrict__ part, int nblocks, float* dw2, float* db2, int i, float* red) {
    const int b = threadIdx.x;
    float v = 0.f;
    for (int k = b; k < nblocks; k += 256) v += part[(size_t)k * 388 + i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((b & 63) == 0) red[b >> 6] = v;
    __syncthreads();
    if (b == 0) {
        const float s = (red[0] + red[1]) + (red[2] + red[3]);
        if (i < 384) { if (dw2) dw2[i] += s; }
        else if (db2) db2[i - 384] += s;
    }
}
__global__ __launch_bounds__(256) void k_bwd_l2_reduce(const float* __restrict__ part, int nblocks, float* dw2, float* db2) {
    __shared__ float red[4];
    bwd_l2_reduce_body(part, nblocks, dw2, db2, blockIdx.x, red);
}

// ---- fp32 MFMA GEMMs -----------------------------------------------------------------------------------------------------
// C[M,N] += A^T B, split over row chunks: A [rows, lda] (gradients, M <= MB*32 <= lda), B [rows, ldb] (activations, ldb % 4 == 0,
// columns >= N up to ldb are zero or ignored). A workgroup (4 waves) owns one row chunk x one 128-column group of N: 32-row
// tiles of A and B go global -> registers (float4, coalesced; the next tile's loads are in flight while the current one is
// multiplied) -> LDS; wave w multiplies all MB row blocks of M against its 32-column block. The chunk's partial product
// is stored to `part` [chunks][MB*32][ldp] and k_gemm_tn_reduce sums the chunks into C (deterministic, no atomics).
template <int MB>
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                 long long rows, int N, float* __restrict__ part, int ldp, int chunk_rows, const unsigned* __restrict__ rows_dev) {
    if (rows_dev) {   // (see k_bwd_l2; chunks re-cut for the actual row count)
        rows = rows < (long long)*rows_dev ? rows : (long long)*rows_dev;
        const long long c = ((rows + gridDim.y - 1) / gridDim.y + 31) / 32 * 32;
        chunk_rows = c < 64 ? 64 : (int)c;
    }
    constexpr int MA = MB * 32;            // staged A columns
    constexpr int A4 = MA / 4;             // float4 per staged A row
    constexpr int NA = (32 * A4) / 256;    // float4 loads per thread for the A tile (MB=4: 4, MB=1: 1)
    __shared__ __attribute__((aligned(16))) float sA[32 * MA];
    __shared__ __attribute__((aligned(16))) float sB[32 * 128];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    const int ng = blockIdx.x;
    const long long r0 = (long long)blockIdx.y * chunk_rows;
    const long long r1 = (r0 + chunk_rows < rows) ? r0 + chunk_rows : rows;
    float4 ra[NA], rb[4];
    auto fetch = [&](long long kt) {
#pragma unroll
        for (int q = 0; q < NA; ++q) {
            const int u = tid + q * 256, row = u / A4, c4 = u % A4;
            const long long r = kt + row;
            ra[q] = r < r1 ? *reinterpret_cast<const float4*>(A + r * lda + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = tid + q * 256, row = u >> 5, c4 = u & 31;
            const long long r = kt + row;
            const int col = ng * 128 + c4 * 4;
            rb[q] = (r < r1 && col < ldb) ? *reinterpret_cast<const float4*>(B + r * ldb + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = f32x16{0};
    if (r0 < r1) fetch(r0);
    for (long long kt = r0; kt < r1; kt += 32) {
        __syncthreads();   // the previous tile has been consumed
#pragma unroll
        for (int q = 0; q < NA; ++q) *reinterpret_cast<float4*>(sA + (size_t)(tid + q * 256) * 4) = ra[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sB + (size_t)(tid + q * 256) * 4) = rb[q];
        __syncthreads();
        if (kt + 32 < r1) fetch(kt + 32);
#pragma unroll 4
        for (int k2 = 0; k2 < 16; ++k2) {
            const int k = 2 * k2 + h;
            const float b = sB[k * 128 + w * 32 + i];
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = mfma(sA[k * MA + m * 32 + i], b, acc[m]);
        }
    }
    float* __restrict__ P = part + (size_t)blockIdx.y * MA * ldp + ng * 128 + w * 32 + i;
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) P[(size_t)(m * 32 + unit_of(v, h)) * ldp] = acc[m][v];
}

// C[M,N] (row-major, ldc) += sum over chunks of part[chunk][MA][ldp]. A workgroup owns 32 consecutive outputs; its 8
// thread groups of 32 each sum every 8th chunk and the partial sums meet in LDS in a fixed order (deterministic, no atomics): a
// 128 x 128 gradient is 512 workgroups instead of 64 (one thread per output left most of the chip idle behind ~500 dependent loads)
__device__ __forceinline__ void gemm_tn_reduce_body(const float* __restrict__ part, int chunks, int MA, int ldp, int M, int N,
                                                    float* __restrict__ C, int ldc, unsigned blk, float (*red)[32]) {
    const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int t = (int)blk * 32 + o;
    const int row = t / ldp, col = t - row * ldp;
    const bool live = row < M && col < N;
    float s0 = 0.f, s1 = 0.f;
    if (live) {
        const float* p = part + (size_t)row * ldp + col;
        const size_t st = (size_t)MA * ldp;
        int c = sl;
        for (; c + 8 < chunks; c += 16) { s0 += p[c * st]; s1 += p[(c + 8) * st]; }
        if (c < chunks) s0 += p[c * st];
    }
    red[sl][o] = s0 + s1;
    __syncthreads();
    if (sl == 0 && live)
        C[(size_t)row * ldc + col] += ((red[0][o] + red[1][o]) + (red[2][o] + red[3][o])) + ((red[4][o] + red[5][o]) + (red[6][o] + red[7][o]));
}
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float* __restrict__ part, int chunks, int MA, int ldp, int M, int N,
                                                        float* __restrict__ C, int ldc) {
    __shared__ float red[8][32];
    gemm_tn_reduce_body(part, chunks, MA, ldp, M, N, C, ldc, blockIdx.x, red);
}
// The fused training step: the partial sums of layer 2 and of the three weight-gradient GEMMs (each in a region of its own) reduced by
// ONE launch behind the last GEMM instead of four launches between them (same per-output order of additions)
struct WgradReduce {
    const float* l2_part; int l2_blocks; float* dw2; float* db2;
    const float* part[3]; int chunks[3], MA[3], ldp[3], M[3], N[3], ldc[3]; float* C[3];
    unsigned block0[5];
    LossReduceArgs loss;   // losses == NULL: none; else the LAST workgroup adds up the loss kernel's partial sums
};
__global__ __launch_bounds__(256) void k_wgrad_reduce(const WgradReduce a) {
    __shared__ float red[8][32];
    const unsigned b = blockIdx.x;
    if (b >= a.block0[4]) { __shared__ float lred[256][3]; loss_reduce_rows(a.loss, lred); return; }
    if (b < a.block0[1]) { bwd_l2_reduce_body(a.l2_part, a.l2_blocks, a.dw2, a.db2, (int)b, &red[0][0]); return; }
    const int t = b < a.block0[2] ? 0 : (b < a.block0[3] ? 1 : 2);
    gemm_tn_reduce_body(a.part[t], a.chunks[t], a.MA[t], a.ldp[t], a.M[t], a.N[t], a.C[t], a.ldc[t], b - a.block0[1 + t], red);
}

// OUT[rows, N] = (IN[rows, K] W[K, N]) (* [ACT > 0] if ACT). A workgroup owns one 128-column group of N: its K x 128 slab
// of W is staged in LDS once (zero-padded), then its 4 waves walk 32-row tiles (grid-stride): rows sit on the MFMA N axis
// (lanes), output columns on M; a lane reads its row of IN as float4 along K (two K-steps per load).
// Requires: K <= 128, ldin % 4 == 0 with IN columns K..round_up(K,4) finite (zero-weighted), ldo % 4 == 0, ldo >= round_up(N, 4).
__global__ __launch_bounds__(256) void k_gemm_nn(const float* __restrict__ IN, int ldin, const float* __restrict__ W, int ldw,
                                                 long long rows, int K, int N, const float* __restrict__ ACT, int ldact,
                                                 float* OUT, int ldo) {
    extern __shared__ __attribute__((aligned(16))) float sW[];   // [K4][128], K4 = round_up(K, 4)
    const int lane = threadIdx.x & 63, s = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int ng = blockIdx.x;
    const int K4 = (K + 3) & ~3;
    // stage the slab with eight independent loads in flight per thread (a one-load-per-iteration loop chained ~64 L2 round
    // trips per workgroup and dominated the kernel)
    for (int base = 0; base < K4 * 128; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int k = idx >> 7, n = ng * 128 + (idx & 127);
            v[u] = (idx < K4 * 128 && k < K && n < N) ? W[(size_t)k * ldw + n] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            if (idx < K4 * 128) sW[idx] = v[u];
        }
    }
    __syncthreads();
    const long long ntiles = (rows + 31) / 32;
    for (long long tile = (long long)blockIdx.y * 4 + w; tile < ntiles; tile += (long long)gridDim.y * 4) {
        const long long r = tile * 32 + s;
        const bool rok = r < rows;
        const float* __restrict__ inr = IN + (rok ? r : 0) * ldin;
        f32x16 acc[4] = {{0}, {0}, {0}, {0}};
        // the lane's row of IN, four float4 (16 K values = 2048 clk of MFMA work) ahead of its use: an L2 round trip is longer
        // than one 8-MFMA step
        constexpr int PF = 4;
        float4 ring[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) ring[p] = 4 * p < K4 ? *reinterpret_cast<const float4*>(inr + 4 * p) : make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < K4; k0 += 4 * PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int kk = k0 + 4 * p;
                if (kk >= K4) break;
                const float4 cur = ring[p];
                if (kk + 4 * PF < K4) ring[p] = *reinterpret_cast<const float4*>(inr + kk + 4 * PF);
                const float b0 = h ? cur.y : cur.x, b1 = h ? cur.w : cur.z;
                const float* w0 = sW + (kk + h) * 128 + s;
                const float* w1 = w0 + 256;
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = mfma(w0[m * 32], b0, acc[m]);
#pragma unroll
                for (int m = 0; m < 4; ++m) acc[m] = mfma(w1[m * 32], b1, acc[m]);
            }
        }
        if (!rok) continue;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = ng * 128 + m * 32 + 8 * g + 4 * h;
                if (n >= ldo) continue;
                float4 v = make_float4(acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]);
                if (ACT) {
                    const float4 t = *reinterpret_cast<const float4*>(ACT + r * ldact + n);
                    v.x = t.x > 0.f ? v.x : 0.f; v.y = t.y > 0.f ? v.y : 0.f; v.z = t.z > 0.f ? v.z : 0.f; v.w = t.w > 0.f ? v.w : 0.f;
                }
                *reinterpret_cast<float4*>(OUT + r * ldo + n) = v;
            }
    }
}

// db[n] += sum_rows G[rows, ld]   (N <= 128; 256 threads: two row phases per block; grid-stride over `chunk`-row tiles so each
// workgroup issues its N atomics once)
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ G, int ld, long long rows, int N, float* db, int chunk) {
    __shared__ float part[256];
    const int n = threadIdx.x & 127, ph = threadIdx.x >> 7;
    float s = 0.f;
    if (n < N)
        for (long long r0 = (long long)blockIdx.x * chunk; r0 < rows; r0 += (long long)gridDim.x * chunk) {
            const long long r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
            float t0 = 0.f, t1 = 0.f;
            long long r = r0 + ph;
            for (; r + 2 < r1; r += 4) { t0 += G[r * ld + n]; t1 += G[(r + 2) * ld + n]; }
            for (; r < r1; r += 2) t0 += G[r * ld + n];
            s += t0 + t1;
        }
    part[threadIdx.x] = s;
    __syncthreads();
    if (ph == 0 && n < N) atomicAdd(&db[n], part[n] + part[128 + n]);
}

// positional encoding forward: feat [rows,32] -> x [rows,352] in the reference's column order (tensorBase.py:11-17)
__global__ __launch_bounds__(256) void k_pe_fwd(const float* __restrict__ feat, long long rows, float* x) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t / 32;
    const int f = (int)(t % 32);
    if (r >= rows) return;
    float* xr = x + r * 352;
    if (f == 31) xr[351] = 0.f;
    if (f >= 27) return;
    const float v = feat[r * 32 + f];
    xr[f] = v;
    float sc = 1.f;
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        float sn, cs;
        sincosf(v * sc, &sn, &cs);
        xr[27 + f * 6 + q] = sn;
        xr[189 + f * 6 + q] = cs;
        sc *= 2.f;
    }
}

// positional encoding backward: gx [rows,352], feat [rows,32] -> gf [rows,32]
__global__ __launch_bounds__(256) void k_pe_bwd(const float* __restrict__ gx, const float* __restrict__ feat, long long rows, float* gf) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t / 32;
    const int f = (int)(t % 32);
    if (r >= rows) return;
    float g = 0.f;
    if (f < 27) {
        const float* gr = gx + r * 352;
        const float v = feat[r * 32 + f];
        g = gr[f];
        float sc = 1.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            float sn, cs;
            sincosf(v * sc, &sn, &cs);
            g = fmaf(gr[27 + f * 6 + q] * sc, cs, g);
            g = fmaf(-(gr[189 + f * 6 + q] * sc), sn, g);
            sc *= 2.f;
        }
    }
    gf[r * 32 + f] = g;
}

// split of a [rows] x (M<=128) x N weight-gradient GEMM into row chunks: ~768 workgroups, chunk a multiple of 32 rows
constexpr int kL2Blocks = 1024;   // workgroups of k_bwd_l2 (four per CU)
struct TnPlan { int chunk_rows, chunks, ng, ldp; };
TnPlan tn_plan(int64_t rows, int N) {
    TnPlan p;
    p.ng = (N + 127) / 128;
    p.ldp = p.ng * 128;
#ifndef T2N_TN_BLOCKS
#define T2N_TN_BLOCKS 512
#endif
    int64_t c = (rows * p.ng + T2N_TN_BLOCKS - 1) / T2N_TN_BLOCKS;   // ~512 workgroups = two per CU (384: GEMM 145 + reduce 54 us per C3 iteration; 512: 123 + 69 before the reduce was widened; 768: 122 + 97)
    c = (c + 31) / 32 * 32;
    if (c < 64) c = 64;
    p.chunk_rows = (int)c;
    p.chunks = (int)((rows + c - 1) / c);
    if (p.chunks < 1) p.chunks = 1;
    return p;
}
// one region per producer (layer 2's per-workgroup sums, then the three GEMMs' chunk partials): the fused training step reduces all four
// with one launch at the end; the composed backward reuses the first region's start for every GEMM in turn
WgradRegions wgrad_regions(int64_t rows, int k0) {
    WgradRegions r;
    size_t o = 0;
    r.l2 = o; o += ((size_t)kL2Blocks * 388 * 4 + 255) / 256 * 256;
    const int shapes[3][2] = {{128, 128}, {128, k0}, {32, 144}};
    for (int i = 0; i < 3; ++i) {
        const TnPlan p = tn_plan(rows, shapes[i][1]);
        r.tn[i] = o; o += ((size_t)p.chunks * shapes[i][0] * p.ldp * 4 + 255) / 256 * 256;
    }
    r.total = o;
    return r;
}
size_t tn_part_bytes(int64_t rows, int k0) { return wgrad_regions(rows, k0).total; }
// layer 2 of the backward: up to kL2Blocks workgroups of 64 rows in flight each, their partial weight / bias sums through `scratch`
// (>= kL2Blocks x 388 floats: the weight-gradient GEMMs' partial buffer, not in use yet)
void launch_bwd_l2(const float4* go, const float* h1, long long rows, const float* w2, float* g1, float* dw2, float* db2,
                          float* scratch, hipStream_t s, const unsigned* rows_dev) {
    const long long tiles = (rows + 63) / 64;
    const unsigned nb = (unsigned)(tiles < kL2Blocks ? tiles : kL2Blocks);
    hipLaunchKernelGGL(k_bwd_l2, dim3(nb), dim3(256), 0, s, go, h1, rows, w2, g1, scratch, rows_dev);
    if (dw2 || db2) hipLaunchKernelGGL(k_bwd_l2_reduce, dim3(387), dim3(256), 0, s, (const float*)scratch, (int)nb, dw2, db2);
}
// the fp32-MFMA GEMMs instead of the f16 / bf16 split ones (t2n_gemm_h.hip, t2n_mlp_bwd_ss.hip): when the field runs its MLP in exact
// fp32 (t2n_field_set_mlp_precision: the backward then keeps fp32 products too)
bool gemm_fp32_mode(const t2n_field* f) { return !f->mlp_split; }
// pe_feat (fused head only): B is the [rows, 352] positional encoding; on the bf16x3 path it is computed from feat [rows, 32] inside the
// GEMM and `B` is never read. db (may be NULL): += column sums of A (the layer's bias gradient).
template <int MB>
static void launch_gemm_tn_t(bool fp32, const float* A, int lda, const float* B, int ldb, long long rows, int M, int N, float* C, int ldc,
                           float* part, hipStream_t s, const float* pe_feat, float* db, const unsigned* rows_dev, bool reduce) {
    const TnPlan p = tn_plan(rows, N);
    if (!fp32 && MB == 4) {   // (the 27-row basis gradient is latency-bound either way: 31 us fp32, 44 us bf16x3)
        (void)launch_gemm_tn_b(A, lda, pe_feat ? pe_feat : B, pe_feat ? 32 : ldb, rows, N, part, p.ldp, p.chunk_rows, p.ng, p.chunks,
                               pe_feat != nullptr, db, s, rows_dev);
    } else {
        hipLaunchKernelGGL((k_gemm_tn<MB>), dim3((unsigned)p.ng, (unsigned)p.chunks), dim3(256), 0, s, A, lda, B, ldb, rows, N, part,
                           p.ldp, p.chunk_rows, rows_dev);
        if (db) hipLaunchKernelGGL(k_colsum, dim3((unsigned)((rows + 127) / 128 < 512 ? (rows + 127) / 128 : 512)), dim3(256), 0, s, A, lda, rows, M, db, 128);
    }
    if (reduce) hipLaunchKernelGGL(k_gemm_tn_reduce, dim3((unsigned)((MB * 32 * p.ldp + 31) / 32)), dim3(256), 0, s, (const float*)part,
                       p.chunks, MB * 32, p.ldp, M, N, C, ldc);
}
void launch_gemm_nn(const float* IN, int ldin, const float* W, int ldw, long long rows, int K, int N, const float* ACT,
                           int ldact, float* OUT, int ldo, hipStream_t s) {
    const int ng = (N + 127) / 128;
    const long long tiles4 = ((rows + 31) / 32 + 3) / 4;
    long long by = 512 / ng;                       // ~2 workgroups per CU (the LDS slab allows two), each staging W once
    if (by > tiles4) by = tiles4;
    if (by < 1) by = 1;
    const size_t lds = (size_t)((K + 3) & ~3) * 128 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_gemm_nn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * 4); attr_set = true; }
    hipLaunchKernelGGL(k_gemm_nn, dim3((unsigned)ng, (unsigned)by), dim3(256), lds, s, IN, ldin, W, ldw, rows, K, N, ACT, ldact, OUT, ldo);
}

// MB = 4: M <= 128 rows of A per workgroup tile; MB = 1: M <= 32 (the 27-row basis gradient)
void launch_gemm_tn(int MB, bool fp32, const float* A, int lda, const float* B, int ldb, long long rows, int M, int N, float* C, int ldc,
                    float* part, hipStream_t s, const float* pe_feat, float* db, const unsigned* rows_dev, bool reduce) {
    if (MB == 4) launch_gemm_tn_t<4>(fp32, A, lda, B, ldb, rows, M, N, C, ldc, part, s, pe_feat, db, rows_dev, reduce);
    else launch_gemm_tn_t<1>(fp32, A, lda, B, ldb, rows, M, N, C, ldc, part, s, pe_feat, db, rows_dev, reduce);
}
// (fused training step) layer 2 + the three GEMMs of the MLP_Fea_noview head, partials at `base` + wgrad_regions(rows, 351): one reduce
void launch_wgrad_reduce(const char* base, long long rows, float* dw2, float* db2, float* dw1, float* dw0, float* dwb, hipStream_t s,
                         const float* loss_part, long long n_rays, float w_depth, float w_trans, float* losses) {
    const WgradRegions R = wgrad_regions(rows, 351);
    WgradReduce a;
    const long long tiles = (rows + 63) / 64;
    a.l2_part = (const float*)(base + R.l2); a.l2_blocks = (int)(tiles < kL2Blocks ? tiles : kL2Blocks); a.dw2 = dw2; a.db2 = db2;
    const int shapes[3][3] = {{128, 128, 128}, {128, 351, 128}, {27, 144, 32}};   // M, N, MA
    float* Cs[3] = {dw1, dw0, dwb};
    unsigned b = 387;
    a.block0[0] = 0; a.block0[1] = b;
    for (int i = 0; i < 3; ++i) {
        const TnPlan p = tn_plan(rows, shapes[i][1]);
        a.part[i] = (const float*)(base + R.tn[i]); a.chunks[i] = p.chunks; a.MA[i] = shapes[i][2]; a.ldp[i] = p.ldp; a.M[i] = shapes[i][0];
        a.N[i] = shapes[i][1]; a.ldc[i] = shapes[i][1]; a.C[i] = Cs[i];
        b += (unsigned)((shapes[i][2] * p.ldp + 31) / 32);
        a.block0[2 + i] = b;
    }
    a.loss = LossReduceArgs{loss_part, (unsigned)((n_rays + 3) / 4), losses, n_rays, w_depth, w_trans};
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(b + (losses ? 1u : 0u)), dim3(256), 0, s, a);
}
static unsigned colsum_grid(long long rows) { return (unsigned)((rows + 127) / 128 < 512 ? (rows + 127) / 128 : 512); }
void launch_colsum(const float* G, int ld, long long rows, int N, float* db, hipStream_t s) {
    hipLaunchKernelGGL(k_colsum, dim3(colsum_grid(rows)), dim3(256), 0, s, G, ld, rows, N, db, 128);
}
void launch_pe_fwd(const float* feat, long long rows, float* x, hipStream_t s) {
    hipLaunchKernelGGL(k_pe_fwd, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, s, feat, rows, x);
}
void launch_pe_bwd(const float* gx, const float* feat, long long rows, float* gf, hipStream_t s) {
    hipLaunchKernelGGL(k_pe_bwd, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, s, gx, feat, rows, gf);
}

}  // namespace t2n
