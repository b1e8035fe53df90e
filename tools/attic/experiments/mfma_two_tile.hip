// Which shape should the sample-stationary head take? One K-step of a 128-unit layer = 12 MFMAs (32x32x16 f16) per 32-sample tile:
// 4 unit tiles x 3 split products, A operands (weights) from LDS as ds_read_b128 (4 per 6 MFMAs and tile set), B operands in registers.
//   TILES = 1, 512 threads: the round-2 kernel's shape (two waves per SIMD, one tile each)
//   TILES = 2 / 4, 256 threads: one wave per SIMD carrying 2 / 4 tiles that SHARE every A operand (accumulators beyond 256 registers
//   go to AccVGPRs)
// FILL = VALU instructions (independent v_fma chains) issued behind every MFMA, order pinned with sched_barrier.
// Prints ns per MFMA and SIMD; 16.2 ns is the bare rate of the part at the clock it holds under this load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int TILES, int FILL, int THREADS, int LDSA>
__global__ __launch_bounds__(THREADS) void k(int iters, float* out, float s) {
    extern __shared__ __attribute__((aligned(16))) uint4 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += THREADS) lds[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x38003800u, 0x3c003c00u);
    __syncthreads();
    unsigned ob = lane * 16u;
    asm volatile("" : "+v"(ob));
    const uint4* __restrict__ LA = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + ob);
    uint4 Bh[TILES], Bl[TILES];
    for (int t = 0; t < TILES; ++t) { Bh[t] = make_uint4(0x3c003c00u + t + lane, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u); Bl[t] = make_uint4(0x1c001c00u + t, 0x1c001c00u, 0x1c001c00u + lane, 0x1c001c00u); }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.01f + e;
    f32x16 acc[TILES][4];
    for (int t = 0; t < TILES; ++t) for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) acc[t][u][e] = 0.f;
    uint4 Ah[2][2], Al[2][2];
    Ah[0][0] = LA[0]; Al[0][0] = LA[64]; Ah[0][1] = LA[128]; Al[0][1] = LA[192];
    Ah[1][0] = LA[256]; Al[1][0] = LA[320]; Ah[1][1] = LA[384]; Al[1][1] = LA[448];
    for (int it = 0; it < iters; ++it) {
        const uint4* __restrict__ cur = LA + (it & 3) * 512;
#pragma unroll
        for (int M = 0; M < 12; ++M) {
            const int g = M / 6, kk = M % 6, p = kk / 2, i = kk % 2, u = 2 * g + i;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, p == 1 ? Al[g][i] : Ah[g][i]),
                                                                   __builtin_bit_cast(h8, p == 2 ? Bl[t] : Bh[t]), acc[t][u], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < FILL; ++e) x[e] = fmaf(x[e], s, 0.25f);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LDSA && kk == 0) {
                const int ng = 1 - g;
                Ah[ng][0] = cur[ng * 256]; Al[ng][0] = cur[ng * 256 + 64]; Ah[ng][1] = cur[ng * 256 + 128]; Al[ng][1] = cur[ng * 256 + 192];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float r = 0.f;
    for (int t = 0; t < TILES; ++t) for (int u = 0; u < 4; ++u) r += acc[t][u][0] + acc[t][u][7];
    for (int e = 0; e < 8; ++e) r += x[e];
    if (r == 12345.f) out[threadIdx.x] = r;
}

template <int TILES, int FILL, int THREADS, int LDSA>
void run(int iters) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<TILES, FILL, THREADS, LDSA><<<256, THREADS, 65536>>>(10, out, 0.999f);
    hipEventRecord(e0);
    k<TILES, FILL, THREADS, LDSA><<<256, THREADS, 65536>>>(iters, out, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)iters * 12 * TILES * (THREADS / 256);   // MFMAs per SIMD
    printf("tiles/wave %d  waves/SIMD %d  fillers/MFMA %d  A from LDS %d: %.2f ns per MFMA per SIMD\n", TILES, THREADS / 256, FILL, LDSA, ms * 1e6 / mf);
    hipFree(out);
}
int main() {
    const int it = 4000;
    run<1, 0, 512, 0>(it); run<1, 0, 512, 1>(it); run<1, 2, 512, 1>(it); run<1, 3, 512, 1>(it); run<1, 4, 512, 1>(it); run<1, 5, 512, 1>(it);
    run<1, 3, 256, 1>(it); run<1, 4, 256, 1>(it);
    run<2, 0, 256, 0>(it); run<2, 0, 256, 1>(it); run<2, 2, 256, 1>(it); run<2, 3, 256, 1>(it); run<2, 4, 256, 1>(it); run<2, 5, 256, 1>(it);
    run<3, 0, 256, 1>(it); run<3, 3, 256, 1>(it); run<3, 4, 256, 1>(it); run<3, 5, 256, 1>(it);
    run<4, 0, 256, 1>(it); run<4, 3, 256, 1>(it); run<4, 4, 256, 1>(it); run<4, 5, 256, 1>(it);
    return 0;
}
