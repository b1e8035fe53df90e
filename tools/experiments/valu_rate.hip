// VALU issue rate on gfx950: how many wave64 fp32 FMA instructions does a SIMD retire per clock? Independent FMA chains (8 accumulators),
// 1..8 waves per SIMD, every CU loaded. Prints wave-instructions per second and per SIMD clock (s_memtime ticks of one wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int PK>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, unsigned long long* ticks) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (PK) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 x0 = {a0, a1}, x1 = {a2, a3}, x2 = {a4, a5}, x3 = {a6, a7};
                const f2 mm = {m, m}, cc = {c, c};
                x0 = __builtin_elementwise_fma(x0, mm, cc); x1 = __builtin_elementwise_fma(x1, mm, cc);
                x2 = __builtin_elementwise_fma(x2, mm, cc); x3 = __builtin_elementwise_fma(x3, mm, cc);
                a0 = x0[0]; a1 = x0[1]; a2 = x1[0]; a3 = x1[1]; a4 = x2[0]; a5 = x2[1]; a6 = x3[0]; a7 = x3[1];
            } else {
                a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    if (blockIdx.x == 0 && threadIdx.x == 0) *ticks = t1 - t0;
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)); hipMalloc(&ticks, 8);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wg_per_cu = 1; wg_per_cu <= 8; wg_per_cu *= 2) {
            const int grid = 256 * wg_per_cu;   // 256-thread workgroups: one wave per SIMD each
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k_valu<1>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                else hipLaunchKernelGGL(k_valu<0>, dim3(grid), dim3(256), 0, 0, out, iters, ticks);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
            const double insts_per_wave = (double)iters * (pk ? 32 : 64);
            const double total = insts_per_wave * grid * 4;
            printf("%s waves/SIMD %d: %.3f ms, %.1f G wave-inst/s, per SIMD %.3f G/s; one wave: %.2f memtime ticks per instruction\n",
                   pk ? "v_pk_fma_f32" : "v_fma_f32  ", wg_per_cu, ms, total / ms / 1e6, total / ms / 1e6 / 1024.0, (double)tk / insts_per_wave);
        }
    return 0;
}
