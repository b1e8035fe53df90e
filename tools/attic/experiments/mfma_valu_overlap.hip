// Do MFMA and VALU instructions of two different waves on one SIMD overlap? 512-thread workgroups (two waves per SIMD); waves 0-3
// run an MFMA-only loop, waves 4-7 a VALU-only loop (v_fma chains, or v_fma_mix + v_cvt_pkrtz like the encoder). Times each role
// alone and both together, for the 16x16x32 and 32x32x16 f16 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, float s) {
    const int w = threadIdx.x >> 6;
    const bool do_mfma = (mode & 1) && w < 4, do_valu = (mode & 2) && w >= 4;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float r = 0.f;
    if (do_mfma) {
        if (KIND == 0) {
            f32x4 c[8] = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int t = 0; t < 8; ++t) c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[t], 0, 0, 0);
            }
            for (int t = 0; t < 8; ++t) r += c[t][0];
        } else {
            f32x16 c[4] = {};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int t = 0; t < 4; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
            }
            for (int t = 0; t < 4; ++t) r += c[t][0];
        }
    }
    if (do_valu) {
        float x[16];
        for (int e = 0; e < 16; ++e) x[e] = threadIdx.x * 0.01f + e;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = fmaf(x[e], s, 0.25f);   // 16 independent v_fma per iteration
        }
        for (int e = 0; e < 16; ++e) r += x[e];
    }
    if (r == 12345.f) out[threadIdx.x] = r;
}

template <int KIND>
float run(int mode, int iters) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(mode, 10, out, 0.999f);
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(mode, iters, out, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}
int main() {
    const int iters = 20000;
    for (int kind = 0; kind < 2; ++kind) {
        float m = kind ? run<1>(1, iters) : run<0>(1, iters);
        float v = kind ? run<1>(2, iters) : run<0>(2, iters);
        float b = kind ? run<1>(3, iters) : run<0>(3, iters);
        const int per = kind ? 4 : 8;
        printf("%s: mfma-only %.3f ms (%.1f cyc/mfma @2.4GHz)  valu-only %.3f ms (%.2f cyc/valu)  both %.3f ms  (sum %.3f, max %.3f)\n",
               kind ? "32x32x16" : "16x16x32", m, m * 2.4e6 / (iters * per), v, v * 2.4e6 / (iters * 16.0), b, m + v, m > v ? m : v);
    }
    return 0;
}
