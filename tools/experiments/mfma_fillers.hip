// How many independent VALU instructions of the SAME wave issue in the shadow of one MFMA? Loop body: 4 (or 8) MFMAs, each followed by
// N v_fma (independent chains), order pinned with sched_barrier. One wave per SIMD (256 threads) and two (512 threads).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, int N, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, float* out, float s) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.5f); }
    float x[8];
    for (int e = 0; e < 8; ++e) x[e] = threadIdx.x * 0.01f + e;
    float r = 0.f;
    if (KIND == 0) {
        f32x4 c[8] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                c[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[t], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < N; ++e) x[e] = fmaf(x[e], s, 0.25f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int t = 0; t < 8; ++t) r += c[t][0];
    } else {
        f32x16 c[4] = {};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                c[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[t], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < N; ++e) x[e] = fmaf(x[e], s, 0.25f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int t = 0; t < 4; ++t) r += c[t][0];
    }
    for (int e = 0; e < 8; ++e) r += x[e];
    if (r == 12345.f) out[threadIdx.x] = r;
}

template <int KIND, int N, int THREADS>
void run(int iters) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, N, THREADS><<<256, THREADS>>>(10, out, 0.999f);
    hipEventRecord(e0);
    k<KIND, N, THREADS><<<256, THREADS>>>(iters, out, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per = KIND ? 4 : 8;
    printf("%s  %d wave(s)/SIMD  %d fillers/MFMA: %.1f ns per MFMA per wave  (%.1f ns per MFMA per SIMD)\n", KIND ? "32x32x16" : "16x16x32", THREADS / 256, N,
           ms * 1e6 / ((double)iters * per), ms * 1e6 / ((double)iters * per * (THREADS / 256)));
    hipFree(out);
}
int main() {
    const int it = 20000;
    run<1, 0, 256>(it); run<1, 2, 256>(it); run<1, 4, 256>(it); run<1, 6, 256>(it); run<1, 8, 256>(it);
    run<1, 0, 512>(it); run<1, 2, 512>(it); run<1, 4, 512>(it); run<1, 6, 512>(it); run<1, 8, 512>(it);
    run<0, 0, 256>(it); run<0, 1, 256>(it); run<0, 2, 256>(it); run<0, 3, 256>(it); run<0, 4, 256>(it);
    run<0, 0, 512>(it); run<0, 1, 512>(it); run<0, 2, 512>(it); run<0, 3, 512>(it); run<0, 4, 512>(it);
    return 0;
}
