// Microbenchmark: LDS atomic throughput on gfx950 (ds_add_f32 vs ds_add_u32 vs plain read-modify-write).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_atomic_bench.hip -o lds_atomic_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int stride) {
    __shared__ float sf[8192];
    unsigned* su = (unsigned*)sf;
    for (int i = threadIdx.x; i < 8192; i += 256) sf[i] = 0.f;
    __syncthreads();
    int a = (threadIdx.x * stride) & 8191;
    float v = 1.0f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) atomicAdd(&sf[a], v);
        if (MODE == 1) atomicAdd(&su[a], (unsigned)it);
        if (MODE == 2) { sf[a] += v; }
        if (MODE == 3) { float o = atomicAdd(&sf[a], v); v += o * 1e-30f; }
        if (MODE == 4) atomicAdd((unsigned long long*)&sf[(a * 2) & 8190], (unsigned long long)it * 77ull);
        if (MODE == 5) atomicAdd((double*)&sf[(a * 2) & 8190], (double)v);
        if (MODE == 6) { asm volatile("ds_add_f32 %0, %1" :: "v"(a * 4), "v"(v) : "memory"); }
        a = (a + 256 * stride + 17) & 8191;
    }
    __syncthreads();
    if (out) out[blockIdx.x * 256 + threadIdx.x] = sf[threadIdx.x];
}
template <int MODE>
void run(const char* name, int stride) {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 1024;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 16, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters;
    printf("%-28s stride %2d: %8.3f ms  %.1f cycles/wave-instr/CU (2.4GHz, 256 CU)\n", name, stride, ms, ms * 1e-3 * 2.4e9 * 256 / winstr);
    hipFree(out);
}
int main() {
    for (int stride : {1, 2, 16, 0}) {
        run<0>("ds_add_f32 (no return)", stride);
        run<1>("ds_add_u32 (no return)", stride);
        run<2>("read + add + write", stride);
        run<3>("ds_add_rtn_f32", stride);
        run<4>("ds_add_u64 (no return)", stride);
        run<5>("ds_add_f64 (no return)", stride);
        run<6>("asm ds_add_f32", stride);
    }
    return 0;
}
