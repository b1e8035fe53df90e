#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float red4(float x) {
    u2 p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    u2 q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__global__ void k(float* out) {
    const int l = threadIdx.x;
    const float x = (float)(1 << (l >> 4)) * 1000.f + (l & 15);     // rows: 1000, 2000, 4000, 8000 + n
    out[l] = red4(x);
}
int main() {
    float* d; hipMalloc(&d, 256); k<<<1, 64>>>(d);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) bad += h[l] != 15000.f + 4 * (l & 15);
    printf("permlane reduce over lanes n, n+16, n+32, n+48: %d mismatches (lane 5 -> %.0f, expected %.0f)\n", bad, h[5], 15000.f + 20);
    return bad != 0;
}
