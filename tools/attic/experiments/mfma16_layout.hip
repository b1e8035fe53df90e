// Operand / result layout of v_mfma_f32_16x16x32_f16 on gfx950: A[i][k]: lane = i + 16 (k / 8), element k % 8; B[k][j]: lane = j + 16 (k / 8),
// element k % 8; D[i][j]: lane = j + 16 (i / 4), register i % 4. Prints the number of mismatches against a host product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {   // A [16][32], B [32][16], D [16][16]
    const int l = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = (_Float16)B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l >> 4) + v) * 16 + (l & 15)] = c[v];
}
int main() {
    float A[16 * 32], B[32 * 16], D[256], R[256];
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < 32; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (float)((k * 5 + j * 2) % 13 - 6);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += A[i * 32 + k] * B[k * 16 + j]; R[i * 16 + j] = s; }
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(A)); hipMalloc(&dB, sizeof(B)); hipMalloc(&dD, sizeof(D));
    hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += D[i] != R[i];
    printf("mfma_f32_16x16x32_f16 layout mismatches: %d of 256\n", bad);
    return bad != 0;
}
