// Accuracy of v_sin_f32 / v_cos_f32 (inputs in revolutions) with a two-constant 1/(2 pi) reduction, against double sin/cos,
// for the positional-encoding arguments f * 2^q (q = 0..5); and whether v_mfma_f32_32x32x16_f16 keeps f16 subnormal inputs.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_sincos(const float* f, int n, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float C1 = 0.15915494309189535f;                 // float(1/(2 pi))
    const float C2 = (float)(0.15915494309189533576888 - (double)C1);
    const float x = f[i];
    const float th = x * C1;
    const float tl = fmaf(x, C1, -th) + x * C2;
    for (int q = 0; q < 6; ++q) {
        const float sc = (float)(1 << q);
        const float fr = __builtin_amdgcn_fractf(th * sc);
        const float a = fmaf(tl, sc, fr);
        out[(size_t)i * 12 + 2 * q] = __builtin_amdgcn_sinf(a);
        out[(size_t)i * 12 + 2 * q + 1] = __builtin_amdgcn_cosf(a);
    }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_subnormal(float* out) {
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.f; b[e] = (_Float16)0.f; }
    const int lane = threadIdx.x;
    if (lane < 32) { a[0] = (_Float16)3e-6f; b[0] = (_Float16)1024.f; }     // 3e-6 is an f16 subnormal (min normal 6.1e-5)
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (lane == 0) { out[0] = c[0]; out[1] = (float)a[0] * 1024.f; }
}

int main() {
    const int n = 1 << 20;
    std::vector<float> f(n);
    for (int i = 0; i < n; ++i) {
        const double u = (i + 0.5) / n;
        f[i] = (float)((i & 1 ? -1 : 1) * (i < n / 2 ? 8.0 * u * 2 : std::pow(10.0, -6 + 10.5 * (u - 0.5) * 2)));   // dense in [-8,8], log-spaced up to 3e4
    }
    float *df, *dout, *dsub;
    hipMalloc(&df, n * 4); hipMalloc(&dout, (size_t)n * 48); hipMalloc(&dsub, 8);
    hipMemcpy(df, f.data(), n * 4, hipMemcpyHostToDevice);
    k_sincos<<<n / 256, 256>>>(df, n, dout);
    k_subnormal<<<1, 64>>>(dsub);
    std::vector<float> out((size_t)n * 12);
    float sub[2];
    hipMemcpy(out.data(), dout, (size_t)n * 48, hipMemcpyDeviceToHost);
    hipMemcpy(sub, dsub, 8, hipMemcpyDeviceToHost);
    double emax[4] = {0, 0, 0, 0};   // [|f|<=8, |f|<=100, |f|<=2000, rest]
    double emaxf[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double x = f[i];
        const int b = std::fabs(x) <= 8 ? 0 : std::fabs(x) <= 100 ? 1 : std::fabs(x) <= 2000 ? 2 : 3;
        for (int q = 0; q < 6; ++q) {
            const double a = (double)(float)(x * (1 << q));   // the reference's fp32 argument (exact product)
            const double es = std::fabs(out[(size_t)i * 12 + 2 * q] - std::sin(a)), ec = std::fabs(out[(size_t)i * 12 + 2 * q + 1] - std::cos(a));
            const double e = es > ec ? es : ec;
            if (e > emax[b]) { emax[b] = e; emaxf[b] = x * (1 << q); }
        }
    }
    printf("hw sin/cos max abs err: |f|<=8: %.3e (arg %.4g)  <=100: %.3e (arg %.4g)  <=2000: %.3e (arg %.4g)  larger: %.3e (arg %.4g)\n",
           emax[0], emaxf[0], emax[1], emaxf[1], emax[2], emaxf[2], emax[3], emaxf[3]);
    printf("mfma f16 subnormal input: got %.6e expected %.6e\n", sub[0], sub[1]);
    return 0;
}
