// SURVEY.md §8 f-1 ("next" row): the dense, HBM-bound tail of the training step that follows the renderer's backward —
// the total-variation regulariser on the VM planes (utils.py:488-504 through models/tensoRF.py:193-203) and the Adam update
// (text2nerf_main.py:453-454,588-590) — as two streaming HIP kernels instead of ~60 eager torch ops:
//   k_tv_grad_add   grad += d/dx [ weight * 2 * (sum_h (x[h]-x[h-1])^2 / count_h + sum_w (x[w]-x[w-1])^2 / count_w) / B ]
//                   (5-point stencil on the reference-layout [1,C,H,W] plane; reads the parameters only)
//   k_adam          torch.optim.Adam's single-tensor update (no weight decay / amsgrad), in place
#include "t2n_device.h"

namespace t2n {

__global__ __launch_bounds__(256) void k_tv_grad_add(const float* __restrict__ x, float* __restrict__ g, int C, int H, int W, float sh,
                                                     float sw) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long n = (long long)C * H * W;
    if (t >= n) return;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const float v = x[t];
    float acc = 0.f;
    if (h > 0) acc += sh * (2.f * (v - x[t - W]));
    if (h < H - 1) acc -= sh * (2.f * (x[t + W] - v));
    if (w > 0) acc += sw * (2.f * (v - x[t - 1]));
    if (w < W - 1) acc -= sw * (2.f * (x[t + 1] - v));
    g[t] += acc;
}

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                              long long n, float lr_over_bc1, float beta1, float beta2, float eps, float inv_bc2_sqrt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - beta1);          // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    p[i] = p[i] - lr_over_bc1 * (mi / denom);
}

// All parameter tensors of the field in ONE launch (19 tensors, most of them tiny: per-tensor launches are launch-bound).
constexpr int kAdamMaxTensors = 32;
struct AdamMulti {
    float* p[kAdamMaxTensors]; const float* g[kAdamMaxTensors]; float* m[kAdamMaxTensors]; float* v[kAdamMaxTensors];
    long long n[kAdamMaxTensors]; unsigned block0[kAdamMaxTensors + 1];
    float lr_over_bc1[kAdamMaxTensors], inv_bc2_sqrt[kAdamMaxTensors];
    int count; float beta1, beta2, eps;
};
__global__ __launch_bounds__(256) void k_adam_multi(const AdamMulti a) {
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < a.count; ++q) t += (a.block0[q] <= blockIdx.x) ? 1 : 0;
    // (scalar selects instead of dynamic indexing of the by-value struct would need 32-way code; the uniform index goes
    //  through s_load from the kernarg segment)
    const long long i = (long long)(blockIdx.x - a.block0[t]) * 256 + threadIdx.x;
    if (i >= a.n[t]) return;
    float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t]; float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * (1.f - a.beta1);
    const float vi = v[i] * a.beta2 + (1.f - a.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * a.inv_bc2_sqrt[t] + a.eps;
    p[i] = p[i] - a.lr_over_bc1[t] * (mi / denom);
}

}  // namespace t2n

using namespace t2n;

extern "C" int t2n_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, const int64_t* sizes, const float* lrs, float beta1, float beta2, float eps,
                                   const int64_t* steps, t2n_stream stream) {
    if (count < 0 || (count && (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !lrs || !steps))) {
        set_error("t2n_adam_step_multi: bad argument");
        return T2N_ERR_INVALID;
    }
    for (int base = 0; base < count; base += kAdamMaxTensors) {
        AdamMulti a;
        memset(&a, 0, sizeof(a));
        const int c = count - base < kAdamMaxTensors ? count - base : kAdamMaxTensors;
        unsigned blocks = 0;
        for (int i = 0; i < c; ++i) {
            const int k = base + i;
            if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || sizes[k] < 0 || steps[k] < 1) {
                set_error("t2n_adam_step_multi: bad tensor %d", k);
                return T2N_ERR_INVALID;
            }
            a.p[i] = params[k]; a.g[i] = grads[k]; a.m[i] = exp_avg[k]; a.v[i] = exp_avg_sq[k]; a.n[i] = sizes[k];
            const double bc1 = 1.0 - pow((double)beta1, (double)steps[k]), bc2 = 1.0 - pow((double)beta2, (double)steps[k]);
            a.lr_over_bc1[i] = (float)((double)lrs[k] / bc1);
            a.inv_bc2_sqrt[i] = (float)(1.0 / sqrt(bc2));
            a.block0[i] = blocks;
            blocks += (unsigned)((sizes[k] + 255) / 256);
        }
        a.block0[c] = blocks;
        a.count = c; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
        if (blocks) hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    }
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_tv_grad_add(const float* param, float* grad, int C, int H, int W, float weight, t2n_stream stream) {
    if (!param || !grad || C <= 0 || H <= 1 || W <= 1) { set_error("t2n_tv_grad_add: bad argument"); return T2N_ERR_INVALID; }
    // TVLoss: weight * 2 * (h_tv / count_h + w_tv / count_w) / batch, batch = 1
    const float sh = weight * 2.f / ((float)C * (float)(H - 1) * (float)W);
    const float sw = weight * 2.f / ((float)C * (float)H * (float)(W - 1));
    const long long n = (long long)C * H * W;
    hipLaunchKernelGGL(k_tv_grad_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, C, H, W, sh, sw);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, int64_t step, t2n_stream stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 0 || step < 1) { set_error("t2n_adam_step: bad argument"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, (float)((double)lr / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2)));
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
