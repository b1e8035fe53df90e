// ---- the driver's loss (text2nerf_main.py:559-575) as one pass over the render outputs --------------------------------------------
// loss = mean((rgb - rgb_t)^2) + w_depth mean((depth - depth_t)^2) + w_trans mean_r(m_r^2), m_r = mean_n(w[r,n] [z[r,n] - depth_t[r] + delta < 0])
// (TransMittanceLoss_mask, utils.py:67-80, target 0), NaN depths count as 0 with no gradient (:559-560). One wave per ray: the
// upstream gradients d_rgb, d_depth, d_weights of t2n_render_backward leave in the same pass; per-workgroup partial sums, then one
// workgroup adds them in a fixed order (deterministic).
#include "t2n_device.h"

namespace t2n {
struct LossArgs {
    const float* rgb; const float* depth; const float* w; const float* z; const float* rgb_t; const float* depth_t;
    long long R; int N; float w_depth, w_trans, delta;
    float* d_rgb; float* d_depth; float* d_w; float* part; float* losses; unsigned nblocks;
};
__device__ __forceinline__ void loss_reduce_body(const LossArgs& a, float (*red)[3]) {
    const LossReduceArgs r{a.part, a.nblocks, a.losses, a.R, a.w_depth, a.w_trans};
    loss_reduce_rows(r, red);
}
__global__ __launch_bounds__(256) void k_train_loss(const LossArgs a) {
    __shared__ float red[4][3];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 4 + wid;
    float e_rgb = 0.f, e_dep = 0.f, e_tr = 0.f;
    if (r < a.R) {
        const float dt = a.depth_t[r];
        const float* wr = a.w + r * a.N;
        const float* zr = a.z + r * a.N;
        float m = 0.f;
        for (int n = lane; n < a.N; n += 64) m += ((zr[n] - dt) + a.delta < 0.f) ? wr[n] : 0.f;
        m = wave_sum(m) / (float)a.N;
        const float gw = (2.f * a.w_trans * m / (float)a.R) / (float)a.N;
        float* dwr = a.d_w + r * a.N;
        for (int n = lane; n < a.N; n += 64) dwr[n] = ((zr[n] - dt) + a.delta < 0.f) ? gw : 0.f;
        if (lane < 3) {
            const float d = a.rgb[r * 3 + lane] - a.rgb_t[r * 3 + lane];
            a.d_rgb[r * 3 + lane] = 2.f * d / (3.f * (float)a.R);
            e_rgb = d * d;
        }
        e_rgb = wave_sum(e_rgb);
        float dep = a.depth[r];
        const bool bad = dep != dep;
        if (bad) dep = 0.f;
        const float dd = dep - dt;
        if (lane == 0) a.d_depth[r] = bad ? 0.f : 2.f * a.w_depth * dd / (float)a.R;
        e_dep = dd * dd;
        e_tr = m * m;
    }
    if (lane == 0) { red[wid][0] = e_rgb; red[wid][1] = e_dep; red[wid][2] = e_tr; }
    __syncthreads();
    if (threadIdx.x < 3) a.part[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_train_loss_reduce(const LossArgs a) {
    __shared__ float red[256][3];
    loss_reduce_body(a, red);
}
int launch_train_loss(const float* rgb, const float* depth, const float* weights, const float* z_vals, const float* rgb_t, const float* depth_t,
                      int64_t n_rays, int n_samples, float w_depth, float w_trans, float delta, float* d_rgb, float* d_depth, float* d_weights,
                      float* losses, float* part, bool reduce, hipStream_t s) {
    LossArgs a;
    a.rgb = rgb; a.depth = depth; a.w = weights; a.z = z_vals; a.rgb_t = rgb_t; a.depth_t = depth_t; a.R = n_rays; a.N = n_samples;
    a.w_depth = w_depth; a.w_trans = w_trans; a.delta = delta; a.d_rgb = d_rgb; a.d_depth = d_depth; a.d_w = d_weights;
    a.part = part; a.losses = losses; a.nblocks = (unsigned)((n_rays + 3) / 4);
    hipLaunchKernelGGL(k_train_loss, dim3(a.nblocks), dim3(256), 0, s, a);
    // reduce = false (fused training step): the partial sums are added up by the step's one reduce launch (k_wgrad_reduce) — the sum only
    // feeds the reported losses. (Folded into this kernel's last workgroup it needs an agent-scope release per workgroup, an L2 write-back
    // each: 357 us instead of 22 measured.)
    if (reduce) hipLaunchKernelGGL(k_train_loss_reduce, dim3(1), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
}  // namespace t2n

using namespace t2n;

extern "C" size_t t2n_train_loss_workspace_bytes(int64_t n_rays) { return n_rays > 0 ? (size_t)((n_rays + 3) / 4) * 3 * sizeof(float) : 0; }

extern "C" int t2n_train_loss(const float* rgb, const float* depth, const float* weights, const float* z_vals, const float* rgb_t,
                              const float* depth_t, int64_t n_rays, int n_samples, float w_depth, float w_trans, float delta, float* d_rgb,
                              float* d_depth, float* d_weights, float* losses, void* workspace, size_t workspace_bytes, t2n_stream stream) {
    if (!rgb || !depth || !weights || !z_vals || !rgb_t || !depth_t || !d_rgb || !d_depth || !d_weights || !losses || !workspace || n_rays <= 0 || n_samples <= 0) {
        set_error("t2n_train_loss: bad argument");
        return T2N_ERR_INVALID;
    }
    if (workspace_bytes < t2n_train_loss_workspace_bytes(n_rays)) { set_error("t2n_train_loss: workspace too small"); return T2N_ERR_WORKSPACE; }
    return launch_train_loss(rgb, depth, weights, z_vals, rgb_t, depth_t, n_rays, n_samples, w_depth, w_trans, delta, d_rgb, d_depth, d_weights,
                             losses, (float*)workspace, true, (hipStream_t)stream);
}

