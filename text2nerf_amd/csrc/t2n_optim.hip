// SURVEY.md §8 f-1 ("next" row): the dense, HBM-bound tail of the training step that follows the renderer's backward —
// the total-variation regulariser on the VM planes (utils.py:488-504 through models/tensoRF.py:193-203) and the Adam update
// (text2nerf_main.py:453-454,588-590) — as two streaming HIP kernels instead of ~60 eager torch ops:
//   k_tv_grad<SET>  grad (+)= d/dx [ weight * 2 * (sum_h (x[h]-x[h-1])^2 / count_h + sum_w (x[w]-x[w-1])^2 / count_w) / B ]
//                   (5-point stencil on the reference-layout [1,C,H,W] plane; reads the parameters only)
//   k_adam          torch.optim.Adam's single-tensor update (no weight decay / amsgrad), in place
#include "t2n_device.h"

namespace t2n {

template <bool SET>
__global__ __launch_bounds__(256) void k_tv_grad(const float* __restrict__ x, float* __restrict__ g, int C, int H, int W, float sh,
                                                 float sw, const float* __restrict__ scale) {
    const float up = (SET && scale) ? *scale : 1.f;   // SET: g = upstream scalar (device) x gradient; ADD: g += gradient
    // one wave per image row (c, y), lanes stride the row: no per-element divisions; same per-element arithmetic and order as before
    const int lane = threadIdx.x & 63;
    const long long rows = (long long)C * H;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        const int h = (int)(r % H);
        const float* __restrict__ row = x + r * W;
        float* __restrict__ grow = g + r * W;
        for (int w = lane; w < W; w += 64) {
            const float v = row[w];
            float acc = 0.f;
            if (h > 0) acc += sh * (2.f * (v - row[w - W]));
            if (h < H - 1) acc -= sh * (2.f * (row[w + W] - v));
            if (w > 0) acc += sw * (2.f * (v - row[w - 1]));
            if (w < W - 1) acc -= sw * (2.f * (row[w + 1] - v));
            if (SET) grow[w] = acc * up; else grow[w] += acc;
        }
    }
}

// TVLoss's two sums of one reference-layout plane [1,C,H,W] (utils.py:488-504): out[0] += sum (x[c,y+1,x] - x[c,y,x])^2,
// out[1] += sum (x[c,y,x+1] - x[c,y,x])^2. Grid-stride blocks, fp32 squares summed in double, two atomics per block.
__global__ __launch_bounds__(256) void k_tv_value(const float* __restrict__ x, int C, int H, int W, double* __restrict__ out) {
    __shared__ double part[8];
    // one wave per image row (c, y): lanes stride the row (coalesced), the row below is the same offset + W; no per-element divisions
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long rows = (long long)C * H;
    float sh = 0.f, sw = 0.f;     // per-lane partial sums of a few hundred squares; combined in double below
    double dh = 0.0, dw = 0.0;
    for (long long r = (long long)blockIdx.x * 4 + wid; r < rows; r += (long long)gridDim.x * 4) {
        const int y = (int)(r % H);
        const float* __restrict__ row = x + r * W;
        const bool below = y < H - 1;
        for (int i = lane; i < W; i += 64) {
            const float v = row[i];
            if (below) { const float d = row[i + W] - v; sh = fmaf(d, d, sh); }
            if (i < W - 1) { const float d = row[i + 1] - v; sw = fmaf(d, d, sw); }
        }
        dh += (double)sh; dw += (double)sw; sh = 0.f; sw = 0.f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dh += __shfl_xor(dh, o); dw += __shfl_xor(dw, o); }
    if (lane == 0) { part[wid] = dh; part[4 + wid] = dw; }
    __syncthreads();
    if (threadIdx.x == 0) {   // T2N_TV_SLOTS pairs of sums: same-address atomics serialise (~88 per microsecond), the caller adds the slots
        double* o = out + 2 * (blockIdx.x & (T2N_TV_SLOTS - 1));
        atomicAdd(&o[0], (part[0] + part[1]) + (part[2] + part[3]));
        atomicAdd(&o[1], (part[4] + part[5]) + (part[6] + part[7]));
    }
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
    // fused training step: scalars and the step's verdict from device memory (tensor t <-> st->lr_over_bc1[st_first + t]); NULL: by value
    const TrainScalars* st; int st_first; int zero_grads; unsigned* zero_words4;   // zero_words4: four words cleared by the launch (the next launch's absmax accumulators)   // zero_grads: the consumed gradient is set to zero (the next step accumulates into it)
};
__global__ __launch_bounds__(256) void k_adam_multi(const AdamMulti a) {
    if (a.zero_words4 && blockIdx.x == 0 && threadIdx.x < 4) a.zero_words4[threadIdx.x] = 0u;
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < a.count; ++q) t += (a.block0[q] <= blockIdx.x) ? 1 : 0;
    // (scalar selects instead of dynamic indexing of the by-value struct would need 32-way code; the uniform index goes
    //  through s_load from the kernarg segment)
    const long long i = (long long)(blockIdx.x - a.block0[t]) * 256 + threadIdx.x;
    if (i >= a.n[t]) return;
    float lr = a.lr_over_bc1[t], ib = a.inv_bc2_sqrt[t];
    if (a.st) {
        if (a.st->skip) { if (a.zero_grads) const_cast<float*>(a.g[t])[i] = 0.f; return; }   // the step's appearance rows did not fit: no tensor moves, no moment decays
        lr = a.st->lr_over_bc1[a.st_first + t]; ib = a.st->inv_bc2_sqrt;
    }
    float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t]; float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
    const float gi = g[i];
    if (a.zero_grads) const_cast<float*>(g)[i] = 0.f;
    const float mi = m[i] + (gi - m[i]) * (1.f - a.beta1);
    const float vi = v[i] * a.beta2 + (1.f - a.beta2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) * ib + a.eps;
    p[i] = p[i] - lr * (mi / denom);
}

// ---- the same step on the field's channel-last master copies -------------------------------------------------------------
// The renderer's backward leaves the factor gradients in channel-last buffers [pos][C] and the forward reads channel-last
// factor copies, so the reference-form step pays four layout passes per iteration around the two streaming kernels above
// (gradients -> [1,C,H,W], zero-filled gradient tensors, parameters -> channel-last). t2n_field_tv_adam_step does the whole
// thing where the data already is: TV stencil on the channel-last parameters (neighbours at +-C and +-W*C), then one pass
// that reads g / m / v / p channel-last, writes p / m / v channel-last and the new values into the caller's reference-layout
// tensor through an LDS tile transpose. Same arithmetic per element as k_tv_grad + k_adam (bit-identical results).
template <bool SET>
__device__ __forceinline__ void tv_grad_cl_body(const float* __restrict__ x, float* __restrict__ g, long long t, int C4, int H, int W, float sh,
                                                float sw) {
    const long long n = (long long)H * W * C4;   // t: one float4 of 4 channels
    if (t >= n) return;
    const long long pos = t / C4;
    const int w = (int)(pos % W), h = (int)(pos / W);
    const float4* __restrict__ X = reinterpret_cast<const float4*>(x);
    const float4 v = X[t];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // (the reference-layout kernel's order: up, down, left, right)
    if (h > 0) { const float4 u = X[t - (long long)W * C4]; acc.x += sh * (2.f * (v.x - u.x)); acc.y += sh * (2.f * (v.y - u.y)); acc.z += sh * (2.f * (v.z - u.z)); acc.w += sh * (2.f * (v.w - u.w)); }
    if (h < H - 1) { const float4 d = X[t + (long long)W * C4]; acc.x -= sh * (2.f * (d.x - v.x)); acc.y -= sh * (2.f * (d.y - v.y)); acc.z -= sh * (2.f * (d.z - v.z)); acc.w -= sh * (2.f * (d.w - v.w)); }
    if (w > 0) { const float4 l = X[t - C4]; acc.x += sw * (2.f * (v.x - l.x)); acc.y += sw * (2.f * (v.y - l.y)); acc.z += sw * (2.f * (v.z - l.z)); acc.w += sw * (2.f * (v.w - l.w)); }
    if (w < W - 1) { const float4 r = X[t + C4]; acc.x -= sw * (2.f * (r.x - v.x)); acc.y -= sw * (2.f * (r.y - v.y)); acc.z -= sw * (2.f * (r.z - v.z)); acc.w -= sw * (2.f * (r.w - v.w)); }
    float4* G = reinterpret_cast<float4*>(g);
    if (SET) { G[t] = acc; return; }   // the gradient buffer STARTS as the TV gradient (t2n_field_tv_seed)
    float4 gv = G[t];
    gv.x += acc.x; gv.y += acc.y; gv.z += acc.z; gv.w += acc.w;
    G[t] = gv;
}

// the same stencil with 32-bit index arithmetic (a factor tensor has < 2^31 float4 elements; checked by the launcher): the 64-bit
// divisions of the form above are ~150 of the seed kernel's instructions per thread, on a kernel that moves 32 bytes per thread
__device__ __forceinline__ void tv_seed_cl_body32(const float* __restrict__ x, float* __restrict__ g, unsigned t, unsigned C4, unsigned H, unsigned W, float sh,
                                                  float sw) {
    const unsigned n = H * W * C4;
    if (t >= n) return;
    const unsigned pos = C4 == 4u ? (t >> 2) : (C4 == 12u ? t / 12u : t / C4);
    const unsigned h = pos / W, w = pos - h * W;
    const float4* __restrict__ X = reinterpret_cast<const float4*>(x);
    const float4 v = X[t];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned row = W * C4;
    if (h > 0) { const float4 u = X[t - row]; acc.x += sh * (2.f * (v.x - u.x)); acc.y += sh * (2.f * (v.y - u.y)); acc.z += sh * (2.f * (v.z - u.z)); acc.w += sh * (2.f * (v.w - u.w)); }
    if (h < H - 1) { const float4 d = X[t + row]; acc.x -= sh * (2.f * (d.x - v.x)); acc.y -= sh * (2.f * (d.y - v.y)); acc.z -= sh * (2.f * (d.z - v.z)); acc.w -= sh * (2.f * (d.w - v.w)); }
    if (w > 0) { const float4 l = X[t - C4]; acc.x += sw * (2.f * (v.x - l.x)); acc.y += sw * (2.f * (v.y - l.y)); acc.z += sw * (2.f * (v.z - l.z)); acc.w += sw * (2.f * (v.w - l.w)); }
    if (w < W - 1) { const float4 r = X[t + C4]; acc.x -= sw * (2.f * (r.x - v.x)); acc.y -= sw * (2.f * (r.y - v.y)); acc.z -= sw * (2.f * (r.z - v.z)); acc.w -= sw * (2.f * (r.w - v.w)); }
    reinterpret_cast<float4*>(g)[t] = acc;
}

__device__ __forceinline__ float adam_one(float p, float gi, float& m, float& v, float lr_over_bc1, float beta1, float beta2, float eps,
                                          float inv_bc2_sqrt) {
    const float mi = m + (gi - m) * (1.f - beta1);
    const float vi = v * beta2 + (1.f - beta2) * gi * gi;
    m = mi; v = vi;
    const float denom = sqrtf(vi) * inv_bc2_sqrt + eps;
    return p - lr_over_bc1 * (mi / denom);
}
// one workgroup = 64 positions x C channels (a contiguous run of the channel-last arrays); tile: [C][65] floats of LDS
template <int C>
__device__ __forceinline__ void adam_cl_tile(float* __restrict__ tile, unsigned blk, float* __restrict__ p_cl, const float* __restrict__ g_cl,
                                             float* __restrict__ m, float* __restrict__ v, float* __restrict__ p_ref, long long npos_total,
                                             float lr_over_bc1, float beta1, float beta2, float eps, float inv_bc2_sqrt, bool relayout_only = false) {
    const long long pos0 = (long long)blk * 64;
    const int npos = (int)(npos_total - pos0 < 64 ? npos_total - pos0 : 64);
    const long long base4 = pos0 * (C / 4);
    const int n4 = npos * (C / 4);
    float4* __restrict__ P = reinterpret_cast<float4*>(p_cl) + base4;
    const float4* __restrict__ G = reinterpret_cast<const float4*>(g_cl) + base4;
    float4* __restrict__ M = reinterpret_cast<float4*>(m) + base4;
    float4* __restrict__ V = reinterpret_cast<float4*>(v) + base4;
    for (int e = threadIdx.x; e < n4; e += 256) {
        float4 pv = P[e];
        if (relayout_only) {
            const int j = e / (C / 4), c = (e - j * (C / 4)) * 4;
            tile[c * 65 + j] = pv.x; tile[(c + 1) * 65 + j] = pv.y; tile[(c + 2) * 65 + j] = pv.z; tile[(c + 3) * 65 + j] = pv.w;
            continue;
        }
        float4 mv = M[e], vv = V[e];
        const float4 gv = G[e];
        pv.x = adam_one(pv.x, gv.x, mv.x, vv.x, lr_over_bc1, beta1, beta2, eps, inv_bc2_sqrt);
        pv.y = adam_one(pv.y, gv.y, mv.y, vv.y, lr_over_bc1, beta1, beta2, eps, inv_bc2_sqrt);
        pv.z = adam_one(pv.z, gv.z, mv.z, vv.z, lr_over_bc1, beta1, beta2, eps, inv_bc2_sqrt);
        pv.w = adam_one(pv.w, gv.w, mv.w, vv.w, lr_over_bc1, beta1, beta2, eps, inv_bc2_sqrt);
        P[e] = pv; M[e] = mv; V[e] = vv;
        const int j = e / (C / 4), c = (e - j * (C / 4)) * 4;
        tile[c * 65 + j] = pv.x; tile[(c + 1) * 65 + j] = pv.y; tile[(c + 2) * 65 + j] = pv.z; tile[(c + 3) * 65 + j] = pv.w;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * 64; e += 256) {
        const int c = e >> 6, j = e & 63;
        if (j < npos) p_ref[(long long)c * npos_total + pos0 + j] = tile[c * 65 + j];
    }
}

// the 12 factor tensors of a field in one launch each for the TV pass and the Adam pass (24 per-tensor launches otherwise:
// the training step is within a few percent of being bound by the host's launch rate)
struct FactorStep {
    float* p[12]; float* g[12]; float* m[12]; float* v[12]; float* ref[12];
    long long npos[12];
    int C[12], H[12], W[12];
    float sh[12], sw[12], lr_over_bc1[12], inv_bc2_sqrt[12];
    unsigned ablock0[13], tblock0[13];   // first workgroup of tensor t in the Adam / TV launch (TV: zero-width for lines and weight 0)
    float beta1, beta2, eps;
    // fused training step (NULL: by value): Adam scalars + verdict; the two TV weights (x 1e-2) of this step in device memory
    const TrainScalars* st; const float* tvw_dev;
    // sharded optimiser (data-parallel, world > 1; include/t2n.h "sharded optimiser"): the first body[t] 64-position blocks of plane t
    // are split evenly over the ranks — this rank owns [own_lo[t], own_hi[t]) — everything behind them (a plane's last blocks, the
    // lines) is replicated on every rank. world <= 1: body = 0, the whole tensor is "replicated" (the single-rank step).
    unsigned own_lo[12], own_hi[12], body[12];
    int world;
    int relayout_only;   // Adam launch: no arithmetic — the blocks of the body this rank does NOT own, channel-last -> reference layout
};
__global__ __launch_bounds__(256) void k_tv_grad_cl_multi(const FactorStep a) {
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < 12; ++q) t += (a.tblock0[q] <= blockIdx.x) ? 1 : 0;
    const long long i = (long long)(blockIdx.x - a.tblock0[t]) * 256 + threadIdx.x;
    tv_grad_cl_body<false>(a.p[t], a.g[t], i, a.C[t] / 4, a.H[t], a.W[t], a.sh[t], a.sw[t]);
}
// every factor gradient tensor initialised: the TV gradient where a tensor has a TV weight, zero elsewhere (tblock0 spans all 12)
__global__ __launch_bounds__(256) void k_tv_seed_cl_multi(const FactorStep a) {
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < 12; ++q) t += (a.tblock0[q] <= blockIdx.x) ? 1 : 0;
    const unsigned i32 = (blockIdx.x - a.tblock0[t]) * 256u + threadIdx.x;
    const long long i = (long long)i32;
    float sh = a.sh[t], sw = a.sw[t];
    if (a.tvw_dev) {   // the host's expressions (t2n_field_tv_seed), on this step's weights; planes only (t = 0..2 density, 6..8 appearance)
        const float tvw = t < 3 ? a.tvw_dev[0] : ((t >= 6 && t < 9) ? a.tvw_dev[1] : 0.f);
        const int C = a.C[t], H = a.H[t], W = a.W[t];
        sh = tvw != 0.f ? tvw * 2.f / ((float)C * (float)(H - 1) * (float)W) : 0.f;
        sw = tvw != 0.f ? tvw * 2.f / ((float)C * (float)H * (float)(W - 1)) : 0.f;
    }
    if (a.world > 1 && (sh != 0.f || sw != 0.f)) {
        // sharded: the owner of a body block seeds `world` times the TV gradient (the ranks' buffers are then AVERAGED), the others zero;
        // replicated blocks carry the TV gradient on every rank
        const unsigned C4 = (unsigned)a.C[t] / 4u;
        const unsigned blk = (C4 == 4u ? (i32 >> 2) : i32 / C4) >> 6;
        if (blk < a.body[t]) {
            if (blk >= a.own_lo[t] && blk < a.own_hi[t]) { sh *= (float)a.world; sw *= (float)a.world; }
            else { sh = 0.f; sw = 0.f; }
        }
    }
    if (sh != 0.f || sw != 0.f) tv_seed_cl_body32(a.p[t], a.g[t], i32, (unsigned)a.C[t] / 4u, (unsigned)a.H[t], (unsigned)a.W[t], sh, sw);
    else if (i < a.npos[t] * (a.C[t] / 4)) reinterpret_cast<float4*>(a.g[t])[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(256) void k_adam_cl_multi(const FactorStep a) {
    __shared__ float tile[48 * 65];
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < 12; ++q) t += (a.ablock0[q] <= blockIdx.x) ? 1 : 0;
    unsigned blk = blockIdx.x - a.ablock0[t];
    // the launch's blocks of tensor t: this rank's slice of the body, then the replicated blocks (relayout_only: the rest of the body)
    if (a.relayout_only) blk = blk < a.own_lo[t] ? blk : blk + (a.own_hi[t] - a.own_lo[t]);
    else { const unsigned own = a.own_hi[t] - a.own_lo[t]; blk = blk < own ? a.own_lo[t] + blk : a.body[t] + (blk - own); }
    float lr = a.lr_over_bc1[t], ib = a.inv_bc2_sqrt[t];
    if (a.st) {
        if (a.st->skip) return;      // (uniform over the launch)
        lr = a.st->lr_over_bc1[t]; ib = a.st->inv_bc2_sqrt;
    }
    const bool ro = a.relayout_only != 0;
    if (a.C[t] == 16) adam_cl_tile<16>(tile, blk, a.p[t], a.g[t], a.m[t], a.v[t], a.ref[t], a.npos[t], lr, a.beta1, a.beta2, a.eps, ib, ro);
    else adam_cl_tile<48>(tile, blk, a.p[t], a.g[t], a.m[t], a.v[t], a.ref[t], a.npos[t], lr, a.beta1, a.beta2, a.eps, ib, ro);
}

}  // namespace t2n

using namespace t2n;

extern "C" int t2n_field_tv_adam_step(t2n_field* f, const t2n_field_params* params, float* const* exp_avg, float* const* exp_avg_sq,
                                      const float* lrs, const int64_t* steps, float beta1, float beta2, float eps,
                                      float tv_weight_density, float tv_weight_app, t2n_stream stream) {
    if (!f || !params || !exp_avg || !exp_avg_sq || !lrs || !steps) { set_error("t2n_field_tv_adam_step: NULL argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded || !f->gbuf_den_plane[0]) {
        set_error("t2n_field_tv_adam_step: needs an uploaded field and the gradients of a t2n_render_backward call");
        return T2N_ERR_INVALID;
    }
    if (f->factor_bf16) { set_error("t2n_field_tv_adam_step: bf16 factor storage keeps no fp32 master copy on the device"); return T2N_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const int* gr = f->desc.grid;
    FactorStep A;
    memset(&A, 0, sizeof(A));
    A.beta1 = beta1; A.beta2 = beta2; A.eps = eps;
    unsigned ab = 0, tb = 0;
    // order of the 12 tensors: density planes, density lines, appearance planes, appearance lines
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const int idx = q * 3 + k;
            const int H = gr[mat1(k)], W = gr[mat0(k)];
            const long long HW = (long long)H * W, L = gr[vecm(k)];
            float* pcl[4] = {f->buf_den_plane[k], f->buf_den_line[k], f->buf_app_plane[k], f->buf_app_line[k]};
            float* gcl[4] = {f->gbuf_den_plane[k], f->gbuf_den_line[k], f->gbuf_app_plane[k], f->gbuf_app_line[k]};
            float* pref[4] = {(float*)params->density_plane[k], (float*)params->density_line[k], (float*)params->app_plane[k], (float*)params->app_line[k]};
            const int C = q < 2 ? 16 : 48;
            const bool plane = (q & 1) == 0;
            const float tvw = q == 0 ? tv_weight_density : (q == 2 ? tv_weight_app : 0.f);
            if (!pref[q] || !exp_avg[idx] || !exp_avg_sq[idx] || steps[idx] < 1) { set_error("t2n_field_tv_adam_step: bad tensor %d", idx); return T2N_ERR_INVALID; }
            A.p[idx] = pcl[q]; A.g[idx] = gcl[q]; A.m[idx] = exp_avg[idx]; A.v[idx] = exp_avg_sq[idx]; A.ref[idx] = pref[q];
            A.npos[idx] = plane ? HW : L; A.C[idx] = C; A.H[idx] = plane ? H : (int)L; A.W[idx] = plane ? W : 1;
            A.tblock0[idx] = tb;
            if (tvw != 0.f) {
                if (H < 2 || W < 2) { set_error("t2n_field_tv_adam_step: TV needs planes of at least 2x2"); return T2N_ERR_INVALID; }
                A.sh[idx] = tvw * 2.f / ((float)C * (float)(H - 1) * (float)W);
                A.sw[idx] = tvw * 2.f / ((float)C * (float)H * (float)(W - 1));
                tb += (unsigned)((HW * (C / 4) + 255) / 256);
            }
            const double bc1 = 1.0 - pow((double)beta1, (double)steps[idx]), bc2 = 1.0 - pow((double)beta2, (double)steps[idx]);
            A.lr_over_bc1[idx] = (float)((double)lrs[idx] / bc1);
            A.inv_bc2_sqrt[idx] = (float)(1.0 / sqrt(bc2));
            A.ablock0[idx] = ab;
            ab += (unsigned)((A.npos[idx] + 63) / 64);
        }
    A.tblock0[12] = tb; A.ablock0[12] = ab;
    if (tb) hipLaunchKernelGGL(k_tv_grad_cl_multi, dim3(tb), dim3(256), 0, s, A);
    hipLaunchKernelGGL(k_adam_cl_multi, dim3(ab), dim3(256), 0, s, A);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// The factor gradient buffer of the next backward call initialised to the TV gradient of the CURRENT device copies (zero for the
// lines and for a zero weight) instead of being zero-filled and TV-incremented after the backward: one pass that only writes the
// buffer, and one that may run on another stream beside the forward (it reads the parameters the forward reads and nothing else).
// Same per-element arithmetic as the TV pass of t2n_field_tv_adam_step; the scatter kernels then accumulate on top.
extern "C" int t2n_field_tv_seed(t2n_field* f, float tv_weight_density, float tv_weight_app, t2n_stream stream) {
    if (!f) { set_error("t2n_field_tv_seed: NULL field"); return T2N_ERR_INVALID; }
    if (!f->uploaded || !f->gbuf_den_plane[0]) { set_error("t2n_field_tv_seed: needs an uploaded field with a gradient buffer"); return T2N_ERR_INVALID; }
    if (f->factor_bf16) { set_error("t2n_field_tv_seed: bf16 factor storage keeps no fp32 master copy on the device"); return T2N_ERR_UNSUPPORTED; }
    const int* gr = f->desc.grid;
    FactorStep A;
    memset(&A, 0, sizeof(A));
    unsigned tb = 0;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const int idx = q * 3 + k;
            const int H = gr[mat1(k)], W = gr[mat0(k)];
            const long long HW = (long long)H * W, L = gr[vecm(k)];
            float* pcl[4] = {f->buf_den_plane[k], f->buf_den_line[k], f->buf_app_plane[k], f->buf_app_line[k]};
            float* gcl[4] = {f->gbuf_den_plane[k], f->gbuf_den_line[k], f->gbuf_app_plane[k], f->gbuf_app_line[k]};
            const int C = q < 2 ? 16 : 48;
            const bool plane = (q & 1) == 0;
            const float tvw = q == 0 ? tv_weight_density : (q == 2 ? tv_weight_app : 0.f);
            A.p[idx] = pcl[q]; A.g[idx] = gcl[q];
            A.npos[idx] = plane ? HW : L; A.C[idx] = C; A.H[idx] = plane ? H : (int)L; A.W[idx] = plane ? W : 1;
            if (tvw != 0.f) {
                if (H < 2 || W < 2) { set_error("t2n_field_tv_seed: TV needs planes of at least 2x2"); return T2N_ERR_INVALID; }
                A.sh[idx] = tvw * 2.f / ((float)C * (float)(H - 1) * (float)W);
                A.sw[idx] = tvw * 2.f / ((float)C * (float)H * (float)(W - 1));
            }
            A.tblock0[idx] = tb;
            tb += (unsigned)((A.npos[idx] * (C / 4) + 255) / 256);
        }
    A.tblock0[12] = tb;
    hipLaunchKernelGGL(k_tv_seed_cl_multi, dim3(tb), dim3(256), 0, (hipStream_t)stream, A);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}


// ---- the fused training step's optimiser launches (t2n_train_step, t2n_backward.hip): the same kernels, with the step's scalars,
// TV weights and verdict read from device memory ----------------------------------------------------------------------------------
namespace t2n {
static void factor_step_geometry(t2n_field* f, FactorStep& A, unsigned& ab, unsigned& tb_all) {
    const int* gr = f->desc.grid;
    ab = 0; tb_all = 0;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const int idx = q * 3 + k;
            const int H = gr[mat1(k)], W = gr[mat0(k)];
            const long long HW = (long long)H * W, L = gr[vecm(k)];
            float* pcl[4] = {f->buf_den_plane[k], f->buf_den_line[k], f->buf_app_plane[k], f->buf_app_line[k]};
            float* gcl[4] = {f->gbuf_den_plane[k], f->gbuf_den_line[k], f->gbuf_app_plane[k], f->gbuf_app_line[k]};
            const int C = q < 2 ? 16 : 48;
            const bool plane = (q & 1) == 0;
            A.p[idx] = pcl[q]; A.g[idx] = gcl[q];
            A.npos[idx] = plane ? HW : L; A.C[idx] = C; A.H[idx] = plane ? H : (int)L; A.W[idx] = plane ? W : 1;
            A.tblock0[idx] = tb_all;
            tb_all += (unsigned)((A.npos[idx] * (C / 4) + 255) / 256);
            A.ablock0[idx] = ab;
            ab += (unsigned)((A.npos[idx] + 63) / 64);
        }
    A.tblock0[12] = tb_all; A.ablock0[12] = ab;
}
// the sharded optimiser's partition of the factor tensors (order: density planes, density lines, appearance planes, appearance lines):
// a plane of n positions has n / 64 whole blocks; chunk = (n / 64) / world of them per rank, body = world * chunk; lines: body = 0
void shard_partition(const t2n_field* f, int world, int rank, unsigned lo[12], unsigned hi[12], unsigned body[12]) {
    const int* gr = f->desc.grid;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const int idx = q * 3 + k;
            const long long HW = (long long)gr[mat1(k)] * gr[mat0(k)];
            const bool plane = (q & 1) == 0;
            const unsigned chunk = (plane && world > 1) ? (unsigned)((HW / 64) / world) : 0u;
            body[idx] = chunk * (unsigned)(world > 1 ? world : 1);
            lo[idx] = chunk * (unsigned)(world > 1 ? rank : 0); hi[idx] = lo[idx] + chunk;
        }
}
static void factor_step_shard(const t2n_field* f, FactorStep& A, int world, int rank, bool relayout_only, unsigned& ab) {
    shard_partition(f, world, rank, A.own_lo, A.own_hi, A.body);
    A.world = world > 1 ? world : 1;
    A.relayout_only = relayout_only ? 1 : 0;
    ab = 0;
    for (int i = 0; i < 12; ++i) {
        const unsigned total = (unsigned)((A.npos[i] + 63) / 64), own = A.own_hi[i] - A.own_lo[i];
        A.ablock0[i] = ab;
        ab += relayout_only ? A.body[i] - own : own + (total - A.body[i]);
    }
    A.ablock0[12] = ab;
}
int launch_tv_seed_dev(t2n_field* f, const float* tvw_dev, hipStream_t s, int world, int rank) {
    if (f->desc.grid[0] < 2 || f->desc.grid[1] < 2 || f->desc.grid[2] < 2) { set_error("t2n_train_step: TV needs planes of at least 2x2"); return T2N_ERR_INVALID; }
    FactorStep A;
    memset(&A, 0, sizeof(A));
    unsigned ab, tb;
    factor_step_geometry(f, A, ab, tb);
    factor_step_shard(f, A, world, rank, false, ab);
    A.tvw_dev = tvw_dev;
    hipLaunchKernelGGL(k_tv_seed_cl_multi, dim3(tb), dim3(256), 0, s, A);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
// Adam on the channel-last copies of the factor tensors first .. first + count - 1 (order: density planes, density lines, appearance
// planes, appearance lines), new values also written to the caller's reference-layout tensors
int launch_factor_adam_dev(t2n_field* f, const t2n_field_params* params, float* const* m, float* const* v, float beta1, float beta2, float eps,
                           const TrainScalars* st, int first, int count, hipStream_t s, int world, int rank, bool relayout_only) {
    FactorStep A;
    memset(&A, 0, sizeof(A));
    unsigned ab, tb;
    factor_step_geometry(f, A, ab, tb);
    factor_step_shard(f, A, world, rank, relayout_only, ab);
    for (int k = 0; k < 3; ++k) {
        A.ref[k] = (float*)params->density_plane[k]; A.ref[3 + k] = (float*)params->density_line[k];
        A.ref[6 + k] = (float*)params->app_plane[k]; A.ref[9 + k] = (float*)params->app_line[k];
    }
    for (int i = 0; i < 12; ++i) { A.m[i] = m ? m[i] : nullptr; A.v[i] = v ? v[i] : nullptr; if (!A.ref[i] || (!relayout_only && (!A.m[i] || !A.v[i]))) { set_error("t2n_train_step: NULL factor tensor / moment %d", i); return T2N_ERR_INVALID; } }
    A.beta1 = beta1; A.beta2 = beta2; A.eps = eps; A.st = st;
    // a sub-range of the tensors: the launch covers their workgroups only (block index rebased by shifting the table)
    const unsigned b0 = A.ablock0[first], b1 = A.ablock0[first + count];
    if (first > 0 || count < 12) {
        for (int i = 0; i <= 12; ++i) A.ablock0[i] = A.ablock0[i] >= b0 ? A.ablock0[i] - b0 : 0u;
        // tensors before `first` get zero-width ranges at 0 (the kernel's search lands on the last tensor whose start is <= the block)
        for (int i = first + count + 1; i <= 12; ++i) A.ablock0[i] = 0xffffffffu;
    }
    if (b1 > b0) hipLaunchKernelGGL(k_adam_cl_multi, dim3(b1 - b0), dim3(256), 0, s, A);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
// Adam on the seven head tensors (reference layout; gradients contiguous in `grads_flat`, parameter order)
int launch_head_adam_dev(const t2n_field_params* params, const float* grads_flat, float* const* m, float* const* v, float beta1, float beta2,
                         float eps, const TrainScalars* st, hipStream_t s, bool zero_grads, unsigned* zero_words4) {
    AdamMulti a;
    memset(&a, 0, sizeof(a));
    float* ps[7] = {(float*)params->basis_weight, (float*)params->mlp_w0, (float*)params->mlp_b0, (float*)params->mlp_w1, (float*)params->mlp_b1,
                    (float*)params->mlp_w2, (float*)params->mlp_b2};
    const long long n[7] = {27 * 144, 128 * 351, 128, 128 * 128, 128, 3 * 128, 3};
    unsigned blocks = 0;
    long long off = 0;
    for (int i = 0; i < 7; ++i) {
        if (!ps[i] || !m[i] || !v[i]) { set_error("t2n_train_step: NULL head tensor / moment %d", i); return T2N_ERR_INVALID; }
        a.p[i] = ps[i]; a.g[i] = grads_flat + off; a.m[i] = m[i]; a.v[i] = v[i]; a.n[i] = n[i];
        a.block0[i] = blocks;
        blocks += (unsigned)((n[i] + 255) / 256);
        off += n[i];
    }
    a.block0[7] = blocks;
    a.count = 7; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.st = st; a.st_first = 12; a.zero_grads = zero_grads ? 1 : 0; a.zero_words4 = zero_words4;
    hipLaunchKernelGGL(k_adam_multi, dim3(blocks), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
}  // namespace t2n

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

extern "C" int t2n_tv_value(const float* param, int C, int H, int W, double* sums, t2n_stream stream) {
    if (!param || !sums || C <= 0 || H <= 1 || W <= 1) { set_error("t2n_tv_value: bad argument"); return T2N_ERR_INVALID; }
    long long blocks = ((long long)C * H + 3) / 4;
    if (blocks > 2048) blocks = 2048;   // same-address atomics serialise: two per block
    hipLaunchKernelGGL(k_tv_value, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, C, H, W, sums);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_tv_grad_add(const float* param, float* grad, int C, int H, int W, float weight, t2n_stream stream) {
    if (!param || !grad || C <= 0 || H <= 1 || W <= 1) { set_error("t2n_tv_grad_add: bad argument"); return T2N_ERR_INVALID; }
    // TVLoss: weight * 2 * (h_tv / count_h + w_tv / count_w) / batch, batch = 1
    const float sh = weight * 2.f / ((float)C * (float)(H - 1) * (float)W);
    const float sw = weight * 2.f / ((float)C * (float)H * (float)(W - 1));
    long long blocks = ((long long)C * H + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_tv_grad<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, C, H, W, sh, sw, (const float*)nullptr);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_tv_grad_set(const float* param, float* grad, int C, int H, int W, float weight, const float* upstream, t2n_stream stream) {
    if (!param || !grad || C <= 0 || H <= 1 || W <= 1) { set_error("t2n_tv_grad_set: bad argument"); return T2N_ERR_INVALID; }
    const float sh = weight * 2.f / ((float)C * (float)(H - 1) * (float)W);
    const float sw = weight * 2.f / ((float)C * (float)H * (float)(W - 1));
    long long blocks = ((long long)C * H + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_tv_grad<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, C, H, W, sh, sw, upstream);
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
