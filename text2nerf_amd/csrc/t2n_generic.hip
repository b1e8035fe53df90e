// General-shape render path: field / head shapes beyond what the tuned kernels hold (more than 16 density or 48 appearance
// components per plane, app_dim > 27 / fea_pe > 6 / featureC > 128 on the MLP heads). The reference is generic in all of them
// (models/tensoRF.py:144-160, models/tensorBase.py:62-159, e_opt.py:83-107); here they run as a plain restatement — one thread per
// ray for the march, one thread per appearance sample for the head, loops over the component / unit counts, parameters read in place
// in the reference's own layouts ([1,C,H,W] planes, [1,C,L,1] lines), gradients accumulated with atomics into those layouts. No MFMA,
// no LDS staging: a path that removes the shape limit, not a fast one (the driver's configuration never comes here).
//
// Replaces: models/tensorBase.py:304-323 (sample_ray), :436-507 (forward), :19-26 (raw2alpha), :406-410 (feature2density), :11-17 +
// :62-159 (the heads), :29-39 (SH / RGB), models/tensoRF.py:205-239 (compute_densityfeature / compute_appfeature), and their autograd.
#include <stdlib.h>

#include "t2n_device.h"

namespace t2n {

constexpr int kGenDimMax = 64, kGenHidMax = 256, kGenInMax = 4096;

struct GenArgs {
    FieldDev F;                       // scalars only (aabb, step, thresholds, activation); the factor sets of F are unused here
    int grid[3], Cd[3], Ca[3], app_dim, shading, fea_pe, view_pe, fC, in0;
    const float* dp[3]; const float* dl[3]; const float* ap[3]; const float* al[3];
    const float* basis; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    float* g_dp[3]; float* g_dl[3]; float* g_ap[3]; float* g_al[3];
    float* g_basis; float* g_w0; float* g_b0; float* g_w1; float* g_b1; float* g_w2; float* g_b2;
    const float* rays; long long n_rays; int ray_stride; int N; int train; int add_bg;
    const float* jitter;
    float* rgb; float* depth;         // outputs
    float* w; float* z;               // [R, N] rows (the caller's tensors or workspace)
    float* sigma; float* T;           // [R, N] context: density and transmittance in front of every sample
    float* rgb_s;                     // [R, N, 3] colours of the appearance samples (context)
    float* acc; float* raw;           // [R] opacity, [R, 3] colour before the clamp
    const float* d_rgb; const float* d_depth; const float* d_w;   // upstream gradients
    float* go;                        // [R, N, 3] dL/d colour of the appearance samples (backward scratch)
    unsigned long long* stats;
};

struct Tap3 { Axis a[3]; };
__device__ __forceinline__ Tap3 gen_taps(const GenArgs& a, float xn, float yn, float zn) {
    Tap3 t;
    t.a[0] = axis_taps(xn, a.grid[0]); t.a[1] = axis_taps(yn, a.grid[1]); t.a[2] = axis_taps(zn, a.grid[2]);
    return t;
}
// plane k of a factor set in the reference layout: value(c, y, x) = P[(c * H + y) * W + x], H = grid[mat1(k)], W = grid[mat0(k)]
__device__ __forceinline__ float gen_plane(const float* __restrict__ P, int c, int H, int W, const Axis& ax, const Axis& ay) {
    const float* __restrict__ p = P + (size_t)c * H * W;
    float v = p[(size_t)ay.i0 * W + ax.i0] * (ay.w0 * ax.w0);
    v = fmaf(p[(size_t)ay.i0 * W + ax.i1], ay.w0 * ax.w1, v);
    v = fmaf(p[(size_t)ay.i1 * W + ax.i0], ay.w1 * ax.w0, v);
    v = fmaf(p[(size_t)ay.i1 * W + ax.i1], ay.w1 * ax.w1, v);
    return v;
}
__device__ __forceinline__ float gen_line(const float* __restrict__ Ln, int c, int Lsz, const Axis& al) {
    const float* __restrict__ p = Ln + (size_t)c * Lsz;
    return fmaf(p[al.i1], al.w1, p[al.i0] * al.w0);
}
__device__ __forceinline__ void gen_plane_add(float* __restrict__ G, int c, int H, int W, const Axis& ax, const Axis& ay, float g) {
    float* __restrict__ p = G + (size_t)c * H * W;
    atomicAdd(p + (size_t)ay.i0 * W + ax.i0, g * (ay.w0 * ax.w0));
    atomicAdd(p + (size_t)ay.i0 * W + ax.i1, g * (ay.w0 * ax.w1));
    atomicAdd(p + (size_t)ay.i1 * W + ax.i0, g * (ay.w1 * ax.w0));
    atomicAdd(p + (size_t)ay.i1 * W + ax.i1, g * (ay.w1 * ax.w1));
}
__device__ __forceinline__ void gen_line_add(float* __restrict__ G, int c, int Lsz, const Axis& al, float g) {
    float* __restrict__ p = G + (size_t)c * Lsz;
    atomicAdd(p + al.i0, g * al.w0);
    atomicAdd(p + al.i1, g * al.w1);
}

__device__ __forceinline__ float gen_density_feature(const GenArgs& a, const Tap3& t) {
    float feat = 0.f;
    for (int k = 0; k < 3; ++k) {
        const int W = a.grid[mat0(k)], H = a.grid[mat1(k)], Lz = a.grid[vecm(k)];
        const Axis& ax = t.a[mat0(k)]; const Axis& ay = t.a[mat1(k)]; const Axis& al = t.a[vecm(k)];
        for (int c = 0; c < a.Cd[k]; ++c) feat = fmaf(gen_plane(a.dp[k], c, H, W, ax, ay), gen_line(a.dl[k], c, Lz, al), feat);
    }
    return feat;
}

__device__ __forceinline__ bool gen_point(const GenArgs& a, const Ray& ray, float z, float& xn, float& yn, float& zn) {
    return a.train ? sample_point<true>(a.F, ray, z, xn, yn, zn) : sample_point<false>(a.F, ray, z, xn, yn, zn);
}
__device__ __forceinline__ float gen_z(const GenArgs& a, const Ray& ray, int i, float u) {
    return a.train ? sample_z<true, true>(a.F, ray, i, u) : sample_z<false, true>(a.F, ray, i, 0.f);
}

// ---- march: one thread per ray --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gen_march(const GenArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rays) return;
    const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
    const float u = a.train ? a.jitter[r] : 0.f;
    const int N = a.N;
    float T = 1.f, acc = 0.f, dep = 0.f;
    unsigned long long nev = 0, napp = 0;
    for (int i = 0; i < N; ++i) {
        const float z = gen_z(a, ray, i, u);
        float xn, yn, zn;
        const bool ok = gen_point(a, ray, z, xn, yn, zn);
        float sg = 0.f;
        if (ok) { sg = feature2density(a.F, gen_density_feature(a, gen_taps(a, xn, yn, zn))); ++nev; }
        const float dist = i < N - 1 ? gen_z(a, ray, i + 1, u) - z : 0.f;                 // :448
        const float alpha = 1.f - expf((-sg) * (dist * a.F.dscale));                      // :19-26
        const float w = alpha * T;
        a.z[r * N + i] = z; a.sigma[r * N + i] = sg; a.T[r * N + i] = T; a.w[r * N + i] = w;
        T = T * ((1.f - alpha) + 1e-10f);
        acc += w;
        dep = fmaf(w, z, dep);
        napp += (w > a.F.thres) ? 1u : 0u;
    }
    a.acc[r] = acc;
    a.depth[r] = dep + (1.f - acc) * ray.last;                                             // :504-505
    if (a.stats) { atomicAdd(&a.stats[T2N_STAT_EVALUATED], nev); atomicAdd(&a.stats[T2N_STAT_APPEARANCE], napp); }
}

// ---- head: one thread per appearance sample ---------------------------------------------------------------------------------------
// X[col] = plane x line of appearance component col (planes concatenated: models/tensoRF.py:223-239), feat = basis_mat X
__device__ __forceinline__ void gen_features(const GenArgs& a, const Tap3& t, float* __restrict__ feat) {
    for (int f = 0; f < a.app_dim; ++f) feat[f] = 0.f;
    int col = 0;
    const int ncol = a.Ca[0] + a.Ca[1] + a.Ca[2];
    for (int k = 0; k < 3; ++k) {
        const int W = a.grid[mat0(k)], H = a.grid[mat1(k)], Lz = a.grid[vecm(k)];
        const Axis& ax = t.a[mat0(k)]; const Axis& ay = t.a[mat1(k)]; const Axis& al = t.a[vecm(k)];
        for (int c = 0; c < a.Ca[k]; ++c, ++col) {
            const float x = gen_plane(a.ap[k], c, H, W, ax, ay) * gen_line(a.al[k], c, Lz, al);
            for (int f = 0; f < a.app_dim; ++f) feat[f] = fmaf(a.basis[(size_t)f * ncol + col], x, feat[f]);
        }
    }
}
// The MLP heads' input row in the reference's column order (models/tensorBase.py:11-17, :75-84, :101-107, :148-155), enumerated
// instead of stored (a 1 024-float row per thread would be scratch memory): fn(j, x_j, f, d) with f >= 0 for a column that depends on
// feature f (d = its derivative with respect to that feature), f = -1 for the view-direction columns.
template <class FN>
__device__ __forceinline__ int gen_inputs(const GenArgs& a, const float* __restrict__ feat, const float* __restrict__ dir, FN&& fn) {
    int n = 0;
    const bool view = a.shading != T2N_SHADE_MLP_FEA_NOVIEW;
    for (int f = 0; f < a.app_dim; ++f) fn(n++, feat[f], f, 1.f);
    if (view) for (int d = 0; d < 3; ++d) fn(n++, dir[d], -1, 0.f);
    const int fpe = a.shading == T2N_SHADE_MLP ? 0 : a.fea_pe;
    if (fpe > 0) {
        for (int f = 0; f < a.app_dim; ++f) for (int o = 0; o < fpe; ++o) { const float sc = (float)(1 << o), t = feat[f] * sc; fn(n++, sinf(t), f, cosf(t) * sc); }
        for (int f = 0; f < a.app_dim; ++f) for (int o = 0; o < fpe; ++o) { const float sc = (float)(1 << o), t = feat[f] * sc; fn(n++, cosf(t), f, -sinf(t) * sc); }
    }
    if (view && a.view_pe > 0) {
        for (int d = 0; d < 3; ++d) for (int o = 0; o < a.view_pe; ++o) fn(n++, sinf(dir[d] * (float)(1 << o)), -1, 0.f);
        for (int d = 0; d < 3; ++d) for (int o = 0; o < a.view_pe; ++o) fn(n++, cosf(dir[d] * (float)(1 << o)), -1, 0.f);
    }
    return n;
}
__device__ __forceinline__ void gen_sh9(const float* dir, float* sh) {   // models/sh.py:4-14,87-112 (degree 2)
    const float dx = dir[0], dy = dir[1], dz = dir[2];
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
    sh[0] = C0; sh[1] = -C1 * dy; sh[2] = C1 * dz; sh[3] = -C1 * dx;
    sh[4] = 1.0925484305920792f * xy; sh[5] = -1.0925484305920792f * yz; sh[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    sh[7] = -1.0925484305920792f * xz; sh[8] = 0.5462742152960396f * (xx - yy);
}
// hidden layers: h = b + W x (row-major [out, in] like nn.Linear), pre-activation values kept (the backward needs the ReLU masks)
__device__ __forceinline__ void gen_linear(const float* __restrict__ W, const float* __restrict__ b, const float* __restrict__ x, int nin, int nout,
                                           bool relu_in, float* __restrict__ h) {
    for (int o = 0; o < nout; ++o) {
        float s = b[o];
        const float* __restrict__ wr = W + (size_t)o * nin;
        for (int j = 0; j < nin; ++j) s = fmaf(wr[j], relu_in ? fmaxf(x[j], 0.f) : x[j], s);
        h[o] = s;
    }
}

template <bool BACKWARD>
__global__ __launch_bounds__(64) void k_gen_shade(const GenArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.N;
    if (t >= a.n_rays * N) return;
    const float w = a.w[t];
    if (!(w > a.F.thres)) return;                                                            // :477
    const long long r = t / N;
    const int i = (int)(t - r * N);
    const float* __restrict__ rp = a.rays + r * a.ray_stride;
    const Ray ray = load_ray(a.F, rp, a.ray_stride);
    const float u = a.train ? a.jitter[r] : 0.f;
    float xn, yn, zn;
    gen_point(a, ray, gen_z(a, ray, i, u), xn, yn, zn);
    const Tap3 tp = gen_taps(a, xn, yn, zn);
    float feat[kGenDimMax];
    gen_features(a, tp, feat);
    const float dir[3] = {rp[3], rp[4], rp[5]};
    float gfeat[kGenDimMax];
    if (BACKWARD) for (int f = 0; f < a.app_dim; ++f) gfeat[f] = 0.f;
    const float* __restrict__ go = BACKWARD ? a.go + t * 3 : nullptr;
    if (a.shading == T2N_SHADE_RGB) {
        if (!BACKWARD) { a.rgb_s[t * 3] = feat[0]; a.rgb_s[t * 3 + 1] = feat[1]; a.rgb_s[t * 3 + 2] = feat[2]; }
        else { gfeat[0] = go[0]; gfeat[1] = go[1]; gfeat[2] = go[2]; }
    } else if (a.shading == T2N_SHADE_SH) {
        float sh[9];
        gen_sh9(dir, sh);
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            for (int b = 0; b < 9; ++b) s = fmaf(sh[b], feat[c * 9 + b], s);
            if (!BACKWARD) a.rgb_s[t * 3 + c] = fmaxf(s + 0.5f, 0.f);                        // :29-33
            else if (s + 0.5f > 0.f) for (int b = 0; b < 9; ++b) gfeat[c * 9 + b] = go[c] * sh[b];
        }
    } else {
        float h0[kGenHidMax], h1[kGenHidMax];
        const int fC = a.fC, nin = a.in0;
        for (int v = 0; v < fC; ++v) h0[v] = a.b0[v];
        gen_inputs(a, feat, dir, [&](int j, float xj, int, float) {
            for (int v = 0; v < fC; ++v) h0[v] = fmaf(a.w0[(size_t)v * nin + j], xj, h0[v]);
        });
        gen_linear(a.w1, a.b1, h0, fC, fC, true, h1);
        float o[3];
        gen_linear(a.w2, a.b2, h1, fC, 3, true, o);
        float c3[3];
        for (int c = 0; c < 3; ++c) c3[c] = 1.f / (1.f + expf(-o[c]));
        if (!BACKWARD) { a.rgb_s[t * 3] = c3[0]; a.rgb_s[t * 3 + 1] = c3[1]; a.rgb_s[t * 3 + 2] = c3[2]; }
        else {
            float d2[3];
            for (int c = 0; c < 3; ++c) d2[c] = go[c] * c3[c] * (1.f - c3[c]);
            // layer 2: dW2, db2, d h1 (h1 is overwritten by its gradient, masked by its own sign)
            for (int c = 0; c < 3; ++c) if (a.g_b2) atomicAdd(a.g_b2 + c, d2[c]);
            for (int v = 0; v < fC; ++v) {
                const float hv = fmaxf(h1[v], 0.f);
                float g = 0.f;
                for (int c = 0; c < 3; ++c) {
                    if (a.g_w2 && hv != 0.f) atomicAdd(a.g_w2 + (size_t)c * fC + v, d2[c] * hv);
                    g = fmaf(a.w2[(size_t)c * fC + v], d2[c], g);
                }
                h1[v] = h1[v] > 0.f ? g : 0.f;
            }
            // layer 1: dW1, db1, d h0
            float g0[kGenHidMax];
            for (int v = 0; v < fC; ++v) g0[v] = 0.f;
            for (int uu = 0; uu < fC; ++uu) {
                const float g = h1[uu];
                if (g == 0.f) continue;
                if (a.g_b1) atomicAdd(a.g_b1 + uu, g);
                const float* __restrict__ wr = a.w1 + (size_t)uu * fC;
                for (int v = 0; v < fC; ++v) {
                    const float hv = fmaxf(h0[v], 0.f);
                    if (a.g_w1 && hv != 0.f) atomicAdd(a.g_w1 + (size_t)uu * fC + v, g * hv);
                    g0[v] = fmaf(wr[v], g, g0[v]);
                }
            }
            for (int v = 0; v < fC; ++v) g0[v] = h0[v] > 0.f ? g0[v] : 0.f;
            // layer 0: dW0, db0, and d features through the raw columns and the encoding's derivative (the view directions carry
            // no gradient)
            for (int v = 0; v < fC; ++v) if (a.g_b0 && g0[v] != 0.f) atomicAdd(a.g_b0 + v, g0[v]);
            gen_inputs(a, feat, dir, [&](int j, float xj, int f, float dxdf) {
                float g = 0.f;
                for (int v = 0; v < fC; ++v) {
                    const float gv = g0[v];
                    if (gv == 0.f) continue;
                    if (a.g_w0) atomicAdd(a.g_w0 + (size_t)v * nin + j, gv * xj);
                    g = fmaf(a.w0[(size_t)v * nin + j], gv, g);
                }
                if (f >= 0) gfeat[f] = fmaf(g, dxdf, gfeat[f]);
            });
        }
    }
    if (BACKWARD) {
        // basis_mat and the appearance factors: X[col] recomputed, dB[f][col] += gfeat[f] X[col], dX[col] = sum_f B[f][col] gfeat[f]
        int col = 0;
        const int ncol = a.Ca[0] + a.Ca[1] + a.Ca[2];
        for (int k = 0; k < 3; ++k) {
            const int W = a.grid[mat0(k)], H = a.grid[mat1(k)], Lz = a.grid[vecm(k)];
            const Axis& ax = tp.a[mat0(k)]; const Axis& ay = tp.a[mat1(k)]; const Axis& al = tp.a[vecm(k)];
            for (int c = 0; c < a.Ca[k]; ++c, ++col) {
                const float pv = gen_plane(a.ap[k], c, H, W, ax, ay), lv = gen_line(a.al[k], c, Lz, al);
                const float xv = pv * lv;
                float gx = 0.f;
                for (int f = 0; f < a.app_dim; ++f) {
                    if (a.g_basis) atomicAdd(a.g_basis + (size_t)f * ncol + col, gfeat[f] * xv);
                    gx = fmaf(a.basis[(size_t)f * ncol + col], gfeat[f], gx);
                }
                if (a.g_ap[k]) gen_plane_add(a.g_ap[k], c, H, W, ax, ay, gx * lv);
                if (a.g_al[k]) gen_line_add(a.g_al[k], c, Lz, al, gx * pv);
            }
        }
    }
}

// ---- composite: one thread per ray (models/tensorBase.py:494-501) ----------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gen_composite(const GenArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rays) return;
    const int N = a.N;
    float c[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < N; ++i) {
        const float w = a.w[r * N + i];
        if (w > a.F.thres) for (int k = 0; k < 3; ++k) c[k] = fmaf(w, a.rgb_s[(r * N + i) * 3 + k], c[k]);
    }
    if (a.add_bg) { const float bg = 1.f - a.acc[r]; c[0] += bg; c[1] += bg; c[2] += bg; }
    for (int k = 0; k < 3; ++k) { a.raw[r * 3 + k] = c[k]; a.rgb[r * 3 + k] = fminf(fmaxf(c[k], 0.f), 1.f); }
}

// ---- backward of composite + raw2alpha + the density factors: one thread per ray -------------------------------------------------
__global__ __launch_bounds__(64) void k_gen_bwd_march(const GenArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rays) return;
    const int N = a.N;
    const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
    float dc[3];
    float dbg = 0.f;
    for (int k = 0; k < 3; ++k) {
        const float raw = a.raw[r * 3 + k];
        dc[k] = (raw >= 0.f && raw <= 1.f) ? a.d_rgb[r * 3 + k] : 0.f;        // clamp(0, 1): autograd passes the gradient on the CLOSED interval (as k_bwd_march)
        dbg += dc[k];
    }
    const float dd = a.d_depth[r];
    // dL/dw_i of the direct terms; rgb_map += 1 - acc and depth_map += (1 - acc) * last put -dbg and -dd * last on every weight
    const float common = (a.add_bg ? -dbg : 0.f) - dd * ray.last;
    float S = 0.f;   // sum over j > i of G_j w_j
    for (int i = N - 1; i >= 0; --i) {
        const long long t = r * N + i;
        const float w = a.w[t], z = a.z[t], sg = a.sigma[t], T = a.T[t];
        float G = common + dd * z + (a.d_w ? a.d_w[t] : 0.f);
        if (w > a.F.thres) {
            for (int k = 0; k < 3; ++k) {
                G = fmaf(dc[k], a.rgb_s[t * 3 + k], G);
                a.go[t * 3 + k] = dc[k] * w;
            }
        }
        const float dist = i < N - 1 ? a.z[t + 1] - z : 0.f;
        const float sd = dist * a.F.dscale;
        const float e = expf((-sg) * sd);                     // 1 - alpha
        const float fct = e + 1e-10f;                          // T_{i+1} = T_i * fct (fct = (1 - alpha) + 1e-10, alpha = 1 - e)
        const float dalpha = G * T - S / fct;
        S = fmaf(G, w, S);
        const float dsg = dalpha * sd * e;                     // d alpha / d sigma = dist * scale * exp(-sigma dist scale)
        if (dsg == 0.f || (sg == 0.f && a.F.act == T2N_ACT_RELU)) continue;
        float xn, yn, zn;
        if (!gen_point(a, ray, z, xn, yn, zn)) continue;       // (sigma is 0 and constant outside the box)
        const Tap3 tp = gen_taps(a, xn, yn, zn);
        // relu: sg > 0 here. softplus (threshold 20: identity above; softplus(x) <= 20 exactly for x <= 20 in fp32): 1 - exp(-softplus)
        const float dfeat = (a.F.act == T2N_ACT_RELU || sg > 20.f) ? dsg : dsg * (-expm1f(-sg));
        for (int k = 0; k < 3; ++k) {
            const int W = a.grid[mat0(k)], H = a.grid[mat1(k)], Lz = a.grid[vecm(k)];
            const Axis& ax = tp.a[mat0(k)]; const Axis& ay = tp.a[mat1(k)]; const Axis& al = tp.a[vecm(k)];
            for (int c = 0; c < a.Cd[k]; ++c) {
                const float pv = gen_plane(a.dp[k], c, H, W, ax, ay), lv = gen_line(a.dl[k], c, Lz, al);
                if (a.g_dp[k]) gen_plane_add(a.g_dp[k], c, H, W, ax, ay, dfeat * lv);
                if (a.g_dl[k]) gen_line_add(a.g_dl[k], c, Lz, al, dfeat * pv);
            }
        }
    }
}

// =====================================================================================================================================
// The forward as kernels (round 6; the backward above stays the plain restatement): what made the first form 1 500 x slower than the tuned
// path per sample was its shape, not its arithmetic — one THREAD walked a whole ray (2 500 waves for a 400 x 400 frame) reading
// [1, C, H, W] factors one component per cache line, and one thread ran a whole MLP out of scratch memory. Here:
//   k_gen_stage      the twelve factor tensors -> channel-last copies in the workspace ([pos][C]: a tap's components are contiguous)
//   k_gen_sigma      one thread per (ray, sample): depth, box test, density feature (components in the reference's order: the same
//                    sums as the plain form), sigma
//   k_gen_scan_compact  one wave per ray: transmittance (wave product scan) / weights / opacity / depth from the sigmas, and the appearance
//                    samples (weight > threshold) -> an index list (one reservation per workgroup of four rays)
//   k_gen_head       one workgroup per 64 list entries: appearance features (each wave a quarter of the components, basis_mat from LDS),
//                    then the head — MLP layers as [units of this wave] x [64 samples] outer products on the VALU with the weights as
//                    wave-uniform (scalar) operands and the activations in LDS; SH / RGB per sample.
//                    MLP heads with <= 512 inputs: the kernel only writes the input rows of one PASS of the list (262 144 entries), layers
//                    0 / 1 run on the matrix cores (k_dense of t2n_heads.hip, exact-fp32 MFMA), k_gen_out finishes (layer 2 + sigmoid);
//                    passes are issued for the worst case and clipped to the device-side count (T2N_GENERIC_VALU_HEAD=1: the VALU form)
// Taken when the workspace has room for the staged copies and the list (t2n_generic_workspace_bytes_desc); same outputs and context as
// the plain form (the backward reads them), values equal to rounding (the density sums are bit-identical).
struct GenFast {
    const float* dp[3]; const float* dl[3]; const float* ap[3]; const float* al[3];   // channel-last copies
    int* list; unsigned* count; unsigned cap;
    int ncol;
    const float* w0p; const float* w1p; int ld0, ld1;   // MLP weights with rows padded to multiples of 16 floats (64-byte aligned rows: s_load_dwordx16)
    // MLP layers on the matrix cores (k_dense of t2n_heads.hip, exact-fp32 MFMA): the head kernel then only WRITES the input rows of
    // one pass of the list — entries [row0, row0 + rows_cap) — to x0 [rows_cap][ldx]; k_gen_out finishes from h1
    float* x0; int ldx; unsigned row0, rows_cap; HeadPlanDev* plan;
    int hbc;   // columns of Hb (a multiple of 16): the head kernel's LDS staging of the input rows
};
__global__ __launch_bounds__(256) void k_gen_padrows(const float* __restrict__ src, float* __restrict__ dst, int rows, int n, int ld) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)rows * ld) return;
    const int r = (int)(t / ld), j = (int)(t - (long long)r * ld);
    dst[t] = j < n ? src[(size_t)r * n + j] : 0.f;
}
struct GenStageArgs { const float* src[12]; float* dst[12]; int C[12]; long long HW[12]; unsigned block0[13]; };
__global__ __launch_bounds__(256) void k_gen_stage(const GenStageArgs a) {
    __shared__ float tile[64][65];
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < 12; ++q) t += (a.block0[q] <= blockIdx.x) ? 1 : 0;
    const int C = a.C[t];
    const long long HW = a.HW[t];
    const int cb = (C + 63) / 64;                       // channel blocks of 64
    const unsigned b = blockIdx.x - a.block0[t];
    const long long p0 = (long long)(b / cb) * 64;
    const int c0 = (int)(b % cb) * 64;
    const int lp = threadIdx.x & 63, q4 = threadIdx.x >> 6;
    for (int c = q4; c < 64; c += 4)
        tile[c][lp] = (c0 + c < C && p0 + lp < HW) ? a.src[t][(long long)(c0 + c) * HW + p0 + lp] : 0.f;
    __syncthreads();
    for (int pp = q4; pp < 64; pp += 4)
        if (c0 + lp < C && p0 + pp < HW) a.dst[t][(p0 + pp) * C + c0 + lp] = tile[lp][pp];
}
// plane / line value of component c from the channel-last copies (same expressions as gen_plane / gen_line)
__device__ __forceinline__ float gen_density_feature_cl(const GenArgs& a, const GenFast& fa, const Tap3& t) {
    float feat = 0.f;
    for (int k = 0; k < 3; ++k) {
        const int W = a.grid[mat0(k)], C = a.Cd[k];
        const Axis& ax = t.a[mat0(k)]; const Axis& ay = t.a[mat1(k)]; const Axis& al = t.a[vecm(k)];
        const float* __restrict__ p00 = fa.dp[k] + ((size_t)ay.i0 * W + ax.i0) * C;
        const float* __restrict__ p01 = fa.dp[k] + ((size_t)ay.i0 * W + ax.i1) * C;
        const float* __restrict__ p10 = fa.dp[k] + ((size_t)ay.i1 * W + ax.i0) * C;
        const float* __restrict__ p11 = fa.dp[k] + ((size_t)ay.i1 * W + ax.i1) * C;
        const float* __restrict__ l0 = fa.dl[k] + (size_t)al.i0 * C;
        const float* __restrict__ l1 = fa.dl[k] + (size_t)al.i1 * C;
        const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
        for (int c = 0; c < C; ++c) {
            float v = p00[c] * w00;
            v = fmaf(p01[c], w01, v); v = fmaf(p10[c], w10, v); v = fmaf(p11[c], w11, v);
            feat = fmaf(v, fmaf(l1[c], al.w1, l0[c] * al.w0), feat);
        }
    }
    return feat;
}
__global__ __launch_bounds__(256) void k_gen_sigma(const GenArgs a, const GenFast fa) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int N = a.N;
    if (t >= a.n_rays * N) return;
    const long long r = t / N;
    const int i = (int)(t - r * N);
    const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
    const float u = a.train ? a.jitter[r] : 0.f;
    const float z = gen_z(a, ray, i, u);
    float xn, yn, zn;
    float sg = 0.f;
    const bool ok = gen_point(a, ray, z, xn, yn, zn);
    if (ok) sg = feature2density(a.F, gen_density_feature_cl(a, fa, gen_taps(a, xn, yn, zn)));
    a.z[t] = z; a.sigma[t] = sg;
    a.T[t] = ok ? 1.f : 0.f;      // (in-box flag for the scan's statistics; overwritten there)
}
__global__ __launch_bounds__(64) void k_gen_scan(const GenArgs a) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.n_rays) return;
    const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
    const int N = a.N;
    float T = 1.f, acc = 0.f, dep = 0.f;
    unsigned long long nev = 0, napp = 0;
    float z = a.z[r * N];
    for (int i = 0; i < N; ++i) {
        const float zn = i < N - 1 ? a.z[r * N + i + 1] : z;
        const float sg = a.sigma[r * N + i];
        nev += a.T[r * N + i] != 0.f ? 1u : 0u;
        const float dist = i < N - 1 ? zn - z : 0.f;
        const float alpha = 1.f - expf((-sg) * (dist * a.F.dscale));
        const float w = alpha * T;
        a.T[r * N + i] = T; a.w[r * N + i] = w;
        T = T * ((1.f - alpha) + 1e-10f);
        acc += w;
        dep = fmaf(w, z, dep);
        napp += (w > a.F.thres) ? 1u : 0u;
        z = zn;
    }
    a.acc[r] = acc;
    a.depth[r] = dep + (1.f - acc) * ray.last;
    if (a.stats) { atomicAdd(&a.stats[T2N_STAT_EVALUATED], nev); atomicAdd(&a.stats[T2N_STAT_APPEARANCE], napp); }
}
// scan + compaction as ONE kernel, a wave per ray: coalesced loads of the ray's sigmas / depths, transmittance by a wave product scan
// (64 samples per round, carried across rounds), weights, opacity, depth; the ray's appearance samples are appended to the list with one
// atomic per WORKGROUP (four rays). Replaces k_gen_scan (a thread per ray: lanes a whole ray apart in memory, 2.4 ms per frame of the
// probe) + k_gen_compact (one atomic per wave of samples on ONE word: 550 K of them, 2.9 ms). The products / sums are formed in scan order
// instead of sample by sample: equal to rounding (1e-7), like the tuned marcher's.
__global__ __launch_bounds__(256) void k_gen_scan_compact(const GenArgs a, const GenFast fa) {
    __shared__ unsigned s_cnt[4];
    __shared__ unsigned s_base;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const long long r = (long long)blockIdx.x * 4 + wid;
    const bool live = r < a.n_rays;
    const int N = a.N;
    float carry = 1.f, acc = 0.f, dep = 0.f;
    unsigned nev = 0, napp = 0;
    if (live) {
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            const bool in = i < N;
            const long long t = r * N + (in ? i : N - 1);
            const float z = a.z[t];
            const float zn = (in && i < N - 1) ? a.z[t + 1] : z;
            const float sg = in ? a.sigma[t] : 0.f;
            const bool box = in && a.T[t] != 0.f;
            const float dist = (in && i < N - 1) ? zn - z : 0.f;
            const float alpha = 1.f - expf((-sg) * (dist * a.F.dscale));
            const float f = in ? (1.f - alpha) + 1e-10f : 1.f;
            const float incl = wave_scan_mul(f, lane);
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.f;
            const float T = carry * excl;
            const float w = in ? alpha * T : 0.f;
            if (in) { a.T[t] = T; a.w[t] = w; }
            carry = carry * __shfl(incl, 63);
            acc += wave_sum(w);
            dep += wave_sum(w * z);
            nev += (unsigned)__popcll(__ballot(box));
            napp += (unsigned)__popcll(__ballot(in && w > a.F.thres));
        }
        if (lane == 0) {
            const Ray ray = load_ray(a.F, a.rays + r * a.ray_stride, a.ray_stride);
            a.acc[r] = acc;
            a.depth[r] = dep + (1.f - acc) * ray.last;
        }
    }
    if (lane == 0) s_cnt[wid] = live ? napp : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        s_base = tot ? atomicAdd(fa.count, tot) : 0u;
        if (a.stats && tot) atomicAdd(&a.stats[T2N_STAT_APPEARANCE], (unsigned long long)tot);
    }
    if (a.stats && lane == 0 && nev) atomicAdd(&a.stats[T2N_STAT_EVALUATED], (unsigned long long)nev);
    __syncthreads();
    if (!live || !napp) return;
    unsigned pos = s_base;
    for (int q = 0; q < wid; ++q) pos += s_cnt[q];
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        const bool app = i < N && a.w[r * N + i] > a.F.thres;       // (written above by this lane)
        const unsigned long long bal = __ballot(app);
        if (app) fa.list[pos + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (int)(r * N + i);
        pos += (unsigned)__popcll(bal);
    }
}
__global__ __launch_bounds__(256) void k_gen_compact(const GenArgs a, const GenFast fa) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool app = t < a.n_rays * a.N && a.w[t] > a.F.thres;
    const unsigned long long bal = __ballot(app);
    if (!bal) return;
    const int lane = threadIdx.x & 63;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(fa.count, (unsigned)__popcll(bal));
    base = __shfl(base, 0);
    if (app) fa.list[base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = (int)t;
}
// column j of the MLP input row (gen_inputs' order) for features `feat` (LDS column of sample s: stride 64) and direction dir
__device__ __forceinline__ float gen_input_col(const GenArgs& a, const float* __restrict__ feat_s, const float* dir, int j) {
    const int D = a.app_dim;
    const bool view = a.shading != T2N_SHADE_MLP_FEA_NOVIEW;
    if (j < D) return feat_s[j * 64];
    j -= D;
    if (view) { if (j < 3) return dir[j]; j -= 3; }
    const int fpe = a.shading == T2N_SHADE_MLP ? 0 : a.fea_pe;
    if (j < 2 * fpe * D) {
        const bool cs = j >= fpe * D;
        if (cs) j -= fpe * D;
        const int f = j / fpe, o = j - f * fpe;
        const float tt = feat_s[f * 64] * (float)(1 << o);
        return cs ? cosf(tt) : sinf(tt);
    }
    j -= 2 * fpe * D;
    const bool cs = j >= 3 * a.view_pe;
    if (cs) j -= 3 * a.view_pe;
    const int d = j / a.view_pe, o = j - d * a.view_pe;
    const float tt = dir[d] * (float)(1 << o);
    return cs ? cosf(tt) : sinf(tt);
}
constexpr int kGenUnitsPerWave = kGenHidMax / 4;   // 64
// wave-uniform weight reads through the CONSTANT address space: the backend then issues scalar loads (s_load_dwordx16 into SGPRs, the FMAs
// take them as scalar operands); through a plain global pointer every weight was a 64-lane vector load of one address — 2 216 of them in
// the head kernel, which made it load-issue-bound
typedef const float __attribute__((address_space(4))) * cfp_t;
__device__ __forceinline__ cfp_t as_const(const float* p) { return (cfp_t)(uintptr_t)p; }
// one dense layer for the workgroup's 64 samples: out unit v = 4 u + g (wave g) of `nout`, inputs X[j][s] (LDS, relu'd if relu_in) for
// j < nin, weights row-major [nout][nin] read wave-uniformly. acc[u] holds unit 4 u + g of sample s.
// W: rows padded to `ld` floats (a multiple of 16, zero beyond nin), 64-byte aligned
template <class XF>
__device__ __forceinline__ void gen_layer(float (&acc)[kGenUnitsPerWave], const float* __restrict__ W, const float* __restrict__ b, int nin, int ld, int nout,
                                          int g, XF&& xchunk) {
#pragma unroll
    for (int u = 0; u < kGenUnitsPerWave; ++u) { const int v = 4 * u + g; acc[u] = v < nout ? b[v] : 0.f; }
    for (int j0 = 0; j0 < nin; j0 += 16) {
        float x[16];
        xchunk(j0, x);                                   // x[e] = input j0 + e of this lane's sample (0 beyond nin)
#pragma unroll
        for (int u = 0; u < kGenUnitsPerWave; ++u) {
            const int v = 4 * u + g;
            if (v < nout) {                              // (wave-uniform)
                const cfp_t wr = as_const((const float*)__builtin_assume_aligned(W + (size_t)v * ld + j0, 64));
                float s_ = acc[u];
#pragma unroll
                for (int e = 0; e < 16; ++e) s_ = fmaf(wr[e], x[e], s_);
                acc[u] = s_;
            }
        }
    }
}
// ROWS: the instantiation that only writes the MLP input rows (the layers run in k_dense): without the VALU layers' 64 accumulators it
// needs a third of the registers (the one kernel for both took 256 VGPRs: one wave per SIMD, nothing to hide the gather's latency behind)
template <bool ROWS, int DMAX>
__device__ __forceinline__ void gen_head_body(const GenArgs& a, const GenFast& fa) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = a.app_dim, ncol = fa.ncol, fC = a.fC;
    float* __restrict__ basisT = sm;                                  // [ncol][D]
    float* __restrict__ feat = basisT + (size_t)ncol * D;             // [D][64]
    float* __restrict__ Xc = feat + D * 64;                           // [16][64] input chunk
    float* __restrict__ Hb = Xc + 16 * 64;                            // [max(4 D, fC)][64]: feature partials, then activations
    const int tid = threadIdx.x, s = tid & 63, g = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned count = *fa.count;
    unsigned ntiles = (count + 63u) / 64u, tile0 = 0;
    if constexpr (ROWS) {   // one pass of the list
        tile0 = fa.row0 / 64u;
        const unsigned te = (fa.row0 + fa.rows_cap) / 64u;
        ntiles = ntiles < te ? ntiles : te;
    }
    if (tile0 + blockIdx.x >= ntiles) return;     // (a pass beyond the count: nothing staged)
    for (int i = tid; i < ncol * D; i += 256) { const int col = i / D, f = i - col * D; basisT[i] = a.basis[(size_t)f * ncol + col]; }
    __syncthreads();
    const bool mlp = a.shading == T2N_SHADE_MLP_FEA_NOVIEW || a.shading == T2N_SHADE_MLP_FEA || a.shading == T2N_SHADE_MLP;
    for (unsigned tile = tile0 + blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned e = tile * 64u + (unsigned)s;
        const bool live = e < count;
        const long long t = fa.list[live ? e : tile * 64u];
        const long long r = t / a.N;
        const int i = (int)(t - r * a.N);
        const float* __restrict__ rp = a.rays + r * a.ray_stride;
        const Ray ray = load_ray(a.F, rp, a.ray_stride);
        const float u = a.train ? a.jitter[r] : 0.f;
        float xn, yn, zn;
        gen_point(a, ray, gen_z(a, ray, i, u), xn, yn, zn);
        const Tap3 tp = gen_taps(a, xn, yn, zn);
        const float dir[3] = {rp[3], rp[4], rp[5]};
        // ---- appearance features: wave g takes a quarter of every plane's components ---------------------------------------------------
        {
            float fp[DMAX];
#pragma unroll
            for (int f = 0; f < DMAX; ++f) fp[f] = 0.f;
            int col0 = 0;
            for (int k = 0; k < 3; ++k) {
                const int W = a.grid[mat0(k)], C = a.Ca[k];
                const Axis& ax = tp.a[mat0(k)]; const Axis& ay = tp.a[mat1(k)]; const Axis& al = tp.a[vecm(k)];
                const float* __restrict__ p00 = fa.ap[k] + ((size_t)ay.i0 * W + ax.i0) * C;
                const float* __restrict__ p01 = fa.ap[k] + ((size_t)ay.i0 * W + ax.i1) * C;
                const float* __restrict__ p10 = fa.ap[k] + ((size_t)ay.i1 * W + ax.i0) * C;
                const float* __restrict__ p11 = fa.ap[k] + ((size_t)ay.i1 * W + ax.i1) * C;
                const float* __restrict__ l0 = fa.al[k] + (size_t)al.i0 * C;
                const float* __restrict__ l1 = fa.al[k] + (size_t)al.i1 * C;
                const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
                auto one = [&](int c, float a00, float a01, float a10, float a11, float b0, float b1) {
                    float v = a00 * w00;
                    v = fmaf(a01, w01, v); v = fmaf(a10, w10, v); v = fmaf(a11, w11, v);
                    const float xv = v * fmaf(b1, al.w1, b0 * al.w0);
                    const float* __restrict__ bt = basisT + (size_t)(col0 + c) * D;
#pragma unroll
                    for (int f = 0; f < DMAX; ++f) if (f < D) fp[f] = fmaf(bt[f], xv, fp[f]);
                };
                if ((C & 3) == 0) {      // four components per load (the staged rows are 16-byte aligned when C is a multiple of 4)
                    const int ng4 = C / 4, per = (ng4 + 3) / 4, qb = g * per, qe = min(ng4, qb + per);
                    for (int q = qb; q < qe; ++q) {
                        const float4 a00 = reinterpret_cast<const float4*>(p00)[q], a01 = reinterpret_cast<const float4*>(p01)[q];
                        const float4 a10 = reinterpret_cast<const float4*>(p10)[q], a11 = reinterpret_cast<const float4*>(p11)[q];
                        const float4 b0 = reinterpret_cast<const float4*>(l0)[q], b1 = reinterpret_cast<const float4*>(l1)[q];
                        one(4 * q, a00.x, a01.x, a10.x, a11.x, b0.x, b1.x); one(4 * q + 1, a00.y, a01.y, a10.y, a11.y, b0.y, b1.y);
                        one(4 * q + 2, a00.z, a01.z, a10.z, a11.z, b0.z, b1.z); one(4 * q + 3, a00.w, a01.w, a10.w, a11.w, b0.w, b1.w);
                    }
                } else {
                    const int per = (C + 3) / 4, cb = g * per, ce = min(C, cb + per);
                    for (int c = cb; c < ce; ++c) one(c, p00[c], p01[c], p10[c], p11[c], l0[c], l1[c]);
                }
                col0 += C;
            }
#pragma unroll
            for (int f = 0; f < DMAX; ++f) if (f < D) Hb[(g * D + f) * 64 + s] = fp[f];
        }
        __syncthreads();
        for (int f = g; f < D; f += 4) feat[f * 64 + s] = (Hb[f * 64 + s] + Hb[(D + f) * 64 + s]) + (Hb[(2 * D + f) * 64 + s] + Hb[(3 * D + f) * 64 + s]);
        __syncthreads();
        float c3[3] = {0.f, 0.f, 0.f};
        if (!mlp) {
            if (g == 0 && live) {
                if (a.shading == T2N_SHADE_RGB) { c3[0] = feat[s]; c3[1] = feat[64 + s]; c3[2] = feat[128 + s]; }
                else {
                    float sh[9];
                    gen_sh9(dir, sh);
                    for (int c = 0; c < 3; ++c) {
                        float q = 0.f;
                        for (int b = 0; b < 9; ++b) q = fmaf(sh[b], feat[(c * 9 + b) * 64 + s], q);
                        c3[c] = fmaxf(q + 0.5f, 0.f);
                    }
                }
                a.rgb_s[t * 3] = c3[0]; a.rgb_s[t * 3 + 1] = c3[1]; a.rgb_s[t * 3 + 2] = c3[2];
            }
            __syncthreads();
            continue;
        }
        const int nin = a.in0;
        if constexpr (ROWS) {
            // ---- the input rows of this tile to memory. Columns are built in rounds of `hbc` of them in Hb (free once the feature partials
            // are reduced): the plain columns one by one; the features' encodings per FEATURE — one sincosf, then angle doubling per octave
            // (sin 2x = 2 s c, cos 2x = c^2 - s^2: <= 5 steps, error ~3e-6) instead of a sinf or cosf per column, which was most of this
            // kernel's time — then 16 columns at a time to memory: thread -> (row tid / 4, four columns), 64-byte runs per row
            const int fpe = a.shading == T2N_SHADE_MLP ? 0 : a.fea_pe;
            const int pe0 = D + (a.shading != T2N_SHADE_MLP_FEA_NOVIEW ? 3 : 0), pe1 = pe0 + 2 * fpe * D;
            const int hbc = fa.hbc;
            float* __restrict__ xr = fa.x0 + (size_t)(tile * 64u - fa.row0 + (unsigned)(tid >> 2)) * fa.ldx + (tid & 3) * 4;
            const int rr = tid >> 2, cc = (tid & 3) * 4;
            for (int c0 = 0; c0 < fa.ldx; c0 += hbc) {
                const int c1 = min(fa.ldx, c0 + hbc);
                __syncthreads();                                        // the previous round has been written out
                for (int j = c0 + g; j < c1; j += 4)
                    if (j < pe0 || j >= pe1) Hb[(j - c0) * 64 + s] = j < nin ? gen_input_col(a, feat + s, dir, j) : 0.f;
                for (int f = g; f < D; f += 4) {
                    const int js = pe0 + f * fpe, jc = js + fpe * D;
                    if ((js >= c1 || js + fpe <= c0) && (jc >= c1 || jc + fpe <= c0)) continue;
                    float sv, cv;
                    sincosf(feat[f * 64 + s], &sv, &cv);
                    for (int o = 0; o < fpe; ++o) {
                        if (js + o >= c0 && js + o < c1) Hb[(js + o - c0) * 64 + s] = sv;
                        if (jc + o >= c0 && jc + o < c1) Hb[(jc + o - c0) * 64 + s] = cv;
                        const float s2 = 2.f * sv * cv, c2 = cv * cv - sv * sv;
                        sv = s2; cv = c2;
                    }
                }
                __syncthreads();
                for (int j0 = c0; j0 < c1; j0 += 16) {
                    const float* __restrict__ h = Hb + (size_t)(j0 - c0 + cc) * 64 + rr;
                    *reinterpret_cast<float4*>(xr + j0) = make_float4(h[0], h[64], h[128], h[192]);
                }
            }
            __syncthreads();
            continue;
        } else {
        // ---- layer 0: the input row in chunks of 16 columns (each wave makes 4 of them, LDS), all waves multiply -------------------------
        float acc[kGenUnitsPerWave];
        gen_layer(acc, fa.w0p, a.b0, nin, fa.ld0, fC, g, [&](int j0, float (&x)[16]) {
            __syncthreads();                                            // the previous chunk has been read
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int j = j0 + 4 * g + q; Xc[(4 * g + q) * 64 + s] = j < nin ? gen_input_col(a, feat + s, dir, j) : 0.f; }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = Xc[e * 64 + s];
        });
        __syncthreads();
#pragma unroll
        for (int uu = 0; uu < kGenUnitsPerWave; ++uu) { const int v = 4 * uu + g; if (v < fC) Hb[v * 64 + s] = fmaxf(acc[uu], 0.f); }
        __syncthreads();
        // ---- layer 1 ------------------------------------------------------------------------------------------------------------------
        gen_layer(acc, fa.w1p, a.b1, fC, fa.ld1, fC, g, [&](int j0, float (&x)[16]) {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = j0 + e < fC ? Hb[(j0 + e) * 64 + s] : 0.f;
        });
        __syncthreads();                                                // every wave has read h0
#pragma unroll
        for (int uu = 0; uu < kGenUnitsPerWave; ++uu) { const int v = 4 * uu + g; if (v < fC) Hb[v * 64 + s] = fmaxf(acc[uu], 0.f); }
        __syncthreads();
        // ---- layer 2 + sigmoid (wave 0) -----------------------------------------------------------------------------------------------
        if (g == 0 && live) {
            float o[3] = {a.b2[0], a.b2[1], a.b2[2]};
            const cfp_t w2 = as_const(a.w2);
            for (int v = 0; v < fC; ++v) {
                const float hv = Hb[v * 64 + s];
                o[0] = fmaf(w2[v], hv, o[0]); o[1] = fmaf(w2[fC + v], hv, o[1]); o[2] = fmaf(w2[2 * fC + v], hv, o[2]);
            }
            for (int c = 0; c < 3; ++c) a.rgb_s[t * 3 + c] = 1.f / (1.f + expf(-o[c]));
        }
        __syncthreads();
        }   // !ROWS
    }
}

__global__ __launch_bounds__(256) void k_gen_head(const GenArgs a, const GenFast fa) { gen_head_body<false, kGenDimMax>(a, fa); }
// (three waves per SIMD: <= 168 VGPRs; DMAX = 32 for app_dim <= 32 keeps half the feature accumulators out of the register file)
template <int DMAX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 4))) void k_gen_head_rows(const GenArgs a, const GenFast fa) { gen_head_body<true, DMAX>(a, fa); }
__global__ void k_gen_plan(const unsigned* __restrict__ count, HeadPlanDev* __restrict__ plan) { plan->rows = *count; }
// layer 2 + sigmoid from h1 [rows_cap][ldh] of one pass: four lanes per row, a quarter of the units each
__global__ __launch_bounds__(256) void k_gen_out(const GenArgs a, const GenFast fa, const float* __restrict__ h1, int ldh) {
    const unsigned count = *fa.count;
    if (fa.row0 >= count) return;
    const int fC = a.fC, per = (fC + 3) / 4;
    for (long long t4 = (long long)blockIdx.x * blockDim.x + threadIdx.x; t4 < (long long)fa.rows_cap * 4; t4 += (long long)gridDim.x * blockDim.x) {
    const unsigned lrow = (unsigned)(t4 >> 2);
    const int p = (int)(t4 & 3);
    const unsigned e = fa.row0 + lrow;
    const bool in = e < count;
    const int v0 = p * per, v1 = min(fC, v0 + per);
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (in) {
        const float* __restrict__ hr = h1 + (size_t)lrow * ldh;
        if ((fC & 15) == 0) {     // a quarter of the row is whole float4s
            for (int v = v0; v < v1; v += 4) {
                const float4 hv = *reinterpret_cast<const float4*>(hr + v);
                const float4 w0 = *reinterpret_cast<const float4*>(a.w2 + v), w1 = *reinterpret_cast<const float4*>(a.w2 + fC + v), w2 = *reinterpret_cast<const float4*>(a.w2 + 2 * fC + v);
                o0 = fmaf(w0.x, hv.x, o0); o0 = fmaf(w0.y, hv.y, o0); o0 = fmaf(w0.z, hv.z, o0); o0 = fmaf(w0.w, hv.w, o0);
                o1 = fmaf(w1.x, hv.x, o1); o1 = fmaf(w1.y, hv.y, o1); o1 = fmaf(w1.z, hv.z, o1); o1 = fmaf(w1.w, hv.w, o1);
                o2 = fmaf(w2.x, hv.x, o2); o2 = fmaf(w2.y, hv.y, o2); o2 = fmaf(w2.z, hv.z, o2); o2 = fmaf(w2.w, hv.w, o2);
            }
        } else {
            for (int v = v0; v < v1; ++v) {
                const float hv = hr[v];
                o0 = fmaf(a.w2[v], hv, o0); o1 = fmaf(a.w2[fC + v], hv, o1); o2 = fmaf(a.w2[2 * fC + v], hv, o2);
            }
        }
    }
    o0 += dpp_quad_xor1(o0); o0 += dpp_quad_xor2(o0);
    o1 += dpp_quad_xor1(o1); o1 += dpp_quad_xor2(o1);
    o2 += dpp_quad_xor1(o2); o2 += dpp_quad_xor2(o2);
    if (in && p == 0) {
        const long long t = fa.list[e];
        a.rgb_s[t * 3] = 1.f / (1.f + expf(-(o0 + a.b2[0])));
        a.rgb_s[t * 3 + 1] = 1.f / (1.f + expf(-(o1 + a.b2[1])));
        a.rgb_s[t * 3 + 2] = 1.f / (1.f + expf(-(o2 + a.b2[2])));
    }
    }
}

static int gen_fill(GenArgs& a, const t2n_generic_desc* d, const t2n_field_params* p, const char* who) {
    if (!d || !p) { set_error("%s: NULL argument", who); return T2N_ERR_INVALID; }
    memset(&a, 0, sizeof(a));
    FieldDev& F = a.F;
    for (int k = 0; k < 3; ++k) {
        F.aabb0[k] = d->aabb_min[k]; F.aabb1[k] = d->aabb_max[k]; F.inv[k] = d->inv_aabb_size[k];
        a.grid[k] = d->grid[k]; a.Cd[k] = d->density_n_comp[k]; a.Ca[k] = d->app_n_comp[k];
        if (d->grid[k] < 2 || a.Cd[k] < 1 || a.Ca[k] < 1) { set_error("%s: bad grid / component counts", who); return T2N_ERR_INVALID; }
        a.dp[k] = p->density_plane[k]; a.dl[k] = p->density_line[k]; a.ap[k] = p->app_plane[k]; a.al[k] = p->app_line[k];
        if (!a.dp[k] || !a.dl[k] || !a.ap[k] || !a.al[k]) { set_error("%s: NULL factor tensor", who); return T2N_ERR_INVALID; }
    }
    F.shift = d->density_shift; F.dscale = d->distance_scale; F.thres = d->weight_thres; F.step = d->step_size;
    F.near = d->near; F.far = d->far; F.zgate = d->z_gate; F.act = d->act; F.shading = d->shading; F.app_dim = d->app_dim;
    a.app_dim = d->app_dim; a.shading = d->shading; a.fea_pe = d->fea_pe; a.view_pe = d->view_pe; a.fC = d->feature_c;
    a.basis = p->basis_weight; a.w0 = p->mlp_w0; a.b0 = p->mlp_b0; a.w1 = p->mlp_w1; a.b1 = p->mlp_b1; a.w2 = p->mlp_w2; a.b2 = p->mlp_b2;
    if (!a.basis) { set_error("%s: NULL basis_weight", who); return T2N_ERR_INVALID; }
    if (a.app_dim < 1 || a.app_dim > kGenDimMax) { set_error("%s: app_dim %d outside [1, %d]", who, a.app_dim, kGenDimMax); return T2N_ERR_UNSUPPORTED; }
    const bool mlp = d->shading == T2N_SHADE_MLP_FEA_NOVIEW || d->shading == T2N_SHADE_MLP_FEA || d->shading == T2N_SHADE_MLP;
    if (d->shading == T2N_SHADE_SH && a.app_dim != 27) { set_error("%s: SH head needs app_dim 27", who); return T2N_ERR_UNSUPPORTED; }
    if (d->shading == T2N_SHADE_RGB && a.app_dim != 3) { set_error("%s: RGB head needs app_dim 3", who); return T2N_ERR_UNSUPPORTED; }
    if (!mlp && d->shading != T2N_SHADE_SH && d->shading != T2N_SHADE_RGB) { set_error("%s: shading head %d", who, d->shading); return T2N_ERR_UNSUPPORTED; }
    if (mlp) {
        if (!a.w0 || !a.b0 || !a.w1 || !a.b1 || !a.w2 || !a.b2) { set_error("%s: MLP head needs all six renderModule tensors", who); return T2N_ERR_INVALID; }
        const bool view = d->shading != T2N_SHADE_MLP_FEA_NOVIEW;
        const int fpe = d->shading == T2N_SHADE_MLP ? 0 : d->fea_pe;
        a.in0 = a.app_dim * (1 + 2 * fpe) + (view ? 3 + 6 * d->view_pe : 0);
        if (a.fC < 1 || a.fC > kGenHidMax || a.in0 > kGenInMax || fpe < 0 || fpe > 16 || d->view_pe < 0 || d->view_pe > 16) {
            set_error("%s: head shape featureC %d / %d inputs beyond %d / %d", who, a.fC, a.in0, kGenHidMax, kGenInMax);
            return T2N_ERR_UNSUPPORTED;
        }
    }
    return T2N_OK;
}

struct GenCarve { size_t w, z, sigma, T, rgb_s, acc, raw, go, total; };
static GenCarve gen_carve(int64_t R, int N, bool own_wz) {
    GenCarve c;
    size_t o = 0;
    const size_t rn = (size_t)R * N * 4;
    auto take = [&](size_t b) { const size_t at = o; o = (o + b + 255) / 256 * 256; return at; };
    c.w = own_wz ? take(rn) : 0; c.z = own_wz ? take(rn) : 0;
    c.sigma = take(rn); c.T = take(rn); c.rgb_s = take(rn * 3); c.acc = take((size_t)R * 4); c.raw = take((size_t)R * 12); c.go = take(rn * 3);
    c.total = o;
    return c;
}

// staged (channel-last) factor copies + the appearance list behind the plain carve
struct GenStageCarve { size_t t[12], list, count, w0p, w1p, plan, x0, h0, h1, total; unsigned rows_cap; int ldx, ldh; };
constexpr long long kGenPassRows = 262144;   // list entries per pass of the matrix-core head (x0 + h0 + h1: 3.4 KB per row at 351 / 256 / 256)
static GenStageCarve gen_stage_carve(const t2n_generic_desc* d, int64_t R, int N, size_t base) {
    GenStageCarve c;
    size_t o = (base + 255) / 256 * 256;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const size_t HW = (size_t)d->grid[mat1(k)] * d->grid[mat0(k)], L = (size_t)d->grid[vecm(k)];
            const size_t C = q < 2 ? (size_t)d->density_n_comp[k] : (size_t)d->app_n_comp[k];
            c.t[q * 3 + k] = o;
            o = (o + ((q & 1) ? L : HW) * C * 4 + 255) / 256 * 256;
        }
    c.list = o; o = (o + (size_t)R * N * 4 + 255) / 256 * 256;
    c.count = o; o += 256;
    // MLP weights, rows padded to 16 floats (in0 <= kGenInMax inputs; sized for the worst case of the limits so that the carve needs no head shape)
    const size_t fc = (size_t)(d->feature_c > 0 ? d->feature_c : 1);
    const int fpe = d->shading == T2N_SHADE_MLP ? 0 : d->fea_pe;
    const size_t in0 = (size_t)d->app_dim * (1 + 2 * (fpe > 0 ? fpe : 0)) + (d->shading != T2N_SHADE_MLP_FEA_NOVIEW ? 3 + 6 * (size_t)(d->view_pe > 0 ? d->view_pe : 0) : 0);
    c.w0p = o; o = (o + fc * ((in0 + 15) / 16 * 16) * 4 + 255) / 256 * 256;
    c.w1p = o; o = (o + fc * ((fc + 15) / 16 * 16) * 4 + 255) / 256 * 256;
    // matrix-core head (MLP heads with <= 512 inputs): the device-side row count and one pass's activation rows
    const long long tot = (long long)R * N;
    c.rows_cap = (unsigned)(((tot < kGenPassRows ? tot : kGenPassRows) + 63) / 64 * 64);
    c.ldx = (int)((in0 + 15) / 16 * 16); c.ldh = (int)((fc + 3) / 4 * 4);
    const bool mlp = d->shading == T2N_SHADE_MLP_FEA_NOVIEW || d->shading == T2N_SHADE_MLP_FEA || d->shading == T2N_SHADE_MLP;
    c.plan = o; o += 256;
    c.x0 = c.h0 = c.h1 = 0;
    if (mlp && in0 <= 512 && fc <= 512) {
        c.x0 = o; o = (o + (size_t)c.rows_cap * c.ldx * 4 + 255) / 256 * 256;
        c.h0 = o; o = (o + (size_t)c.rows_cap * c.ldh * 4 + 255) / 256 * 256;
        c.h1 = o; o = (o + (size_t)c.rows_cap * c.ldh * 4 + 255) / 256 * 256;
    }
    c.total = o;
    return c;
}
// rows_only: the kernel writes the input rows and leaves the layers to k_dense — Hb then holds the feature partials and the staging
// rounds only (4 D columns rounded up to 16), which lets two workgroups share a CU's LDS (one wave per SIMD hid none of the gather's latency)
static size_t gen_head_lds(const t2n_generic_desc* d, bool rows_only = false) {
    const size_t ncol = (size_t)d->app_n_comp[0] + d->app_n_comp[1] + d->app_n_comp[2], D = (size_t)d->app_dim;
    const size_t hb = rows_only ? (4 * D + 15) / 16 * 16 : (4 * D > (size_t)d->feature_c ? 4 * D : (size_t)d->feature_c);
    return (ncol * D + D * 64 + 16 * 64 + hb * 64) * sizeof(float);
}

}  // namespace t2n

using namespace t2n;

// with room for the channel-last staging of the factors and the appearance list: the forward then runs as the kernels of round 6
// (k_gen_stage ... k_gen_head) instead of the plain per-ray / per-sample restatement
extern "C" size_t t2n_generic_workspace_bytes_desc(const t2n_generic_desc* desc, int64_t n_rays, int n_samples) {
    if (!desc || n_rays <= 0 || n_samples <= 0) return 0;
    return gen_stage_carve(desc, n_rays, n_samples, gen_carve(n_rays, n_samples, true).total).total;
}

extern "C" size_t t2n_generic_workspace_bytes(int64_t n_rays, int n_samples) {
    if (n_rays <= 0 || n_samples <= 0) return 0;
    return gen_carve(n_rays, n_samples, true).total;
}

static int gen_bind(GenArgs& a, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags, const float* jitter,
                    float* weights, float* z_vals, void* workspace, size_t workspace_bytes, const char* who) {
    if (!rays || !workspace || n_rays <= 0 || ray_stride < 6 || n_samples < 1) { set_error("%s: bad argument", who); return T2N_ERR_INVALID; }
    if (flags & T2N_FLAG_NDC) { set_error("%s: NDC sampling is not available on the general-shape path", who); return T2N_ERR_UNSUPPORTED; }
    if ((flags & T2N_FLAG_TRAIN) && !jitter) { set_error("%s: train mode needs the jitter draws", who); return T2N_ERR_INVALID; }
    if ((uint64_t)n_rays * (uint64_t)n_samples > 0x7fffffffull) { set_error("%s: more than 2^31 samples per call (chunk the rays)", who); return T2N_ERR_UNSUPPORTED; }
    const GenCarve c = gen_carve(n_rays, n_samples, true);
    if (c.total > workspace_bytes) { set_error("%s: workspace %zu B < %zu B", who, workspace_bytes, c.total); return T2N_ERR_WORKSPACE; }
    char* ws = (char*)workspace;
    a.rays = rays; a.n_rays = n_rays; a.ray_stride = ray_stride; a.N = n_samples;
    a.train = (flags & T2N_FLAG_TRAIN) ? 1 : 0; a.add_bg = (flags & T2N_FLAG_ADD_BG) ? 1 : 0; a.jitter = jitter;
    a.w = weights ? weights : (float*)(ws + c.w); a.z = z_vals ? z_vals : (float*)(ws + c.z);
    a.sigma = (float*)(ws + c.sigma); a.T = (float*)(ws + c.T); a.rgb_s = (float*)(ws + c.rgb_s); a.acc = (float*)(ws + c.acc);
    a.raw = (float*)(ws + c.raw); a.go = (float*)(ws + c.go);
    return T2N_OK;
}

extern "C" int t2n_generic_forward(const t2n_generic_desc* desc, const t2n_field_params* params, const float* rays, int64_t n_rays,
                                   int ray_stride, int n_samples, uint32_t flags, const float* jitter, float* rgb, float* depth,
                                   float* weights, float* z_vals, uint64_t* stats, void* workspace, size_t workspace_bytes, t2n_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) { if (stats) T2N_HIP(hipMemsetAsync(stats, 0, sizeof(uint64_t) * T2N_STAT_COUNT, s)); return T2N_OK; }
    GenArgs a;
    int rc = gen_fill(a, desc, params, "t2n_generic_forward");
    if (rc) return rc;
    if (!rgb || !depth) { set_error("t2n_generic_forward: NULL output"); return T2N_ERR_INVALID; }
    if ((rc = gen_bind(a, rays, n_rays, ray_stride, n_samples, flags, jitter, weights, z_vals, workspace, workspace_bytes, "t2n_generic_forward"))) return rc;
    a.rgb = rgb; a.depth = depth; a.stats = (unsigned long long*)stats;
    if (stats) T2N_HIP(hipMemsetAsync(stats, 0, sizeof(uint64_t) * T2N_STAT_COUNT, s));
    const long long tot = (long long)n_rays * n_samples;
    const GenStageCarve sc = gen_stage_carve(desc, n_rays, n_samples, gen_carve(n_rays, n_samples, true).total);
    const size_t lds = gen_head_lds(desc);
    static const bool plain = getenv("T2N_GENERIC_PLAIN") && atoi(getenv("T2N_GENERIC_PLAIN")) != 0;
    if (!plain && sc.total <= workspace_bytes && lds <= 160 * 1024) {
        char* ws = (char*)workspace;
        GenFast fa;
        GenStageArgs sa;
        unsigned blocks = 0;
        for (int q = 0; q < 4; ++q)
            for (int k = 0; k < 3; ++k) {
                const int idx = q * 3 + k;
                const float* src[4] = {a.dp[k], a.dl[k], a.ap[k], a.al[k]};
                const long long HW = (long long)desc->grid[mat1(k)] * desc->grid[mat0(k)], L = desc->grid[vecm(k)];
                sa.src[idx] = src[q]; sa.dst[idx] = (float*)(ws + sc.t[idx]); sa.C[idx] = q < 2 ? a.Cd[k] : a.Ca[k]; sa.HW[idx] = (q & 1) ? L : HW;
                sa.block0[idx] = blocks;
                blocks += (unsigned)(((sa.HW[idx] + 63) / 64) * ((sa.C[idx] + 63) / 64));
            }
        sa.block0[12] = blocks;
        for (int k = 0; k < 3; ++k) { fa.dp[k] = sa.dst[k]; fa.dl[k] = sa.dst[3 + k]; fa.ap[k] = sa.dst[6 + k]; fa.al[k] = sa.dst[9 + k]; }
        fa.list = (int*)(ws + sc.list); fa.count = (unsigned*)(ws + sc.count); fa.cap = (unsigned)tot; fa.ncol = a.Ca[0] + a.Ca[1] + a.Ca[2];
        T2N_HIP(hipMemsetAsync(fa.count, 0, 4, s));
        hipLaunchKernelGGL(k_gen_stage, dim3(blocks), dim3(256), 0, s, sa);
        fa.w0p = fa.w1p = nullptr; fa.ld0 = fa.ld1 = 0;
        if (a.w0 && a.w1 && a.in0 > 0) {
            fa.ld0 = (a.in0 + 15) / 16 * 16; fa.ld1 = (a.fC + 15) / 16 * 16;
            fa.w0p = (const float*)(ws + sc.w0p); fa.w1p = (const float*)(ws + sc.w1p);
            hipLaunchKernelGGL(k_gen_padrows, dim3((unsigned)(((long long)a.fC * fa.ld0 + 255) / 256)), dim3(256), 0, s, a.w0, (float*)(ws + sc.w0p), a.fC, a.in0, fa.ld0);
            hipLaunchKernelGGL(k_gen_padrows, dim3((unsigned)(((long long)a.fC * fa.ld1 + 255) / 256)), dim3(256), 0, s, a.w1, (float*)(ws + sc.w1p), a.fC, a.fC, fa.ld1);
        }
        hipLaunchKernelGGL(k_gen_sigma, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a, fa);
        static const bool split_scan = getenv("T2N_GENERIC_SPLIT_SCAN") && atoi(getenv("T2N_GENERIC_SPLIT_SCAN")) != 0;   // (A/B: round 6's first form)
        if (split_scan) {
            hipLaunchKernelGGL(k_gen_scan, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, s, a);
            hipLaunchKernelGGL(k_gen_compact, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, a, fa);
        } else
            hipLaunchKernelGGL(k_gen_scan_compact, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, s, a, fa);
        static bool attr_set = false;
        if (!attr_set) { T2N_HIP(hipFuncSetAttribute((const void*)k_gen_head, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                         T2N_HIP(hipFuncSetAttribute((const void*)k_gen_head_rows<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                         T2N_HIP(hipFuncSetAttribute((const void*)k_gen_head_rows<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr_set = true; }
        const unsigned long long wt = ((unsigned long long)tot + 63) / 64;
        fa.x0 = nullptr; fa.ldx = 0; fa.row0 = 0; fa.rows_cap = 0; fa.plan = nullptr; fa.hbc = 0;
        static const bool valu_head = getenv("T2N_GENERIC_VALU_HEAD") && atoi(getenv("T2N_GENERIC_VALU_HEAD")) != 0;
        if (sc.x0 && fa.w0p && !valu_head) {
            // MLP layers 0 / 1 on the matrix cores: passes of rows_cap list entries, issued for the worst case (every sample an appearance
            // sample) and clipped to the count on the device — a pass beyond the count costs its empty launches
            fa.x0 = (float*)(ws + sc.x0); fa.ldx = sc.ldx; fa.rows_cap = sc.rows_cap; fa.plan = (HeadPlanDev*)(ws + sc.plan);
            fa.hbc = (4 * a.app_dim + 15) / 16 * 16;
            const size_t lds_rows = gen_head_lds(desc, true);
            float* h0 = (float*)(ws + sc.h0); float* h1 = (float*)(ws + sc.h1);
            hipLaunchKernelGGL(k_gen_plan, dim3(1), dim3(1), 0, s, (const unsigned*)fa.count, fa.plan);
            const unsigned long long pt = ((unsigned long long)sc.rows_cap + 63) / 64;
            for (long long row0 = 0; row0 < tot; row0 += sc.rows_cap) {
                fa.row0 = (unsigned)row0;
                if (a.app_dim <= 32) hipLaunchKernelGGL((k_gen_head_rows<32>), dim3((unsigned)(pt < 2048 ? pt : 2048)), dim3(256), lds_rows, s, a, fa);
                else hipLaunchKernelGGL((k_gen_head_rows<64>), dim3((unsigned)(pt < 2048 ? pt : 2048)), dim3(256), lds_rows, s, a, fa);
                if ((rc = launch_dense_rows(fa.x0, sc.ldx, a.w0, a.in0, a.fC, a.b0, 1, sc.rows_cap, h0, sc.ldh, s, fa.plan, row0))) return rc;
                if ((rc = launch_dense_rows(h0, sc.ldh, a.w1, a.fC, a.fC, a.b1, 1, sc.rows_cap, h1, sc.ldh, s, fa.plan, row0))) return rc;
                hipLaunchKernelGGL(k_gen_out, dim3((unsigned)min(((unsigned long long)sc.rows_cap * 4 + 255) / 256, 2048ull)), dim3(256), 0, s, a, fa, (const float*)h1, sc.ldh);
            }
        } else
            hipLaunchKernelGGL(k_gen_head, dim3((unsigned)(wt < 1024 ? (wt ? wt : 1) : 1024)), dim3(256), lds, s, a, fa);
        hipLaunchKernelGGL(k_gen_composite, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, s, a);
        T2N_HIP(hipGetLastError());
        return T2N_OK;
    }
    hipLaunchKernelGGL(k_gen_march, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_gen_shade<false>, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, s, a);
    hipLaunchKernelGGL(k_gen_composite, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_generic_backward(const t2n_generic_desc* desc, const t2n_field_params* params, const float* rays, int64_t n_rays,
                                    int ray_stride, int n_samples, uint32_t flags, const float* jitter, const float* weights,
                                    const float* z_vals, const float* d_rgb, const float* d_depth, const float* d_weights,
                                    const t2n_field_grads* g, void* workspace, size_t workspace_bytes, t2n_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n_rays == 0) return T2N_OK;
    GenArgs a;
    int rc = gen_fill(a, desc, params, "t2n_generic_backward");
    if (rc) return rc;
    if (!g || !d_rgb || !d_depth) { set_error("t2n_generic_backward: NULL argument"); return T2N_ERR_INVALID; }
    if ((rc = gen_bind(a, rays, n_rays, ray_stride, n_samples, flags, jitter, const_cast<float*>(weights), const_cast<float*>(z_vals), workspace,
                       workspace_bytes, "t2n_generic_backward"))) return rc;
    for (int k = 0; k < 3; ++k) { a.g_dp[k] = g->density_plane[k]; a.g_dl[k] = g->density_line[k]; a.g_ap[k] = g->app_plane[k]; a.g_al[k] = g->app_line[k]; }
    a.g_basis = g->basis_weight; a.g_w0 = g->mlp_w0; a.g_b0 = g->mlp_b0; a.g_w1 = g->mlp_w1; a.g_b1 = g->mlp_b1; a.g_w2 = g->mlp_w2; a.g_b2 = g->mlp_b2;
    a.d_rgb = d_rgb; a.d_depth = d_depth; a.d_w = d_weights;
    hipLaunchKernelGGL(k_gen_bwd_march, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, s, a);
    const long long tot = (long long)n_rays * n_samples;
    hipLaunchKernelGGL(k_gen_shade<true>, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
