/*
 * ORACLE (test infrastructure — NOT the product): plain-C restatement of the forward render path of Text2NeRF's TensoRF
 * VM-split renderer (eval and train sampling), fp32, one ray at a time, OpenMP over rays.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, as the checker / the timed
 * CPU port. It is independent of the PyTorch oracle (oracle_torch.py): no ATen calls, reference-layout [1,C,H,W] weights
 * read in place. Parity: PINNED against the golden vectors generated from the reference (tests/test_oracle_c.py).
 *
 * Each function cites the reference lines it follows (paths relative to the reference repository). The arithmetic of
 * F.grid_sample / F.softplus / cumprod / nn.Linear (PyTorch ATen, torch==1.13.1+cu116, requirements.txt:164) is restated
 * from the published algorithm: unnormalise ((g+1)/2)*(size-1), floor, weights by subtraction, zero padding.
 * Build: make -C oracle   (gcc -O2 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float aabb0[3], aabb1[3], inv[3];
    int grid[3];
    int density_c, app_c, app_dim, shading /*0 MLP_Fea_noview, 1 SH, 2 RGB*/, fea_pe, feature_c, act;
    float density_shift, distance_scale, weight_thres, step, near_, far_, z_gate;
    const float* density_plane[3]; const float* density_line[3];
    const float* app_plane[3]; const float* app_line[3];
    const float* basis; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
} t2n_oracle_field;

static const int MAT0[3] = {0, 0, 1}, MAT1[3] = {1, 2, 2}, VEC[3] = {2, 1, 0};   /* models/tensorBase.py:190-191 */

/* ATen grid_sampler (bilinear, zeros, align_corners=True) for one axis */
static void axis_taps(float g, int size, int* i0, int* i1, float* w0, float* w1, int* ok0, int* ok1) {
    float ix = ((g + 1.f) / 2.f) * (float)(size - 1);
    float f0 = floorf(ix);
    *w1 = ix - f0;
    *w0 = 1.f - *w1;
    *ok0 = (f0 >= 0.f) && (f0 <= (float)(size - 1));
    *ok1 = (f0 >= -1.f) && (f0 <= (float)(size - 2));
    float fc = f0 < -1.f ? -1.f : (f0 > (float)(size - 1) ? (float)(size - 1) : f0);
    int i = (int)fc;
    *i0 = i < 0 ? 0 : i;
    *i1 = i + 1 > size - 1 ? size - 1 : i + 1;
}

/* one channel of plane k (bilinear) and line k (linear) at the normalised point: models/tensoRF.py:214-217,232-235 */
static void plane_line(const float* plane, const float* line, int C, int W, int H, int L, int c, float gx, float gy, float gv,
                       float* pv, float* lv) {
    int x0, x1, y0, y1, l0, l1, ox0, ox1, oy0, oy1, ol0, ol1;
    float wx0, wx1, wy0, wy1, wl0, wl1;
    axis_taps(gx, W, &x0, &x1, &wx0, &wx1, &ox0, &ox1);
    axis_taps(gy, H, &y0, &y1, &wy0, &wy1, &oy0, &oy1);
    axis_taps(gv, L, &l0, &l1, &wl0, &wl1, &ol0, &ol1);
    const float* p = plane + (size_t)c * H * W;
    float nw = (ox0 && oy0) ? p[(size_t)y0 * W + x0] : 0.f, ne = (ox1 && oy0) ? p[(size_t)y0 * W + x1] : 0.f;
    float sw = (ox0 && oy1) ? p[(size_t)y1 * W + x0] : 0.f, se = (ox1 && oy1) ? p[(size_t)y1 * W + x1] : 0.f;
    *pv = nw * (wy0 * wx0) + ne * (wy0 * wx1) + sw * (wy1 * wx0) + se * (wy1 * wx1);
    const float* q = line + (size_t)c * L;
    *lv = (ol0 ? q[l0] : 0.f) * wl0 + (ol1 ? q[l1] : 0.f) * wl1;
    (void)C;
}

/* models/tensoRF.py:205-220 */
static float density_feature(const t2n_oracle_field* f, const float xn[3]) {
    float feat = 0.f;
    for (int k = 0; k < 3; ++k) {
        int W = f->grid[MAT0[k]], H = f->grid[MAT1[k]], L = f->grid[VEC[k]];
        float s = 0.f;
        for (int c = 0; c < f->density_c; ++c) {
            float pv, lv;
            plane_line(f->density_plane[k], f->density_line[k], f->density_c, W, H, L, c, xn[MAT0[k]], xn[MAT1[k]], xn[VEC[k]], &pv, &lv);
            s += pv * lv;
        }
        feat = feat + s;
    }
    return feat;
}

/* models/tensorBase.py:406-410 */
static float feature2density(const t2n_oracle_field* f, float feat) {
    if (f->act == 1) return feat > 0.f ? feat : 0.f;
    float x = feat + f->density_shift;
    return x > 20.f ? x : log1pf(expf(x));
}

/* models/tensoRF.py:223-239 + models/tensorBase.py:11-17,29-33,88-109 */
static void shade(const t2n_oracle_field* f, const float xn[3], const float dir[3], float rgb[3]) {
    float x144[256], feat[64];
    int K = 3 * f->app_c;
    for (int k = 0; k < 3; ++k) {
        int W = f->grid[MAT0[k]], H = f->grid[MAT1[k]], L = f->grid[VEC[k]];
        for (int c = 0; c < f->app_c; ++c) {
            float pv, lv;
            plane_line(f->app_plane[k], f->app_line[k], f->app_c, W, H, L, c, xn[MAT0[k]], xn[MAT1[k]], xn[VEC[k]], &pv, &lv);
            x144[k * f->app_c + c] = pv * lv;
        }
    }
    for (int i = 0; i < f->app_dim; ++i) {
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += f->basis[(size_t)i * K + k] * x144[k];
        feat[i] = s;
    }
    if (f->shading == 2) { rgb[0] = feat[0]; rgb[1] = feat[1]; rgb[2] = feat[2]; return; }
    if (f->shading == 1) {   /* models/sh.py:4-14,87-112 (deg 2) */
        const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
        const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
        float x = dir[0], y = dir[1], z = dir[2], sh[9];
        sh[0] = C0; sh[1] = -C1 * y; sh[2] = C1 * z; sh[3] = -C1 * x;
        sh[4] = C2[0] * (x * y); sh[5] = C2[1] * (y * z); sh[6] = C2[2] * (2.0f * (z * z) - x * x - y * y);
        sh[7] = C2[3] * (x * z); sh[8] = C2[4] * (x * x - y * y);
        for (int c = 0; c < 3; ++c) {
            float s = 0.f;
            for (int b = 0; b < 9; ++b) s += sh[b] * feat[c * 9 + b];
            s += 0.5f;
            rgb[c] = s > 0.f ? s : 0.f;
        }
        return;
    }
    /* MLP_Fea_noview */
    float in[512], h0[256], h1[256];
    int D = f->app_dim, P = f->fea_pe, n = 0;
    for (int i = 0; i < D; ++i) in[n++] = feat[i];
    for (int i = 0; i < D; ++i) for (int q = 0; q < P; ++q) in[n++] = sinf(feat[i] * (float)(1 << q));
    for (int i = 0; i < D; ++i) for (int q = 0; q < P; ++q) in[n++] = cosf(feat[i] * (float)(1 << q));
    int FC = f->feature_c;
    for (int u = 0; u < FC; ++u) {
        float s = f->b0[u];
        for (int k = 0; k < n; ++k) s += f->w0[(size_t)u * n + k] * in[k];
        h0[u] = s > 0.f ? s : 0.f;
    }
    for (int u = 0; u < FC; ++u) {
        float s = f->b1[u];
        for (int k = 0; k < FC; ++k) s += f->w1[(size_t)u * FC + k] * h0[k];
        h1[u] = s > 0.f ? s : 0.f;
    }
    for (int c = 0; c < 3; ++c) {
        float s = f->b2[c];
        for (int k = 0; k < FC; ++k) s += f->w2[(size_t)c * FC + k] * h1[k];
        rgb[c] = 1.f / (1.f + expf(-s));
    }
}

/* models/tensorBase.py:436-507 for one ray (ndc_ray=False, alphaMask=None). scratch: 2*N floats. */
static void render_ray(const t2n_oracle_field* f, const float* ray, int stride, int N, int is_train, int add_bg, float u,
                       float* rgb, float* depth, float* w_out, float* z_out, float* scratch, long* n_eval, long* n_app) {
    const float* o = ray; const float* d = ray + 3;
    /* sample_ray :304-323 */
    float tmin = -INFINITY;
    for (int k = 0; k < 3; ++k) {
        float v = d[k] == 0.f ? 1e-6f : d[k];
        float ra = (f->aabb1[k] - o[k]) / v, rb = (f->aabb0[k] - o[k]) / v;
        float m = ra < rb ? ra : rb;
        if (m > tmin) tmin = m;
    }
    if (tmin < f->near_) tmin = f->near_;
    if (tmin > f->far_) tmin = f->far_;
    float* sigma = scratch; float* zs = scratch + N;
    float T = 1.f, acc = 0.f, dep = 0.f, c[3] = {0.f, 0.f, 0.f};
    for (int i = 0; i < N; ++i) {
        float rng = (float)i;
        if (is_train) rng = rng + u;
        float st = f->step * rng;
        zs[i] = tmin + st;
    }
    for (int i = 0; i < N; ++i) {
        float z = zs[i], p[3], xn[3];
        int ok = 1;
        for (int k = 0; k < 3; ++k) {
            float m = d[k] * z;
            p[k] = o[k] + m;
            if (f->aabb0[k] > p[k] || p[k] > f->aabb1[k]) ok = 0;
        }
        if (!is_train && !(p[2] > f->z_gate)) ok = 0;   /* :459-462 */
        float sg = 0.f;
        if (ok) {
            for (int k = 0; k < 3; ++k) { float s = p[k] - f->aabb0[k]; float t = s * f->inv[k]; xn[k] = t - 1.f; }
            sg = feature2density(f, density_feature(f, xn));
            (*n_eval)++;
        }
        sigma[i] = sg;
    }
    for (int i = 0; i < N; ++i) {
        float z = zs[i];
        float dist = i < N - 1 ? zs[i + 1] - z : 0.f;              /* :448 */
        float dd = dist * f->distance_scale;
        float alpha = 1.f - expf((-sigma[i]) * dd);                /* raw2alpha :19-26 */
        float w = alpha * T;
        float tt = (1.f - alpha) + 1e-10f;
        T = T * tt;
        if (w_out) w_out[i] = w;
        if (z_out) z_out[i] = z;
        acc += w;
        dep += w * z;
        if (w > f->weight_thres) {                                 /* :477,489-492 */
            float p[3], xn[3], col[3];
            for (int k = 0; k < 3; ++k) { float m = d[k] * z; p[k] = o[k] + m; float s = p[k] - f->aabb0[k]; float t = s * f->inv[k]; xn[k] = t - 1.f; }
            shade(f, xn, d, col);
            for (int k = 0; k < 3; ++k) c[k] += w * col[k];
            (*n_app)++;
        }
    }
    for (int k = 0; k < 3; ++k) {                                  /* :494-501 */
        float v = c[k];
        if (add_bg) v = v + (1.f - acc);
        rgb[k] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    }
    *depth = dep + (1.f - acc) * ray[stride - 1];                  /* :504-505 */
}

/* renderer.py:28-42 without the chunking: rays [n, stride]; jitter [n] (train) or NULL; weights/z_vals [n,N] or NULL.
 * stats[0] += evaluated samples, stats[1] += appearance samples. Returns 0. */
int t2n_oracle_render(const t2n_oracle_field* f, const float* rays, long n_rays, int stride, int n_samples, int is_train,
                      int add_bg, const float* jitter, float* rgb, float* depth, float* weights, float* z_vals, long* stats) {
    long ev = 0, ap = 0;
#pragma omp parallel reduction(+ : ev, ap)
    {
        float* scratch = (float*)malloc(sizeof(float) * 2 * (size_t)n_samples);
#pragma omp for schedule(dynamic, 64)
        for (long r = 0; r < n_rays; ++r) {
            long e = 0, a = 0;
            render_ray(f, rays + (size_t)r * stride, stride, n_samples, is_train, add_bg, jitter ? jitter[r] : 0.f, rgb + 3 * r,
                       depth + r, weights ? weights + (size_t)r * n_samples : NULL, z_vals ? z_vals + (size_t)r * n_samples : NULL,
                       scratch, &e, &a);
            ev += e; ap += a;
        }
        free(scratch);
    }
    if (stats) { stats[0] += ev; stats[1] += ap; }
    return 0;
}

int t2n_oracle_threads(void) {
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp master
        n = omp_get_num_threads();
    }
#endif
    return n;
}
