// The view-dependent rendering heads of models/tensorBase.py — MLPRender_Fea (:62-86), MLPRender_PE (:111-135), MLPRender
// (:137-159) — on a general (unfused) path: the appearance features come from the shade kernel's gather + basis stage, then
//   k_head_in        builds the MLP input row of every appearance sample in the reference's column order
//                    ([features, viewdirs, PE(features | pts), PE(viewdirs)], positional_encoding :11-17),
//   k_dense<2>       layers 0 and 1: OUT = relu(IN W^T + b) on exact-fp32 MFMA, W^T slab staged in LDS,
//   k_dense3_sigmoid layer 2 + sigmoid into the appearance list's rgb.
// The Text2NeRF driver always uses MLP_Fea_noview (fused, t2n_shade.hip); these heads exist for upstream TensoRF
// configurations and checkpoints ("support cheaply"). Backward: k_head_in_bwd folds dL/d(input row) back to
// dL/dfeatures (only the feature columns carry parameter gradients: view directions and points are inputs).
#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ constexpr int unit_of_h(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma_h(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

HeadDims head_dims(const t2n_field_desc& d) {
    HeadDims h;
    memset(&h, 0, sizeof(h));
    h.shading = d.shading; h.C = d.app_dim; h.fea_pe = d.fea_pe; h.view_pe = d.view_pe; h.pos_pe = d.pos_pe;
    int o = 0;
    h.o_feat = o; o += h.C;
    h.o_view = o; o += 3;
    if (d.shading == T2N_SHADE_MLP_FEA) { h.o_pe_a = o; h.n_pe_a = d.fea_pe > 0 ? h.C * d.fea_pe : 0; o += 2 * h.n_pe_a; }
    else if (d.shading == T2N_SHADE_MLP_PE) { h.o_pe_a = o; h.n_pe_a = d.pos_pe > 0 ? 3 * d.pos_pe : 0; o += 2 * h.n_pe_a; }
    else { h.o_pe_a = o; h.n_pe_a = 0; }
    h.o_pe_v = o; h.n_pe_v = d.view_pe > 0 ? 3 * d.view_pe : 0; o += 2 * h.n_pe_v;
    h.K0 = o;
    h.K0pad = (o + 3) & ~3;
    return h;
}

struct TilePrefixH { unsigned t[kLists + 1]; };
// The forward's device-side plan (t2n_render_forward reads no count on the host): tile prefix of the sub-lists and the row count, derived
// from the march kernel's counters by one wave; the head kernels below take their rows in PASSES of a fixed capacity — pass p covers rows
// [row0, row0 + cap) of the call, activation row r lives at scratch row r - row0 — and clip to the count on the device. The host issues
// the passes for the worst case (every sample an appearance sample); a pass beyond the count costs its empty launches.
__global__ __launch_bounds__(64) void k_head_plan(const unsigned* __restrict__ counters, unsigned list_cap, HeadPlanDev* __restrict__ plan) {
    const int lane = threadIdx.x;
    unsigned cnt = lane < kLists ? counters[lane * kCounterStride] : 0u;
    if (cnt > list_cap) cnt = list_cap;
    unsigned incl = (cnt + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane < kLists) plan->t[lane] = incl - (cnt + 31u) / 32u;
    if (lane == kLists - 1) { plan->t[kLists] = incl; plan->rows = incl * 32u; }
}

// row -> appearance-list entry (rows are 32-entry tiles, sub-list by sub-list); returns false for padding rows
__device__ __forceinline__ bool row_entry(long long row, const TilePrefixH& tp, const unsigned* counters, unsigned list_cap, unsigned& idx) {
    const unsigned tile = (unsigned)(row >> 5);
    int l = 0;
#pragma unroll
    for (int q = 1; q < kLists; ++q) l += (tp.t[q] <= tile) ? 1 : 0;
    const unsigned slot = (unsigned)(row - (long long)tp.t[l] * 32);
    unsigned cnt = counters[l * kCounterStride];
    if (cnt > list_cap) cnt = list_cap;
    idx = (unsigned)l * list_cap + slot;
    return slot < cnt;
}

struct HeadInArgs {
    HeadDims H; const float* feat32; const float4* app_pos; const int* app_ray; const float* rays; int ray_stride; int ndc;
    const unsigned* counters; unsigned list_cap; TilePrefixH tp; long long rows; float* x0;
    const HeadPlanDev* plan; long long row0;   // plan != NULL: tile prefix / row count from device memory, rows = the pass's capacity (see k_head_plan)
};
// one thread per (row, column): column c of the reference's torch.cat
__global__ __launch_bounds__(256) void k_head_in(const HeadInArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Kp = a.H.K0pad;
    const long long lrow = t / Kp;
    const int c = (int)(t - lrow * Kp);
    const long long row = a.row0 + lrow;
    if (lrow >= a.rows || (a.plan && row >= (long long)a.plan->rows)) return;
    unsigned idx;
    const bool live = a.plan ? row_entry(row, *reinterpret_cast<const TilePrefixH*>(a.plan->t), a.counters, a.list_cap, idx)
                             : row_entry(row, a.tp, a.counters, a.list_cap, idx);
    float v = 0.f;
    if (live && c < a.H.K0) {
        const HeadDims& H = a.H;
        auto view = [&](int k) {
            const float* rp = a.rays + (size_t)a.app_ray[idx] * a.ray_stride;
            float d = rp[3 + k];
            if (a.ndc) { const float n = sqrtf((rp[3] * rp[3] + rp[4] * rp[4]) + rp[5] * rp[5]); d = d / n; }
            return d;
        };
        auto base_a = [&](int k) {   // the tensor PE block A encodes: features (MLP_Fea) or normalised points (MLP_PE)
            if (H.shading == T2N_SHADE_MLP_FEA) return a.feat32[lrow * 32 + k];
            const float4 p = a.app_pos[idx];
            return k == 0 ? p.x : (k == 1 ? p.y : p.z);
        };
        if (c < H.o_view) v = a.feat32[lrow * 32 + c];
        else if (c < H.o_view + 3) v = view(c - H.o_view);
        else if (c < H.o_pe_v) {            // PE block A: [sin (k-major, octave-minor) | cos]
            const int freqs = H.shading == T2N_SHADE_MLP_FEA ? H.fea_pe : H.pos_pe;
            int e = c - H.o_pe_a;
            const bool cs = e >= H.n_pe_a;
            if (cs) e -= H.n_pe_a;
            const float arg = base_a(e / freqs) * exp2f((float)(e % freqs));
            v = cs ? cosf(arg) : sinf(arg);
        } else {                             // PE of the view directions
            int e = c - H.o_pe_v;
            const bool cs = e >= H.n_pe_v;
            if (cs) e -= H.n_pe_v;
            const float arg = view(e / H.view_pe) * exp2f((float)(e % H.view_pe));
            v = cs ? cosf(arg) : sinf(arg);
        }
    }
    a.x0[lrow * Kp + c] = v;
}

// gf [rows,32] = dL/dfeatures from gx [rows,K0pad] = dL/d(input row): the direct feature columns, plus (MLP_Fea) the chain
// through sin / cos of features * 2^q
__global__ __launch_bounds__(256) void k_head_in_bwd(const HeadDims H, const float* __restrict__ gx, const float* __restrict__ feat32,
                                                     long long rows, float* gf) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t / 32;
    const int f = (int)(t % 32);
    if (r >= rows) return;
    float g = 0.f;
    if (f < H.C) {
        const float* gr = gx + r * H.K0pad;
        g = gr[H.o_feat + f];
        if (H.shading == T2N_SHADE_MLP_FEA && H.fea_pe > 0) {
            const float v = feat32[r * 32 + f];
            float sc = 1.f;
            for (int q = 0; q < H.fea_pe; ++q) {
                const float arg = v * sc;
                g = fmaf(gr[H.o_pe_a + f * H.fea_pe + q] * sc, cosf(arg), g);
                g = fmaf(-(gr[H.o_pe_a + H.n_pe_a + f * H.fea_pe + q] * sc), sinf(arg), g);
                sc *= 2.f;
            }
        }
    }
    gf[r * 32 + f] = g;
}

// Backward of the parameter-free heads: gf [rows,32] = dL/dfeatures from go [rows] = dL/drgb of the appearance samples.
// RGBRender (models/tensorBase.py:35-38): rgb = features -> gf[c] = go[c]. SHRender (:29-33, models/sh.py:87-112, degree 2):
// rgb[c] = relu(sum_b sh_b(viewdir) feat[9 c + b] + 0.5) -> gf[9 c + b] = go[c] [rgb[c] > 0] sh_b.
struct SimpleHeadBwdArgs {
    int shading, ndc; const float4* go; const float4* app_rgb; const int* app_ray; const float* rays; int ray_stride;
    const unsigned* counters; unsigned list_cap; TilePrefixH tp; long long rows; float* gf;
};
__global__ __launch_bounds__(256) void k_simple_head_bwd(const SimpleHeadBwdArgs a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long r = t / 32;
    const int f = (int)(t % 32);
    if (r >= a.rows) return;
    unsigned idx;
    const bool live = row_entry(r, a.tp, a.counters, a.list_cap, idx);
    float g = 0.f;
    if (live) {
        const float4 go = a.go[r];
        if (a.shading == T2N_SHADE_RGB) {
            g = f == 0 ? go.x : (f == 1 ? go.y : (f == 2 ? go.z : 0.f));
        } else if (f < 27) {
            const int c = f / 9, b = f - 9 * c;
            const float4 rgb = a.app_rgb[idx];
            const float gc = c == 0 ? go.x : (c == 1 ? go.y : go.z), vc = c == 0 ? rgb.x : (c == 1 ? rgb.y : rgb.z);
            if (vc > 0.f) {
                const float* rp = a.rays + (size_t)a.app_ray[idx] * a.ray_stride;
                float dx = rp[3], dy = rp[4], dz = rp[5];
                if (a.ndc) { const float n = sqrtf((dx * dx + dy * dy) + dz * dz); dx = dx / n; dy = dy / n; dz = dz / n; }
                const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
                const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                            C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
                float sh;
                switch (b) {
                    case 0: sh = C0; break;
                    case 1: sh = -C1 * dy; break;
                    case 2: sh = C1 * dz; break;
                    case 3: sh = -C1 * dx; break;
                    case 4: sh = C20 * (dx * dy); break;
                    case 5: sh = C21 * (dy * dz); break;
                    case 6: sh = C22 * (2.0f * (dz * dz) - dx * dx - dy * dy); break;
                    case 7: sh = C23 * (dx * dz); break;
                    default: sh = C24 * (dx * dx - dy * dy); break;
                }
                g = gc * sh;
            }
        }
    }
    a.gf[r * 32 + f] = g;
}

int launch_simple_head_bwd(t2n_field* f, const unsigned tiles_before[kLists + 1], long long rows, const float4* go, const float4* app_rgb,
                           const int* app_ray, const float* rays, int ray_stride, const unsigned* counters, unsigned list_cap, float* gf,
                           hipStream_t s) {
    SimpleHeadBwdArgs a;
    a.shading = f->desc.shading; a.ndc = f->dev.ztab ? 1 : 0; a.go = go; a.app_rgb = app_rgb; a.app_ray = app_ray; a.rays = rays;
    a.ray_stride = ray_stride; a.counters = counters; a.list_cap = list_cap; a.rows = rows; a.gf = gf;
    for (int l = 0; l <= kLists; ++l) a.tp.t[l] = tiles_before[l];
    hipLaunchKernelGGL(k_simple_head_bwd, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// OUT[rows, N] = act(IN[rows, K] Wt[N, K]^T + bias): a workgroup owns one 64-column group of N; its K x 64 slab of W^T is
// staged once (row stride 65: conflict-free both ways), 4 waves walk 32-row tiles, rows on the MFMA N axis, columns on M.
// K <= 512 (LDS), ldin % 4 == 0 with finite padding columns, ldo % 4 == 0.
__global__ __launch_bounds__(256) void k_dense(const float* __restrict__ IN, int ldin, const float* __restrict__ Wt, int K, int N,
                                               const float* __restrict__ bias, int relu, long long rows, float* OUT, int ldo,
                                               const HeadPlanDev* __restrict__ plan, long long row0) {
    if (plan) {   // rows = the pass's capacity: clip to what the call really has
        const long long left = (long long)plan->rows - row0;
        rows = left < rows ? left : rows;
        if (rows <= 0) return;
    }
    extern __shared__ __attribute__((aligned(16))) float sW[];   // [K4][65]
    const int lane = threadIdx.x & 63, s = lane & 31, h = lane >> 5, w = threadIdx.x >> 6;
    const int ng = blockIdx.x;
    const int K4 = (K + 3) & ~3;
    for (int base = 0; base < K4 * 64; base += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int n = idx / K4, k = idx - n * K4;          // k fastest: coalesced along a row of Wt
            v[u] = (idx < K4 * 64 && k < K && ng * 64 + n < N) ? Wt[(size_t)(ng * 64 + n) * K + k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            const int n = idx / K4, k = idx - n * K4;
            if (idx < K4 * 64) sW[k * 65 + n] = v[u];
        }
    }
    __syncthreads();
    const long long ntiles = (rows + 31) / 32;
    for (long long tile = (long long)blockIdx.y * 4 + w; tile < ntiles; tile += (long long)gridDim.y * 4) {
        const long long r = tile * 32 + s;
        const bool rok = r < rows;
        const float* __restrict__ inr = IN + (rok ? r : 0) * ldin;
        f32x16 acc[2] = {{0}, {0}};
        auto mac4 = [&](const float4 cur, int k0) {
            const float b0 = h ? cur.y : cur.x, b1 = h ? cur.w : cur.z;
            const float* w0 = sW + (k0 + h) * 65 + s;
            const float* w1 = w0 + 2 * 65;
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = mfma_h(w0[m * 32], b0, acc[m]);
#pragma unroll
            for (int m = 0; m < 2; ++m) acc[m] = mfma_h(w1[m * 32], b1, acc[m]);
        };
        // the input row in blocks of 16 columns, the NEXT block's four loads in flight under this block's 16 MFMAs (1 024 matrix-pipe cycles):
        // with one load per four MFMAs issued where it is needed, the wave — alone on its SIMD: the W slab takes the CU's LDS — stood
        // waiting for memory two thirds of the time (0.32 of the fp32 MFMA rate on the general path's 351 -> 256 -> 256 layers)
        const int K16 = K4 & ~15;
        float4 n0, n1, n2, n3;
        if (K16 > 0) { n0 = *reinterpret_cast<const float4*>(inr); n1 = *reinterpret_cast<const float4*>(inr + 4); n2 = *reinterpret_cast<const float4*>(inr + 8); n3 = *reinterpret_cast<const float4*>(inr + 12); }
        for (int k0 = 0; k0 < K16; k0 += 16) {
            const float4 c0 = n0, c1 = n1, c2 = n2, c3 = n3;
            if (k0 + 16 < K16) {
                n0 = *reinterpret_cast<const float4*>(inr + k0 + 16); n1 = *reinterpret_cast<const float4*>(inr + k0 + 20);
                n2 = *reinterpret_cast<const float4*>(inr + k0 + 24); n3 = *reinterpret_cast<const float4*>(inr + k0 + 28);
            }
            mac4(c0, k0); mac4(c1, k0 + 4); mac4(c2, k0 + 8); mac4(c3, k0 + 12);
        }
        for (int k0 = K16; k0 < K4; k0 += 4) mac4(*reinterpret_cast<const float4*>(inr + k0), k0);
        if (!rok) continue;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = ng * 64 + m * 32 + 8 * g + 4 * h;
                if (n >= N) continue;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[m][4 * g + e] + (n + e < N ? bias[n + e] : 0.f);
                    o[e] = relu ? fmaxf(x, 0.f) : x;
                }
                *reinterpret_cast<float4*>(OUT + r * ldo + n) = make_float4(o[0], o[1], o[2], o[3]);
            }
    }
}

// layer 2 (3 outputs) + sigmoid -> app_rgb of the row's list entry; 4 lanes per row
struct Dense3Args {
    const float* h1; const float* w2; const float* b2; const unsigned* counters; unsigned list_cap; TilePrefixH tp; long long rows; float4* app_rgb;
    const float4* app_pos;   // the entry's compositing weight (.w) rides along in app_rgb.w: k_composite reads one array
    const HeadPlanDev* plan; long long row0;
};
__global__ __launch_bounds__(256) void k_dense3_sigmoid(const Dense3Args a) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long lrow = t >> 2;
    const long long row = a.row0 + lrow;
    const int p = (int)(t & 3);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    const bool in = lrow < a.rows && (!a.plan || row < (long long)a.plan->rows);
    if (in) {
        const float* hr = a.h1 + lrow * 128 + p * 32;
        for (int j = 0; j < 32; ++j) {
            const float x = hr[j];
            s0 = fmaf(x, a.w2[p * 32 + j], s0); s1 = fmaf(x, a.w2[128 + p * 32 + j], s1); s2 = fmaf(x, a.w2[256 + p * 32 + j], s2);
        }
    }
    s0 += dpp_quad_xor1(s0); s0 += dpp_quad_xor2(s0);
    s1 += dpp_quad_xor1(s1); s1 += dpp_quad_xor2(s1);
    s2 += dpp_quad_xor1(s2); s2 += dpp_quad_xor2(s2);
    if (in && p == 0) {
        unsigned idx;
        if (a.plan ? row_entry(row, *reinterpret_cast<const TilePrefixH*>(a.plan->t), a.counters, a.list_cap, idx) : row_entry(row, a.tp, a.counters, a.list_cap, idx)) {
            const float r = 1.f / (1.f + expf(-(s0 + a.b2[0]))), g = 1.f / (1.f + expf(-(s1 + a.b2[1]))), b = 1.f / (1.f + expf(-(s2 + a.b2[2])));
            a.app_rgb[idx] = make_float4(r, g, b, a.app_pos[idx].w);
        }
    }
}

static int launch_dense(const float* IN, int ldin, const float* Wt, int K, int N, const float* bias, int relu, long long rows, float* OUT,
                        int ldo, hipStream_t s, const HeadPlanDev* plan = nullptr, long long row0 = 0) {
    if (K > 512) { set_error("generic head: %d MLP inputs > 512", K); return T2N_ERR_UNSUPPORTED; }
    const int ng = (N + 63) / 64;
    const long long tiles4 = ((rows + 31) / 32 + 3) / 4;
    long long by = 256 / ng;
    if (by > tiles4) by = tiles4;
    if (by < 1) by = 1;
    const size_t lds = (size_t)((K + 3) & ~3) * 65 * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_dense, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 65 * 4); attr_set = true; }
    hipLaunchKernelGGL(k_dense, dim3((unsigned)ng, (unsigned)by), dim3(256), lds, s, IN, ldin, Wt, K, N, bias, relu, rows, OUT, ldo, plan, row0);
    return T2N_OK;
}

// (the general-shape path's MLP layers, t2n_generic.hip)
int launch_dense_rows(const float* IN, int ldin, const float* Wt, int K, int N, const float* bias, int relu, long long rows, float* OUT, int ldo,
                      hipStream_t s, const HeadPlanDev* plan, long long row0) {
    return launch_dense(IN, ldin, Wt, K, N, bias, relu, rows, OUT, ldo, s, plan, row0);
}

// features (already in feat32) -> X0 -> h0 -> h1 -> app_rgb. tiles_before[l] = 32-row tiles before sub-list l.
int launch_head_forward(t2n_field* f, const unsigned tiles_before[kLists + 1], long long rows, const float* feat32, const float4* app_pos,
                        const int* app_ray, const float* rays, int ray_stride, const unsigned* counters, unsigned list_cap, float* x0,
                        float* h0, float* h1, float4* app_rgb, hipStream_t s, const HeadPlanDev* plan, long long row0) {
    if (rows <= 0) return T2N_OK;
    const HeadDims H = head_dims(f->desc);
    const t2n_field_params* P = &f->params_ref;
    HeadInArgs a;
    a.H = H; a.feat32 = feat32; a.app_pos = app_pos; a.app_ray = app_ray; a.rays = rays; a.ray_stride = ray_stride;
    a.ndc = f->dev.ztab ? 1 : 0; a.counters = counters; a.list_cap = list_cap; a.rows = rows; a.x0 = x0; a.plan = plan; a.row0 = row0;
    for (int l = 0; l <= kLists; ++l) a.tp.t[l] = tiles_before ? tiles_before[l] : 0u;
    const long long n = rows * H.K0pad;
    hipLaunchKernelGGL(k_head_in, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a);
    int rc;
    if ((rc = launch_dense(x0, H.K0pad, P->mlp_w0, H.K0, 128, P->mlp_b0, 1, rows, h0, 128, s, plan, row0))) return rc;
    if ((rc = launch_dense(h0, 128, P->mlp_w1, 128, 128, P->mlp_b1, 1, rows, h1, 128, s, plan, row0))) return rc;
    Dense3Args d;
    d.h1 = h1; d.w2 = P->mlp_w2; d.b2 = P->mlp_b2; d.counters = counters; d.list_cap = list_cap; d.tp = a.tp; d.rows = rows; d.app_rgb = app_rgb; d.app_pos = app_pos; d.plan = plan; d.row0 = row0;
    hipLaunchKernelGGL(k_dense3_sigmoid, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, s, d);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

int launch_head_plan(const unsigned* counters, unsigned list_cap, HeadPlanDev* plan, hipStream_t s) {
    hipLaunchKernelGGL(k_head_plan, dim3(1), dim3(64), 0, s, counters, list_cap, plan);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

int launch_head_in_bwd(t2n_field* f, const float* gx, const float* feat32, long long rows, float* gf, hipStream_t s) {
    const HeadDims H = head_dims(f->desc);
    hipLaunchKernelGGL(k_head_in_bwd, dim3((unsigned)((rows * 32 + 255) / 256)), dim3(256), 0, s, H, gx, feat32, rows, gf);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n
