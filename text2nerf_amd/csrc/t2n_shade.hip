// Appearance: VM-split appearance-factor gather, basis_mat, positional encoding, the 351->128->128->3 MLP (or the SH /
// RGB heads) for a compacted list of samples (K2).
//
// Replaces (reference): models/tensoRF.py:223-239 (compute_appfeature), models/tensorBase.py:11-17 (positional_encoding),
// :88-109 (MLPRender_Fea_noview), :29-33 + models/sh.py:87-112 (SHRender), :36-39 (RGBRender).
//
// Mapping (gfx950, wave64): one wave owns a tile of 32 samples; the dense contractions run on the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so results do not depend on how samples are grouped into tiles).
// Samples sit on the MFMA N axis (lane & 31), output units on M: lane (s, h = lane>>5) ends a layer holding, for
// sample s, the units u = (v&3) + 8*(v>>2) + 4*h of each 32-row block in accumulator register v. Activations go through
// the wave's private LDS tile ([unit][33], k-major, +1 pad: conflict-free) to become the next layer's B operand
// (K-step t pairs units 2t and 2t+1 on the two half-waves). The weight matrices are re-packed once per upload
// (k_pack_mlp) into that K order, one dword per lane per (K-step, M-block), so every A fetch is a coalesced 256-B read.
//
//   gather   384 (sample, channel-quad) items per plane over 64 lanes: each tap is a contiguous 192-B read;
//            plane x line products go to the LDS tile X[144][33]
//   basis    72 K-steps,  A = basisA[t][lane],  B = X[2t+h][s]            -> 27(32) features per sample -> LDS Fe[32][33]
//   layer 0  14 feature pairs x (raw, sin/cos x 6 octaves) = 182 K-steps + 1 bias step, 4 M-blocks each
//   layer 1  64 K-steps + bias, B = relu(h0) from LDS;  layer 2: 64 K-steps + bias, 1 M-block (3 live rows)
//   sigmoid, store rgb for the wave's 32 samples.
#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXld = 33;                 // padded row length of the LDS tiles
constexpr int kAppK = 144;               // 3 * 48
constexpr int kTileFloats = kAppK * kXld;  // per wave
constexpr int kPE = 6;
constexpr int kL0Pairs = 14;
constexpr int kL0Steps = kL0Pairs * (1 + 2 * kPE) + 1;   // 183
constexpr int kL1Steps = 65, kL2Steps = 65, kBasisSteps = kAppK / 2;

__host__ __device__ constexpr int unit_of(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

struct ShadeArgs {
    FieldDev F;
    const float4* app_pos; const int* app_ray; const float* rays; int ray_stride;
    const float* xyz; const float* viewdirs;      // explicit-point mode (t2n_shade_at): xyz [n,3], viewdirs [n,3] or null
    const unsigned* counters; unsigned list_cap; int nlists; unsigned count_max;   // list mode: counters[nlists]; point mode: count_max
    float4* app_rgb; float* feat_out; float* rgb_out;   // rgb_out: packed [n,3] (explicit-point mode)
    ShadeCtx ctx;   // activation rows for the backward pass (all NULL in the normal forward)
};

template <int K>
__device__ __forceinline__ void gather_plane(const FactorSet& S, float* __restrict__ X, int lane, const float4* pos_l,
                                             const float* xyz, unsigned base, unsigned count, float* ctx_x, unsigned row0) {
    constexpr int CQ = 12;   // 48 channels / 4
#pragma unroll 2
    for (int it = 0; it < 6; ++it) {
        const int item = it * 64 + lane;
        const int s = item / CQ, q = item - s * CQ;
        const unsigned idx = base + (unsigned)s;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < count) {
            float xn, yn, zn;
            if (pos_l) { const float4 p = pos_l[idx]; xn = p.x; yn = p.y; zn = p.z; }
            else { xn = xyz[(size_t)idx * 3]; yn = xyz[(size_t)idx * 3 + 1]; zn = xyz[(size_t)idx * 3 + 2]; }
            QuadTaps t;
            issue_taps<K>(S, CQ, q, xn, yn, zn, t);
            const float4 p = taps_plane(t), l = taps_line(t);
            v = make_float4(p.x * l.x, p.y * l.y, p.z * l.z, p.w * l.w);
        }
        float* dst = X + (size_t)(K * 48 + q * 4) * kXld + s;
        dst[0] = v.x; dst[kXld] = v.y; dst[2 * kXld] = v.z; dst[3 * kXld] = v.w;
        if (ctx_x) *reinterpret_cast<float4*>(ctx_x + (size_t)(row0 + s) * kAppK + K * 48 + q * 4) = v;
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256, 2) void k_shade(const ShadeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int s = lane & 31, h = lane >> 5;
    float* __restrict__ X = smem + (size_t)wid * kTileFloats;
    float* __restrict__ Fe = X;   // reused after the basis contraction
    const FieldDev& F = a.F;
    // tile enumeration over the appearance sub-lists (list l occupies [l*list_cap, l*list_cap + counters[l]))
    unsigned cnt_l = 0;
    if (lane < a.nlists) {
        cnt_l = a.counters ? a.counters[lane] : a.count_max;
        if (a.counters && cnt_l > a.list_cap) cnt_l = a.list_cap;
    }
    unsigned incl = (cnt_l + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned ntiles = __shfl(incl, a.nlists - 1);
    const unsigned wave_stride = gridDim.x * 4u;

    for (unsigned tile = blockIdx.x * 4u + wid; tile < ntiles; tile += wave_stride) {
        const int li = (int)__popcll(__ballot((lane < a.nlists) & (incl <= tile)));
        const unsigned before = li ? __shfl(incl, li - 1) : 0u;
        const unsigned lbase = (unsigned)li * a.list_cap;
        const unsigned base = lbase + (tile - before) * 32u;
        const unsigned count = lbase + __shfl(cnt_l, li);      // one past the last live entry of this sub-list
        // ---- gather: plane x line products for 32 samples x 144 channels -> X ---------------------------------------
        const unsigned row0 = tile * 32u;   // activation row of lane sample 0 (ctx mode)
        gather_plane<0>(F.app, X, lane, a.app_pos, a.xyz, base, count, a.ctx.x144, row0);
        gather_plane<1>(F.app, X, lane, a.app_pos, a.xyz, base, count, a.ctx.x144, row0);
        gather_plane<2>(F.app, X, lane, a.app_pos, a.xyz, base, count, a.ctx.x144, row0);
        wave_lds_sync();

        // ---- basis_mat: feat[i][s] = sum_k Wb[i][k] X[k][s] ----------------------------------------------------------
        f32x16 accb = {0};
        {
            const float* __restrict__ ap = F.basisA + lane;
            const float* __restrict__ bp = X + (size_t)h * kXld + s;
#pragma unroll 8
            for (int t = 0; t < kBasisSteps; ++t) accb = mfma(ap[t * 64], bp[(size_t)2 * t * kXld], accb);
        }
        wave_lds_sync();
#pragma unroll
        for (int v = 0; v < 16; ++v) Fe[unit_of(v, h) * kXld + s] = accb[v];
        wave_lds_sync();

        const unsigned idx = base + (unsigned)s;
        const bool live = idx < count;
        if (a.ctx.feat32) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(a.ctx.feat32 + (size_t)(row0 + s) * 32 + 8 * g + 4 * h) =
                    make_float4(accb[4 * g], accb[4 * g + 1], accb[4 * g + 2], accb[4 * g + 3]);
        }
        if (a.feat_out && live) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int fidx = unit_of(v, h);
                if (fidx < F.app_dim) a.feat_out[(size_t)idx * F.app_dim + fidx] = accb[v];
            }
        }

        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (F.shading == T2N_SHADE_MLP_FEA_NOVIEW) {
            // ---- layer 0: PE on the fly, 4 M-blocks -------------------------------------------------------------------
            f32x16 acc0[4] = {{0}, {0}, {0}, {0}};
            const float* __restrict__ wp = F.w0A + lane;
#pragma unroll 1
            for (int r = 0; r < kL0Pairs; ++r) {
                const float fv = Fe[(2 * r + h) * kXld + s];
                {
                    const float a0 = wp[0], a1 = wp[64], a2 = wp[128], a3 = wp[192];
                    acc0[0] = mfma(a0, fv, acc0[0]); acc0[1] = mfma(a1, fv, acc0[1]);
                    acc0[2] = mfma(a2, fv, acc0[2]); acc0[3] = mfma(a3, fv, acc0[3]);
                    wp += 256;
                }
                float scale = 1.f;
#pragma unroll 1
                for (int q = 0; q < kPE; ++q) {
                    float sn, cs;
                    sincosf(fv * scale, &sn, &cs);
                    scale *= 2.f;
                    const float a0 = wp[0], a1 = wp[64], a2 = wp[128], a3 = wp[192];
                    const float c0 = wp[256], c1 = wp[320], c2 = wp[384], c3 = wp[448];
                    acc0[0] = mfma(a0, sn, acc0[0]); acc0[1] = mfma(a1, sn, acc0[1]);
                    acc0[2] = mfma(a2, sn, acc0[2]); acc0[3] = mfma(a3, sn, acc0[3]);
                    acc0[0] = mfma(c0, cs, acc0[0]); acc0[1] = mfma(c1, cs, acc0[1]);
                    acc0[2] = mfma(c2, cs, acc0[2]); acc0[3] = mfma(c3, cs, acc0[3]);
                    wp += 512;
                }
            }
            {   // bias step: B = 1 on the low half-wave
                const float one = h == 0 ? 1.f : 0.f;
                acc0[0] = mfma(wp[0], one, acc0[0]); acc0[1] = mfma(wp[64], one, acc0[1]);
                acc0[2] = mfma(wp[128], one, acc0[2]); acc0[3] = mfma(wp[192], one, acc0[3]);
            }
            // ---- layer 1: relu(h0) through LDS, 64 K-steps --------------------------------------------------------------
            float* __restrict__ Hs = X;
            wave_lds_sync();   // Fe reads done
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
                for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc0[ms][v], 0.f);
            }
            if (a.ctx.h0) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(a.ctx.h0 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc0[ms][4 * g], 0.f), fmaxf(acc0[ms][4 * g + 1], 0.f),
                                        fmaxf(acc0[ms][4 * g + 2], 0.f), fmaxf(acc0[ms][4 * g + 3], 0.f));
            }
            wave_lds_sync();
            f32x16 acc1[4] = {{0}, {0}, {0}, {0}};
            {
                const float* __restrict__ p = F.w1A + lane;
                const float* __restrict__ bp = Hs + (size_t)h * kXld + s;
#pragma unroll 4
                for (int t = 0; t < 64; ++t) {
                    const float b = bp[(size_t)2 * t * kXld];
                    acc1[0] = mfma(p[0], b, acc1[0]); acc1[1] = mfma(p[64], b, acc1[1]);
                    acc1[2] = mfma(p[128], b, acc1[2]); acc1[3] = mfma(p[192], b, acc1[3]);
                    p += 256;
                }
                const float one = h == 0 ? 1.f : 0.f;
                acc1[0] = mfma(p[0], one, acc1[0]); acc1[1] = mfma(p[64], one, acc1[1]);
                acc1[2] = mfma(p[128], one, acc1[2]); acc1[3] = mfma(p[192], one, acc1[3]);
            }
            // ---- layer 2 (3 live rows) ---------------------------------------------------------------------------------
            wave_lds_sync();   // h0 reads done
#pragma unroll
            for (int ms = 0; ms < 4; ++ms) {
#pragma unroll
                for (int v = 0; v < 16; ++v) Hs[(ms * 32 + unit_of(v, h)) * kXld + s] = fmaxf(acc1[ms][v], 0.f);
            }
            if (a.ctx.h1) {
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<float4*>(a.ctx.h1 + (size_t)(row0 + s) * 128 + ms * 32 + 8 * g + 4 * h) =
                            make_float4(fmaxf(acc1[ms][4 * g], 0.f), fmaxf(acc1[ms][4 * g + 1], 0.f),
                                        fmaxf(acc1[ms][4 * g + 2], 0.f), fmaxf(acc1[ms][4 * g + 3], 0.f));
            }
            wave_lds_sync();
            f32x16 acc2 = {0};
            {
                const float* __restrict__ p = F.w2A + lane;
                const float* __restrict__ bp = Hs + (size_t)h * kXld + s;
#pragma unroll 8
                for (int t = 0; t < 64; ++t) acc2 = mfma(p[t * 64], bp[(size_t)2 * t * kXld], acc2);
                acc2 = mfma(p[64 * 64], h == 0 ? 1.f : 0.f, acc2);
            }
            cr = sigmoidf_(acc2[0]); cg = sigmoidf_(acc2[1]); cb = sigmoidf_(acc2[2]);   // rows 0..2 live on h == 0
        } else if (F.shading == T2N_SHADE_SH) {
            if (h == 0 && live) {
                float dx, dy, dz;
                if (a.viewdirs) { dx = a.viewdirs[(size_t)idx * 3]; dy = a.viewdirs[(size_t)idx * 3 + 1]; dz = a.viewdirs[(size_t)idx * 3 + 2]; }
                else { const float* rp = a.rays + (size_t)a.app_ray[idx] * a.ray_stride; dx = rp[3]; dy = rp[4]; dz = rp[5]; }
                // models/sh.py:4-14,87-112 (degree 2)
                const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
                const float C20 = 1.0925484305920792f, C21 = -1.0925484305920792f, C22 = 0.31539156525252005f,
                            C23 = -1.0925484305920792f, C24 = 0.5462742152960396f;
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz, xy = dx * dy, yz = dy * dz, xz = dx * dz;
                float sh[9];
                sh[0] = C0; sh[1] = -C1 * dy; sh[2] = C1 * dz; sh[3] = -C1 * dx;
                sh[4] = C20 * xy; sh[5] = C21 * yz; sh[6] = C22 * (2.0f * zz - xx - yy); sh[7] = C23 * xz; sh[8] = C24 * (xx - yy);
                float o[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.f;
#pragma unroll
                    for (int b = 0; b < 9; ++b) acc = fmaf(sh[b], Fe[(c * 9 + b) * kXld + s], acc);
                    o[c] = fmaxf(acc + 0.5f, 0.f);
                }
                cr = o[0]; cg = o[1]; cb = o[2];
            }
        } else {   // T2N_SHADE_RGB: features are the colour
            if (h == 0) { cr = Fe[0 * kXld + s]; cg = Fe[1 * kXld + s]; cb = Fe[2 * kXld + s]; }
        }
        if (h == 0 && live) {
            if (a.app_rgb) a.app_rgb[idx] = make_float4(cr, cg, cb, 0.f);
            if (a.rgb_out) { a.rgb_out[(size_t)idx * 3] = cr; a.rgb_out[(size_t)idx * 3 + 1] = cg; a.rgb_out[(size_t)idx * 3 + 2] = cb; }
        }
        wave_lds_sync();   // Fe reads done before the next tile's gather overwrites X
    }
}

// ---- parameter packing ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* basis; const float* w0; const float* b0; const float* w1; const float* b1; const float* w2; const float* b2;
    float* basisA; float* w0A; float* w1A; float* w2A;
    int app_dim, has_mlp;
};

__global__ __launch_bounds__(256) void k_pack_mlp(const PackArgs a) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = kBasisSteps * 64, n0 = kL0Steps * 256, n1 = kL1Steps * 256, n2 = kL2Steps * 64;
    if (gid < nb) {
        const int t = gid / 64, l = gid % 64, i = l & 31, h = l >> 5;
        a.basisA[gid] = i < a.app_dim ? a.basis[i * kAppK + 2 * t + h] : 0.f;
        return;
    }
    if (!a.has_mlp) return;
    int g = gid - nb;
    if (g < n0) {
        const int t = g / 256, mb = (g / 64) % 4, l = g % 64, i = l & 31, h = l >> 5;
        const int out = mb * 32 + i;
        float v;
        if (t == kL0Steps - 1) v = h == 0 ? a.b0[out] : 0.f;
        else {
            const int r = t / 13, j = t % 13;
            const int f = 2 * r + h;
            int col;
            if (j == 0) col = f;
            else { const int q = (j - 1) >> 1; col = ((j - 1) & 1) ? 27 + 27 * kPE + f * kPE + q : 27 + f * kPE + q; }
            v = f < 27 ? a.w0[out * (27 + 2 * 27 * kPE) + col] : 0.f;
        }
        a.w0A[g] = v;
        return;
    }
    g -= n0;
    if (g < n1) {
        const int t = g / 256, mb = (g / 64) % 4, l = g % 64, i = l & 31, h = l >> 5;
        const int out = mb * 32 + i;
        float v;
        if (t == 64) v = h == 0 ? a.b1[out] : 0.f;
        else v = a.w1[out * 128 + 2 * t + h];
        a.w1A[g] = v;
        return;
    }
    g -= n1;
    if (g < n2) {
        const int t = g / 64, l = g % 64, i = l & 31, h = l >> 5;
        float v = 0.f;
        if (i < 3) {
            if (t == 64) v = h == 0 ? a.b2[i] : 0.f;
            else v = a.w2[i * 128 + 2 * t + h];
        }
        a.w2A[g] = v;
    }
}

int launch_pack_mlp(t2n_field* f, const t2n_field_params* p, hipStream_t s) {
    const size_t nb = (size_t)kBasisSteps * 64, n0 = (size_t)kL0Steps * 256, n1 = (size_t)kL1Steps * 256, n2 = (size_t)kL2Steps * 64;
    if (!f->buf_mlp) {
        T2N_HIP(hipMalloc((void**)&f->buf_mlp, (nb + n0 + n1 + n2) * sizeof(float)));
        f->dev.basisA = f->buf_mlp;
        f->dev.w0A = f->buf_mlp + nb;
        f->dev.w1A = f->buf_mlp + nb + n0;
        f->dev.w2A = f->buf_mlp + nb + n0 + n1;
    }
    PackArgs a;
    a.basis = p->basis_weight; a.w0 = p->mlp_w0; a.b0 = p->mlp_b0; a.w1 = p->mlp_w1; a.b1 = p->mlp_b1; a.w2 = p->mlp_w2; a.b2 = p->mlp_b2;
    a.basisA = f->buf_mlp; a.w0A = f->buf_mlp + nb; a.w1A = f->buf_mlp + nb + n0; a.w2A = f->buf_mlp + nb + n0 + n1;
    a.app_dim = f->desc.app_dim;
    a.has_mlp = f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW;
    const size_t total = nb + n0 + n1 + n2;
    hipLaunchKernelGGL(k_pack_mlp, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

static int shade_grid(unsigned long long count_max) {
    const unsigned long long tiles = (count_max + 31u) / 32u;
    unsigned long long blocks = (tiles + 3u) / 4u;
    const unsigned cap = 256u * 2u;   // 2 workgroups per CU (LDS: 4 x 19 KB each)
    if (blocks > cap) blocks = cap;
    if (blocks == 0) blocks = 1;
    return (int)blocks;
}

int launch_shade_list(t2n_field* f, const float4* app_pos, const int* app_ray, const float* rays, int ray_stride,
                      const unsigned* counters_dev, unsigned list_cap, float4* app_rgb, const ShadeCtx* ctx, hipStream_t s) {
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    if (ctx) a.ctx = *ctx;
    a.F = f->dev;
    a.app_pos = app_pos; a.app_ray = app_ray; a.rays = rays; a.ray_stride = ray_stride;
    a.counters = counters_dev; a.list_cap = list_cap; a.nlists = kLists; a.app_rgb = app_rgb;
    const size_t lds = (size_t)4 * kTileFloats * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        T2N_HIP(hipFuncSetAttribute((const void*)k_shade, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    timing_begin(f, T2N_K_SHADE, s);
    hipLaunchKernelGGL(k_shade, dim3(shade_grid((unsigned long long)list_cap * kLists)), dim3(256), lds, s, a);
    timing_end(f, T2N_K_SHADE, s);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

}  // namespace t2n

using namespace t2n;

extern "C" size_t t2n_shade_workspace_bytes(int64_t n) { (void)n; return 256; }

extern "C" int t2n_shade_at(const t2n_field* fc, const float* xyz_norm, const float* viewdirs, int64_t n, float* app_feat,
                            float* rgb, void* workspace, size_t workspace_bytes, t2n_stream stream) {
    (void)workspace; (void)workspace_bytes;
    t2n_field* f = const_cast<t2n_field*>(fc);
    if (!f || !xyz_norm || n < 0 || n > 0x7fffffff) { set_error("t2n_shade_at: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_shade_at: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (f->desc.shading == T2N_SHADE_SH && rgb && !viewdirs) { set_error("t2n_shade_at: SH head needs viewdirs"); return T2N_ERR_INVALID; }
    if (n == 0) return T2N_OK;
    ShadeArgs a;
    memset(&a, 0, sizeof(a));
    a.F = f->dev;
    a.xyz = xyz_norm; a.viewdirs = viewdirs; a.count_max = (unsigned)n; a.nlists = 1; a.feat_out = app_feat; a.rgb_out = rgb;
    const size_t lds = (size_t)4 * kTileFloats * sizeof(float);
    T2N_HIP(hipFuncSetAttribute((const void*)k_shade, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_shade, dim3(shade_grid((unsigned)n)), dim3(256), lds, (hipStream_t)stream, a);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
