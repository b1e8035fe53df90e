// Device-side building blocks shared by the march / shade / backward kernels (gfx950, wave64).
// All arithmetic that decides WHICH samples exist (t_min, z, point, box test, z gate, normalisation) is written as
// separately rounded fp32 operations in the reference's order (this library is compiled with -ffp-contract=off);
// FMAs appear only where written explicitly.
#pragma once
#include "t2n_internal.h"

namespace t2n {

__device__ __forceinline__ float dpp_quad_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xf, 0xf, true));  // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_quad_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xf, 0xf, true));  // quad_perm [2,3,0,1]
}

// Sum over the LPS consecutive lanes that share one sample (LPS = channels/4).
template <int LPS>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (LPS >= 2) v += dpp_quad_xor1(v);
    if constexpr (LPS >= 4) v += dpp_quad_xor2(v);
    if constexpr (LPS >= 8) v += __shfl_xor(v, 4);
    if constexpr (LPS >= 16) v += __shfl_xor(v, 8);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Inclusive product scan over the 64 lanes of a wave (lane 0 first).
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o);
        if (lane >= o) v *= t;
    }
    return v;
}

struct Ray {
    float ox, oy, oz, dx, dy, dz, tmin, last, norm;   // norm: |d| on the NDC path (dists are scaled by it, :443-444), else 1
};

// models/tensorBase.py:308-311
__device__ __forceinline__ float ray_tmin(const FieldDev& F, float ox, float oy, float oz, float dx, float dy, float dz) {
    const float vx = dx == 0.f ? 1e-6f : dx, vy = dy == 0.f ? 1e-6f : dy, vz = dz == 0.f ? 1e-6f : dz;
    const float ax = (F.aabb1[0] - ox) / vx, bx = (F.aabb0[0] - ox) / vx;
    const float ay = (F.aabb1[1] - oy) / vy, by = (F.aabb0[1] - oy) / vy;
    const float az = (F.aabb1[2] - oz) / vz, bz = (F.aabb0[2] - oz) / vz;
    float t = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    return fminf(fmaxf(t, F.near), F.far);
}

__device__ __forceinline__ Ray load_ray(const FieldDev& F, const float* __restrict__ rp, int stride) {
    Ray r;
    r.ox = rp[0]; r.oy = rp[1]; r.oz = rp[2]; r.dx = rp[3]; r.dy = rp[4]; r.dz = rp[5];
    r.last = rp[stride - 1];
    r.tmin = ray_tmin(F, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz);
    r.norm = F.ztab ? sqrtf((r.dx * r.dx + r.dy * r.dy) + r.dz * r.dz) : 1.f;
    return r;
}

// models/tensorBase.py:313-318: z_i = t_min + stepSize * (i [+ u])
// NOTAB: the caller's launches never carry an NDC depth table (the tile marcher): no run-time test in the step loop
template <bool TRAIN, bool NOTAB = false>
__device__ __forceinline__ float sample_z(const FieldDev& F, const Ray& r, int i, float u) {
    if (!NOTAB && F.ztab) return F.ztab[i];          // NDC: torch.linspace(near, far, N) [+ shared jitter], built by the host mirror
    float rng = (float)i;
    if (TRAIN) rng = rng + u;
    const float st = F.step * rng;
    return r.tmin + st;
}
// raw2alpha's distance argument (:475): dists * distance_scale, with dists first scaled by |d| on the NDC path (:444)
__device__ __forceinline__ float scaled_dist(const FieldDev& F, const Ray& r, float dist) {
    return F.ztab ? (dist * r.norm) * F.dscale : dist * F.dscale;
}

// Point, box mask (:320-321), eval z gate (:459-462), normalisation (:245-246). Returns validity.
template <bool TRAIN>
__device__ __forceinline__ bool sample_point(const FieldDev& F, const Ray& r, float z, float& xn, float& yn, float& zn) {
    const float mx = r.dx * z, my = r.dy * z, mz = r.dz * z;
    const float px = r.ox + mx, py = r.oy + my, pz = r.oz + mz;
    bool out = (F.aabb0[0] > px) | (px > F.aabb1[0]) | (F.aabb0[1] > py) | (py > F.aabb1[1]) | (F.aabb0[2] > pz) |
               (pz > F.aabb1[2]);
    bool ok = !out;
    if (!TRAIN) ok = ok & (pz > F.zgate);
    const float sx = px - F.aabb0[0], sy = py - F.aabb0[1], sz = pz - F.aabb0[2];
    const float tx = sx * F.inv[0], ty = sy * F.inv[1], tz = sz * F.inv[2];
    xn = tx - 1.f; yn = ty - 1.f; zn = tz - 1.f;
    return ok;
}

// ATen grid_sampler_2d (bilinear, zeros, align_corners=True) restated for one axis:
// unnormalise ((g+1)/2)*(size-1), corner = floor, weights by subtraction, out-of-range taps weigh 0.
struct Axis {
    int i0, i1;     // clamped tap indices
    float w0, w1;   // weights (0 for out-of-range taps)
};
__device__ __forceinline__ Axis axis_taps(float g, int size) {
    const float h = (g + 1.f) / 2.f;
    const float ix = h * (float)(size - 1);
    const float f0 = floorf(ix);
    const float w1 = ix - f0;
    const float w0 = 1.f - w1;
    const float hi = (float)(size - 1);
    Axis a;
    const bool ok0 = (f0 >= 0.f) & (f0 <= hi);
    const bool ok1 = (f0 >= -1.f) & (f0 <= hi - 1.f);
    const float fc = fminf(fmaxf(f0, -1.f), hi);
    const int i = (int)fc;
    a.i0 = max(i, 0);
    a.i1 = min(i + 1, size - 1);
    a.w0 = ok0 ? w0 : 0.f;
    a.w1 = ok1 ? w1 : 0.f;
    return a;
}

// axis_taps for a coordinate known to lie in [-1, 1] (a sample that passed the box test): ix is in [0, size - 1], so both range
// guards hold (the high tap's weight is already 0 where it would fail: f0 == size - 1 means ix == size - 1) and the clamps reduce
// to one min. Same values, six instructions fewer per axis.
__device__ __forceinline__ Axis axis_taps_inbox(float g, int size) {
    const float h = (g + 1.f) / 2.f;
    const float ix = h * (float)(size - 1);
    const float f0 = floorf(ix);
    Axis a;
    a.w1 = ix - f0;
    a.w0 = 1.f - a.w1;
    a.i0 = (int)f0;
    a.i1 = min(a.i0 + 1, size - 1);
    return a;
}

// The clamped floor index of axis_taps (same arithmetic): low tap at cell i (weight 0 when i == -1), high tap at i + 1
// (weight 0 when i == size - 1).
__device__ __forceinline__ int axis_cell(float g, int size) {
    const float h = (g + 1.f) / 2.f;
    const float ix = h * (float)(size - 1);
    const float f0 = floorf(ix);
    return (int)fminf(fmaxf(f0, -1.f), (float)(size - 1));
}

__device__ __forceinline__ float4 f4_mul(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 f4_fma(float4 a, float s, float4 c) {
    return make_float4(fmaf(a.x, s, c.x), fmaf(a.y, s, c.y), fmaf(a.z, s, c.z), fmaf(a.w, s, c.w));
}

// One channel-quad q of plane k's bilinear value and line k's linear value at normalised point (xn,yn,zn).
// CQ = channels/4 (float4 per tap).
struct QuadTaps {
    float4 nw, ne, sw, se, l0, l1;
    float wnw, wne, wsw, wse, wl0, wl1;
};

template <int K>
__device__ __forceinline__ void plane_line_coords(const float xn, const float yn, const float zn, float& gx, float& gy,
                                                  float& gv) {
    // matMode = [[0,1],[0,2],[1,2]], vecMode = [2,1,0]
    if constexpr (K == 0) { gx = xn; gy = yn; gv = zn; }
    if constexpr (K == 1) { gx = xn; gy = zn; gv = yn; }
    if constexpr (K == 2) { gx = yn; gy = zn; gv = xn; }
}

// Tap addresses (in float4 units inside the channel-last plane / line of factor pair K) and interpolation weights.
struct TapIdx {
    unsigned nw, ne, sw, se, l0, l1;
    float wnw, wne, wsw, wse, wl0, wl1;
};

template <int K>
__device__ __forceinline__ void compute_taps(const FactorSet& S, int CQ, int q, float xn, float yn, float zn, TapIdx& o) {
    float gx, gy, gv;
    plane_line_coords<K>(xn, yn, zn, gx, gy, gv);
    const Axis ax = axis_taps(gx, S.W[K]);
    const Axis ay = axis_taps(gy, S.H[K]);
    const Axis al = axis_taps(gv, S.L[K]);
    const unsigned W = (unsigned)S.W[K];
    const unsigned r0 = (unsigned)ay.i0 * W, r1 = (unsigned)ay.i1 * W;
    o.nw = (r0 + (unsigned)ax.i0) * CQ + q;
    o.ne = (r0 + (unsigned)ax.i1) * CQ + q;
    o.sw = (r1 + (unsigned)ax.i0) * CQ + q;
    o.se = (r1 + (unsigned)ax.i1) * CQ + q;
    o.l0 = (unsigned)al.i0 * CQ + q;
    o.l1 = (unsigned)al.i1 * CQ + q;
    o.wnw = ay.w0 * ax.w0; o.wne = ay.w0 * ax.w1; o.wsw = ay.w1 * ax.w0; o.wse = ay.w1 * ax.w1;
    o.wl0 = al.w0; o.wl1 = al.w1;
}

template <int K>
__device__ __forceinline__ void issue_taps(const FactorSet& S, int CQ, int q, float xn, float yn, float zn, QuadTaps& t) {
    TapIdx o;
    compute_taps<K>(S, CQ, q, xn, yn, zn, o);
    const float4* __restrict__ P = reinterpret_cast<const float4*>(S.plane[K]);
    const float4* __restrict__ Ln = reinterpret_cast<const float4*>(S.line[K]);
    t.nw = P[o.nw]; t.ne = P[o.ne]; t.sw = P[o.sw]; t.se = P[o.se];
    t.l0 = Ln[o.l0]; t.l1 = Ln[o.l1];
    t.wnw = o.wnw; t.wne = o.wne; t.wsw = o.wsw; t.wse = o.wse; t.wl0 = o.wl0; t.wl1 = o.wl1;
}

// Axis a of the grid always pairs with coordinate a (plane k: W = grid[mat0], H = grid[mat1]; line: L = grid[vec]), so a
// sample needs only THREE distinct tap computations, shared by its three plane/line pairs.
struct Axes3 { Axis a[3]; };
__device__ __forceinline__ Axes3 sample_axes(const FactorSet& S, float xn, float yn, float zn) {
    Axes3 A;
    A.a[0] = axis_taps(xn, S.W[0]);   // grid[0] = W of planes 0 and 1, L of line 2
    A.a[1] = axis_taps(yn, S.H[0]);   // grid[1] = H of plane 0, W of plane 2, L of line 1
    A.a[2] = axis_taps(zn, S.H[1]);   // grid[2] = H of planes 1 and 2, L of line 0
    return A;
}
__device__ __forceinline__ Axes3 sample_axes_inbox(const FactorSet& S, float xn, float yn, float zn) {
    Axes3 A;
    A.a[0] = axis_taps_inbox(xn, S.W[0]);
    A.a[1] = axis_taps_inbox(yn, S.H[0]);
    A.a[2] = axis_taps_inbox(zn, S.H[1]);
    return A;
}
// four bf16 channels (8 B, channel 4q in the low half of .x) -> fp32: exact (bf16 is the upper half of an fp32)
__device__ __forceinline__ float4 bf16x4_to_f4(uint2 u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
// a[23:0] * b[23:0] + c as ONE full-rate instruction (the compiler turns the plain expression into v_mul_lo_u32 / v_mad_u64_u32, quarter rate)
__device__ __forceinline__ unsigned umad24(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// HALF: the factor set is read from its bf16 copy (8-B quads, half the bytes through the texture addresser / L1).
template <int K, bool HALF = false>
__device__ __forceinline__ void issue_taps_ax(const FactorSet& S, int CQ, int q, const Axes3& A, QuadTaps& t) {
    const Axis& ax = A.a[mat0(K)];
    const Axis& ay = A.a[mat1(K)];
    const Axis& al = A.a[vecm(K)];
    // 32-bit BYTE offsets from the (scalar) plane base: lets the loads use the SGPR-base + VGPR-offset addressing form
    const unsigned W = (unsigned)S.W[K];
    constexpr unsigned QB = HALF ? 8u : 16u;
    const unsigned tb = (unsigned)CQ * QB, qb = (unsigned)q * QB;
    // 24-bit multiplies: full-rate VALU ops (v_mul_lo_u32 is quarter rate); tap indices and widths are < 2^24
    if constexpr (HALF) {
        const unsigned r0 = __umul24((unsigned)ay.i0, W), r1 = __umul24((unsigned)ay.i1, W);
        const char* __restrict__ P = reinterpret_cast<const char*>(S.plane_h[K]);
        const char* __restrict__ Ln = reinterpret_cast<const char*>(S.line_h[K]);
        const uint2 nw = *reinterpret_cast<const uint2*>(P + ((r0 + (unsigned)ax.i0) * tb + qb));
        const uint2 ne = *reinterpret_cast<const uint2*>(P + ((r0 + (unsigned)ax.i1) * tb + qb));
        const uint2 sw = *reinterpret_cast<const uint2*>(P + ((r1 + (unsigned)ax.i0) * tb + qb));
        const uint2 se = *reinterpret_cast<const uint2*>(P + ((r1 + (unsigned)ax.i1) * tb + qb));
        const uint2 l0 = *reinterpret_cast<const uint2*>(Ln + ((unsigned)al.i0 * tb + qb));
        const uint2 l1 = *reinterpret_cast<const uint2*>(Ln + ((unsigned)al.i1 * tb + qb));
        t.nw = bf16x4_to_f4(nw); t.ne = bf16x4_to_f4(ne); t.sw = bf16x4_to_f4(sw); t.se = bf16x4_to_f4(se);
        t.l0 = bf16x4_to_f4(l0); t.l1 = bf16x4_to_f4(l1);
    } else {
        const char* __restrict__ P = reinterpret_cast<const char*>(S.plane[K]);
        const char* __restrict__ Ln = reinterpret_cast<const char*>(S.line[K]);
        // (texel index < 2^24 — grids up to 4096^2 — and texel bytes < 2^24: v_mad_u32_u24, a full-rate op; the plain 32-bit multiply-add
        // compiles to v_mad_u64_u32)
        t.nw = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i0, W, (unsigned)ax.i0), tb, qb));
        t.ne = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i0, W, (unsigned)ax.i1), tb, qb));
        t.sw = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i1, W, (unsigned)ax.i0), tb, qb));
        t.se = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i1, W, (unsigned)ax.i1), tb, qb));
        t.l0 = *reinterpret_cast<const float4*>(Ln + umad24((unsigned)al.i0, tb, qb));
        t.l1 = *reinterpret_cast<const float4*>(Ln + umad24((unsigned)al.i1, tb, qb));
    }
    t.wnw = ay.w0 * ax.w0; t.wne = ay.w0 * ax.w1; t.wsw = ay.w1 * ax.w0; t.wse = ay.w1 * ax.w1;
    t.wl0 = al.w0; t.wl1 = al.w1;
}

// The second sample of a pair on the taps of the first: consecutive entries of an appearance list are consecutive steps of one ray
// (half a voxel apart on the C2 grid), so the four texels of a plane - or the two of a line - are very often THE SAME as the previous
// entry's. The loads run only in the lanes whose cell changed (the others keep the registers the first sample loaded: a masked load
// leaves inactive lanes untouched); the weights are always this sample's. Same values, same arithmetic: bit-identical rows.
template <int K>
__device__ __forceinline__ void issue_taps_ax_changed(const FactorSet& S, int CQ, int q, const Axes3& A, bool plane_changed, bool line_changed,
                                                      QuadTaps& t) {
    const Axis& ax = A.a[mat0(K)];
    const Axis& ay = A.a[mat1(K)];
    const Axis& al = A.a[vecm(K)];
    const unsigned W = (unsigned)S.W[K];
    const unsigned tb = (unsigned)CQ * 16u, qb = (unsigned)q * 16u;
    const char* __restrict__ P = reinterpret_cast<const char*>(S.plane[K]);
    const char* __restrict__ Ln = reinterpret_cast<const char*>(S.line[K]);
    if (plane_changed) {
        t.nw = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i0, W, (unsigned)ax.i0), tb, qb));
        t.ne = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i0, W, (unsigned)ax.i1), tb, qb));
        t.sw = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i1, W, (unsigned)ax.i0), tb, qb));
        t.se = *reinterpret_cast<const float4*>(P + umad24(umad24((unsigned)ay.i1, W, (unsigned)ax.i1), tb, qb));
    }
    if (line_changed) {
        t.l0 = *reinterpret_cast<const float4*>(Ln + umad24((unsigned)al.i0, tb, qb));
        t.l1 = *reinterpret_cast<const float4*>(Ln + umad24((unsigned)al.i1, tb, qb));
    }
    t.wnw = ay.w0 * ax.w0; t.wne = ay.w0 * ax.w1; t.wsw = ay.w1 * ax.w0; t.wse = ay.w1 * ax.w1;
    t.wl0 = al.w0; t.wl1 = al.w1;
}

__device__ __forceinline__ float4 taps_plane(const QuadTaps& t) {
    float4 v = f4_mul(t.nw, t.wnw);
    v = f4_fma(t.ne, t.wne, v);
    v = f4_fma(t.sw, t.wsw, v);
    v = f4_fma(t.se, t.wse, v);
    return v;
}
__device__ __forceinline__ float4 taps_line(const QuadTaps& t) {
    float4 v = f4_mul(t.l0, t.wl0);
    v = f4_fma(t.l1, t.wl1, v);
    return v;
}


// AlphaGridMask.sample_alpha(xyz) > 0 (models/tensorBase.py:52-59,451-456): 3-D grid_sample, trilinear, zeros padding,
// align_corners=True, of the occupancy volume at a WORLD point. Called only when the field carries a mask.
__device__ __forceinline__ float alpha_value(const FieldDev& F, float px, float py, float pz) {
    const float sx = px - F.a_min[0], sy = py - F.a_min[1], sz = pz - F.a_min[2];
    const float tx = sx * F.a_inv[0], ty = sy * F.a_inv[1], tz = sz * F.a_inv[2];
    const Axis ax = axis_taps(tx - 1.f, F.aW), ay = axis_taps(ty - 1.f, F.aH), az = axis_taps(tz - 1.f, F.aD);
    const float* __restrict__ V = F.alpha;
    const size_t r00 = ((size_t)az.i0 * F.aH + ay.i0) * F.aW, r01 = ((size_t)az.i0 * F.aH + ay.i1) * F.aW;
    const size_t r10 = ((size_t)az.i1 * F.aH + ay.i0) * F.aW, r11 = ((size_t)az.i1 * F.aH + ay.i1) * F.aW;
    float v = V[r00 + ax.i0] * (ax.w0 * ay.w0 * az.w0);
    v += V[r00 + ax.i1] * (ax.w1 * ay.w0 * az.w0);
    v += V[r01 + ax.i0] * (ax.w0 * ay.w1 * az.w0);
    v += V[r01 + ax.i1] * (ax.w1 * ay.w1 * az.w0);
    v += V[r10 + ax.i0] * (ax.w0 * ay.w0 * az.w1);
    v += V[r10 + ax.i1] * (ax.w1 * ay.w0 * az.w1);
    v += V[r11 + ax.i0] * (ax.w0 * ay.w1 * az.w1);
    v += V[r11 + ax.i1] * (ax.w1 * ay.w1 * az.w1);
    return v;
}
__device__ __forceinline__ bool alpha_pass(const FieldDev& F, const Ray& r, float z) {
    const float mx = r.dx * z, my = r.dy * z, mz = r.dz * z;
    return alpha_value(F, r.ox + mx, r.oy + my, r.oz + mz) > 0.f;
}

// Conservative sample-index interval [lo, hi] outside of which no sample of the ray can pass the box test (and the eval
// z gate): slab test in t with a +-3 sample margin (fp32 rounding of the analytic bounds is ~1e-6 relative, the jitter
// shifts samples by < 1). Exact validity is always re-tested per sample; hi < lo means "no candidate".
template <bool TRAIN>
__device__ __forceinline__ void ray_interval(const FieldDev& F, const Ray& ray, int N, int& lo, int& hi) {
    lo = N; hi = -1;
    float t0 = -3.0e38f, t1 = 3.0e38f;
    const float o[3] = {ray.ox, ray.oy, ray.oz}, d[3] = {ray.dx, ray.dy, ray.dz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (d[k] != 0.f) {
            const float ta = (F.aabb0[k] - o[k]) / d[k], tb = (F.aabb1[k] - o[k]) / d[k];
            t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
        } else if (o[k] < F.aabb0[k] || o[k] > F.aabb1[k]) { t1 = -3.0e38f; }
    }
    if (!TRAIN) {   // eval gate: world z > zgate
        if (ray.dz != 0.f) {
            const float tg = (F.zgate - ray.oz) / ray.dz;
            if (ray.dz > 0.f) t0 = fmaxf(t0, tg); else t1 = fminf(t1, tg);
        } else if (!(ray.oz > F.zgate)) { t1 = -3.0e38f; }
    }
    if (t1 >= t0) {
        const float flo = floorf((t0 - ray.tmin) / F.step) - 3.f, fhi = ceilf((t1 - ray.tmin) / F.step) + 3.f;
        lo = (int)fminf(fmaxf(flo, 0.f), (float)N);
        hi = (int)fminf(fmaxf(fhi, -1.f), (float)(N - 1));
    }
}

// expf / logf as the device library evaluates them (extended-precision product with log2 e around v_exp_f32; v_log_f32 times ln 2
// in two pieces), WITHOUT its guards for arguments no caller here has: exp_finite(x) for finite x <= 88 (the library adds the overflow
// branch, a flush below -103.3 that v_ldexp_f32 performs by itself — the argument is clamped at -200, so -inf gives 0 as well —, and
// denormal-input scaling the reduced argument never needs),
// log_ge1(u) for finite u >= 1 (no denormal pre-scaling, no infinity test). Same operations in the same order on the values that
// remain: the results are the library's bit for bit, at 10 + 5 instead of 13 + 13 instructions per evaluation — the tile marcher runs
// three of them per sample (softplus = log1p(exp(.)), then 1 - exp(-sigma dist): models/tensorBase.py:19-26,406-410).
// NaN (deliberate deviation): v_max_f32 returns the other operand for a quiet NaN, so exp_finite(NaN) = exp(-200) = 0 where
// expf(NaN) is NaN — a NaN density feature renders as empty space (sigma = 0, alpha = 0) on the marchers that use this form, where the
// reference propagates NaN into the ray's colour. A diverged field therefore shows as missing geometry and a finite loss, not as NaN
// pixels (tests/test_hip_range.py::test_nan_density_feature_renders_as_empty_space pins the behaviour); keeping the NaN would cost a
// compare + select in front of each of the three evaluations per sample of the frame's most issue-bound kernel.
__device__ __forceinline__ float exp_finite(float x0) {
    float x;                                                        // max(x0, -200): -inf would turn ph - n into NaN; below -103.3 the result is 0
    asm("v_max_f32 %0, 0xc3480000, %1" : "=v"(x) : "v"(x0));        // either way (one instruction: fmaxf adds a canonicalising v_max in front)
    const float l2e = __uint_as_float(0x3fb8aa3bu);
    const float ph = x * l2e;
    const float n = __builtin_rintf(ph);
    float pl = fmaf(x, l2e, -ph);
    pl = fmaf(x, __uint_as_float(0x32a5705fu), pl);                 // + x * (log2 e - l2e)
    const float a = (ph - n) + pl;
    return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)n);
}
__device__ __forceinline__ float log_ge1(float u) {
    const float ln2 = __uint_as_float(0x3f317217u);                 // (the library's high piece: NOT the float nearest to ln 2)
    const float y = __builtin_amdgcn_logf(u);
    const float r0 = y * ln2;
    float r1 = fmaf(y, ln2, -r0);
    r1 = fmaf(y, __uint_as_float(0x3377d1cfu), r1);                 // + y * (ln 2 - ln2)
    return r0 + r1;
}

// log1p(t) for t >= 0 as log(u) + (t - (u - 1)) / u with u = fl(1 + t): the second term restores what the rounding of 1 + t
// dropped (first order; the neglected term is below 2^-48 relative), so the result is within ~1 ulp of the accurate logf —
// at ~35 instructions instead of the ~130 of the library's double-float log1pf (a quarter of the tile marcher's step).
__device__ __forceinline__ float log1p_pos(float t) {
    const float u = 1.f + t;
    return fmaf(t - (u - 1.f), __builtin_amdgcn_rcpf(u), log_ge1(u));
}

// feature2density (models/tensorBase.py:406-410): softplus(beta=1, threshold=20) of feat+shift, or relu(feat).
// ACT: the activation as a compile-time constant (-1: F.act at run time)
template <int ACT = -1>
__device__ __forceinline__ float feature2density(const FieldDev& F, float feat) {
    if (ACT == T2N_ACT_RELU || (ACT < 0 && F.act == T2N_ACT_RELU)) return fmaxf(feat, 0.f);
    const float x = feat + F.shift;
    return x > 20.f ? x : log1p_pos(exp_finite(x));
}

// XCD-aware block -> logical tile map: the dispatcher places block b on XCD b % 8; give every XCD a contiguous run of
// logical tiles so neighbouring rays (which read neighbouring texels) share one L2. Bijective for any nblocks.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned nblocks) {
    const unsigned xcd = b & 7u, j = b >> 3;
    const unsigned q = nblocks >> 3, rem = nblocks & 7u;
    const unsigned start = xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return start + j;
}


// the sum of k_train_loss's per-workgroup partials -> losses[4] (one workgroup of 256 threads, fixed order: deterministic). Shared by
// k_train_loss_reduce (t2n_loss.hip) and the fused training step's one reduce launch (k_wgrad_reduce, t2n_bwd_mlp.hip).
struct LossReduceArgs { const float* part; unsigned nblocks; float* losses; long long R; float w_depth, w_trans; };
__device__ __forceinline__ void loss_reduce_rows(const LossReduceArgs& a, float (*red)[3]) {
    const float* __restrict__ part = a.part;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (unsigned b = threadIdx.x; b < a.nblocks; b += 256) { s0 += part[(size_t)b * 3]; s1 += part[(size_t)b * 3 + 1]; s2 += part[(size_t)b * 3 + 2]; }
    red[threadIdx.x][0] = s0; red[threadIdx.x][1] = s1; red[threadIdx.x][2] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[threadIdx.x][0] += red[threadIdx.x + o][0]; red[threadIdx.x][1] += red[threadIdx.x + o][1]; red[threadIdx.x][2] += red[threadIdx.x + o][2]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float mse = red[0][0] / (3.f * (float)a.R), dl = red[0][1] / (float)a.R, tl = red[0][2] / (float)a.R;
        a.losses[0] = mse; a.losses[1] = dl; a.losses[2] = tl; a.losses[3] = mse + a.w_depth * dl + a.w_trans * tl;
    }
}
}  // namespace t2n
