// Ray marching: sampling + density lookup + transmittance scan + appearance-sample compaction (K1), the per-ray
// composite (K3), and the point-wise density / raw2alpha / ray-filter entry points.
//
// Replaces (reference, relative paths): models/tensorBase.py:304-323 (sample_ray), :448 (dists), :459-462 (z gate),
// :245-246 (normalize_coord), models/tensoRF.py:205-220 (compute_densityfeature), models/tensorBase.py:406-410
// (feature2density), :19-26 (raw2alpha), :477 (app mask), :494-505 (composite).
//
// K1 mapping (gfx950, wave64): one wave per ray, four waves per workgroup.
//   pass A  64 samples per step, one lane each: exact box/z-gate test -> first/last valid sample of the ray (the mask is
//           an interval because every coordinate is monotone in the sample index); writes z_vals when asked.
//   pass B  16 samples per step, 4 lanes per sample: lane q of a sample loads channel quad q (16 B) of each of the
//           4 plane taps + 2 line taps of the 3 factor pairs -> every tap is one contiguous 64-B read; bilinear / linear
//           combine in registers, DPP quad reduction -> sigma into the wave's LDS window.
//   pass C  64 samples per step: alpha, wave-level product scan for the transmittance (carry across steps), weights into
//           the LDS window, acc/depth partial sums, appearance-sample count.
//   pass D  one atomicAdd per ray reserves a contiguous, sample-ordered slice of the appearance list; ballot/prefix
//           compaction writes (normalised point, weight) there.
#include "t2n_device.h"

namespace t2n {

struct MarchArgs {
    FieldDev F;
    const float* rays; long long n_rays; int ray_stride; int n_samples; int npad;
    const float* jitter;
    float* depth; float* acc; float* weights; float* z_vals;
    float4* app_pos; int* app_ray; int4* ray_app; unsigned* counters; unsigned list_cap;
    unsigned long long* stats;
    float* sigma_ctx;    // [n_rays, n_samples] kept for the backward pass (KEEP_CTX) or NULL
    unsigned nblocks;
};

// HALF: density factors are read from their bf16 copies (t2n_field_set_factor_storage)
template <bool TRAIN, int DC, bool HALF>
__global__ __launch_bounds__(256) void k_march(const MarchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LPS = DC / 4;          // lanes per sample
    constexpr int SPW = kWave / LPS;     // samples per wave step in pass B
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    float* __restrict__ sig = smem + (size_t)wid * 2 * a.npad;
    float* __restrict__ wts = sig + a.npad;
    const FieldDev& F = a.F;

    // Appearance list: 8 independent sub-lists (one per XCD-contiguous run of ray tiles, see xcd_tile) so that the
    // per-ray reservation atomics of concurrently running workgroups spread over 8 addresses instead of one.
    const unsigned list = blockIdx.x & 7u;
    const long long r = (long long)xcd_tile(blockIdx.x, a.nblocks) * 4 + wid;
    if (r >= a.n_rays) return;
    const int N = a.n_samples;
    const Ray ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
    const float u = (TRAIN && !F.ztab) ? a.jitter[r] : 0.f;   // NDC: `jitter` is the depth table

    // ---- pass A: validity interval. The mask is an interval (every coordinate is monotone in the sample index), so its
    // ends are found by exact per-sample tests on the 32 candidates at either end of a conservative analytic range; the
    // full exact scan is kept as the fallback for rays where that does not bracket the interval.
    int first = N, last = -1;
    {
        int lo, hi;
        ray_interval<TRAIN>(F, ray, N, lo, hi);
        bool done = hi < lo;
        if (F.ztab) { done = false; lo = 0; hi = -1; }   // NDC depths are a table: no analytic interval, exact scan below
        if (!done && !F.ztab) {
            const bool narrow = hi - lo < 64;
            const int i = narrow ? lo + lane : (lane < 32 ? lo + lane : hi - (lane - 32));
            bool ok = false;
            if (i >= lo && i <= hi) {
                const float z = sample_z<TRAIN>(F, ray, i, u);
                float xn, yn, zn;
                ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
            }
            const unsigned long long m = __ballot(ok);
            if (narrow) {
                if (m) { first = lo + (int)__ffsll((long long)m) - 1; last = lo + 63 - (int)__clzll((long long)m); }
                // exact if the range's ends are invalid or are the ray's first / last sample
                const bool lo_ok = !(m & 1ull) || lo == 0;
                const bool hi_ok = !((m >> (hi - lo)) & 1ull) || hi == N - 1;
                done = lo_ok && hi_ok;
                if (!done) { first = N; last = -1; }
            } else {
                const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
                // provably exact only if both ends are bracketed: the outermost candidates are invalid (or are the
                // first / last sample of the ray) and each group holds a valid sample; otherwise scan everything
                const bool lo_ok = !(mlo & 1u) || lo == 0, hi_ok = !(mhi & 1u) || hi == N - 1;
                if (mlo && mhi && lo_ok && hi_ok) {
                    first = lo + (int)__ffs((int)mlo) - 1;
                    last = hi - ((int)__ffs((int)mhi) - 1);
                    done = true;
                }
            }
        }
        if (!done) {   // exact scan over all samples
            for (int base = 0; base < N; base += 64) {
                const int i = base + lane;
                bool ok = false;
                if (i < N) {
                    const float z = sample_z<TRAIN>(F, ray, i, u);
                    float xn, yn, zn;
                    ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
                }
                const unsigned long long m = __ballot(ok);
                if (m) {
                    if (first == N) first = base + (int)__ffsll((long long)m) - 1;
                    last = base + 63 - (int)__clzll((long long)m);
                }
            }
        }
    }
    const unsigned nvalid = last >= first ? (unsigned)(last - first + 1) : 0u;
    if (a.z_vals) {
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            if (i < N) a.z_vals[r * N + i] = sample_z<TRAIN>(F, ray, i, u);
        }
    }

    float acc = 0.f, dep = 0.f;
    unsigned napp = 0;
    const int Lw = last - first + 1;   // window length (<= 0: empty ray)

    unsigned nmask = 0;   // in-interval samples rejected by the alpha mask
    int Le = Lw;          // evaluated part of the window (shorter than Lw after an early termination)
    if (Lw > 0) {
      // early termination (t2n_field_set_early_termination; eval without weights / context only): passes B and C alternate over
      // 64-sample blocks and the ray stops behind the block that took its transmittance below term_eps
      const bool et = !TRAIN && F.term_eps > 0.f;
      const int blk = et ? 64 : Lw;
      float carry = 1.f;
      for (int b0 = 0; b0 < Lw; b0 += blk) {
        const int b1 = min(Lw, b0 + blk);
        // ---- pass B: density -------------------------------------------------------------------------------------
        const int q = lane & (LPS - 1);
        const int sl = lane / LPS;
        for (int base = b0; base < b1; base += SPW) {
            const int j = base + sl;
            const int i = first + j;
            float xn = 0.f, yn = 0.f, zn = 0.f;
            bool ok = false;
            if (j < b1) {
                const float z = sample_z<TRAIN>(F, ray, i, u);
                ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
                if (F.alpha && ok) ok = alpha_pass(F, ray, z);      // models/tensorBase.py:451-456
            }
            if (F.alpha) nmask += (unsigned)__popcll(__ballot((j < b1) & !ok & (q == 0)));
            float part = 0.f;
            if (ok) {
                QuadTaps t0, t1, t2;
                const Axes3 A = sample_axes_inbox(F.den, xn, yn, zn);   // ok: the sample passed the box test
                issue_taps_ax<0, HALF>(F.den, LPS, q, A, t0);
                issue_taps_ax<1, HALF>(F.den, LPS, q, A, t1);
                issue_taps_ax<2, HALF>(F.den, LPS, q, A, t2);
                float4 p, l;
                p = taps_plane(t0); l = taps_line(t0);
                part = p.x * l.x; part = fmaf(p.y, l.y, part); part = fmaf(p.z, l.z, part); part = fmaf(p.w, l.w, part);
                p = taps_plane(t1); l = taps_line(t1);
                part = fmaf(p.x, l.x, part); part = fmaf(p.y, l.y, part); part = fmaf(p.z, l.z, part); part = fmaf(p.w, l.w, part);
                p = taps_plane(t2); l = taps_line(t2);
                part = fmaf(p.x, l.x, part); part = fmaf(p.y, l.y, part); part = fmaf(p.z, l.z, part); part = fmaf(p.w, l.w, part);
            }
            const float feat = group_sum<LPS>(part);
            if (q == 0 && j < b1) {
                const float sg = ok ? feature2density(F, feat) : 0.f;
                sig[j] = sg;
                if (a.sigma_ctx) a.sigma_ctx[r * N + i] = sg;
            }
        }
        // LDS window written by lanes of this wave only; a wave is in lock-step, but the compiler needs the fence.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ---- pass C: alpha, transmittance scan, weights ------------------------------------------------------------
        for (int base = b0; base < b1; base += 64) {
            const int j = base + lane;
            const int i = first + j;
            float sg = 0.f, z = 0.f, dist = 0.f;
            if (j < b1) {
                sg = sig[j];
                z = sample_z<TRAIN>(F, ray, i, u);
                if (i < N - 1) dist = sample_z<TRAIN>(F, ray, i + 1, u) - z;   // :448, last sample gets 0
            }
            const float d = scaled_dist(F, ray, dist);
            const float nsd = (-sg) * d;
            const float alpha = 1.f - exp_finite(nsd);   // (nsd <= 0)
            const float f = (1.f - alpha) + 1e-10f;
            const float incl = wave_scan_mul(f, lane);
            float excl = __shfl_up(incl, 1);
            if (lane == 0) excl = 1.f;
            const float T = carry * excl;
            const float w = alpha * T;
            carry = carry * __shfl(incl, 63);
            if (j < b1) {
                wts[j] = w;
                acc += w;
                dep = fmaf(w, z, dep);
            }
            napp += (unsigned)__popcll(__ballot((j < b1) & (w > F.thres)));
        }
        if (et && carry < F.term_eps) { Le = b1; break; }
      }
        acc = wave_sum(acc);
        dep = wave_sum(dep);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ---- outputs per ray ---------------------------------------------------------------------------------------------
    unsigned slot0 = 0;
    if (lane == 0) {
        if (napp) slot0 = atomicAdd(&a.counters[list * kCounterStride], napp);
        const bool fits = slot0 + napp <= a.list_cap;
        slot0 += list * a.list_cap;
        a.ray_app[r] = make_int4((int)slot0, fits ? (int)napp : 0, (int)((Le < Lw ? (unsigned)Le : nvalid) - nmask), Le > 0 ? (first | (Le << 11)) : 0);
        a.acc[r] = acc;
        a.depth[r] = dep + (1.f - acc) * ray.last;   // :504-505
        if (!fits && a.stats) a.stats[T2N_STAT_OVERFLOW] = 1ull;   // cannot happen: list_cap is the worst case on this path
        if (!fits) a.counters[kOverflowWord] = 1u;
        if (!fits) napp = 0;
    }
    napp = __shfl(napp, 0);
    slot0 = __shfl(slot0, 0);

    if (a.weights) {
        for (int base = 0; base < N; base += 64) {
            const int i = base + lane;
            if (i < N) {
                const int j = i - first;
                a.weights[r * N + i] = (j >= 0 && j < Lw) ? wts[j] : 0.f;
            }
        }
    }

    // ---- pass D: appearance list -------------------------------------------------------------------------------------
    if (napp) {
        unsigned run = 0;
        for (int base = 0; base < Le; base += 64) {
            const int j = base + lane;
            const float w = j < Le ? wts[j] : 0.f;
            const bool m = (j < Le) & (w > F.thres);
            const unsigned long long bal = __ballot(m);
            if (m) {
                const unsigned pre = (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
                const float z = sample_z<TRAIN>(F, ray, first + j, u);
                float xn, yn, zn;
                sample_point<TRAIN>(F, ray, z, xn, yn, zn);
                const unsigned s = slot0 + run + pre;
                a.app_pos[s] = make_float4(xn, yn, zn, w);
                a.app_ray[s] = (int)r;
            }
            run += (unsigned)__popcll(bal);
        }
    }
}

// K3: per-ray composite of the shaded appearance samples, in sample order (models/tensorBase.py:494-501).
struct CompositeArgs {
    long long n_rays; const int4* ray_app; const float4* app_pos; const float4* app_rgb; const float* acc; float* rgb;
    float4* rgb_raw;   // pre-clamp colour kept for the backward pass, or NULL
    int add_bg;
    int img_w, img_h;   // > 0: the sub-launch is whole rows of an image and its lists were written by the tile marcher
};
// One thread per ray. After the tile marcher the 64 rays of an 8x8-pixel tile own ONE contiguous region of the list (their slices
// back to back), so a wave takes a tile (lane -> pixel as in k_march_tiles): its reads stay inside that region and every line is
// fetched once. (Row-major waves cut across eight tiles' regions: the same lines were fetched by eight waves, 312 MB of HBM
// traffic per C2 frame for 94 MB of entries and outputs.)
__global__ __launch_bounds__(256) void k_composite(const CompositeArgs a) {
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (a.img_w > 0) {
        const int lane = threadIdx.x & 63;
        const long long tile = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
        const int tiles_x = (a.img_w + 7) >> 3;
        const int ty = (int)(tile / tiles_x), tx = (int)(tile - (long long)ty * tiles_x);
        const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
        if (px >= a.img_w || py >= a.img_h) return;
        r = (long long)py * a.img_w + px;
    }
    if (r >= a.n_rays) return;
    const int4 ra = a.ray_app[r];
    if (ra.x < 0) return;   // a ray whose entries fitted no sub-list (budgeted lists): k_finish_rays, behind this kernel, writes its colour
    float cr = 0.f, cg = 0.f, cb = 0.f;
    // (r, g, b, w) entries: the shading kernels copy the entry's weight next to its colour. Eight loads in flight per trip (a wave
    // runs as long as its longest slice: one dependent round trip per entry was most of the kernel's 57 us); the sum keeps the
    // entries' order
    for (int k = 0; k < ra.y; k += 8) {
        float4 c[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) c[j] = a.app_rgb[ra.x + min(k + j, ra.y - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (k + j < ra.y) { cr = fmaf(c[j].w, c[j].x, cr); cg = fmaf(c[j].w, c[j].y, cg); cb = fmaf(c[j].w, c[j].z, cb); }
    }
    if (a.add_bg) {
        const float bg = 1.f - a.acc[r];
        cr += bg; cg += bg; cb += bg;
    }
    if (a.rgb_raw) a.rgb_raw[r] = make_float4(cr, cg, cb, 0.f);
    a.rgb[r * 3 + 0] = fminf(fmaxf(cr, 0.f), 1.f);
    a.rgb[r * 3 + 1] = fminf(fmaxf(cg, 0.f), 1.f);
    a.rgb[r * 3 + 2] = fminf(fmaxf(cb, 0.f), 1.f);
}

// Per-call counters: V (evaluated samples) and A (appearance samples) summed over the rays of one sub-launch.
__global__ __launch_bounds__(256) void k_ray_stats(const int4* __restrict__ ray_app, long long n, unsigned long long* stats) {
    __shared__ unsigned long long part[8];
    unsigned long long v = 0, ap = 0;
    for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (long long)gridDim.x * blockDim.x) {
        const int4 ra = ray_app[r];
        v += (unsigned)ra.z; ap += (unsigned)ra.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v += __shfl_xor((long long)v, o);
        ap += __shfl_xor((long long)ap, o);
    }
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[wid] = v; part[4 + wid] = ap; }
    __syncthreads();
    if (threadIdx.x == 0) {   // one atomic pair per block: same-address atomics serialise (~88 per microsecond)
        atomicAdd(&stats[T2N_STAT_EVALUATED], part[0] + part[1] + part[2] + part[3]);
        atomicAdd(&stats[T2N_STAT_APPEARANCE], part[4] + part[5] + part[6] + part[7]);
    }
}

// Point-wise density: 4 lanes per point (same gather as pass B).
template <int DC>
__global__ __launch_bounds__(256) void k_density_at(const FieldDev F, const float* __restrict__ xyz, long long n,
                                                    float* feat_out, float* sigma_out) {
    constexpr int LPS = DC / 4;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long p = t / LPS;
    const int q = (int)(t % LPS);
    const bool ok = p < n;
    float part = 0.f;
    if (ok) {
        const float xn = xyz[p * 3 + 0], yn = xyz[p * 3 + 1], zn = xyz[p * 3 + 2];
        QuadTaps t0, t1, t2;
        const Axes3 A = sample_axes(F.den, xn, yn, zn);
        issue_taps_ax<0>(F.den, LPS, q, A, t0);
        issue_taps_ax<1>(F.den, LPS, q, A, t1);
        issue_taps_ax<2>(F.den, LPS, q, A, t2);
        float4 pv, l;
        pv = taps_plane(t0); l = taps_line(t0);
        part = pv.x * l.x; part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
        pv = taps_plane(t1); l = taps_line(t1);
        part = fmaf(pv.x, l.x, part); part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
        pv = taps_plane(t2); l = taps_line(t2);
        part = fmaf(pv.x, l.x, part); part = fmaf(pv.y, l.y, part); part = fmaf(pv.z, l.z, part); part = fmaf(pv.w, l.w, part);
    }
    const float feat = group_sum<LPS>(part);
    if (ok && q == 0) {
        if (feat_out) feat_out[p] = feat;
        if (sigma_out) sigma_out[p] = feature2density(F, feat);
    }
}

// raw2alpha on explicit [R,N] tensors: one wave per ray.
__global__ __launch_bounds__(256) void k_raw2alpha(const float* __restrict__ sigma, const float* __restrict__ dist,
                                                   long long R, int N, float* alpha_o, float* w_o, float* bg_o) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float carry = 1.f;
    for (int base = 0; base < N; base += 64) {
        const int i = base + lane;
        float sg = 0.f, d = 0.f;
        if (i < N) { sg = sigma[r * N + i]; d = dist[r * N + i]; }
        const float nsd = (-sg) * d;
        const float alpha = 1.f - expf(nsd);
        const float f = (1.f - alpha) + 1e-10f;
        const float incl = wave_scan_mul(f, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        carry = carry * __shfl(incl, 63);
        if (i < N) {
            if (alpha_o) alpha_o[r * N + i] = alpha;
            if (w_o) w_o[r * N + i] = alpha * T;
        }
    }
    if (bg_o && lane == 0) bg_o[r] = carry;
}

__global__ __launch_bounds__(256) void k_alpha_at(const FieldDev F, const float* __restrict__ xyz, long long n, float* out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    out[t] = alpha_value(F, xyz[t * 3], xyz[t * 3 + 1], xyz[t * 3 + 2]);
}

// filtering_rays(bbox_only=True): models/tensorBase.py:385-391
__global__ __launch_bounds__(256) void k_filter_bbox(const FieldDev F, const float* __restrict__ rays, long long n,
                                                     int stride, uint8_t* mask) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const float* rp = rays + r * stride;
    const float ox = rp[0], oy = rp[1], oz = rp[2], dx = rp[3], dy = rp[4], dz = rp[5];
    const float vx = dx == 0.f ? 1e-6f : dx, vy = dy == 0.f ? 1e-6f : dy, vz = dz == 0.f ? 1e-6f : dz;
    const float ax = (F.aabb1[0] - ox) / vx, bx = (F.aabb0[0] - ox) / vx;
    const float ay = (F.aabb1[1] - oy) / vy, by = (F.aabb0[1] - oy) / vy;
    const float az = (F.aabb1[2] - oz) / vz, bz = (F.aabb0[2] - oz) / vz;
    const float tmin = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    const float tmax = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    mask[r] = tmax > tmin ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
int launch_march(t2n_field* f, const RenderLaunch& L, hipStream_t s) {
    MarchArgs a;
    a.F = f->dev;
    a.rays = L.rays; a.n_rays = L.n_rays; a.ray_stride = L.ray_stride; a.n_samples = L.n_samples;
    a.npad = (L.n_samples + 63) & ~63;
    a.jitter = L.jitter;
    a.depth = L.depth; a.acc = L.acc; a.weights = L.weights; a.z_vals = L.z_vals;
    a.app_pos = L.app_pos; a.app_ray = L.app_ray; a.ray_app = L.ray_app; a.counters = L.counters; a.list_cap = L.list_cap;
    a.stats = (unsigned long long*)L.stats;
    a.sigma_ctx = L.sigma_ctx;
    a.nblocks = (unsigned)((L.n_rays + 3) / 4);
    const size_t lds = (size_t)4 * 2 * a.npad * sizeof(float);
    const bool train = (L.flags & T2N_FLAG_TRAIN) != 0;
    a.F.term_eps = (!train && !L.weights && !L.z_vals && !L.sigma_ctx) ? f->term_eps : 0.f;
    timing_begin(f, T2N_K_MARCH, s);
    const bool half = f->factor_bf16 && f->dev.den.plane_h[0];
    if (train) {
        if (half) hipLaunchKernelGGL((k_march<true, 16, true>), dim3(a.nblocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_march<true, 16, false>), dim3(a.nblocks), dim3(256), lds, s, a);
    } else {
        if (half) hipLaunchKernelGGL((k_march<false, 16, true>), dim3(a.nblocks), dim3(256), lds, s, a);
        else hipLaunchKernelGGL((k_march<false, 16, false>), dim3(a.nblocks), dim3(256), lds, s, a);
    }
    timing_end(f, T2N_K_MARCH, s);
    T2N_HIP(hipGetLastError());
    return launch_ray_stats(L, s);
}

#ifndef T2N_STATS_BLOCKS
#define T2N_STATS_BLOCKS 128
#endif
int launch_ray_stats(const RenderLaunch& L, hipStream_t s) {
    if (L.stats) {
        unsigned nb = (unsigned)((L.n_rays + 255) / 256);
        if (nb > T2N_STATS_BLOCKS) nb = T2N_STATS_BLOCKS;
        hipLaunchKernelGGL(k_ray_stats, dim3(nb), dim3(256), 0, s, (const int4*)L.ray_app, (long long)L.n_rays,
                           (unsigned long long*)L.stats);
        T2N_HIP(hipGetLastError());
    }
    return T2N_OK;
}

int launch_composite(t2n_field* f, const RenderLaunch& L, hipStream_t s, int img_w, int img_h) {
    CompositeArgs c;
    c.n_rays = L.n_rays; c.ray_app = L.ray_app; c.app_pos = L.app_pos; c.app_rgb = L.app_rgb; c.acc = L.acc; c.rgb = L.rgb; c.rgb_raw = L.rgb_raw;
    c.add_bg = (L.flags & T2N_FLAG_ADD_BG) ? 1 : 0;
    c.img_w = img_w; c.img_h = img_h;
    long long blocks = (L.n_rays + 255) / 256;
    if (img_w > 0) blocks = ((long long)((img_w + 7) >> 3) * ((img_h + 7) >> 3) + 3) / 4;   // four 8x8 tiles per workgroup
    timing_begin(f, T2N_K_COMPOSITE, s);
    hipLaunchKernelGGL(k_composite, dim3((unsigned)blocks), dim3(256), 0, s, c);
    timing_end(f, T2N_K_COMPOSITE, s);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// TensorBase.sample_ray / sample_ray_ndc as a stage of their own (models/tensorBase.py:293-323): one thread per (ray, sample)
__global__ __launch_bounds__(256) void k_sample_ray(FieldDev F, const float* __restrict__ ro, const float* __restrict__ rd, long long n, int N,
                                                    const float* __restrict__ u, float* __restrict__ pts, float* __restrict__ zs,
                                                    unsigned char* __restrict__ valid) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * N) return;
    const long long r = t / N;
    const int i = (int)(t - r * N);
    Ray ray;
    ray.ox = ro[r * 3]; ray.oy = ro[r * 3 + 1]; ray.oz = ro[r * 3 + 2];
    ray.dx = rd[r * 3]; ray.dy = rd[r * 3 + 1]; ray.dz = rd[r * 3 + 2];
    ray.tmin = F.ztab ? 0.f : ray_tmin(F, ray.ox, ray.oy, ray.oz, ray.dx, ray.dy, ray.dz);
    const float z = u ? sample_z<true>(F, ray, i, u[r]) : sample_z<false>(F, ray, i, 0.f);
    const float mx = ray.dx * z, my = ray.dy * z, mz = ray.dz * z;
    const float px = ray.ox + mx, py = ray.oy + my, pz = ray.oz + mz;
    const bool out = (F.aabb0[0] > px) | (px > F.aabb1[0]) | (F.aabb0[1] > py) | (py > F.aabb1[1]) | (F.aabb0[2] > pz) | (pz > F.aabb1[2]);
    pts[t * 3] = px; pts[t * 3 + 1] = py; pts[t * 3 + 2] = pz;
    if (zs) zs[t] = z;
    valid[t] = out ? 0 : 1;
}

}  // namespace t2n

using namespace t2n;

extern "C" int t2n_sample_ray(const t2n_field* f, const float* rays_o, const float* rays_d, int64_t n, int n_samples, const float* jitter,
                              int ndc, float* pts, float* z_vals, uint8_t* valid, t2n_stream stream) {
    if (!f || !rays_o || !rays_d || !pts || !valid || n < 0 || n_samples <= 0 || (ndc && !jitter)) {
        set_error("t2n_sample_ray: bad argument");
        return T2N_ERR_INVALID;
    }
    if (n == 0) return T2N_OK;
    FieldDev F = f->dev;
    F.ztab = ndc ? jitter : nullptr;
    const long long total = (long long)n * n_samples;
    hipLaunchKernelGGL(k_sample_ray, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, F, rays_o, rays_d, (long long)n,
                       n_samples, ndc ? nullptr : jitter, pts, ndc ? nullptr : z_vals, valid);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_density_at(const t2n_field* f, const float* xyz_norm, int64_t n, float* feat, float* sigma,
                              t2n_stream stream) {
    if (!f || !xyz_norm || n < 0) { set_error("t2n_density_at: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded) { set_error("t2n_density_at: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (n == 0) return T2N_OK;
    const long long threads = (long long)n * 4;
    hipLaunchKernelGGL((k_density_at<16>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       f->dev, xyz_norm, (long long)n, feat, sigma);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_raw2alpha(const float* sigma, const float* dist, int64_t n_rays, int n_samples, float* alpha,
                             float* weights, float* bg, t2n_stream stream) {
    if (!sigma || !dist || n_rays < 0 || n_samples <= 0) { set_error("t2n_raw2alpha: bad argument"); return T2N_ERR_INVALID; }
    if (n_rays == 0) return T2N_OK;
    hipLaunchKernelGGL(k_raw2alpha, dim3((unsigned)((n_rays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sigma, dist,
                       (long long)n_rays, n_samples, alpha, weights, bg);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_filter_rays_bbox(const t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, uint8_t* mask,
                                    t2n_stream stream) {
    if (!f || !rays || !mask || n_rays < 0 || ray_stride < 6) { set_error("t2n_filter_rays_bbox: bad argument"); return T2N_ERR_INVALID; }
    if (n_rays == 0) return T2N_OK;
    hipLaunchKernelGGL(k_filter_bbox, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f->dev,
                       rays, (long long)n_rays, ray_stride, mask);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_alpha_at(const t2n_field* f, const float* xyz_world, int64_t n, float* alpha, t2n_stream stream) {
    if (!f || !xyz_world || !alpha || n < 0) { set_error("t2n_alpha_at: bad argument"); return T2N_ERR_INVALID; }
    if (!f->dev.alpha) { set_error("t2n_alpha_at: the field has no alpha mask"); return T2N_ERR_STATE; }
    if (n == 0) return T2N_OK;
    hipLaunchKernelGGL(k_alpha_at, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f->dev, xyz_world,
                       (long long)n, alpha);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}
