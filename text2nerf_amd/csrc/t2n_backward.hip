// Backward of the render call w.r.t. every field parameter: what autograd derives for the reference's
// TensorBase.forward (text2nerf_main.py:589). Sample coordinates are detached in the reference
// (models/tensoRF.py:208-210,226-228), so gradients flow only into the factor tensors, basis_mat and the MLP.
//
// Pipeline (all on the caller's stream; `rows` = padded appearance-sample rows, 32 per shade tile):
//   1  k_shade (ctx mode)      re-run the appearance forward, keeping X[rows,144], feat[rows,32], h0/h1[rows,128]
//   2  k_bwd_march             per ray: recompute alpha/T/w from the kept sigma, dL/dw -> dL/dalpha (reverse scan) -> dL/dsigma
//                              -> dL/dfeature; emits per appearance sample go = dL/d(pre-sigmoid rgb); counts the samples of every
//                              15^3-cell block of the grid
//   2b k_bin_scan, k_bwd_bin,  density scatter (side stream): one record per sample sorted by block; per block the trilinear splat of
//      k_bwd_den_block         dL/dfeature into a 16^3 corner field in LDS, from which all six density gradients are small MFMA
//                              contractions with the block's line rows / plane texels (global-atomic path: k_bwd_march<.,false>)
//   3  k_bwd_l2                layer 2 (3 outputs): dW2, db2 (per-workgroup partials + k_bwd_l2_reduce), g1 = (go W2) * [h1 > 0]
//   3-4 (MLP_Fea_noview head) k_mlp_bwd_ss: the whole input-gradient chain g1 -> g0 -> gx -> gf -> gX in one kernel (t2n_mlp_bwd_ss.hip);
//                              k_bwd_l2 then only accumulates dW2 / db2, the weight gradients stay GEMMs (t2n_gemm_h.hip)
//   4  gemm_tn / gemm_nn       fp32-MFMA GEMMs: dW1 = g1^T h0, g0 = (g1 W1) * [h0 > 0], dW0 = g0^T PE(feat), gx = g0 W0,
//                              PE backward -> gf, dWb = gf^T X, gX = gf Wb; k_colsum for the biases
//   5  k_app_bin, k_bin_scan,  appearance scatter: records sorted by 16x16-texel plane tile, accumulated in LDS (ds_add_f64), flushed once per
//      k_bwd_tile_accum<48>    tile segment (global-atomic path: k_bwd_app_scatter); runs beside the weight-gradient GEMMs (third stream)
//   6  k_relayout_add          channel-last gradient buffers -> += reference-layout [1,C,H,W] gradient tensors
// This file: k_bwd_march, the scatters (density blocks, appearance tiles, the atomic paths) and the driver (workspace carve, streams,
// t2n_render_backward). Steps 3-4 live in t2n_bwd_mlp.hip (+ t2n_gemm_h.hip, t2n_mlp_bwd_ss.hip), the training loss in t2n_loss.hip.
#include <stdlib.h>

#include <mutex>

#include "t2n_device.h"

namespace t2n {

typedef float f32x16 __attribute__((ext_vector_type(16)));
__host__ __device__ constexpr int unit_of(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }
__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct TilePrefix { unsigned t[kLists + 1]; };   // tiles before each sub-list (host-computed from the counters)
struct GradSet { float* plane[3]; float* line[3]; };
// The same in device memory, for a backward that never reads the counters on the host (T2N_FLAG_DEVICE_ROWS): tile prefix and row
// count from the forward's counters (k_bwd_plan), clipped to the row CAPACITY the caller's buffers hold; overflow = the count
// exceeded it (the rows beyond take no part in this backward: the caller learns it from the forward's posted counters)
struct BwdPlan { TilePrefix tp; unsigned rows; unsigned overflow; };

__device__ __forceinline__ unsigned slot_to_row(unsigned slot, unsigned list_cap, const TilePrefix& tp) {
    const unsigned l = slot / list_cap;
    return tp.t[l] * 32u + (slot - l * list_cap);
}

// Scatter d(feature)/d(plane taps, line taps) for factor pair K, ONE CHANNEL PER LANE: the 16 lanes of a sample cover 16
// consecutive channels = one 64-B line per tap, so every atomic instruction touches 4 lines x 16 lanes (a 4-lane/float4
// mapping would visit each line four times). `coff` = channel index of this lane inside the C-channel texel, `g` =
// dL/d(P_c * L_c) for that channel. Value taps are re-gathered (scalar, coalesced per 16 lanes).
template <int K>
__device__ __forceinline__ void scatter_chan(const FactorSet& S, const GradSet& G, int C, int coff, float xn, float yn, float zn,
                                             float g) {
    TapIdx o;
    compute_taps<K>(S, C / 4, 0, xn, yn, zn, o);   // float4-unit offsets of channel 0 of each tap
    const float* __restrict__ P = S.plane[K];
    const float* __restrict__ Ln = S.line[K];
    const unsigned nw = o.nw * 4 + coff, ne = o.ne * 4 + coff, sw = o.sw * 4 + coff, se = o.se * 4 + coff;
    const unsigned l0 = o.l0 * 4 + coff, l1 = o.l1 * 4 + coff;
    float pv = P[nw] * o.wnw;
    pv = fmaf(P[ne], o.wne, pv); pv = fmaf(P[sw], o.wsw, pv); pv = fmaf(P[se], o.wse, pv);
    const float lv = fmaf(Ln[l1], o.wl1, Ln[l0] * o.wl0);
    const float gp = g * lv, gl = g * pv;
    float* gP = G.plane[K];
    float* gL = G.line[K];
    if (o.wnw != 0.f) atomicAdd(gP + nw, gp * o.wnw);
    if (o.wne != 0.f) atomicAdd(gP + ne, gp * o.wne);
    if (o.wsw != 0.f) atomicAdd(gP + sw, gp * o.wsw);
    if (o.wse != 0.f) atomicAdd(gP + se, gp * o.wse);
    if (o.wl0 != 0.f) atomicAdd(gL + l0, gl * o.wl0);
    if (o.wl1 != 0.f) atomicAdd(gL + l1, gl * o.wl1);
}

// Sliding-window scatter: consecutive samples of a ray (and consecutive appearance samples of the list) move by less
// than a texel per step, so their 2x2 plane footprints and 2-row line footprints mostly coincide or shift by one. Each
// lane (one channel) keeps the footprint's accumulators in registers and issues a global atomic only when a texel leaves
// the window — the L2 fp32-atomic rate (~1 dword / 2.7 clk / channel) bounds this pass, so fewer atomics is the lever.
constexpr int kWinEmpty = -(1 << 30);
struct PlaneWin {
    int x, y;                    // base texel (x0, y0), may be -1 at the low border; kWinEmpty: empty
    float a00, a01, a10, a11;    // [dy][dx]
};
struct LineWin {
    int r;                       // base row; kWinEmpty: empty
    float a0, a1;
};
__device__ __forceinline__ void flush1(float* g, int W, int C, int x, int y, int coff, float v) {
    if (v != 0.f && x >= 0 && y >= 0) atomicAdd(g + ((size_t)y * W + x) * C + coff, v);
}
__device__ __forceinline__ void plane_flush(PlaneWin& w, float* g, int W, int C, int coff) {
    if (w.x != kWinEmpty) {
        flush1(g, W, C, w.x, w.y, coff, w.a00); flush1(g, W, C, w.x + 1, w.y, coff, w.a01);
        flush1(g, W, C, w.x, w.y + 1, coff, w.a10); flush1(g, W, C, w.x + 1, w.y + 1, coff, w.a11);
    }
    w.x = kWinEmpty; w.a00 = w.a01 = w.a10 = w.a11 = 0.f;
}
// add the contributions (v00 v01 / v10 v11) of a sample whose footprint starts at (x, y)
__device__ __forceinline__ void plane_add(PlaneWin& w, float* g, int W, int C, int coff, int x, int y, float v00, float v01,
                                          float v10, float v11) {
    if (w.x != kWinEmpty && !(x == w.x && y == w.y)) {
        const int dx = x - w.x, dy = y - w.y;
        if (dy == 0 && dx == 1) {           // shift right: column x leaves
            flush1(g, W, C, w.x, w.y, coff, w.a00); flush1(g, W, C, w.x, w.y + 1, coff, w.a10);
            w.a00 = w.a01; w.a10 = w.a11; w.a01 = 0.f; w.a11 = 0.f; w.x = x;
        } else if (dy == 0 && dx == -1) {
            flush1(g, W, C, w.x + 1, w.y, coff, w.a01); flush1(g, W, C, w.x + 1, w.y + 1, coff, w.a11);
            w.a01 = w.a00; w.a11 = w.a10; w.a00 = 0.f; w.a10 = 0.f; w.x = x;
        } else if (dx == 0 && dy == 1) {
            flush1(g, W, C, w.x, w.y, coff, w.a00); flush1(g, W, C, w.x + 1, w.y, coff, w.a01);
            w.a00 = w.a10; w.a01 = w.a11; w.a10 = 0.f; w.a11 = 0.f; w.y = y;
        } else if (dx == 0 && dy == -1) {
            flush1(g, W, C, w.x, w.y + 1, coff, w.a10); flush1(g, W, C, w.x + 1, w.y + 1, coff, w.a11);
            w.a10 = w.a00; w.a11 = w.a01; w.a00 = 0.f; w.a01 = 0.f; w.y = y;
        } else {
            plane_flush(w, g, W, C, coff);
        }
    }
    w.x = x; w.y = y;
    w.a00 += v00; w.a01 += v01; w.a10 += v10; w.a11 += v11;
}
__device__ __forceinline__ void line_flush(LineWin& w, float* g, int C, int coff) {
    if (w.r != kWinEmpty) {
        if (w.a0 != 0.f && w.r >= 0) atomicAdd(g + (size_t)w.r * C + coff, w.a0);
        if (w.a1 != 0.f) atomicAdd(g + (size_t)(w.r + 1) * C + coff, w.a1);
    }
    w.r = kWinEmpty; w.a0 = w.a1 = 0.f;
}
__device__ __forceinline__ void line_add(LineWin& w, float* g, int C, int coff, int r, float v0, float v1) {
    if (w.r != kWinEmpty && r != w.r) {
        if (r == w.r + 1) {
            if (w.a0 != 0.f && w.r >= 0) atomicAdd(g + (size_t)w.r * C + coff, w.a0);
            w.a0 = w.a1; w.a1 = 0.f;
        } else if (r == w.r - 1) {
            if (w.a1 != 0.f) atomicAdd(g + (size_t)(w.r + 1) * C + coff, w.a1);
            w.a1 = w.a0; w.a0 = 0.f;
        } else {
            line_flush(w, g, C, coff);
        }
    }
    w.r = r;
    w.a0 += v0; w.a1 += v1;
}

// One channel of factor pair K for one sample: re-gather the value taps, form dL/dP and dL/dL for this channel and push
// them into the sliding windows. The footprint origin is the UNCLAMPED floor index (taps outside the grid carry weight 0
// and accumulate exact zeros, which flush1 never writes).
template <int K>
__device__ __forceinline__ void scatter_win(const FactorSet& S, const GradSet& G, int C, int coff, float xn, float yn, float zn,
                                            float g, PlaneWin& pw, LineWin& lw) {
    float gx, gy, gv;
    plane_line_coords<K>(xn, yn, zn, gx, gy, gv);
    const Axis ax = axis_taps(gx, S.W[K]);
    const Axis ay = axis_taps(gy, S.H[K]);
    const Axis al = axis_taps(gv, S.L[K]);
    const int W = S.W[K];
    const float* __restrict__ P = S.plane[K];
    const float* __restrict__ Ln = S.line[K];
    const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
    float pv = P[((size_t)ay.i0 * W + ax.i0) * C + coff] * w00;
    pv = fmaf(P[((size_t)ay.i0 * W + ax.i1) * C + coff], w01, pv);
    pv = fmaf(P[((size_t)ay.i1 * W + ax.i0) * C + coff], w10, pv);
    pv = fmaf(P[((size_t)ay.i1 * W + ax.i1) * C + coff], w11, pv);
    const float lv = fmaf(Ln[(size_t)al.i1 * C + coff], al.w1, Ln[(size_t)al.i0 * C + coff] * al.w0);
    const float gp = g * lv, gl = g * pv;
    // window origins: i1 - 1 where the high tap is live, else i0 (clamped borders: the dead tap's weight is 0)
    const int ox = ax.w1 != 0.f ? ax.i1 - 1 : ax.i0, oy = ay.w1 != 0.f ? ay.i1 - 1 : ay.i0, ol = al.w1 != 0.f ? al.i1 - 1 : al.i0;
    // re-express the four products on the window grid (a clamped low tap at the -1 border sits at origin + 0 with weight 0)
    const float vx0 = ax.i0 == ox ? 1.f : 0.f, vy0 = ay.i0 == oy ? 1.f : 0.f;
    plane_add(pw, G.plane[K], W, C, coff, ox, oy, gp * w00 * vx0 * vy0, gp * w01 * vy0, gp * w10 * vx0, gp * w11);
    line_add(lw, G.line[K], C, coff, ol, al.i0 == ol ? gl * al.w0 : 0.f, gl * al.w1);
}


// ---- binned scatters -------------------------------------------------------------------------------------------------------
// A training batch puts ~28 samples into every texel of every density plane, from unrelated rays: global fp32 atomics per
// tap (even merged along a ray) run at the L2 atomic rate. Instead: (1) count the samples of every bin, (2) prefix-sum,
// (3) write a 16-B record per sample into its bin's run, (4) one workgroup per segment of one bin's records accumulates in
// LDS and flushes once. Appearance: bins = 16x16-texel plane tiles (one record per sample AND plane; ds_add_f64 on a staged
// 17x17x16 tile + the whole line, k_bwd_tile_accum). Density: bins = 15^3-cell blocks, one record per sample (k_bwd_den_block).
constexpr int kBinTile = 16;      // texels per tile edge (footprints reach one texel further: 17 staged)
constexpr int kBinCopies = 32;    // privatised histogram / cursor copies (every ray starts in the camera's tile)
constexpr int kBinSegApp = 512;   // smallest segment (sizes the segment list); the scan picks the actual size per call
// (512 work items of >= 512 records until the staging of a segment got cheap — round 6 — and the accounting showed its fixed cost
// per segment: 256 items of >= 1 024 records now, -1 ... -2 % of the 16 384-ray step; 256-record segments: +8 %; profiles/round6_train_ab.txt)
#ifndef T2N_ACC_TARGET_SEGS_APP
#define T2N_ACC_TARGET_SEGS_APP 256
#endif
#ifndef T2N_ACC_SEG_MIN
#define T2N_ACC_SEG_MIN 1024
#endif
constexpr unsigned kAccTargetSegsApp = T2N_ACC_TARGET_SEGS_APP;   // accumulate work items aimed at: whole rounds over 256 CUs
constexpr unsigned kAccGrid = 2048;   // accumulate workgroups launched (grid-stride over the segment list)

struct BinGeom { int tw[3], before[3], total; };
static BinGeom bin_geom(const FactorSet& S) {
    BinGeom g;
    int t = 0;
    for (int k = 0; k < 3; ++k) {
        g.tw[k] = (S.W[k] + kBinTile) / kBinTile;              // cell + 1 in [0, W]
        g.before[k] = t;
        t += g.tw[k] * ((S.H[k] + kBinTile) / kBinTile);
    }
    g.total = t;
    return g;
}
// tile ids (global over the three planes) of a sample; axis a of the grid pairs with coordinate a (see sample_axes)
__device__ __forceinline__ void bin_keys(const FactorSet& S, const BinGeom& G, float xn, float yn, float zn, int key[3]) {
    const int cx = (axis_cell(xn, S.W[0]) + 1) / kBinTile, cy = (axis_cell(yn, S.H[0]) + 1) / kBinTile,
              cz = (axis_cell(zn, S.H[1]) + 1) / kBinTile;
    key[0] = G.before[0] + cy * G.tw[0] + cx;
    key[1] = G.before[1] + cz * G.tw[1] + cx;
    key[2] = G.before[2] + cz * G.tw[2] + cy;
}
// Density: 3-D blocks of kBlk^3 cells (k_bwd_den_block). The three density pairs share their axes (plane k spans two of them, line k the
// third), so ONE key per sample serves all six gradients. cell + 1 lies in [0, size]: (size + kBlk) / kBlk blocks per axis.
constexpr int kBlk = 15;          // cells per block edge: taps reach one further, 16 per axis = the MFMA tile edge
struct BlockGeom { int nb[3]; int size[3]; int total; int copies; };
static BlockGeom block_geom(const FactorSet& S) {
    BlockGeom g;
    g.size[0] = S.W[0]; g.size[1] = S.H[0]; g.size[2] = S.H[1];
    for (int a = 0; a < 3; ++a) g.nb[a] = (g.size[a] + kBlk) / kBlk;
    g.total = g.nb[0] * g.nb[1] * g.nb[2];
    g.copies = kBinCopies;   // privatised histogram copies (every ray starts in the camera's block)
    return g;
}
// the density pairs really share their axes (TensorVMSplit: plane k = grid[mat1] x grid[mat0], line k = grid[vec])
static bool block_geom_ok(const FactorSet& S) {
    return S.C == 16 && S.W[1] == S.W[0] && S.L[2] == S.W[0] && S.W[2] == S.H[0] && S.L[1] == S.H[0] && S.H[2] == S.H[1] && S.L[0] == S.H[1];
}
__device__ __forceinline__ int block_key(const BlockGeom& G, float xn, float yn, float zn) {
    const int bx = (axis_cell(xn, G.size[0]) + 1) / kBlk, by = (axis_cell(yn, G.size[1]) + 1) / kBlk, bz = (axis_cell(zn, G.size[2]) + 1) / kBlk;
    return (bz * G.nb[1] + by) * G.nb[0] + bx;
}
// Runs of equal keys along the 64 lanes (consecutive samples of a ray stay in a tile for many steps): the first lane of a
// run is its leader and reserves for the whole run. Must be called by all 64 lanes.
__device__ __forceinline__ void run_leader(int key, int lane, bool& leader, int& runlen, int& ll) {
    const int prev = __shfl_up(key, 1);
    leader = (lane == 0) | (key != prev);
    const unsigned long long lm = __ballot(leader);
    const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
    const unsigned long long higher = lm & ~upto;
    const int nxt = higher ? (__ffsll((long long)higher) - 1) : 64;
    runlen = nxt - lane;
    ll = 63 - __clzll((long long)(lm & upto));
}

struct BwdMarchArgs {
    FieldDev F;
    GradSet gden;
    const float* rays; long long n_rays; int ray_stride; int n_samples; int npad;
    const float* jitter; const float* sigma; const int4* ray_app; const float4* app_rgb; const float4* rgb_raw;
    const float* d_rgb; const float* d_depth; const float* d_w;
    float4* go;   // [rows] dL/d(pre-sigmoid rgb) per appearance row
    unsigned list_cap; TilePrefix tp; int add_bg;
    const BwdPlan* plan;   // when set: the tile prefix comes from device memory (T2N_FLAG_DEVICE_ROWS)
    // BIN: dL/dfeature per sample goes to gfeat [n_rays, N] (aliases sigma) and the (plane, tile) histogram is counted
    float* gfeat; unsigned* hist; BlockGeom geom;
    // LOSS (fused training step): the driver's loss (t2n_loss.hip, text2nerf_main.py:559-575) evaluated here instead of by a kernel of its
    // own between forward and backward — the weights and sample depths it needs are recomputed by this kernel anyway, so the forward
    // materialises neither, and d_rgb / d_depth / d_weights never exist in memory. rgb / depth: the composite's outputs; part: the
    // per-workgroup partial sums [ceil(n_rays / 4)][3] of k_train_loss (same grouping, same order: the reported losses are bit-identical)
    const float* rgb; const float* depth; const float* rgb_t; const float* depth_t; float w_depth, w_trans, delta; float* loss_part;
};

// BIN = false: scatter with sliding windows + global atomics (any grid). BIN = true: first pass of the tile-binned scatter.
// COUNT = false (fused step): the block histogram is counted BEFORE the backward, from the forward's windows (k_den_count on the GEMM
// stream beside the shade kernel), for every in-box sample whatever its gradient: this kernel then only leaves dL/dfeature
template <bool TRAIN, bool BIN, bool LOSS = false, bool COUNT = true>
__global__ __launch_bounds__(256) void k_bwd_march(const BwdMarchArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float* __restrict__ Sg = smem + (size_t)wid * 4 * a.npad;   // sigma
    float* __restrict__ Al = Sg + a.npad;                       // alpha
    float* __restrict__ Tw = Al + a.npad;                       // transmittance before the sample
    float* __restrict__ Gw = Tw + a.npad;                       // dL/dw, later dL/dfeature
    const FieldDev& F = a.F;
    const long long r = (long long)blockIdx.x * 4 + wid;
    if (!LOSS && r >= a.n_rays) return;
    const bool live_ray = r < a.n_rays;
    const long long rs = live_ray ? r : 0;
    const int4 ra = a.ray_app[rs];
    const int first = ra.w & 2047, Lw = live_ray ? (ra.w >> 11) : 0;
    if (!LOSS && Lw <= 0) return;
    const int N = a.n_samples;
    const Ray ray = load_ray(F, a.rays + rs * a.ray_stride, a.ray_stride);
    const float u = (TRAIN && !F.ztab) ? a.jitter[rs] : 0.f;   // NDC: `jitter` is the depth table
    // upstream gradients of this ray; clamp(0,1) passes gradient on the closed interval
    const float4 rr = a.rgb_raw[rs];
    float d_r, d_g, d_b, gd;
    float e_rgb = 0.f, e_dep = 0.f, dt = 0.f;
    if constexpr (LOSS) {
        // k_train_loss's per-ray arithmetic (same expressions, same order)
        const float R = (float)a.n_rays;
        const float x0 = a.rgb[rs * 3 + 0] - a.rgb_t[rs * 3 + 0], x1 = a.rgb[rs * 3 + 1] - a.rgb_t[rs * 3 + 1], x2 = a.rgb[rs * 3 + 2] - a.rgb_t[rs * 3 + 2];
        d_r = 2.f * x0 / (3.f * R); d_g = 2.f * x1 / (3.f * R); d_b = 2.f * x2 / (3.f * R);
        const float q = lane == 0 ? x0 * x0 : (lane == 1 ? x1 * x1 : (lane == 2 ? x2 * x2 : 0.f));
        e_rgb = wave_sum(q);
        dt = a.depth_t[rs];
        float dep = a.depth[rs];
        const bool bad = dep != dep;
        if (bad) dep = 0.f;
        const float dd = dep - dt;
        gd = bad ? 0.f : 2.f * a.w_depth * dd / R;
        e_dep = dd * dd;
    } else {
        d_r = a.d_rgb[r * 3 + 0]; d_g = a.d_rgb[r * 3 + 1]; d_b = a.d_rgb[r * 3 + 2];
        gd = a.d_depth[r];
    }
    const float gr = (rr.x >= 0.f && rr.x <= 1.f) ? d_r : 0.f;
    const float gg = (rr.y >= 0.f && rr.y <= 1.f) ? d_g : 0.f;
    const float gb = (rr.z >= 0.f && rr.z <= 1.f) ? d_b : 0.f;
    const float bg = a.add_bg ? 1.f : 0.f;
    float m_lane = 0.f;   // LOSS: sum of the masked weights of the samples this lane saw (sample index = first + lane mod 64)

    // ---- forward recompute (same arithmetic as k_march pass C) + dL/dw ---------------------------------------------------
    float carry = 1.f;
    unsigned run = 0;
    for (int base = 0; base < Lw; base += 64) {
        const int j = base + lane, i = first + j;
        float sg = 0.f, z = 0.f, dist = 0.f;
        if (j < Lw) {
            sg = a.sigma[r * N + i];
            z = sample_z<TRAIN>(F, ray, i, u);
            if (i < N - 1) dist = sample_z<TRAIN>(F, ray, i + 1, u) - z;
        }
        const float d = scaled_dist(F, ray, dist);
        const float alpha = 1.f - expf((-sg) * d);
        const float f = (1.f - alpha) + 1e-10f;
        const float incl = wave_scan_mul(f, lane);
        float excl = __shfl_up(incl, 1);
        if (lane == 0) excl = 1.f;
        const float T = carry * excl;
        const float w = alpha * T;
        carry = carry * __shfl(incl, 63);
        const bool app = (j < Lw) & (w > F.thres);
        const unsigned long long bal = __ballot(app);
        float cr = 0.f, cg = 0.f, cb = 0.f;
        if (app) {
            const unsigned slot = (unsigned)ra.x + run + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            const float4 c = a.app_rgb[slot];
            cr = c.x; cg = c.y; cb = c.z;
            // MLP heads: rgb = sigmoid(o): dL/do = dL/drgb * rgb (1 - rgb), dL/drgb_sample = g_c * w. SH / RGB heads end without a
            // sigmoid: their backward (k_simple_head_bwd) takes dL/drgb itself
            const bool sig = F.shading != T2N_SHADE_SH && F.shading != T2N_SHADE_RGB;
            const unsigned row = a.plan ? slot_to_row(slot, a.list_cap, a.plan->tp) : slot_to_row(slot, a.list_cap, a.tp);
            if (!a.plan || row < a.plan->rows) a.go[row] =
                make_float4(gr * w * (sig ? cr * (1.f - cr) : 1.f), gg * w * (sig ? cg * (1.f - cg) : 1.f),
                            gb * w * (sig ? cb * (1.f - cb) : 1.f), 0.f);
        }
        run += (unsigned)__popcll(bal);
        if (j < Lw) {
            float G = gr * (cr - bg) + gg * (cg - bg) + gb * (cb - bg) + gd * (z - ray.last);
            if constexpr (LOSS) m_lane += ((z - dt) + a.delta < 0.f) ? w : 0.f;
            else if (a.d_w) G += a.d_w[r * N + i];
            Sg[j] = sg; Al[j] = alpha; Tw[j] = T; Gw[j] = G;
        }
    }
    float gw = 0.f;
    if constexpr (LOSS) {
        // m_r = mean_n(w [z - depth_t + delta < 0]): k_train_loss's lane n mod 64 summed the samples this kernel's lane (n - first) mod 64
        // saw, in the same (ascending) order — moved to that lane, the wave sum is k_train_loss's bit for bit
        float m = __shfl(m_lane, (lane - first) & 63);
        if (!live_ray) m = 0.f;
        m = wave_sum(m) / (float)N;
        gw = (2.f * a.w_trans * m / (float)a.n_rays) / (float)N;
        __shared__ float red[4][3];
        if (lane == 0) { red[wid][0] = live_ray ? e_rgb : 0.f; red[wid][1] = live_ray ? e_dep : 0.f; red[wid][2] = live_ray ? m * m : 0.f; }
        __syncthreads();
        if (threadIdx.x < 3) a.loss_part[(size_t)blockIdx.x * 3 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (Lw <= 0) return;
    }
    wave_lds_sync();

    // ---- reverse scan: dL/dalpha_i = G_i T_i - (sum_{k>i} G_k w_k) / f_i ---------------------------------------------------
    float suffix = 0.f;
    const int nchunk = (Lw + 63) / 64;
    for (int c = nchunk - 1; c >= 0; --c) {
        const int j = c * 64 + lane, i = first + j;
        float sg = 0.f, al = 0.f, T = 0.f, G = 0.f, dist = 0.f;
        if (j < Lw) {
            sg = Sg[j]; al = Al[j]; T = Tw[j]; G = Gw[j];
            const float z = sample_z<TRAIN>(F, ray, i, u);
            if (i < N - 1) dist = sample_z<TRAIN>(F, ray, i + 1, u) - z;
            if constexpr (LOSS) G += ((z - dt) + a.delta < 0.f) ? gw : 0.f;   // d_weights of TransMittanceLoss_mask, never materialised
        }
        const float val = G * (al * T);
        float s = val;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float t = __shfl_down(s, o);
            if (lane + o < 64) s += t;
        }
        const float after = suffix + (s - val);
        suffix += __shfl(s, 0);
        if (j < Lw) {
            const float f = (1.f - al) + 1e-10f;
            const float dalpha = G * T - after / f;
            const float dsigma = dalpha * scaled_dist(F, ray, dist) * (1.f - al);
            // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x)), from the kept sigma. As -expm1(-sigma): in empty space sigma is
            // softplus(-10) = 4.5e-5, and 1.f - expf(-4.5e-5f) carries the rounding of a value next to 1 (6e-8 absolute = 1.3e-3 of the
            // result) into every density gradient — what the round-3 / round-4 fuzz campaigns saw on nearly empty batches (seeds 3027,
            // 4295: all six density tensors 6e-4 ... 1.6e-3 off the float64 oracle, both scatters alike; tools/experiments/fuzz_scatter_diag.py)
            const float dact = F.act == T2N_ACT_RELU ? (sg > 0.f ? 1.f : 0.f) : -expm1f(-sg);
            Gw[j] = dsigma * dact;
        }
    }
    wave_lds_sync();

    if constexpr (BIN) {
        // ---- dL/dfeature of every live sample -> gfeat; count the samples of each block bin -----------------------------------
        const unsigned copy = (unsigned)(r >> 2) & (unsigned)(a.geom.copies - 1);
        for (int base = 0; base < Lw; base += 64) {
            const int j = base + lane, i = first + j;
            int key = -1;
            if (j < Lw) {
                float gf = Gw[j];
                float xn, yn, zn;
                const float z = sample_z<TRAIN>(F, ray, i, u);
                bool ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
                if (F.alpha && ok) ok = alpha_pass(F, ray, z);
                if constexpr (COUNT) { if (ok && gf != 0.f) key = block_key(a.geom, xn, yn, zn); else gf = 0.f; }
                else if (!ok) gf = 0.f;
                a.gfeat[r * N + i] = gf;
            }
            if constexpr (COUNT) {
                bool leader; int runlen, ll;
                run_leader(key, lane, leader, runlen, ll);
                if (leader && key >= 0) atomicAdd(&a.hist[copy * (unsigned)a.geom.total + (unsigned)key], (unsigned)runlen);
            }
        }
        return;
    }

    // ---- scatter: 16 lanes per sample (one channel each); each 16-lane group walks a CONTIGUOUS quarter of the window
    // so that successive samples are neighbours along the ray and the sliding windows merge their shared texels ----------
    const int ch = lane & 15, sl = lane >> 4;
    const int Q = (Lw + 3) >> 2;
    PlaneWin pw0, pw1, pw2;
    LineWin lw0, lw1, lw2;
    pw0.x = kWinEmpty; pw0.y = 0; pw0.a00 = pw0.a01 = pw0.a10 = pw0.a11 = 0.f; pw1 = pw0; pw2 = pw0;
    lw0.r = kWinEmpty; lw0.a0 = lw0.a1 = 0.f; lw1 = lw0; lw2 = lw0;
    for (int t = 0; t < Q; ++t) {
        const int j = sl * Q + t, i = first + j;
        if (j < Lw) {
            const float gf = Gw[j];
            float xn, yn, zn;
            const float z = sample_z<TRAIN>(F, ray, i, u);
            bool ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
            if (F.alpha && ok) ok = alpha_pass(F, ray, z);
            if (ok && gf != 0.f) {
                scatter_win<0>(F.den, a.gden, 16, ch, xn, yn, zn, gf, pw0, lw0);
                scatter_win<1>(F.den, a.gden, 16, ch, xn, yn, zn, gf, pw1, lw1);
                scatter_win<2>(F.den, a.gden, 16, ch, xn, yn, zn, gf, pw2, lw2);
            }
        }
    }
    plane_flush(pw0, a.gden.plane[0], F.den.W[0], 16, ch); plane_flush(pw1, a.gden.plane[1], F.den.W[1], 16, ch);
    plane_flush(pw2, a.gden.plane[2], F.den.W[2], 16, ch);
    line_flush(lw0, a.gden.line[0], 16, ch); line_flush(lw1, a.gden.line[1], 16, ch); line_flush(lw2, a.gden.line[2], 16, ch);
}


// (2) single workgroup: exclusive scan of the [tile][copy] histogram -> absolute cursors (in place), tile starts, and the
// segment list of the accumulate pass. The histogram is pulled into LDS with coalesced loads first (LDS = true) so the
// per-thread serial runs do not chain ~70 dependent global round trips.
// inclusive scan of one value per thread over a 1024-thread workgroup: shuffles inside the waves, the 16 wave totals through LDS (two
// barriers; the Hillis-Steele form over LDS took twenty). `total` = the sum over the workgroup. sh: 17 words.
__device__ __forceinline__ unsigned block_scan_1024(unsigned v, unsigned* sh, unsigned& total) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = (unsigned)__shfl_up((int)v, o);
        if (lane >= o) v += u;
    }
    __syncthreads();            // the previous use of sh is over
    if (lane == 63) sh[w] = v;
    __syncthreads();
    unsigned before = 0, all = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const unsigned x = sh[q]; all += x; before += q < w ? x : 0u; }
    total = all;
    return v + before;
}
// (2a) the privatised copies of every bin's counter -> within-bin offsets (in place) and the bin's total. The counters lie copy-major
// ([copy][bin]): side by side, the copies of a hot bin share one 128-byte line, and same-LINE atomics serialise at the L2 like
// same-address ones (k_bwd_bin 122 -> 65 us, k_bwd_march 124 -> 64 us per C3 iteration when four copies moved apart; 32 copies here).
__global__ __launch_bounds__(256) void k_bin_reduce(unsigned* __restrict__ hist, int n_bins, int copies, unsigned* __restrict__ bin_total) {
    if (copies == 32) {   // lane = copy: 32 lanes per bin, a shuffle scan instead of a 32-step loop per thread (1 083 appearance bins are 5 workgroups of loops)
        const int j = blockIdx.x * 8 + (threadIdx.x >> 5), c = threadIdx.x & 31;
        const bool ok = j < n_bins;
        const unsigned v = ok ? hist[(size_t)c * n_bins + j] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const unsigned u = (unsigned)__shfl_up((int)incl, o, 32);
            if (c >= o) incl += u;
        }
        if (ok) {
            hist[(size_t)c * n_bins + j] = incl - v;
            if (c == 31) bin_total[j] = incl;
        }
        return;
    }
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n_bins) return;
    unsigned run = 0;
    for (int c = 0; c < copies; ++c) {
        const unsigned v = hist[(size_t)c * n_bins + j];
        hist[(size_t)c * n_bins + j] = run;
        run += v;
    }
    bin_total[j] = run;
}
// (2b) single workgroup: exclusive scan of the bin totals -> bin starts (+ sentinel) and the segment list of the accumulate pass. A record's
// position is start[bin] + the running within-bin cursor of its copy.
// copies > 0 (few bins: the appearance tiles): step (2a) runs here too — thread j walks bin j's privatised counters `hist_copies`
// [copy][bin] (offsets in place) — instead of as a launch of its own in front of this one
template <bool LDS>
__global__ __launch_bounds__(1024) void k_bin_scan(unsigned* hist, int n_tiles, unsigned* tile_start, int4* segs, unsigned* nseg_out,
                                                   unsigned seg_cap, unsigned seg_size_in, unsigned target_segs, unsigned seg_min,
                                                   unsigned* hist_copies, int copies) {
    extern __shared__ unsigned sh_hist[];
    __shared__ unsigned sh[17];
    const int t = threadIdx.x;
    const int n = n_tiles;
    if (copies > 0) {
        for (int j = t; j < n; j += 1024) {
            unsigned run = 0;
            for (int c = 0; c < copies; ++c) {
                const unsigned v = hist_copies[(size_t)c * n + j];
                hist_copies[(size_t)c * n + j] = run;
                run += v;
            }
            if (LDS) sh_hist[j] = run; else hist[j] = run;
        }
        __syncthreads();
    } else if (LDS) {
        for (int j = t; j < n; j += 1024) sh_hist[j] = hist[j];
        __syncthreads();
    }
    unsigned* H = LDS ? sh_hist : hist;
    const int per = (n + 1023) / 1024;
    const int b = t * per, e = min(n, b + per);
    unsigned sum = 0;
    for (int i = b; i < e; ++i) sum += H[i];
    unsigned total;
    unsigned run = block_scan_1024(sum, sh, total) - sum;
    for (int i = b; i < e; ++i) {
        const unsigned c = H[i];
        H[i] = run;
        run += c;
    }
    __syncthreads();
    if (LDS) for (int j = t; j < n; j += 1024) hist[j] = sh_hist[j];
    // tile starts (+ sentinel); non-empty tiles; segment size such that the accumulate pass gets ~target_segs work items
    // (a whole number of rounds over the CUs: with fixed 8192-record segments 1079 items ran as 4.2 rounds on 256 CUs)
    const int pt = (n_tiles + 1023) / 1024;
    const int tb = t * pt, te = min(n_tiles, tb + pt);
    unsigned ne = 0;
    for (int j = tb; j < te; ++j) {
        const unsigned s0 = H[j], s1 = j + 1 < n_tiles ? H[j + 1] : total;
        tile_start[j] = s0;
        ne += s1 > s0 ? 1u : 0u;
    }
    if (t == 0) tile_start[n_tiles] = total;
    unsigned seg_size = seg_size_in;
    if (seg_size == 0) {   // (uniform over the workgroup)
        unsigned nonempty;
        (void)block_scan_1024(ne, sh, nonempty);
        const unsigned room = target_segs > nonempty + 64 ? target_segs - nonempty / 2 : 64;   // every tile ends in a partial segment
        seg_size = (total + room - 1) / room;
        seg_size = (seg_size + 63) / 64 * 64;
        seg_size = seg_size < seg_min ? seg_min : (seg_size > 16384 ? 16384 : seg_size);
    }
    unsigned ns = 0;
    for (int j = tb; j < te; ++j) {
        const unsigned s0 = H[j], s1 = j + 1 < n_tiles ? H[j + 1] : total;
        ns += (s1 - s0 + seg_size - 1) / seg_size;
    }
    unsigned nsegs;
    unsigned si = block_scan_1024(ns, sh, nsegs) - ns;
    for (int j = tb; j < te; ++j) {
        const unsigned s0 = H[j], s1 = j + 1 < n_tiles ? H[j + 1] : total;
        for (unsigned s = s0; s < s1; s += seg_size) {
            if (si < seg_cap) segs[si] = make_int4(j, (int)s, (int)min(s1, s + seg_size), 0);
            ++si;
        }
    }
    if (t == 0) *nseg_out = min(nsegs, seg_cap);
}
static void launch_bin_scan(unsigned* hist, int n_tiles, int copies, unsigned* bin_total, unsigned* tile_start, int4* segs, unsigned* nseg, unsigned seg_cap,
                            unsigned seg_size, unsigned target_segs, unsigned seg_min, hipStream_t s, bool one_launch = false) {
    const bool merged = one_launch && (long long)n_tiles * copies <= 65536;   // (a single workgroup walks every counter: small tables only)
    if (!merged) hipLaunchKernelGGL(k_bin_reduce, dim3((unsigned)(copies == 32 ? (n_tiles + 7) / 8 : (n_tiles + 255) / 256)), dim3(256), 0, s, hist, n_tiles, copies, bin_total);
    const size_t lds = (size_t)n_tiles * 4;
    if (lds <= 150 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)k_bin_scan<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
        hipLaunchKernelGGL((k_bin_scan<true>), dim3(1), dim3(1024), lds, s, bin_total, n_tiles, tile_start, segs, nseg, seg_cap, seg_size, target_segs, seg_min,
                           hist, merged ? copies : 0);
    } else {
        hipLaunchKernelGGL((k_bin_scan<false>), dim3(1), dim3(1024), 0, s, bin_total, n_tiles, tile_start, segs, nseg, seg_cap, seg_size, target_segs, seg_min,
                           hist, merged ? copies : 0);
    }
}

// (3) per ray (same ray -> wave -> copy map as the counting pass): write the records into their tiles' runs
struct BinArgs {
    FieldDev F; BlockGeom geom;
    const float* rays; long long n_rays; int ray_stride; int n_samples;
    const float* jitter; const float* gfeat; const int4* ray_app; unsigned* cursor; const unsigned* tile_start; float4* recs;
};
// the counting pass on its own (fused step): every in-box sample of the forward's window, same ray -> wave -> copy map as k_bwd_bin<.., true>
template <bool TRAIN>
__device__ __forceinline__ void den_count_body(const BinArgs& a, unsigned bx) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const FieldDev& F = a.F;
    const long long r = (long long)bx * 4 + wid;
    if (r >= a.n_rays) return;
    const int4 ra = a.ray_app[r];
    const int first = ra.w & 2047, Lw = ra.w >> 11;
    if (Lw <= 0) return;
    const Ray ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
    const float u = (TRAIN && !F.ztab) ? a.jitter[r] : 0.f;
    const unsigned copy = (unsigned)(r >> 2) & (unsigned)(a.geom.copies - 1);
    for (int base = 0; base < Lw; base += 64) {
        const int j = base + lane, i = first + j;
        int key = -1;
        if (j < Lw) {
            float xn, yn, zn;
            const float z = sample_z<TRAIN>(F, ray, i, u);
            bool ok = sample_point<TRAIN>(F, ray, z, xn, yn, zn);
            if (F.alpha && ok) ok = alpha_pass(F, ray, z);
            if (ok) key = block_key(a.geom, xn, yn, zn);
        }
        bool leader; int runlen, ll;
        run_leader(key, lane, leader, runlen, ll);
        if (leader && key >= 0) atomicAdd(&a.cursor[copy * (unsigned)a.geom.total + (unsigned)key], (unsigned)runlen);
    }
}
// ALL: a record for every in-box sample (k_den_count's predicate; a zero gradient adds zeros) instead of every non-zero gradient
template <bool TRAIN, bool ALL = false>
__global__ __launch_bounds__(256) void k_bwd_bin(const BinArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const FieldDev& F = a.F;
    const long long r = (long long)blockIdx.x * 4 + wid;
    if (r >= a.n_rays) return;
    const int4 ra = a.ray_app[r];
    const int first = ra.w & 2047, Lw = ra.w >> 11;
    if (Lw <= 0) return;
    const int N = a.n_samples;
    const Ray ray = load_ray(F, a.rays + r * a.ray_stride, a.ray_stride);
    const float u = (TRAIN && !F.ztab) ? a.jitter[r] : 0.f;   // NDC: `jitter` is the depth table
    const unsigned copy = (unsigned)(r >> 2) & (unsigned)(a.geom.copies - 1);
    for (int base = 0; base < Lw; base += 64) {
        const int j = base + lane, i = first + j;
        int key = -1;
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Lw) {
            rec.w = a.gfeat[r * N + i];
            if constexpr (ALL) {
                const float z = sample_z<TRAIN>(F, ray, i, u);
                bool ok = sample_point<TRAIN>(F, ray, z, rec.x, rec.y, rec.z);
                if (F.alpha && ok) ok = alpha_pass(F, ray, z);
                if (ok) key = block_key(a.geom, rec.x, rec.y, rec.z);
            } else if (rec.w != 0.f) {
                const float z = sample_z<TRAIN>(F, ray, i, u);
                (void)sample_point<TRAIN>(F, ray, z, rec.x, rec.y, rec.z);
                key = block_key(a.geom, rec.x, rec.y, rec.z);
            }
        }
        bool leader; int runlen, ll;
        run_leader(key, lane, leader, runlen, ll);
        unsigned pos = 0;
        if (leader && key >= 0) pos = a.tile_start[key] + atomicAdd(&a.cursor[copy * (unsigned)a.geom.total + (unsigned)key], (unsigned)runlen);
        pos = __shfl(pos, ll) + (unsigned)(lane - ll);
        if (key >= 0) a.recs[pos] = rec;
    }
}

// (4d) density: accumulate one segment of one block's records. The density feature is sum_k sum_c P_k[c](two axes) L_k[c](third axis) with
// ONE gradient g per sample for all channels and pairs, so every gradient of the block is a contraction of the same scalar field
//     G[z][y][x] = sum over samples of g * (trilinear weight of the sample at corner (x, y, z)):
//     dP_0[y][x][c] = sum_z G L_0[z][c],  dP_1[z][x][c] = sum_y G L_1[y][c],  dP_2[z][y][c] = sum_x G L_2[x][c],
//     dL_0[z][c] = sum_yx G P_0[y][x][c], dL_1[y][c] = sum_zx G P_1[z][x][c], dL_2[x][c] = sum_zy G P_2[z][y][c].
// Phase 1: lane = record, eight ds_add_f64 into the 16^3 corner field of the block (padded strides: the three contractions read it along
// different axes). Phase 2: the six contractions as 16x16x4 fp32 MFMAs (A = G as fp32, B = line rows from LDS / plane texels straight
// from global memory), 16 x 16 result tiles flushed with global atomics. The per-plane tile form of this scatter (one record per sample
// AND plane, 6 x 16 LDS atomics per record) spent 430 us per C3 iteration at the LDS atomic rate.
constexpr int kBlkE = kBlk + 1;                    // taps per axis
constexpr int kGsy = kBlkE + 1, kGsz = kBlkE * kGsy + 1;   // strides of G in doubles (x: 1): bank-conflict-free along every axis
constexpr int kGsize = kBlkE * kGsz;
constexpr int kDenThreads = 512;
// records per segment: a block's six result tiles are flushed (12 288 + 1 536 global atomics) once per SEGMENT, and the splat itself is
// cheap (~8 ns per 1000 records per CU), so blocks are split only when they hold very many records (the camera's block of a C3
// batch: 57 000 of 2.5 M): 4 000 segments of ~700 records took 262 us, 1 300 of up to 16 384 take 73
constexpr unsigned kDenSeg = 16384;
struct DenBlockArgs {
    FactorSet S; GradSet G; BlockGeom geom; const int4* segs; const unsigned* nseg; const float4* recs;
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// -DDB_PROF (timing-only build): cycle sums per region, lane 0 of every wave: [0] zero + line rows, [1] barrier, [2] splat, [3] barrier,
// [4] plane contractions + flush, [5] line contractions + flush, [6] segments x waves, [7] records (wave 0 counts them)
#ifdef DB_PROF
__device__ unsigned long long g_db_prof[16];
#define DB_T(i) do { const unsigned long long tn = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) db_acc[i] += tn - db_t; db_t = tn; } while (0)
#else
#define DB_T(i) do {} while (0)
#endif
__global__ __launch_bounds__(kDenThreads) void k_bwd_den_block(const DenBlockArgs a) {
    __shared__ double Gs[kGsize];
    __shared__ float Ls[3][kBlkE][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const unsigned nseg = *a.nseg;
    const int X = a.geom.size[0], Y = a.geom.size[1], Z = a.geom.size[2];
#ifdef DB_PROF
    unsigned long long db_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, db_t = 0;
#endif
    for (unsigned seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        if (seg != blockIdx.x) __syncthreads();   // the previous segment's contractions have finished reading G and the lines
#ifdef DB_PROF
        db_t = __builtin_readcyclecounter();
        if (lane == 0) db_acc[6] += 1;
#endif
        const int4 sg = a.segs[seg];
#ifdef DB_PROF
        if (tid == 0) db_acc[7] += (unsigned long long)(sg.z - sg.y);
#endif
        const int bx = sg.x % a.geom.nb[0], byz = sg.x / a.geom.nb[0], by = byz % a.geom.nb[1], bz = byz / a.geom.nb[1];
        const int ox = bx * kBlk - 1, oy = by * kBlk - 1, oz = bz * kBlk - 1;   // cell of local (0, 0, 0)
        for (int i = tid; i < kGsize; i += kDenThreads) Gs[i] = 0.0;
        if (tid < 3 * kBlkE * 4) {   // line rows of the block: L_0 over z, L_1 over y, L_2 over x
            const int k = tid / (kBlkE * 4), rq = tid % (kBlkE * 4), row = rq >> 2, q = rq & 3;
            const int g = (k == 0 ? oz : (k == 1 ? oy : ox)) + row, n = k == 0 ? Z : (k == 1 ? Y : X);
            const float* __restrict__ Ln = k == 0 ? a.S.line[0] : (k == 1 ? a.S.line[1] : a.S.line[2]);
            const float4 v = (g >= 0 && g < n) ? *reinterpret_cast<const float4*>(Ln + (size_t)g * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&Ls[k][row][q * 4]) = v;
        }
#ifdef DB_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        DB_T(0);
        __syncthreads();
        DB_T(1);
        // ---- phase 1: the trilinear splat of the segment's records ---------------------------------------------------------------
        for (int i = sg.y + tid; i < sg.z; i += kDenThreads) {
            const float4 p = a.recs[i];
            const Axis ax = axis_taps(p.x, X), ay = axis_taps(p.y, Y), az = axis_taps(p.z, Z);
            const int lx = axis_cell(p.x, X) - ox, ly = axis_cell(p.y, Y) - oy, lz = axis_cell(p.z, Z) - oz;
            const int b = lz * kGsz + ly * kGsy + lx;
            const float g0 = p.w * az.w0, g1 = p.w * az.w1;
            const float w00 = ay.w0 * ax.w0, w01 = ay.w0 * ax.w1, w10 = ay.w1 * ax.w0, w11 = ay.w1 * ax.w1;
            atomicAdd(&Gs[b], (double)(g0 * w00)); atomicAdd(&Gs[b + 1], (double)(g0 * w01));
            atomicAdd(&Gs[b + kGsy], (double)(g0 * w10)); atomicAdd(&Gs[b + kGsy + 1], (double)(g0 * w11));
            atomicAdd(&Gs[b + kGsz], (double)(g1 * w00)); atomicAdd(&Gs[b + kGsz + 1], (double)(g1 * w01));
            atomicAdd(&Gs[b + kGsz + kGsy], (double)(g1 * w10)); atomicAdd(&Gs[b + kGsz + kGsy + 1], (double)(g1 * w11));
        }
#ifdef DB_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        DB_T(2);
        __syncthreads();
        DB_T(3);
        // ---- phase 2: contractions. MFMA 16x16x4 fp32: A lane (i, kk) = A[i][kk], B lane (j, kk) = B[kk][j], D lane (j, q) regs r = D[4q + r][j]
        const int li = lane & 15, kk = lane >> 4;
        // planes: 48 tiles of 16 cells x 16 channels (pair k, tile t): wave w takes tiles w, w + 8, ...
        for (int pt = w; pt < 48; pt += 8) {
            const int k = pt >> 4, t = pt & 15;
            // pair 0: tile = y row t, cell i = x, contraction over z; pair 1: tile = z row t, cell i = x, over y; pair 2: tile = z row t, cell i = y, over x
            const int abase = k == 0 ? t * kGsy + li : (k == 1 ? t * kGsz + li : t * kGsz + li * kGsy);
            const int astep = k == 0 ? kGsz : (k == 1 ? kGsy : 1);
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int c = 4 * ks + kk;
                d = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Gs[abase + c * astep], Ls[k][c][li], d, 0, 0, 0);
            }
            // D[cell 4 kk + r][channel li]
            const int W = k == 2 ? Y : X, H = k == 0 ? Y : Z;
            const int gy = (k == 0 ? oy : oz) + t, gx0 = (k == 2 ? oy : ox) + 4 * kk;
            float* __restrict__ gP = k == 0 ? a.G.plane[0] : (k == 1 ? a.G.plane[1] : a.G.plane[2]);
            if (gy >= 0 && gy < H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gx = gx0 + r;
                    if (d[r] != 0.f && gx >= 0 && gx < W) atomicAdd(gP + ((size_t)gy * W + gx) * 16 + li, d[r]);
                }
            }
        }
#ifdef DB_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        DB_T(4);
        // lines: pair k = w / 2 for waves 0..5, half (w & 1) of the 256 plane cells each: D[line row][channel] += sum_cells G P
        if (w < 6) {
            const int k = w >> 1, half = w & 1;
            const int W = k == 2 ? Y : X, H = k == 0 ? Y : Z;
            const float* __restrict__ P = k == 0 ? a.S.plane[0] : (k == 1 ? a.S.plane[1] : a.S.plane[2]);
            const int py0 = k == 0 ? oy : oz, px0 = k == 2 ? oy : ox;
            // A[i = line row][cell (u, v)]: pair 0 row z, cell (y, x); pair 1 row y, cell (z, x); pair 2 row x, cell (z, y)
            const int arow = k == 0 ? li * kGsz : (k == 1 ? li * kGsy : li);
            const int au = k == 0 ? kGsy : kGsz, av = k == 2 ? kGsy : 1;
            f32x4_t d = {0.f, 0.f, 0.f, 0.f};
            // (all 32 plane values in flight at once instead of four groups of eight: 59.8 K -> 43.2 K cycles for this region in the DB_PROF
            // accounting and nothing off the step: profiles/round6_train_ab.txt)
            for (int s8 = 0; s8 < 32; s8 += 8) {
                float bv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cell = (half * 32 + s8 + e) * 4 + kk, u = cell >> 4, v = cell & 15;
                    const int gy = py0 + u, gx = px0 + v;
                    bv[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? P[((size_t)gy * W + gx) * 16 + li] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int cell = (half * 32 + s8 + e) * 4 + kk, u = cell >> 4, v = cell & 15;
                    d = __builtin_amdgcn_mfma_f32_16x16x4f32((float)Gs[arow + u * au + v * av], bv[e], d, 0, 0, 0);
                }
            }
            const int n = k == 0 ? Z : (k == 1 ? Y : X), g0 = (k == 0 ? oz : (k == 1 ? oy : ox)) + 4 * kk;
            float* __restrict__ gL = k == 0 ? a.G.line[0] : (k == 1 ? a.G.line[1] : a.G.line[2]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = g0 + r;
                if (d[r] != 0.f && g >= 0 && g < n) atomicAdd(gL + (size_t)g * 16 + li, d[r]);
            }
        }
#ifdef DB_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        DB_T(5);
    }
#ifdef DB_PROF
    if (lane == 0) for (int i = 0; i < 8; ++i) if (db_acc[i]) atomicAdd(&g_db_prof[i], db_acc[i]);
#endif
}
#ifdef DB_PROF
}  // namespace t2n
extern "C" int t2n_debug_db_prof(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(t2n::g_db_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(t2n::g_db_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
namespace t2n {
#endif

// (4) accumulate one segment of one tile in LDS. Measured on gfx950 (tools/experiments/lds_atomic_bench.hip): ds_add_f32
// retires ~1 lane per 3 clocks (194 clk per wave instruction, whatever the addresses) while ds_add_f64 / ds_add_u64 run at
// 8-9 clk conflict-free — so the tile and line accumulators are DOUBLES. Per wave and batch of 64 records: lane = record
// computes the taps once (cell offset, line row, six weights, g) into a wave-private LDS table; then 16 lanes per record
// (one channel each, C/16 channel groups in turn) read their record's table entry (broadcast), the staged plane / line
// values, and issue 4 + 2 ds_add_f64.
constexpr int kAccThreads = 512;
struct TileAccumArgs {
    FactorSet S; GradSet G; BinGeom geom; const int4* segs; const unsigned* nseg; const float4* recs;
    // appearance: rec.w holds the activation row (int bits) and the gradient of channel c of pair K is gx[row * gx_ld + K * CT + c];
    // density (gx == NULL): rec.w is dL/dfeature itself, the same for all channels
    const float* gx; int gx_ld;
};
// CT = channels of the factor set (texel stride); a workgroup covers the 16 channels [coff, coff + 16).
// CG = channels per workgroup (16: one workgroup per CU with 138 KB of LDS at 300^3; 8 with NT = 256 threads: 69 KB — two workgroups per CU,
// one accumulates while the other stages / zeroes / flushes)
// -DTA_PROF (timing-only build, tools/r6_ta_prof.sh): cycle sums per region of k_bwd_tile_accum, lane 0 of every wave, summed over the launch.
// [0] stage + zero, [1] barrier, [2] record tables, [3] gradient loads (forced wait), [4] accumulate loop, [5] barrier, [6] flush,
// [7] segments x waves, [8] batches, [9] records
#ifdef TA_PROF
__device__ unsigned long long g_ta_prof[16];
#define TA_T(i) do { const unsigned long long tn = __builtin_readcyclecounter(); if ((threadIdx.x & 63) == 0) ta_acc[i] += tn - ta_t; ta_t = tn; } while (0)
#define TA_DECL unsigned long long ta_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ta_t = __builtin_readcyclecounter()
#else
#define TA_T(i) do {} while (0)
#endif
#ifndef T2N_TA_PIPE
#define T2N_TA_PIPE 1
#endif
template <int CT, int K, int CG, int NT>
__device__ __forceinline__ void tile_accum_records(const TileAccumArgs& a, const int4 sg, int x0, int y0, int coff, const float* Pv,
                                                   double* Pa, const float* Lv, double* La, float4* tab
#ifdef TA_PROF
                                                   , unsigned long long (&ta_acc)[10], unsigned long long& ta_t
#endif
                                                   ) {
    constexpr int C = CG;   // 64 / CG records per instruction, CG instructions per batch of 64 records
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, ch = lane & (CG - 1), sub = lane / CG;
    constexpr int NW = NT / 64;
    const int W = a.S.W[K], H = a.S.H[K], L = a.S.L[K];
    const int per = ((sg.z - sg.y + NW * 64 - 1) / (NW * 64)) * 64;     // records per wave, whole batches
    const int rb = sg.y + wid * per, re = min(sg.z, rb + per);
    float4* T = tab + wid * 64 * 3;
#if T2N_TA_PIPE
    // the gradient rows of a batch are loaded ONE BATCH AHEAD (the row of record ri sits in lane ri's record register: a bpermute), the
    // records two batches ahead — every load UNCONDITIONAL, at clamped addresses, so that the compiler can count the loads in flight: a first
    // form with the loads under divergent conditions waited for vmcnt(0) in front of every use and was slower than no prefetch at all
    // (0.830 -> 0.897 ms per step). Kernel under the train loop: 314.7 -> 292.3 us at 16 384 rays, 89.3 -> 85.4 at 2 048 (tools/r6_kernel_ab.sh);
    // the per-batch round trip for these rows was a third of the kernel's time (profiles/round6_tile_accum_accounting.txt).
    // T2N_TA_PIPE=0: the loads at the top of their own batch (both scatter forms pass a gradient array: a.gx is never NULL here)
    if (rb >= re) return;
    const int last = re - 1;
    float4 p = a.recs[min(rb + lane, last)];
    float4 pn = a.recs[min(rb + 64 + lane, last)];
    auto issue_g = [&](const float4& rec, float (&g)[CG]) {
#pragma unroll
        for (int q = 0; q < CG; ++q) {
            const int row = __shfl(__float_as_int(rec.w), sub * CG + q);
            g[q] = a.gx[(size_t)row * a.gx_ld + K * CT + coff + ch];
        }
    };
    float gcur[CG];
    issue_g(p, gcur);
#else
    float4 p = rb + lane < re ? a.recs[rb + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
#endif
    for (int b0 = rb; b0 < re; b0 += 64) {
        float gx, gy, gv;
        plane_line_coords<K>(p.x, p.y, p.z, gx, gy, gv);
        const Axis ax = axis_taps(gx, W), ay = axis_taps(gy, H), al = axis_taps(gv, L);
        const int lx = axis_cell(gx, W) - x0, ly = axis_cell(gy, H) - y0, fl = axis_cell(gv, L);
        const bool live = b0 + lane < re;
        T[lane * 3 + 0] = make_float4(__int_as_float(live ? (ly * (kBinTile + 1) + lx) * C : 0), __int_as_float(live ? (fl + 1) * C : 0),
                                      live ? p.w : 0.f, live ? 1.f : 0.f);
        T[lane * 3 + 1] = make_float4(ay.w0 * ax.w0, ay.w0 * ax.w1, ay.w1 * ax.w0, ay.w1 * ax.w1);
        T[lane * 3 + 2] = make_float4(al.w0, al.w1, 0.f, 0.f);
#if T2N_TA_PIPE
        const float4 pnn = a.recs[min(b0 + 128 + lane, last)];
        wave_lds_sync();
        TA_T(2);
        float gnx[CG];
        issue_g(pn, gnx);
        float gpre[CG];
#pragma unroll
        for (int q = 0; q < CG; ++q) gpre[q] = gcur[q];
#else
        if (b0 + 64 + lane < re) p = a.recs[b0 + 64 + lane];   // next batch in flight while this one is accumulated
        wave_lds_sync();
        TA_T(2);
        // appearance: the CG per-channel gradients this lane needs for the batch, all in flight before the accumulate loop
        // (a dependent global load per record inside the loop serialised ~1 us round trips). Loading them one batch AHEAD (rows by
        // bpermute from the next batch's record registers, records two batches ahead) measured slower: 0.830 -> 0.897 ms per 16 384-ray step
        float gpre[CG];
        if (a.gx) {
#pragma unroll
            for (int q = 0; q < CG; ++q) {
                const float4 t0 = T[(sub * CG + q) * 3];
                gpre[q] = t0.w != 0.f ? a.gx[(size_t)__float_as_int(t0.z) * a.gx_ld + K * CT + coff + ch] : 0.f;
            }
        }
#endif
#ifdef TA_PROF
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TA_T(3);
        if (lane == 0) { ta_acc[8] += 1; ta_acc[9] += (unsigned long long)min(64, re - b0); }
#endif
#pragma unroll
        for (int q = 0; q < CG; ++q) {
            // the RPI records an instruction handles are CG apart in the batch: neighbours in the list are neighbouring
            // samples of ONE ray and would hit the same accumulator words (same-address LDS atomics serialise)
            const int ri = sub * CG + q;
            const float4 t0 = T[ri * 3], wp = T[ri * 3 + 1], wl = T[ri * 3 + 2];
            const int c00 = __float_as_int(t0.x) + ch, c01 = c00 + C, c10 = c00 + (kBinTile + 1) * C, c11 = c10 + C;
            const int r0 = __float_as_int(t0.y) + ch, r1 = r0 + C;
#if T2N_TA_PIPE
            const float g2 = t0.w != 0.f ? gpre[q] : 0.f;
#else
            const float g2 = a.gx ? gpre[q] : t0.z;
#endif
            if (g2 != 0.f) {
                float pv = Pv[c00] * wp.x;
                pv = fmaf(Pv[c01], wp.y, pv); pv = fmaf(Pv[c10], wp.z, pv); pv = fmaf(Pv[c11], wp.w, pv);
                const float lv = fmaf(Lv[r1], wl.y, Lv[r0] * wl.x);
                const float gp = g2 * lv, gl = g2 * pv;
                atomicAdd(&Pa[c00], (double)(gp * wp.x)); atomicAdd(&Pa[c01], (double)(gp * wp.y));
                atomicAdd(&Pa[c10], (double)(gp * wp.z)); atomicAdd(&Pa[c11], (double)(gp * wp.w));
                atomicAdd(&La[r0], (double)(gl * wl.x)); atomicAdd(&La[r1], (double)(gl * wl.y));
            }
        }
        wave_lds_sync();
#ifdef TA_PROF
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        TA_T(4);
#if T2N_TA_PIPE
        p = pn; pn = pnn;
#pragma unroll
        for (int q = 0; q < CG; ++q) gcur[q] = gnx[q];
#endif
    }
}
// grid: (segments, CT / 16)
template <int CT, int CG = 16, int NT = kAccThreads>
__global__ __launch_bounds__(NT) void k_bwd_tile_accum(const TileAccumArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const unsigned nseg = *a.nseg;
    const int coff = blockIdx.y * CG;
#ifdef TA_PROF
    TA_DECL;
#endif
    for (unsigned seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
    if (seg != blockIdx.x) __syncthreads();   // the previous segment's flush has finished reading the accumulators
#ifdef TA_PROF
    ta_t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) ta_acc[7] += 1;
#endif
    const int4 sg = a.segs[seg];
    const int k = sg.x >= a.geom.before[2] ? 2 : (sg.x >= a.geom.before[1] ? 1 : 0);
    const int tile = sg.x - a.geom.before[k];
    int W, H, L, tw; const float* __restrict__ P; const float* __restrict__ Ln; float* gP; float* gL;
    // (no dynamic indexing of the by-value argument struct: that would move it to scratch)
    if (k == 0) { W = a.S.W[0]; H = a.S.H[0]; L = a.S.L[0]; tw = a.geom.tw[0]; P = a.S.plane[0]; Ln = a.S.line[0]; gP = a.G.plane[0]; gL = a.G.line[0]; }
    else if (k == 1) { W = a.S.W[1]; H = a.S.H[1]; L = a.S.L[1]; tw = a.geom.tw[1]; P = a.S.plane[1]; Ln = a.S.line[1]; gP = a.G.plane[1]; gL = a.G.line[1]; }
    else { W = a.S.W[2]; H = a.S.H[2]; L = a.S.L[2]; tw = a.geom.tw[2]; P = a.S.plane[2]; Ln = a.S.line[2]; gP = a.G.plane[2]; gL = a.G.line[2]; }
    const int x0 = (tile % tw) * kBinTile - 1, y0 = (tile / tw) * kBinTile - 1;   // texel of local (0, 0)
    constexpr int C = CG, T1 = kBinTile + 1, TP = T1 * T1 * C, C4 = C / 4;
    double* Pa = reinterpret_cast<double*>(smem);
    double* La = Pa + TP;
    float* Pv = reinterpret_cast<float*>(La + (size_t)(L + 2) * C);
    float* Lv = Pv + TP;
    float4* tab = reinterpret_cast<float4*>(Lv + (size_t)(L + 2) * C);
    // staging: EVERY load of the segment's plane tile and line issued before the first LDS store (a thread has up to 3 + 4 of them), the
    // accumulators zeroed underneath. (The form `cond ? *ptr : zero4` made the compiler select between the global pointer and the ADDRESS
    // of the zero constant — in scratch memory — and load through it: a flat load + a full wait per element, and the zero-fill loop read
    // its zeros from scratch with a wait per store: staging was 43 % (16 384 rays) to 56 % (2 048 rays) of this kernel's time, TA_PROF
    // accounting in profiles/round6_tile_accum_accounting.txt.)
    constexpr int PIT = (TP / 4 + NT - 1) / NT, LIT = 4;
    float4 pvr[PIT], lvr[LIT];
#pragma unroll
    for (int it = 0; it < PIT; ++it) {
        const int idx = threadIdx.x + it * NT;
        const int cell = idx / C4, q = idx - cell * C4, ly = cell / T1, lx = cell - ly * T1, y = y0 + ly, x = x0 + lx;
        pvr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < TP / 4 && x >= 0 && x < W && y >= 0 && y < H) pvr[it] = *reinterpret_cast<const float4*>(P + ((size_t)y * W + x) * CT + coff + q * 4);
    }
    const int nl4 = (L + 2) * C4;
#pragma unroll
    for (int it = 0; it < LIT; ++it) {
        const int idx = threadIdx.x + it * NT;
        const int row = idx / C4 - 1, q = idx % C4;
        lvr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nl4 && row >= 0 && row < L) lvr[it] = *reinterpret_cast<const float4*>(Ln + (size_t)row * CT + coff + q * 4);
    }
    for (int idx = threadIdx.x; idx < (TP + (L + 2) * C) / 2; idx += NT) reinterpret_cast<float4*>(Pa)[idx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int it = 0; it < PIT; ++it) { const int idx = threadIdx.x + it * NT; if (idx < TP / 4) reinterpret_cast<float4*>(Pv)[idx] = pvr[it]; }
#pragma unroll
    for (int it = 0; it < LIT; ++it) { const int idx = threadIdx.x + it * NT; if (idx < nl4) reinterpret_cast<float4*>(Lv)[idx] = lvr[it]; }
    for (int idx = threadIdx.x + LIT * NT; idx < nl4; idx += NT) {     // (lines beyond LIT x NT quads: grids past ~500)
        const int row = idx / C4 - 1, q = idx % C4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row >= 0 && row < L) v = *reinterpret_cast<const float4*>(Ln + (size_t)row * CT + coff + q * 4);
        reinterpret_cast<float4*>(Lv)[idx] = v;
    }
#ifdef TA_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TA_T(0);
    __syncthreads();
    TA_T(1);
    if (k == 0) tile_accum_records<CT, 0, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab, ta_acc, ta_t);
    else if (k == 1) tile_accum_records<CT, 1, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab, ta_acc, ta_t);
    else tile_accum_records<CT, 2, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab, ta_acc, ta_t);
    __syncthreads();
    TA_T(5);
#else
    __syncthreads();
    if (k == 0) tile_accum_records<CT, 0, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab);
    else if (k == 1) tile_accum_records<CT, 1, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab);
    else tile_accum_records<CT, 2, CG, NT>(a, sg, x0, y0, coff, Pv, Pa, Lv, La, tab);
    __syncthreads();
#endif
    for (int idx = threadIdx.x; idx < TP; idx += NT) {
        const float v = (float)Pa[idx];
        if (v != 0.f) {
            const int cell = idx / C, c = idx - cell * C, ly = cell / T1, lx = cell - ly * T1, y = y0 + ly, x = x0 + lx;
            if (x >= 0 && x < W && y >= 0 && y < H) atomicAdd(gP + ((size_t)y * W + x) * CT + coff + c, v);
        }
    }
    for (int idx = threadIdx.x; idx < L * C; idx += NT) {
        const float v = (float)La[C + idx];
        if (v != 0.f) atomicAdd(gL + (size_t)(idx / C) * CT + coff + (idx % C), v);
    }
#ifdef TA_PROF
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TA_T(6);
#endif
    }   // segments
#ifdef TA_PROF
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < 10; ++i) if (ta_acc[i]) atomicAdd(&g_ta_prof[i], ta_acc[i]);
#endif
}
#ifdef TA_PROF
}  // namespace t2n
extern "C" int t2n_debug_ta_prof(unsigned long long out[16], int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(t2n::g_ta_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(t2n::g_ta_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
namespace t2n {
#endif

// Appearance records: one per appearance-list entry and plane. PASS 0 counts, PASS 1 writes (same lane -> entry -> copy map).
// Row r of the activation buffers <-> list entry: tile r / 32 belongs to sub-list l (tp.t[l] <= tile < tp.t[l + 1]).
struct AppBinArgs {
    FactorSet S; BinGeom geom; const float4* app_pos; const unsigned* counters; unsigned list_cap; TilePrefix tp; long long rows;
    unsigned* hist; const unsigned* tile_start; float4* recs;
    const BwdPlan* plan;   // when set: tile prefix and row count from device memory (rows = the capacity the grid was sized for)
};
template <int PASS>
__device__ __forceinline__ void app_bin_body(const AppBinArgs& a, unsigned bx) {
    const int lane = threadIdx.x & 63;
    const long long wv = (long long)bx * 4 + (threadIdx.x >> 6);
    const long long row = wv * 64 + lane;
    const long long rows = a.plan ? (long long)a.plan->rows : a.rows;
    if (wv * 64 >= rows) return;
    // (a reference picked between the device plan and the by-value argument: flat loads, but `tp.t[l]` below stays ONE load at a computed
    // address — a private copy turned it into a dynamically indexed register array and k_app_bin<1> from 12 into 33 us)
    const TilePrefix& tp = a.plan ? a.plan->tp : a.tp;
    int key[3] = {-1, -1, -1};
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) {
        const unsigned tile = (unsigned)(row >> 5);
        int l = 0;
#pragma unroll
        for (int q = 1; q < kLists; ++q) l += (tp.t[q] <= tile) ? 1 : 0;
        const unsigned slot = (unsigned)(row - (long long)tp.t[l] * 32);
        unsigned cnt = a.counters[l * kCounterStride];
        if (cnt > a.list_cap) cnt = a.list_cap;
        if (slot < cnt) {
            const float4 p = a.app_pos[(size_t)l * a.list_cap + slot];
            rec = make_float4(p.x, p.y, p.z, __int_as_float((int)row));
            bin_keys(a.S, a.geom, p.x, p.y, p.z, key);
        }
    }
    const unsigned copy = (unsigned)wv & (kBinCopies - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        bool leader; int runlen, ll;
        run_leader(key[k], lane, leader, runlen, ll);
        if (PASS == 0) {
            if (leader && key[k] >= 0) atomicAdd(&a.hist[copy * (unsigned)a.geom.total + (unsigned)key[k]], (unsigned)runlen);
        } else {
            unsigned pos = 0;
            if (leader && key[k] >= 0) pos = a.tile_start[key[k]] + atomicAdd(&a.hist[copy * (unsigned)a.geom.total + (unsigned)key[k]], (unsigned)runlen);
            pos = __shfl(pos, ll) + (unsigned)(lane - ll);
            if (key[k] >= 0) a.recs[pos] = rec;
        }
    }
}
template <int PASS>
__global__ __launch_bounds__(256) void k_app_bin(const AppBinArgs a) { app_bin_body<PASS>(a, blockIdx.x); }
// fused step: both counting passes that need nothing but the forward's outputs as ONE launch — the appearance records' tile histogram
// (workgroups [0, nb_app)) and the density samples' block histogram (the rest)
__global__ __launch_bounds__(256) void k_early_bins(const AppBinArgs ab, const BinArgs ca, unsigned nb_app) {
    if (blockIdx.x < nb_app) app_bin_body<0>(ab, blockIdx.x);
    else den_count_body<true>(ca, blockIdx.x - nb_app);
}
// doubles: tile + line accumulators; floats: staged values; per-wave tap tables (64 records x 3 float4)
static size_t tile_accum_lds(int C, int Lmax, int threads = kAccThreads) {
    const size_t cells = (size_t)(kBinTile + 1) * (kBinTile + 1) * C + (size_t)(Lmax + 2) * C;
    return cells * 8 + cells * 4 + (size_t)(threads / 64) * 64 * 3 * 16;
}

// appearance taps: re-gather and scatter-add with gX [rows,144]
struct AppScatterArgs {
    FieldDev F; GradSet gapp; const float4* app_pos; const unsigned* counters; unsigned list_cap; const float* gxapp;
};
template <int K>
__device__ __forceinline__ void app_scatter_plane(const AppScatterArgs& a, int lane, unsigned base, unsigned count, unsigned row0) {
    // 16 lanes per sample; each 16-lane group walks 8 CONSECUTIVE list entries (neighbouring samples of a ray)
    const int ch = lane & 15, sl = lane >> 4;
    PlaneWin pw[3];
    LineWin lw[3];
#pragma unroll
    for (int cg = 0; cg < 3; ++cg) { pw[cg].x = kWinEmpty; pw[cg].y = 0; pw[cg].a00 = pw[cg].a01 = pw[cg].a10 = pw[cg].a11 = 0.f; lw[cg].r = kWinEmpty; lw[cg].a0 = lw[cg].a1 = 0.f; }
    for (int t = 0; t < 8; ++t) {
        const int s = sl * 8 + t;
        const unsigned idx = base + (unsigned)s;
        if (idx < count) {
            const float4 p = a.app_pos[idx];
            const float* gr = a.gxapp + (size_t)(row0 + s) * 144 + K * 48 + ch;
#pragma unroll
            for (int cg = 0; cg < 3; ++cg) {
                const float g = gr[cg * 16];
                if (g != 0.f) scatter_win<K>(a.F.app, a.gapp, 48, cg * 16 + ch, p.x, p.y, p.z, g, pw[cg], lw[cg]);
            }
        }
    }
#pragma unroll
    for (int cg = 0; cg < 3; ++cg) {
        plane_flush(pw[cg], a.gapp.plane[K], a.F.app.W[K], 48, cg * 16 + ch);
        line_flush(lw[cg], a.gapp.line[K], 48, cg * 16 + ch);
    }
}
__global__ __launch_bounds__(256) void k_bwd_app_scatter(const AppScatterArgs a) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned cnt_l = 0;
    if (lane < kLists) { cnt_l = a.counters[lane * kCounterStride]; if (cnt_l > a.list_cap) cnt_l = a.list_cap; }
    unsigned incl = (cnt_l + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned ntiles = __shfl(incl, kLists - 1);
    for (unsigned tile = blockIdx.x * 4u + wid; tile < ntiles; tile += gridDim.x * 4u) {
        const int li = (int)__popcll(__ballot((lane < kLists) & (incl <= tile)));
        const unsigned before = li ? __shfl(incl, li - 1) : 0u;
        const unsigned lbase = (unsigned)li * a.list_cap;
        const unsigned base = lbase + (tile - before) * 32u;
        const unsigned count = lbase + __shfl(cnt_l, li);
        app_scatter_plane<0>(a, lane, base, count, tile * 32u);
        app_scatter_plane<1>(a, lane, base, count, tile * 32u);
        app_scatter_plane<2>(a, lane, base, count, tile * 32u);
    }
}

// channel-last gradient buffer [HW][C] -> += reference layout [1,C,H,W], through an LDS tile of 64 texels; the (up to) 12 factor
// tensors of a backward call in ONE launch (twelve launches of mostly tiny grids cost ~9 us each)
struct RelayoutAddMulti { const float* src[12]; float* dst[12]; int C[12]; long long HW[12]; unsigned block0[13]; int count; };
__global__ __launch_bounds__(256) void k_relayout_add(const RelayoutAddMulti a) {
    __shared__ float tile[64 * 49];
    int t = 0;
#pragma unroll 1
    for (int q = 1; q < a.count; ++q) t += (a.block0[q] <= blockIdx.x) ? 1 : 0;
    const float* __restrict__ src = a.src[t];
    float* dst = a.dst[t];
    const int C = a.C[t];
    const long long HW = a.HW[t];
    const long long pix0 = (long long)(blockIdx.x - a.block0[t]) * 64;
    const int ld = C + 1;
    const long long n = (HW - pix0 < 64 ? HW - pix0 : 64) * C;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int px = i / C, c = i - px * C;
        tile[px * ld + c] = src[pix0 * C + i];
    }
    __syncthreads();
    const int lp = threadIdx.x & 63, cs = threadIdx.x >> 6;
    if (pix0 + lp < HW)
        for (int c = cs; c < C; c += 4) dst[(long long)c * HW + pix0 + lp] += tile[lp * ld + c];
}

// The library's two side streams are PROCESS-wide (per device), created together at first use: HIP maps streams onto a handful of
// hardware queues in creation order, and chains that share a queue run one behind the other. With a pair of streams per field, a field
// created late in a process (after other fields, the caller's copy streams, ...) could get both of its side streams — or one of them and
// the caller's — on one queue: the same 8 192-ray step took 0.79 ms in a long-running bench process and 0.60 ms in a fresh one.
static std::mutex g_stream_mutex;
static hipStream_t g_side[16][2];
int shared_side_streams(void** a, void** b) {
    int dev = 0;
    T2N_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { set_error("device ordinal %d beyond the side-stream table", dev); return T2N_ERR_UNSUPPORTED; }
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    if (!g_side[dev][0]) {
        hipStream_t x, y;
        T2N_HIP(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        T2N_HIP(hipStreamCreateWithFlags(&y, hipStreamNonBlocking));
        g_side[dev][0] = x; g_side[dev][1] = y;
    }
    if (a && !*a) *a = (void*)g_side[dev][0];
    if (b && !*b) *b = (void*)g_side[dev][1];
    return T2N_OK;
}

// Activation / gradient rows of the backward pass. Buffers whose lifetimes do not overlap (or that are rewritten
// element-in-place by the same thread) share storage: g1 over h1, g0 over h0, gx over xpe, gf over feat32, gX over x144.
struct BwdCarve { size_t x144, feat32, h0, h1, go, xpe, part, gpack, hist, bin_total, tile_start, nseg, segs, recs, a_hist, a_bin_total, a_tile_start, a_nseg, a_segs, a_recs, plan, total; unsigned seg_cap, a_seg_cap; };
static size_t al256(size_t x) { return (x + 255) / 256 * 256; }
static BwdCarve bwd_carve(int64_t rows, int64_t n_rays, int n_samples, int n_tiles, int n_blocks, int k0 = 351, bool rows_kept = false) {   // k0: inputs of MLP layer 0; n_blocks: density bins x copies; rows_kept: the activation rows live in the forward's workspace (fused step): none here
    BwdCarve c;
    size_t o = 0;
    const size_t R = (size_t)rows;
    const size_t RA = rows_kept ? 0 : R;
    c.x144 = o; o = al256(o + RA * 144 * 4);
    c.feat32 = o; o = al256(o + RA * 32 * 4);
    c.h0 = o; o = al256(o + RA * 128 * 4);
    c.h1 = o; o = al256(o + RA * 128 * 4);
    c.go = o; o = al256(o + R * 16);
    // (fused step: the encoding is never materialised — the rows hold G0 [128] | GF [32] | GX [144] only)
    c.xpe = o; o = al256(o + R * (size_t)(rows_kept ? 304 : ((k0 + 3) & ~3)) * 4);
    c.part = o; o = al256(o + tn_part_bytes(rows, k0));
    c.gpack = o; o = al256(o + (gemm_h_pack_bytes(k0) > mlp_bwd_ss_pack_bytes() ? gemm_h_pack_bytes(k0) : mlp_bwd_ss_pack_bytes()));   // packed W^T operands of the input-gradient GEMMs (t2n_gemm_h.hip / t2n_mlp_bwd_ss.hip)
    // block-binned density scatter: worst case one record per sample
    const size_t cap = (size_t)n_rays * (size_t)n_samples;
    c.seg_cap = (unsigned)(cap / kDenSeg + (size_t)n_blocks + 1);
    c.hist = o; o = al256(o + (size_t)n_blocks * 4);
    c.bin_total = o; o = al256(o + (size_t)n_blocks * 4);
    c.tile_start = o; o = al256(o + ((size_t)n_blocks + 1) * 4);
    c.nseg = o; o = al256(o + 4);
    c.segs = o; o = al256(o + (size_t)c.seg_cap * 16);
    c.recs = o; o = al256(o + cap * 16);
    // the same for the appearance samples (one record per activation row and plane)
    c.a_seg_cap = (unsigned)(3 * R / kBinSegApp + (size_t)n_tiles + 1);
    c.a_hist = o; o = al256(o + (size_t)n_tiles * kBinCopies * 4);
    c.a_bin_total = o; o = al256(o + (size_t)n_tiles * 4);
    c.a_tile_start = o; o = al256(o + ((size_t)n_tiles + 1) * 4);
    c.a_nseg = o; o = al256(o + 4);
    c.a_segs = o; o = al256(o + (size_t)c.a_seg_cap * 16);
    c.a_recs = o; o = al256(o + 3 * R * 16);
    c.plan = o; o = al256(o + sizeof(BwdPlan));
    c.total = o;
    return c;
}

// the 12 channel-last gradient buffers are slices of ONE allocation (density planes, density lines, appearance planes, appearance
// lines; 256-B aligned slices): one memset, one in-place all-reduce
// [density planes 0..2 | density lines 0..2 | appearance planes 0..2 | appearance lines 0..2]: the density gradients (final when the
// density scatter is done, ~0.5 ms before the appearance scatter) are ONE contiguous prefix — the first bucket of a data-parallel
// all-reduce that overlaps the rest of the backward (t2n_field_wait_density_grads). *den_bytes: the prefix's size.
static size_t grad_layout(const t2n_field* f, size_t (&off)[12], size_t* den_bytes = nullptr) {
    const int* g = f->desc.grid;
    size_t o = 0;
    for (int q = 0; q < 4; ++q) {
        if (q == 2 && den_bytes) *den_bytes = o;
        for (int k = 0; k < 3; ++k) {
            const size_t HW = (size_t)g[mat1(k)] * g[mat0(k)], L = (size_t)g[vecm(k)];
            const size_t sz[4] = {HW * 16 * 4, L * 16 * 4, HW * 48 * 4, L * 48 * 4};
            off[q * 3 + k] = o; o += (sz[q] + 255) / 256 * 256;
        }
    }
    return o;
}
static void grad_slices(t2n_field* f, const size_t (&off)[12]) {
    char* b = (char*)f->gbuf_all;
    for (int k = 0; k < 3; ++k) {
        f->gbuf_den_plane[k] = (float*)(b + off[0 + k]); f->gbuf_den_line[k] = (float*)(b + off[3 + k]);
        f->gbuf_app_plane[k] = (float*)(b + off[6 + k]); f->gbuf_app_line[k] = (float*)(b + off[9 + k]);
    }
}
static int ensure_grad_buffers(t2n_field* f) {
    if (f->gbuf_all) return T2N_OK;
    size_t off[12];
    const size_t o = grad_layout(f, off);
    T2N_HIP(hipMalloc((void**)&f->gbuf_all, o));
    f->gbuf_bytes = o;
    f->gbuf_external = false;
    grad_slices(f, off);
    return T2N_OK;
}

// The plan of a backward that reads no count on the host: one wave turns the forward's sub-list counters into the tile prefix and the
// row count, clipped to the capacity; the forward's statement that it kept the activation rows with exactly this capacity is checked here
// too (a mismatch leaves rows = 0: nothing of the appearance branch runs, overflow says why)
// host_rec (pinned host memory of the field, may be NULL): words 0..7 = the rows the last eight such backwards NEEDED (before clipping),
// word 8 = how many of them overflowed their capacity so far, word 9 = sequence number of the newest record + 1. A caller sizes its
// next capacity from it without ever waiting: every record arrives, late at worst.
__global__ __launch_bounds__(64) void k_bwd_plan(const unsigned* __restrict__ counters, unsigned list_cap, unsigned rows_cap, unsigned kept_rows,
                                                 BwdPlan* __restrict__ plan, unsigned* host_rec, unsigned seq) {
    const int lane = threadIdx.x;
    unsigned cnt = lane < kLists ? counters[lane * kCounterStride] : 0u;
    if (cnt > list_cap) cnt = list_cap;
    unsigned incl = (cnt + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned excl = incl - (cnt + 31u) / 32u;
    if (lane < kLists) plan->tp.t[lane] = excl;
    const unsigned total = __shfl(incl, kLists - 1);
    if (lane == 0) {
        plan->tp.t[kLists] = total;
        const bool stated = counters[kKeptMagicWord] == kKeptMagic && counters[kKeptRowsWord] == kept_rows;
        const unsigned rows = total * 32u;
        const unsigned ovf = (!stated || rows > rows_cap) ? 1u : 0u;
        plan->overflow = ovf;
        plan->rows = !stated ? 0u : (rows < rows_cap ? rows : rows_cap);
        if (host_rec) {
            volatile unsigned* h = host_rec;
            h[seq & 7u] = rows;
            if (ovf) h[8] = h[8] + 1u;      // (plan kernels of one field run one after the other)
            __threadfence_system();
            h[9] = seq + 1u;
        }
    }
}

}  // namespace t2n

using namespace t2n;

// The appearance counts of a KEEP_CTX forward travel to pinned host memory right behind the march kernel (ctx_counts_post, called by
// t2n_render_forward) with an event behind the copy: the backward waits for THAT event, not for the stream — the loss ops queued
// between forward and backward keep the GPU busy while the host sizes and launches the backward. Slots are matched by workspace
// pointer; a forward from another library build / an evicted slot falls back to the stream-draining read.
// The ring is process-wide (the C-ABI's t2n_render_ctx_rows takes the workspace, not the field): slots are keyed by (device,
// workspace), guarded by a mutex (fields driven from several host threads), and a slot's event and pinned buffer belong to the device
// that was current when they were made — a slot taken over by another device re-creates them there. 32 slots: a forward whose slot
// has been evicted before its backward only falls back to the stream-draining read.
struct CtxSlot { const void* ws; hipEvent_t ev; unsigned* host; bool valid; int dev; };
constexpr int kCtxSlots = 32;
static CtxSlot g_ctx[kCtxSlots];
static unsigned g_ctx_next = 0;
static std::mutex g_ctx_mutex;
int t2n::ctx_counts_post(const void* ws, const unsigned* counters_dev, hipStream_t s) {
    int dev = 0;
    T2N_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    for (auto& q : g_ctx) if (q.valid && q.ws == ws && q.dev == dev) q.valid = false;
    CtxSlot& c = g_ctx[g_ctx_next++ % kCtxSlots];
    if (c.host && c.dev != dev) {
        (void)hipEventDestroy(c.ev);
        (void)hipHostFree(c.host);
        c.host = nullptr;
    }
    if (!c.host) {
        T2N_HIP(hipHostMalloc((void**)&c.host, sizeof(unsigned) * kLists * kCounterStride, hipHostMallocDefault));
        T2N_HIP(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
        c.dev = dev;
    }
    c.ws = ws;
    { const int rc = post_counts(counters_dev, c.host, s); if (rc) return rc; }   // (a device store into the pinned buffer: no blit kernel)
    T2N_HIP(hipEventRecord(c.ev, s));
    c.valid = true;
    return T2N_OK;
}

static int read_counts(const void* fwd_ws, int64_t n_rays, int n_samples, hipStream_t s, unsigned counts[kLists], TilePrefix* tp,
                       int64_t* rows, unsigned* kept_rows_stated = nullptr, bool consume = false) {
    const Carve c = carve_workspace(n_rays, n_samples, true, false);
    unsigned raw[kLists * kCounterStride];
    CtxSlot* slot = nullptr;
    int dev = 0;
    T2N_HIP(hipGetDevice(&dev));
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        for (auto& q : g_ctx) if (q.valid && q.ws == fwd_ws && q.dev == dev) slot = &q;
        if (slot) ev = slot->ev;
    }
    if (slot) {
        T2N_HIP(hipEventSynchronize(ev));     // (outside the lock: other threads keep posting)
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        if (slot->valid && slot->ws == fwd_ws && slot->dev == dev) memcpy(raw, slot->host, sizeof(raw));
        else slot = nullptr;                    // taken over meanwhile: read from the device below
    }
    if (!slot) {
        T2N_HIP(hipMemcpyAsync(raw, (const char*)fwd_ws + c.counters, sizeof(raw), hipMemcpyDeviceToHost, s));
        T2N_HIP(hipStreamSynchronize(s));
    }
    for (int l = 0; l < kLists; ++l) counts[l] = raw[l * kCounterStride];
    if (kept_rows_stated) *kept_rows_stated = raw[kKeptMagicWord] == kKeptMagic ? raw[kKeptRowsWord] : 0u;
    if (consume) {   // the backward overwrites the kept rows in place: a second backward on this workspace must recompute
        if (slot) { std::lock_guard<std::mutex> lock(g_ctx_mutex); if (slot->valid && slot->ws == fwd_ws) slot->host[kKeptMagicWord] = 0u; }
        // (the device word is cleared by the caller's setup kernel: a 4-byte hipMemsetAsync is a 6-us fill kernel of its own on the stream)
    }
    unsigned t = 0;
    for (int l = 0; l < kLists; ++l) {
        if (counts[l] > c.list_cap) counts[l] = c.list_cap;
        if (tp) tp->t[l] = t;
        t += (counts[l] + 31u) / 32u;
    }
    if (tp) tp->t[kLists] = t;
    *rows = (int64_t)t * 32;
    return T2N_OK;
}

extern "C" int t2n_render_ctx_rows(const void* fwd_workspace, int64_t n_rays, int n_samples, t2n_stream stream, int64_t* rows) {
    if (!fwd_workspace || !rows || n_rays <= 0 || n_samples <= 0) { set_error("t2n_render_ctx_rows: bad argument"); return T2N_ERR_INVALID; }
    unsigned counts[kLists];
    return read_counts(fwd_workspace, n_rays, n_samples, (hipStream_t)stream, counts, nullptr, rows);
}

// What the T2N_FLAG_DEVICE_ROWS backwards of this field recorded (k_bwd_plan): a plain read of pinned host memory, never waits.
extern "C" int t2n_field_device_rows_record(const t2n_field* f, uint32_t out[10]) {
    if (!f || !out) { set_error("t2n_field_device_rows_record: NULL argument"); return T2N_ERR_INVALID; }
    if (!f->plan_host) { for (int i = 0; i < 10; ++i) out[i] = 0u; return T2N_OK; }
    const volatile unsigned* h = f->plan_host;
    for (int i = 0; i < 10; ++i) out[i] = h[i];
    return T2N_OK;
}

extern "C" size_t t2n_backward_workspace_bytes(const t2n_field* f, int64_t rows, int64_t n_rays, int n_samples) {
    if (!f || n_rays <= 0 || n_samples <= 0) return 0;
    const int k0 = head_is_generic(f->desc.shading) ? head_dims(f->desc).K0 : 351;
    const BlockGeom bg = block_geom(f->dev.den);
    return bwd_carve(rows < 32 ? 32 : rows, n_rays, n_samples, bin_geom(f->dev.den).total, bg.total * bg.copies, k0).total;
}

extern "C" int t2n_render_backward(t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags,
                                   const float* jitter, const float* d_rgb, const float* d_depth, const float* d_weights,
                                   const t2n_field_grads* g, void* fwd_workspace, size_t fwd_workspace_bytes, void* bwd_workspace,
                                   size_t bwd_workspace_bytes, t2n_stream stream) {
    if (!f || !rays || !d_rgb || !d_depth || !g || !fwd_workspace || !bwd_workspace || n_rays <= 0 || ray_stride < 6) {
        set_error("t2n_render_backward: bad argument");
        return T2N_ERR_INVALID;
    }
    if (!f->uploaded) { set_error("t2n_render_backward: field has no uploaded parameters"); return T2N_ERR_STATE; }
    const bool generic = head_is_generic(f->desc.shading);
    // SH / RGB: no head parameters, but the colour gradients still reach basis_mat and the appearance factors
    const bool simple = f->desc.shading == T2N_SHADE_SH || f->desc.shading == T2N_SHADE_RGB;
    if (f->desc.shading != T2N_SHADE_MLP_FEA_NOVIEW && !generic && !simple) { set_error("t2n_render_backward: unknown shading head %d", f->desc.shading); return T2N_ERR_UNSUPPORTED; }
    const int K0 = generic ? head_dims(f->desc).K0 : 351, K0pad = (K0 + 3) & ~3;
    if (!(flags & T2N_FLAG_KEEP_CTX)) { set_error("t2n_render_backward: forward was not run with T2N_FLAG_KEEP_CTX"); return T2N_ERR_STATE; }
    if ((flags & (T2N_FLAG_TRAIN | T2N_FLAG_NDC)) && !jitter) { set_error("t2n_render_backward: train / NDC mode needs the jitter draws / depth table"); return T2N_ERR_INVALID; }
    struct ZtabScope { t2n_field* f; ~ZtabScope() { f->dev.ztab = nullptr; } } ztab_scope{f};
    f->dev.ztab = (flags & T2N_FLAG_NDC) ? jitter : nullptr;
    if (n_samples > 1024) { set_error("t2n_render_backward: n_samples %d > 1024", n_samples); return T2N_ERR_UNSUPPORTED; }
    const Carve c = carve_workspace(n_rays, n_samples, true, false);
    if (c.total > fwd_workspace_bytes) { set_error("t2n_render_backward: forward workspace too small"); return T2N_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    unsigned counts[kLists];
    TilePrefix tp;
    memset(&tp, 0, sizeof(tp));
    int64_t rows = 0;
    unsigned kept_stated = 0;
    int rc = T2N_OK;
    const BinGeom geom = bin_geom(f->dev.den);
    const BlockGeom bgeom = block_geom(f->dev.den);
    // T2N_FLAG_DEVICE_ROWS: nothing is read back. The row CAPACITY is what both workspaces hold (the forward's kept activation rows, the
    // backward's row buffers); k_bwd_plan derives tile prefix and row count on the device and every row-streaming kernel clips to it.
    const bool dev_rows = (flags & T2N_FLAG_DEVICE_ROWS) != 0;
    if (dev_rows) {
        const KeptRows kc = kept_rows(c.total, fwd_workspace_bytes);
        if (generic || simple || gemm_fp32_mode(f) || f->desc.app_dim != 27 || K0 != 351 || kc.rows < 32 || f->dev.app.C != 48) {
            set_error("t2n_render_backward: T2N_FLAG_DEVICE_ROWS needs the fused MLP_Fea_noview head in split-f16 mode and a forward workspace with kept activation rows");
            return T2N_ERR_UNSUPPORTED;
        }
        // EVERY precondition of the device-side plan is checked here, before anything is launched or forked (the checks further
        // down stay as assertions: a refusal there would leave queued kernels and unjoined side streams behind)
        {
            int Lm = 0;
            for (int k = 0; k < 3; ++k) Lm = f->dev.den.L[k] > Lm ? f->dev.den.L[k] : Lm;
            const bool force_atomic_env = getenv("T2N_BWD_ATOMIC_SCATTER") && atoi(getenv("T2N_BWD_ATOMIC_SCATTER")) != 0;
            if (force_atomic_env || tile_accum_lds(16, Lm) > 160 * 1024 || !block_geom_ok(f->dev.den) || (uint64_t)n_rays * n_samples * 3 >= 0x7fffffffull) {
                set_error("t2n_render_backward: T2N_FLAG_DEVICE_ROWS needs the binned scatters (grid lines within the LDS budget)");
                return T2N_ERR_UNSUPPORTED;
            }
            if (!g->mlp_w1 || !g->mlp_w0) { set_error("t2n_render_backward: T2N_FLAG_DEVICE_ROWS needs the weight-gradient tensors of both hidden layers"); return T2N_ERR_INVALID; }
        }
        int64_t lo = 0, hi = (int64_t)kc.rows / 32;     // largest capacity (in tiles) whose backward buffers fit
        while (lo < hi) {
            const int64_t mid = (lo + hi + 1) / 2;
            if (bwd_carve(mid * 32, n_rays, n_samples, geom.total, bgeom.total * bgeom.copies, K0).total <= bwd_workspace_bytes) lo = mid; else hi = mid - 1;
        }
        if (lo < 1) { set_error("t2n_render_backward: backward workspace %zu B holds no row", bwd_workspace_bytes); return T2N_ERR_WORKSPACE; }
        rows = lo * 32;
        kept_stated = kc.rows;       // (checked on the device by k_bwd_plan)
    } else {
        rc = read_counts(fwd_workspace, n_rays, n_samples, s, counts, &tp, &rows, &kept_stated, true);
        if (rc) return rc;
    }
    const int64_t rows_alloc = rows < 32 ? 32 : rows;
    const BwdCarve b = bwd_carve(rows_alloc, n_rays, n_samples, geom.total, bgeom.total * bgeom.copies, K0);
    if (b.total > bwd_workspace_bytes) { set_error("t2n_render_backward: backward workspace %zu B < %zu B", bwd_workspace_bytes, b.total); return T2N_ERR_WORKSPACE; }
    if ((rc = ensure_grad_buffers(f))) return rc;

    char* fw = (char*)fwd_workspace;
    char* bw = (char*)bwd_workspace;
    float* x144 = (float*)(bw + b.x144); float* feat32 = (float*)(bw + b.feat32); float* h0 = (float*)(bw + b.h0);
    float* h1 = (float*)(bw + b.h1); float4* go = (float4*)(bw + b.go); float* xpe = (float*)(bw + b.xpe);
    // the forward kept the activation rows (spare KEEP_CTX workspace, all rows fit): use them in place, skip the recompute
    const KeptRows kr = generic ? KeptRows{0, 0, 0, 0, 0} : kept_rows(c.total, fwd_workspace_bytes);
    // ... and only when the forward SAID it kept them with this very capacity: a caller that hands the backward a larger (pooled) buffer
    // than the forward saw would otherwise read uninitialised activations
    const bool kept = kr.rows >= 32 && (int64_t)kr.rows >= rows_alloc && kept_stated == kr.rows;
    if (kept) { x144 = (float*)(fw + kr.x144); feat32 = (float*)(fw + kr.feat32); h0 = (float*)(fw + kr.h0); h1 = (float*)(fw + kr.h1); }
    float* part = (float*)(bw + b.part);
    float* g1 = h1;      // k_bwd_l2 rewrites each element in place
    float* g0 = h0;      // gemm_nn reads the ReLU mask and writes the masked product at the same element
    float* gx = xpe;     // xpe is dead once dW0 has been accumulated
    float* gf = feat32;  // k_pe_bwd: element-in-place
    float* gxapp = x144; // x144 is dead once dWb has been accumulated
    const float4* app_pos = (const float4*)(fw + c.app_pos);
    const int* app_ray = (const int*)(fw + c.app_ray);
    float4* app_rgb = (float4*)(fw + c.app_rgb);
    const unsigned* counters = (const unsigned*)(fw + c.counters);
    BwdPlan* plan = dev_rows ? (BwdPlan*)(bw + b.plan) : nullptr;
    const unsigned* rows_dev = plan ? &plan->rows : nullptr;
    if (plan) {
        if (!f->plan_host) {
            T2N_HIP(hipHostMalloc((void**)&f->plan_host, 16 * sizeof(unsigned), hipHostMallocDefault));
            for (int i = 0; i < 16; ++i) f->plan_host[i] = 0u;
        }
        hipLaunchKernelGGL(k_bwd_plan, dim3(1), dim3(64), 0, s, counters, c.list_cap, (unsigned)rows, kept_stated, plan, f->plan_host, f->plan_seq++);
        T2N_HIP(hipGetLastError());
    }

    const int* gr = f->desc.grid;
    // a library-owned gradient buffer holds the gradients of THIS call; a caller-owned one (t2n_field_set_grad_buffer) accumulates
    // and is zeroed by its owner
    if (!f->gbuf_external) T2N_HIP(hipMemsetAsync(f->gbuf_all, 0, f->gbuf_bytes, s));
    SetupOps so;   // go, the two bin histograms: zeroed by one kernel once the scatter paths are known (below)
    so.zero(go, (size_t)rows_alloc * 16);
    so.zero(fw + c.counters + kKeptMagicWord * 4, 4);   // the kept rows are consumed by this call (read_counts cleared the host copy)

    // 1. appearance forward recompute with activations kept
    timing_begin(f, T2N_K_BWD_MLP, s);
    if (rows > 0 && !generic && !kept) {
        ShadeCtx ctx{x144, feat32, h0, h1};
        if ((rc = launch_shade_list(f, app_pos, app_ray, rays, ray_stride, counters, c.list_cap, app_rgb, &ctx, s))) return rc;
    } else if (rows > 0 && generic) {   // general heads: features from the shade kernel's first two stages, then the unfused MLP (xpe holds X0)
        ShadeCtx ctx{x144, feat32, nullptr, nullptr};
        if ((rc = launch_shade_list(f, app_pos, app_ray, rays, ray_stride, counters, c.list_cap, app_rgb, &ctx, s, true))) return rc;
        if ((rc = launch_head_forward(f, tp.t, rows, feat32, app_pos, app_ray, rays, ray_stride, counters, c.list_cap, xpe, h0, h1, app_rgb, s))) return rc;
    }
    timing_end(f, T2N_K_BWD_MLP, s);

    // 2. per-ray backward + density scatter
    bool side = false;   // the density scatter was put on the side stream: joined before step 6
    hipStream_t den_stream = s;   // the stream the density gradients are finished on
    bool side_gemm = false;   // the weight-gradient GEMMs were put on the third stream: joined before step 6
    bool packed_early = false;   // k_mlp_bwd_ss's operands were packed in front of k_bwd_march
    bool pack_side = false;      // ... on the third stream: joined in front of k_mlp_bwd_ss
    bool bin = false;
    size_t lds_bin = 0;
    {
        BwdMarchArgs a;
        a.F = f->dev;
        for (int k = 0; k < 3; ++k) { a.gden.plane[k] = f->gbuf_den_plane[k]; a.gden.line[k] = f->gbuf_den_line[k]; }
        a.rays = rays; a.n_rays = n_rays; a.ray_stride = ray_stride; a.n_samples = n_samples; a.npad = (n_samples + 63) & ~63;
        a.jitter = jitter; a.sigma = (const float*)(fw + c.sigma); a.ray_app = (const int4*)(fw + c.ray_app);
        a.app_rgb = app_rgb; a.rgb_raw = (const float4*)(fw + c.rgb_raw);
        a.d_rgb = d_rgb; a.d_depth = d_depth; a.d_w = d_weights; a.go = go; a.list_cap = c.list_cap; a.tp = tp; a.plan = plan;
        a.add_bg = (flags & T2N_FLAG_ADD_BG) ? 1 : 0;
        const size_t lds = (size_t)4 * 4 * a.npad * sizeof(float);
        const unsigned nb = (unsigned)((n_rays + 3) / 4);
        // binned scatters (density: 3-D blocks; appearance: plane tiles, as long as the grid's lines fit the LDS budget) unless
        // T2N_BWD_ATOMIC_SCATTER=1 asks for the sliding-window global-atomic path
        int Lmax = 0;
        for (int k = 0; k < 3; ++k) Lmax = f->dev.den.L[k] > Lmax ? f->dev.den.L[k] : Lmax;
        const size_t lds_acc = tile_accum_lds(16, Lmax);
        static const bool force_atomic = getenv("T2N_BWD_ATOMIC_SCATTER") && atoi(getenv("T2N_BWD_ATOMIC_SCATTER")) != 0;
        bin = !force_atomic && lds_acc <= 160 * 1024 && block_geom_ok(f->dev.den) && (uint64_t)n_rays * n_samples * 3 < 0x7fffffffull;
        lds_bin = lds_acc;
        if (plan && !bin) { set_error("t2n_render_backward: T2N_FLAG_DEVICE_ROWS needs the binned scatters (grid lines within the LDS budget)"); return T2N_ERR_UNSUPPORTED; }
        a.gfeat = (float*)(fw + c.sigma); a.hist = (unsigned*)(bw + b.hist); a.geom = bgeom;
        if (bin) {
            so.zero(a.hist, (size_t)bgeom.total * bgeom.copies * 4);
            if (rows > 0 && f->dev.app.C == 48) so.zero(bw + b.a_hist, (size_t)bin_geom(f->dev.app).total * kBinCopies * 4);
        }
        // the fused input-gradient chain's operands are packed now, while the stream is alone on the GPU
        const bool pack_now = rows > 0 && !generic && !simple && !gemm_fp32_mode(f) && f->desc.app_dim == 27 && K0 == 351;
        if (pack_now) so.zero(mlp_bwd_ss_absmax_words((void*)(bw + b.gpack)), 16);
        if ((rc = launch_setup(so, s))) return rc;
        static const bool serial = getenv("T2N_BWD_SERIAL") != nullptr;
        if (pack_now) {
            // ... on the third stream (idle until the weight-gradient GEMMs): two 6-us kernels off the caller's stream, joined in front
            // of k_mlp_bwd_ss, behind k_bwd_march and layer 2
            hipStream_t sp = s;
#ifndef T2N_PACK_ON_CALLER_STREAM
#define T2N_PACK_ON_CALLER_STREAM 0
#endif
            if (!serial && bin && !T2N_PACK_ON_CALLER_STREAM) {
                if (!f->gemm_stream || !f->ev_fork2) {
                    hipEvent_t e0, e1;
                    { const int rcs = shared_side_streams(&f->side_stream, &f->gemm_stream); if (rcs) return rcs; }
                    T2N_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
                    T2N_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                    f->ev_fork2 = (void*)e0; f->ev_join2 = (void*)e1;
                }
                if (!f->ev_pack) { hipEvent_t e; T2N_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); f->ev_pack = (void*)e; }
                sp = (hipStream_t)f->gemm_stream;
                T2N_HIP(hipEventRecord((hipEvent_t)f->ev_fork2, s));
                T2N_HIP(hipStreamWaitEvent(sp, (hipEvent_t)f->ev_fork2, 0));
                pack_side = true;
            }
            if ((rc = mlp_bwd_ss_pack(f, (void*)(bw + b.gpack), sp, true, true))) return rc;
            if (pack_side) T2N_HIP(hipEventRecord((hipEvent_t)f->ev_pack, sp));
        }
        packed_early = pack_now;
        timing_begin(f, T2N_K_BWD_MARCH, s);
        if (!bin) {
            if (flags & T2N_FLAG_TRAIN) hipLaunchKernelGGL((k_bwd_march<true, false>), dim3(nb), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((k_bwd_march<false, false>), dim3(nb), dim3(256), lds, s, a);
        } else {
            if (flags & T2N_FLAG_TRAIN) hipLaunchKernelGGL((k_bwd_march<true, true>), dim3(nb), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((k_bwd_march<false, true>), dim3(nb), dim3(256), lds, s, a);
            // The density scatter (scan -> records -> LDS accumulate: atomic-latency- and LDS-bound, little VALU, no MFMA) shares
            // nothing with the MLP backward and the appearance scatter below but the finished k_bwd_march: it runs on a side stream
            // beside them and is joined before the gradients leave this call (T2N_BWD_SERIAL=1: one stream).
            hipStream_t sd = s;
            if (!serial && rows > 0) {
                if (!f->side_stream || !f->ev_fork) {
                    hipEvent_t e0, e1;
                    { const int rcs = shared_side_streams(&f->side_stream, &f->gemm_stream); if (rcs) return rcs; }
                    T2N_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
                    T2N_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                    f->ev_fork = (void*)e0; f->ev_join = (void*)e1;
                }
                sd = (hipStream_t)f->side_stream;
                T2N_HIP(hipEventRecord((hipEvent_t)f->ev_fork, s));
                T2N_HIP(hipStreamWaitEvent(sd, (hipEvent_t)f->ev_fork, 0));
                side = true;
            }
            launch_bin_scan(a.hist, bgeom.total, bgeom.copies, (unsigned*)(bw + b.bin_total), (unsigned*)(bw + b.tile_start), (int4*)(bw + b.segs), (unsigned*)(bw + b.nseg), b.seg_cap, kDenSeg,
                            0u, kDenSeg, sd);
            BinArgs ba;
            ba.F = f->dev; ba.geom = bgeom; ba.rays = rays; ba.n_rays = n_rays; ba.ray_stride = ray_stride; ba.n_samples = n_samples;
            ba.jitter = jitter; ba.gfeat = a.gfeat; ba.ray_app = a.ray_app; ba.cursor = a.hist; ba.tile_start = (const unsigned*)(bw + b.tile_start); ba.recs = (float4*)(bw + b.recs);
            if (flags & T2N_FLAG_TRAIN) hipLaunchKernelGGL((k_bwd_bin<true>), dim3(nb), dim3(256), 0, sd, ba);
            else hipLaunchKernelGGL((k_bwd_bin<false>), dim3(nb), dim3(256), 0, sd, ba);
            DenBlockArgs da;
            da.S = f->dev.den; da.G = a.gden; da.geom = bgeom; da.segs = (const int4*)(bw + b.segs);
            da.nseg = (const unsigned*)(bw + b.nseg); da.recs = (const float4*)(bw + b.recs);
            hipLaunchKernelGGL(k_bwd_den_block, dim3(b.seg_cap < kAccGrid ? b.seg_cap : kAccGrid), dim3(kDenThreads), 0, sd, da);
            if (side) T2N_HIP(hipEventRecord((hipEvent_t)f->ev_join, sd));
            den_stream = sd;
        }
        // the density gradients of this call are final behind this point of `den_stream` (t2n_field_wait_density_grads)
        if (!f->ev_den) { hipEvent_t e; T2N_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); f->ev_den = (void*)e; }
        T2N_HIP(hipEventRecord((hipEvent_t)f->ev_den, den_stream));
        timing_end(f, T2N_K_BWD_MARCH, s);
        T2N_HIP(hipGetLastError());
    }

    if (rows > 0) {
        const t2n_field_params* P = &f->params_ref;
        timing_begin(f, T2N_K_BWD_MLP, s);
        const bool gemm_fp32 = gemm_fp32_mode(f);
        const bool fused = !generic && !gemm_fp32 && f->desc.app_dim == 27 && K0 == 351;   // (the five-launch form below serves the general heads and the exact mode)
        void* gpack = (void*)(bw + b.gpack);
        if (simple) {
            // parameter-free heads: dL/dfeatures straight from the colour gradients, then basis_mat's two products on the exact path
            if ((rc = launch_simple_head_bwd(f, tp.t, rows, (const float4*)go, app_rgb, app_ray, rays, ray_stride, counters, c.list_cap, gf, s))) return rc;
            if (g->basis_weight) launch_gemm_tn(1, true, gf, 32, x144, 144, rows, f->desc.app_dim, 144, g->basis_weight, 144, part, s);
            launch_gemm_nn(gf, 32, P->basis_weight, 144, rows, f->desc.app_dim, 144, nullptr, 0, gxapp, 144, s);
        } else if (fused) {
            // the input-gradient chain as ONE kernel (t2n_mlp_bwd_ss.hip): k_bwd_l2 only accumulates dW2 / db2 (h1 stays intact for it),
            // the chain writes g1 over h1 and g0 / gf / gX into the (otherwise unused) encoding buffer
            float* G0 = xpe; float* GF = xpe + (size_t)rows * 128; float* GX = xpe + (size_t)rows * 160;
            launch_bwd_l2((const float4*)go, (const float*)h1, rows, P->mlp_w2, nullptr, g->mlp_w2, g->mlp_b2, part, s, rows_dev);
            if (pack_side) T2N_HIP(hipStreamWaitEvent(s, (hipEvent_t)f->ev_pack, 0));
            if ((rc = launch_mlp_bwd_ss(f, gpack, (const float4*)go, h1, h0, feat32, G0, GF, GX, rows, s, packed_early, rows_dev))) return rc;
            // From here two chains share nothing but read-only rows: the weight-gradient GEMMs (g1 / G0 / GF with h0 / features / x144
            // -> the MLP gradients, through `part`) and the appearance scatter (GX -> the factor gradient buffers). The GEMMs wait on
            // memory latency and workgroup barriers, the scatter on LDS atomics: they run side by side, the GEMMs on a third stream
            // that is joined before the gradients leave this call (T2N_BWD_SERIAL=1: one stream).
            hipStream_t sg = s;
            if (side) {
                if (!f->gemm_stream || !f->ev_fork2) {
                    hipEvent_t e0, e1;
                    { const int rcs = shared_side_streams(&f->side_stream, &f->gemm_stream); if (rcs) return rcs; }
                    T2N_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
                    T2N_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                    f->ev_fork2 = (void*)e0; f->ev_join2 = (void*)e1;
                }
                sg = (hipStream_t)f->gemm_stream;
                T2N_HIP(hipEventRecord((hipEvent_t)f->ev_fork2, s));
                T2N_HIP(hipStreamWaitEvent(sg, (hipEvent_t)f->ev_fork2, 0));
                side_gemm = true;
            }
            if (plan && (!g->mlp_w1 || !g->mlp_w0)) { set_error("t2n_render_backward: T2N_FLAG_DEVICE_ROWS needs the weight-gradient tensors of both hidden layers"); return T2N_ERR_INVALID; }
            if (g->mlp_w1) launch_gemm_tn(4, gemm_fp32, g1, 128, h0, 128, rows, 128, 128, g->mlp_w1, 128, part, sg, nullptr, g->mlp_b1, rows_dev);
            else if (g->mlp_b1) launch_colsum((const float*)g1, 128, (long long)rows, 128, g->mlp_b1, sg);
            if (g->mlp_w0) launch_gemm_tn(4, gemm_fp32, G0, 128, xpe, K0pad, rows, 128, K0, g->mlp_w0, K0, part, sg, feat32, g->mlp_b0, rows_dev);
            else if (g->mlp_b0) launch_colsum((const float*)G0, 128, (long long)rows, 128, g->mlp_b0, sg);
            if (g->basis_weight) launch_gemm_tn(1, gemm_fp32, GF, 32, x144, 144, rows, f->desc.app_dim, 144, g->basis_weight, 144, part, sg, nullptr, nullptr, rows_dev);
            if (side_gemm) T2N_HIP(hipEventRecord((hipEvent_t)f->ev_join2, sg));
            gxapp = GX;
        } else {
        // 3. layer 2
        launch_bwd_l2((const float4*)go, (const float*)h1, rows, P->mlp_w2, g1, g->mlp_w2, g->mlp_b2, part, s);
        // 4. layers 1, 0, PE, basis
        // (a bias gradient without its weight gradient does not occur: the column sums ride in the weight-gradient GEMM)
        if (g->mlp_w1) launch_gemm_tn(4, gemm_fp32, g1, 128, h0, 128, rows, 128, 128, g->mlp_w1, 128, part, s, nullptr, g->mlp_b1);
        else if (g->mlp_b1) launch_colsum((const float*)g1, 128, (long long)rows, 128, g->mlp_b1, s);
        // input-gradient GEMMs: split-f16 MFMA products with a power-of-two scale per row (t2n_gemm_h.hip)
        // keeps the fp32-MFMA form
        if (!gemm_fp32 && (rc = gemm_h_pack(f, gpack, K0, s))) return rc;
        if (gemm_fp32) launch_gemm_nn(g1, 128, P->mlp_w1, 128, rows, 128, 128, h0, 128, g0, 128, s);
        else if ((rc = launch_gemm_nn_h(gpack, 0, K0, g1, 128, rows, h0, 128, g0, 128, s))) return rc;
        const bool pe_in_gemm = !generic && !gemm_fp32 && g->mlp_w0;   // the encoding is computed inside the weight-gradient GEMM
        if (!generic && !pe_in_gemm && g->mlp_w0) launch_pe_fwd((const float*)feat32, (long long)rows, xpe, s);
        if (g->mlp_w0) launch_gemm_tn(4, gemm_fp32, g0, 128, xpe, K0pad, rows, 128, K0, g->mlp_w0, K0, part, s, pe_in_gemm ? feat32 : nullptr, g->mlp_b0);
        else if (g->mlp_b0) launch_colsum((const float*)g0, 128, (long long)rows, 128, g->mlp_b0, s);
        if (gemm_fp32) launch_gemm_nn(g0, 128, P->mlp_w0, K0, rows, 128, K0, nullptr, 0, gx, K0pad, s);
        else if ((rc = launch_gemm_nn_h(gpack, 1, K0, g0, 128, rows, nullptr, 0, gx, K0pad, s))) return rc;
        if (!generic) launch_pe_bwd((const float*)gx, (const float*)feat32, (long long)rows, gf, s);
        else if ((rc = launch_head_in_bwd(f, gx, feat32, rows, gf, s))) return rc;
        if (g->basis_weight) launch_gemm_tn(1, gemm_fp32, gf, 32, x144, 144, rows, f->desc.app_dim, 144, g->basis_weight, 144, part, s);
        if (gemm_fp32) launch_gemm_nn(gf, 32, P->basis_weight, 144, rows, f->desc.app_dim, 144, nullptr, 0, gxapp, 144, s);
        else if ((rc = launch_gemm_nn_h(gpack, 2, K0, gf, 32, rows, nullptr, 0, gxapp, 144, s))) return rc;
        }
        timing_end(f, T2N_K_BWD_MLP, s);
        T2N_HIP(hipGetLastError());
        // 5. appearance scatter
        AppScatterArgs sa;
        sa.F = f->dev;
        for (int k = 0; k < 3; ++k) { sa.gapp.plane[k] = f->gbuf_app_plane[k]; sa.gapp.line[k] = f->gbuf_app_line[k]; }
        sa.app_pos = app_pos; sa.counters = counters; sa.list_cap = c.list_cap; sa.gxapp = gxapp;
        timing_begin(f, T2N_K_BWD_SCATTER, s);
        if (bin && f->dev.app.C == 48) {
            // tile-binned: count -> scan -> write records -> LDS accumulate, 16 channels per workgroup
            AppBinArgs ab;
            ab.S = f->dev.app; ab.geom = bin_geom(f->dev.app); ab.app_pos = app_pos; ab.counters = counters; ab.list_cap = c.list_cap;
            ab.plan = plan; ab.tp = tp; ab.rows = rows; ab.hist = (unsigned*)(bw + b.a_hist); ab.tile_start = (const unsigned*)(bw + b.a_tile_start); ab.recs = (float4*)(bw + b.a_recs);
            const unsigned nbk = (unsigned)((rows + 255) / 256);
            hipLaunchKernelGGL((k_app_bin<0>), dim3(nbk), dim3(256), 0, s, ab);
            launch_bin_scan(ab.hist, ab.geom.total, kBinCopies, (unsigned*)(bw + b.a_bin_total), (unsigned*)(bw + b.a_tile_start), (int4*)(bw + b.a_segs), (unsigned*)(bw + b.a_nseg),
                            b.a_seg_cap, 0u, kAccTargetSegsApp, (unsigned)T2N_ACC_SEG_MIN, s);
            hipLaunchKernelGGL((k_app_bin<1>), dim3(nbk), dim3(256), 0, s, ab);
            TileAccumArgs ta;
            ta.S = f->dev.app; ta.G = sa.gapp; ta.geom = ab.geom; ta.segs = (const int4*)(bw + b.a_segs);
            ta.nseg = (const unsigned*)(bw + b.a_nseg); ta.recs = (const float4*)(bw + b.a_recs);
            ta.gx = gxapp; ta.gx_ld = 144;
            T2N_HIP(hipFuncSetAttribute((const void*)k_bwd_tile_accum<48>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bin));
            hipLaunchKernelGGL((k_bwd_tile_accum<48>), dim3(b.a_seg_cap < kAccGrid ? b.a_seg_cap : kAccGrid, 3), dim3(kAccThreads), lds_bin, s, ta);
        } else {
            unsigned blocks = (unsigned)((rows / 32 + 3) / 4);
            if (blocks > 2048) blocks = 2048;
            if (blocks == 0) blocks = 1;
            hipLaunchKernelGGL(k_bwd_app_scatter, dim3(blocks), dim3(256), 0, s, sa);
        }
        timing_end(f, T2N_K_BWD_SCATTER, s);
        T2N_HIP(hipGetLastError());
    }

    if (side) T2N_HIP(hipStreamWaitEvent(s, (hipEvent_t)f->ev_join, 0));
    if (side_gemm) T2N_HIP(hipStreamWaitEvent(s, (hipEvent_t)f->ev_join2, 0));
    // 6. channel-last gradient buffers -> += reference layouts
    {
        RelayoutAddMulti ra;
        memset(&ra, 0, sizeof(ra));
        unsigned blocks = 0;
        auto add = [&](const float* src, float* dst, int C, long long n) {
            if (!dst) return;
            ra.src[ra.count] = src; ra.dst[ra.count] = dst; ra.C[ra.count] = C; ra.HW[ra.count] = n; ra.block0[ra.count] = blocks;
            blocks += (unsigned)((n + 63) / 64);
            ra.count++;
        };
        for (int k = 0; k < 3; ++k) {
            const long long HW = (long long)gr[mat1(k)] * gr[mat0(k)], L = gr[vecm(k)];
            add(f->gbuf_den_plane[k], g->density_plane[k], 16, HW);
            add(f->gbuf_den_line[k], g->density_line[k], 16, L);
            add(f->gbuf_app_plane[k], g->app_plane[k], 48, HW);
            add(f->gbuf_app_line[k], g->app_line[k], 48, L);
        }
        ra.block0[ra.count] = blocks;
        if (blocks) hipLaunchKernelGGL(k_relayout_add, dim3(blocks), dim3(256), 0, s, ra);
    }
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" size_t t2n_field_grad_buffer_bytes(const t2n_field* f) {
    if (!f) return 0;
    size_t off[12];
    return grad_layout(f, off);
}

extern "C" int t2n_field_set_grad_buffer(t2n_field* f, void* buf, size_t bytes) {
    if (!f) { set_error("t2n_field_set_grad_buffer: NULL field"); return T2N_ERR_INVALID; }
    size_t off[12];
    const size_t need = grad_layout(f, off);
    if (!buf) {   // back to a library-owned buffer (allocated on the next backward)
        if (f->gbuf_external) { f->gbuf_all = nullptr; f->gbuf_bytes = 0; f->gbuf_external = false; }
        return T2N_OK;
    }
    if (bytes < need || ((uintptr_t)buf & 255u)) { set_error("t2n_field_set_grad_buffer: needs %zu bytes, 256-B aligned", need); return T2N_ERR_INVALID; }
    if (f->gbuf_all && !f->gbuf_external) (void)hipFree(f->gbuf_all);
    f->gbuf_all = (float*)buf; f->gbuf_bytes = need; f->gbuf_external = true;
    grad_slices(f, off);
    return T2N_OK;
}

extern "C" size_t t2n_field_grad_buffer_density_bytes(const t2n_field* f) {
    if (!f) return 0;
    size_t off[12], den = 0;
    (void)grad_layout(f, off, &den);
    return den;
}

extern "C" int t2n_field_wait_density_grads(const t2n_field* f, t2n_stream waiter) {
    if (!f) { set_error("t2n_field_wait_density_grads: NULL field"); return T2N_ERR_INVALID; }
    if (f->ev_den) T2N_HIP(hipStreamWaitEvent((hipStream_t)waiter, (hipEvent_t)f->ev_den, 0));
    return T2N_OK;
}

extern "C" int t2n_field_shard_layout(const t2n_field* f, int world, int64_t out[36]) {
    if (!f || !out || world < 1) { set_error("t2n_field_shard_layout: bad argument"); return T2N_ERR_INVALID; }
    size_t off[12];
    (void)grad_layout(f, off);
    unsigned lo[12], hi[12], body[12];
    shard_partition(f, world, 0, lo, hi, body);
    const int* gr = f->desc.grid;
    for (int q = 0; q < 4; ++q)
        for (int k = 0; k < 3; ++k) {
            const int t = q * 3 + k, C = q < 2 ? f->desc.density_n_comp : f->desc.app_n_comp;
            const int64_t n = (q & 1) == 0 ? (int64_t)gr[mat1(k)] * gr[mat0(k)] : (int64_t)gr[vecm(k)];
            out[3 * t] = (int64_t)(off[t] / 4); out[3 * t + 1] = (int64_t)(hi[t] - lo[t]) * 64 * C; out[3 * t + 2] = n * C;
        }
    return T2N_OK;
}

extern "C" int t2n_field_factor_buffer(const t2n_field* f, int t, void** ptr) {
    if (!f || !ptr || t < 0 || t >= 12) { set_error("t2n_field_factor_buffer: bad argument"); return T2N_ERR_INVALID; }
    if (!f->uploaded || f->factor_bf16) { set_error("t2n_field_factor_buffer: needs an uploaded field with fp32 factor storage"); return T2N_ERR_STATE; }
    float* const* tab[4] = {f->buf_den_plane, f->buf_den_line, f->buf_app_plane, f->buf_app_line};
    *ptr = (void*)tab[t / 3][t % 3];
    return T2N_OK;
}

// =====================================================================================================================================
// t2n_train_step: one optimisation step of text2nerf_main.py:547-601 as ONE submission (include/t2n.h). The kernels are those of
// t2n_render_forward (KEEP_CTX, train), t2n_train_loss, t2n_render_backward (T2N_FLAG_DEVICE_ROWS) and the optimiser entry points; what is
// new is the ORDER: every launch that does not lie on the step's dependency chain runs on a side stream beside it, and nothing about a
// step is decided on the host.
//   s  (caller)  setup -> march -> shade (kept rows) -> composite -> loss -> bwd_march -> mlp_bwd_ss -> tile_accum -> Adam (factors)
//   sa (side)    TV seed of the gradient buffer (beside the march) ........ density binning + scatter (behind bwd_march)
//   sb (bin)     plan (rows, tile prefix, verdict, Adam scalars, host record) -> appearance binning (needs the forward's lists only)
//   sg (gemm)    layer-2 gradients (behind bwd_march) -> weight-gradient GEMMs (behind mlp_bwd_ss) -> Adam (head) -> operand re-packs
// =====================================================================================================================================
#ifndef T2N_ACC_CG8
#define T2N_ACC_CG8 1
#endif
#ifndef T2N_SEED_AFTER_MARCH
#define T2N_SEED_AFTER_MARCH 1
#endif
namespace t2n {

// zero fills of one step as ONE launch: region r = blockIdx.y (a uniform index into the kernel arguments)
constexpr int kZeroRegions = 12;
struct ZeroOps { unsigned* ptr[kZeroRegions]; unsigned long long words[kZeroRegions]; int n = 0;
    const uint4* up_src = nullptr; uint4* up_dst = nullptr; unsigned long long up_n16 = 0;   // optional upload: 16-byte words from pinned host memory (grid row n)
    bool add(void* p, size_t bytes) { if (!p || !bytes) return true; if (n >= kZeroRegions) return false; ptr[n] = (unsigned*)p; words[n] = (bytes + 3) / 4; ++n; return true; } };
__global__ __launch_bounds__(256) void k_zero_regions(const ZeroOps o) {
    const int r = blockIdx.y;
    if (r == o.n) {   // the batch: read over the host link by the kernel itself (same steady-state time as an engine copy in front of the launch, which stalled the stream for ~5 ms once per process)
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < o.up_n16; i += (unsigned long long)gridDim.x * 256)
            o.up_dst[i] = o.up_src[i];
        return;
    }
    unsigned* __restrict__ p = o.ptr[r];
    const unsigned long long n = o.words[r];
    if ((((uintptr_t)p) & 15u) == 0) {
        uint4* __restrict__ p4 = reinterpret_cast<uint4*>(p);
        const unsigned long long n4 = n / 4;
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (unsigned long long)gridDim.x * 256) p4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (unsigned long long i = n4 * 4 + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) p[i] = 0u;
    } else {
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) p[i] = 0u;
    }
}

// The plan of a fused step (one wave, right behind the forward's march): k_bwd_plan's tile prefix / row count / overflow from the
// sub-list counters, and with them the step's VERDICT — a step whose rows exceed the capacity applies no update — Adam's step count and
// this step's bias-corrected scalars (torch.optim.Adam: step_size = lr / (1 - beta1^t), denom = sqrt(v) / sqrt(1 - beta2^t) + eps, in
// double like the host entry points), the vote word behind the head gradients and the record in pinned host memory.
__global__ __launch_bounds__(64) void k_train_plan(const unsigned* __restrict__ counters, unsigned list_cap, unsigned rows_cap, BwdPlan* __restrict__ plan,
                                                   TrainState* __restrict__ st, const float* __restrict__ hyper, float beta1, float beta2, float* vote,
                                                   unsigned* host_rec, unsigned slot) {
    TrainScalars* __restrict__ sc = &st->sc[slot];
    const int lane = threadIdx.x;
    unsigned cnt = lane < kLists ? counters[lane * kCounterStride] : 0u;
    if (cnt > list_cap) cnt = list_cap;
    unsigned incl = (cnt + 31u) / 32u;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    const unsigned excl = incl - (cnt + 31u) / 32u;
    if (lane < kLists) plan->tp.t[lane] = excl;
    const unsigned total = __shfl(incl, kLists - 1);
    const unsigned rows = total * 32u;
    const unsigned ovf = rows > rows_cap ? 1u : 0u;
    const unsigned step = st->step + (ovf ? 0u : 1u);     // (every lane reads the old value; lane 0 writes below)
    if (lane < 19 && !ovf) {
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        sc->lr_over_bc1[lane] = (float)((double)hyper[lane] / bc1);
        if (lane == 0) sc->inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    }
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        plan->tp.t[kLists] = total;
        plan->overflow = ovf;
        plan->rows = rows < rows_cap ? rows : rows_cap;
        const unsigned seq = st->seq;
        st->seq = seq + 1u;
        sc->skip = ovf;
        st->step = step;
        if (ovf) st->skipped = st->skipped + 1u;
        if (vote) *vote = ovf ? 1.f : 0.f;
        if (host_rec) {
            // ONE 8-byte store per record into fine-grained pinned memory: {rows needed, (sequence + 1) << 1 | withheld}. No fence: a
            // system-scope release writes the L2 back (60 us behind the march's 50 MB of output), and the host needs no ordering between
            // records — an entry whose sequence number is not the one it waits for has simply not arrived yet
            volatile unsigned long long* h = reinterpret_cast<volatile unsigned long long*>(host_rec);
            h[2 + (seq & 15u)] = ((unsigned long long)(((seq + 1u) << 1) | ovf) << 32) | rows;
            host_rec[1] = step; host_rec[2] = st->skipped;
        }
    }
}
// Data-parallel steps (phases 1 | all-reduce | 2): the vote word came back from the all-reduce of the head gradients. A rank whose own
// rows fitted but whose peers' did not withholds its update too: the optimistic count of k_train_plan is taken back.
__global__ __launch_bounds__(64) void k_train_commit(TrainState* __restrict__ st, const float* __restrict__ vote, unsigned* host_rec, unsigned slot) {
    if (threadIdx.x != 0) return;
    if (*vote != 0.f && !st->sc[slot].skip) {
        st->sc[slot].skip = 1u; st->step = st->step - 1u; st->skipped = st->skipped + 1u;
        if (host_rec) {
            volatile unsigned long long* h = reinterpret_cast<volatile unsigned long long*>(host_rec);
            const unsigned seq = st->seq - 1u;
            h[2 + (seq & 15u)] = h[2 + (seq & 15u)] | (1ull << 32);
            host_rec[1] = st->step; host_rec[2] = st->skipped;
        }
    }
}
__global__ __launch_bounds__(64) void k_train_set_step(TrainState* st, unsigned step, unsigned* host_rec) {
    if (threadIdx.x == 0) { st->step = step; if (host_rec) ((volatile unsigned*)host_rec)[1] = step; }
}

struct TrainCarve { size_t fwd, fwd_bytes, bwd, bwd_bytes, g1, rgb, depth, part, total; };
static TrainCarve train_carve(const t2n_field* f, int64_t R, int N, int64_t rows) {
    TrainCarve t;
    const BinGeom geom = bin_geom(f->dev.den);
    const BlockGeom bg = block_geom(f->dev.den);
    size_t o = 0;
    const Carve c = carve_workspace(R, N, true, false);
    t.fwd = o; t.fwd_bytes = al256(c.total) + (size_t)rows * (144 + 32 + 128 + 128) * 4; o = al256(o + t.fwd_bytes);
    t.bwd = o; t.bwd_bytes = bwd_carve(rows, R, N, geom.total, bg.total * bg.copies, 351, true).total; o = al256(o + t.bwd_bytes);
    t.g1 = o; o = al256(o + (size_t)rows * 128 * 4);
    t.rgb = o; o = al256(o + (size_t)R * 12);
    t.depth = o; o = al256(o + (size_t)R * 4);
    t.part = o; o = al256(o + (size_t)((R + 3) / 4) * 12);
    t.total = o;
    return t;
}
static size_t train_dev_bytes() { return 256 + al256(mlp_bwd_ss_pack_bytes()); }
static bool train_supported(const t2n_field* f) {
    return f->desc.shading == T2N_SHADE_MLP_FEA_NOVIEW && f->desc.app_dim == 27 && f->dev.app.C == 48 && f->dev.den.C == 16 && f->mlp_split &&
           !f->factor_bf16 && !f->dev.alpha && block_geom_ok(f->dev.den);
}
static int train_ensure(t2n_field* f, hipStream_t s) {
    if (!f->train_dev) {
        T2N_HIP(hipMalloc(&f->train_dev, train_dev_bytes()));
        T2N_HIP(hipMemsetAsync(f->train_dev, 0, 256, s));
        T2N_HIP(hipHostMalloc((void**)&f->train_host, 64 * sizeof(unsigned), hipHostMallocDefault));
        for (int i = 0; i < 64; ++i) f->train_host[i] = 0u;
        f->train_packed = false;
    }
    int rc;
    if ((rc = shared_side_streams(&f->side_stream, &f->gemm_stream))) return rc;
    for (auto& e : f->train_ev) if (!e) { hipEvent_t x; T2N_HIP(hipEventCreateWithFlags(&x, hipEventDisableTiming)); e = (void*)x; }
    if (!f->ev_den) { hipEvent_t e; T2N_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); f->ev_den = (void*)e; }
    return T2N_OK;
}
// the backward chain's transposed split-f16 operands + the forward's packed head operands from the CURRENT head weights
// (the forward pack reads every weight anyway: it leaves the four matrices' largest magnitudes in the chain pack's absmax words — zeroed
// by the head Adam launch in front of it — so the chain pack is one short launch; a pack kernel that scanned the matrices itself took
// 60-105 us at the end of the step's longest chain)
static int train_repack(t2n_field* f, const t2n_field_params* p, hipStream_t s) {
    int rc;
    void* gpack = (char*)f->train_dev + 256;
    if ((rc = launch_pack_mlp(f, p, s, (unsigned*)mlp_bwd_ss_absmax_words(gpack)))) return rc;
    if ((rc = mlp_bwd_ss_pack(f, gpack, s, true, false, true))) return rc;
    f->train_packed = true;
    return T2N_OK;
}

}  // namespace t2n

extern "C" size_t t2n_train_step_workspace_bytes(const t2n_field* f, int64_t n_rays, int n_samples, int64_t rows_capacity) {
    if (!f || n_rays <= 0 || n_samples <= 0 || rows_capacity < 32) return 0;
    return train_carve(f, n_rays, n_samples, rows_capacity / 32 * 32).total;
}

extern "C" int t2n_field_train_set_step(t2n_field* f, uint32_t step, t2n_stream stream) {
    if (!f) { set_error("t2n_field_train_set_step: NULL field"); return T2N_ERR_INVALID; }
    int rc;
    if ((rc = train_ensure(f, (hipStream_t)stream))) return rc;
    hipLaunchKernelGGL(k_train_set_step, dim3(1), dim3(64), 0, (hipStream_t)stream, (TrainState*)f->train_dev, (unsigned)step, f->train_host);
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

extern "C" int t2n_field_train_record(const t2n_field* f, uint32_t out[36]) {
    if (!f || !out) { set_error("t2n_field_train_record: NULL argument"); return T2N_ERR_INVALID; }
    if (!f->train_host) { for (int i = 0; i < 36; ++i) out[i] = 0u; return T2N_OK; }
    // pinned layout: words 1, 2 = steps applied / withheld (informational: written without ordering), then 16 eight-byte records
    // {rows needed, (sequence + 1) << 1 | withheld} at sequence & 15. out[0] = number of consecutive records present from the field's
    // first step on, i.e. records of sequence numbers < out[0] have all landed (the host keeps its own count of what it consumed).
    const volatile unsigned long long* h = reinterpret_cast<const volatile unsigned long long*>(f->train_host);
    out[1] = f->train_host[1]; out[2] = f->train_host[2]; out[3] = 0u;
    unsigned newest = 0u;
    for (int k = 0; k < 16; ++k) {
        const unsigned long long e = h[2 + k];
        const unsigned tag = (unsigned)(e >> 32);
        out[4 + 2 * k] = (unsigned)e;           // rows needed
        out[5 + 2 * k] = tag;                   // (sequence + 1) << 1 | withheld; 0 = never written
        if ((tag >> 1) > newest) newest = tag >> 1;
    }
    out[0] = newest;   // newest sequence number + 1 seen in the ring (records arrive in order: one plan kernel after the other)
    return T2N_OK;
}

extern "C" int t2n_train_step(t2n_field* f, const t2n_train_step_args* A, t2n_stream stream) {
    if (!f || !A || !A->rays || !A->jitter || !A->rgb_target || !A->depth_target || !A->hyper || !A->head_grads || !A->workspace || !A->losses ||
        A->n_rays <= 0 || A->ray_stride < 6 || A->n_samples <= 0 || !(A->phases & 7u) || ((A->phases & 4u) && A->phases != 4u)) {
        set_error("t2n_train_step: bad argument");
        return T2N_ERR_INVALID;
    }
    const int sworld = A->shard_world > 1 ? A->shard_world : 1, srank = A->shard_world > 1 ? A->shard_rank : 0;
    if (srank < 0 || srank >= sworld || (sworld > 1 && (A->phases & 3u) == 3u)) {
        set_error("t2n_train_step: shard_rank %d of %d, or a sharded step as one call (the gradient exchange lies between phases 1 and 2)", A->shard_rank, A->shard_world);
        return T2N_ERR_INVALID;
    }
    if (A->phases == 4u) {   // sharded optimiser: the gathered body blocks of the other ranks -> the caller's reference-layout tensors
        if (!f->uploaded || !train_supported(f)) { set_error("t2n_train_step: phases = 4 needs an uploaded field of the fused step's shape"); return T2N_ERR_STATE; }
        if (sworld < 2) return T2N_OK;
        return launch_factor_adam_dev(f, &A->params, nullptr, nullptr, 0.f, 0.f, 0.f, nullptr, 0, 12, (hipStream_t)stream, sworld, srank, true);
    }
    if (!f->uploaded) { set_error("t2n_train_step: field has no uploaded parameters"); return T2N_ERR_STATE; }
    if (!train_supported(f)) {
        set_error("t2n_train_step: needs the fused MLP_Fea_noview head (27 / 6 / 128) in split-f16 mode, 16 + 48 components, fp32 factor storage, no alpha mask");
        return T2N_ERR_UNSUPPORTED;
    }
    const int64_t R = A->n_rays;
    const int N = A->n_samples;
    const int64_t rows = A->rows_capacity / 32 * 32;
    if (N > 1024 || rows < 32 || (uint64_t)R * N * 3 >= 0x7fffffffull || (uint64_t)list_capacity(R, N) * kLists > 0x7fffffffull) {
        set_error("t2n_train_step: n_samples %d (<= 1024), rows_capacity %lld (>= 32) or n_rays x n_samples out of range", N, (long long)A->rows_capacity);
        return T2N_ERR_UNSUPPORTED;
    }
    int Lmax = 0;
    for (int k = 0; k < 3; ++k) Lmax = f->dev.den.L[k] > Lmax ? f->dev.den.L[k] : Lmax;
    const size_t lds_acc = tile_accum_lds(16, Lmax);
    if (lds_acc > 160 * 1024) { set_error("t2n_train_step: grid lines beyond the LDS budget of the binned scatter"); return T2N_ERR_UNSUPPORTED; }
    const TrainCarve T = train_carve(f, R, N, rows);
    if (T.total > A->workspace_bytes) { set_error("t2n_train_step: workspace %zu B < %zu B", A->workspace_bytes, T.total); return T2N_ERR_WORKSPACE; }
    if (!f->gbuf_all) { set_error("t2n_train_step: the field has no gradient buffer (t2n_field_set_grad_buffer)"); return T2N_ERR_STATE; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if ((rc = train_ensure(f, s))) return rc;
    // three streams beside the caller's (one per hardware queue): sa = early part + TV seed + density scatter / Adam, sg = layer 2 +
    // weight-gradient GEMMs + head step. (With separate streams for the plan / binning, the early part and the caller's batch copy, seven
    // streams shared four queues: whichever chain landed behind the GEMM chain's queue waited for it — 0.86 or 0.99 ms per step depending
    // on the order the streams had been created in.)
    hipStream_t sa = (hipStream_t)f->side_stream, sg = (hipStream_t)f->gemm_stream, sb = sa;
    hipEvent_t* ev = (hipEvent_t*)f->train_ev;   // 0 begin, 1 seed, 2 march, 3 plan, 4 appbin, 5 bwd_march, 6 chain, 7 gemm-stream join, 8 loss
    TrainState* st = (TrainState*)f->train_dev;
    void* gpack = (char*)f->train_dev + 256;
    float* vote = A->head_grads + (T2N_TRAIN_HEAD_GRAD_FLOATS - 1);
    const bool do_grad = (A->phases & 1u) != 0, do_opt = (A->phases & 2u) != 0;
    // pipelined form: the early part of this step on `se`, beside the previous step's tail (see include/t2n.h)
    // density bins counted from the forward's windows, beside the shade kernel, instead of behind the backward march (as t2n_render_backward
    // does): takes ~110 us of small kernels off the chain backward march -> density scatter -> density Adam -> next march, at the price of
    // records for zero-gradient samples and a second evaluation of the sample positions. Small batches are bound by that chain (2 048 rays:
    // 0.404 -> 0.392 ms; 8 192 rays: 0.575 -> 0.564), large ones by the machine's throughput (16 384 rays: 0.851 -> 0.857, later 0.829 / 0.845 ->
    // 0.838 / 0.826: nothing either way): taken up to 4 096 rays (the configuration every record of the round ran)
    // (T2N_DEN_EARLY=0 / 1 forces it; profiles/round6_train_ab.txt)
    static const int den_env = getenv("T2N_DEN_EARLY") ? atoi(getenv("T2N_DEN_EARLY")) : -1;
    bool den_early = den_env >= 0 ? den_env != 0 : A->n_rays <= 4096;
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap_status);
    if (cap_status != hipStreamCaptureStatusNone) den_early = false;   // (the captured step keeps the one DAG its replay was tested with: the early form's extra fork crashed hipStreamEndCapture on ROCm 7.2)
    const bool pipe = (A->flags & T2N_FLAG_PIPELINE) && A->host_batch && do_grad && do_opt && A->workspace_bytes >= 2 * T.total && cap_status == hipStreamCaptureStatusNone;
    const bool chain_prev = f->train_chain;               // the previous call was a full step of this form and left its density-Adam event
    f->train_chain = false;
    const unsigned par = f->train_calls & 1u;            // scalar slot (always alternates) and, pipelined, workspace half
    if (do_grad) f->train_calls++;
    const unsigned slot = do_grad ? par : ((f->train_calls - 1u) & 1u);   // an optimiser-only call takes the slot its gradient call wrote
    TrainScalars* sc = &st->sc[slot];
    hipStream_t se = pipe ? sa : s;     // (on sa the early part is ordered behind the previous step's density Adam by the stream itself)
    // the caller's tensors are the parameters from here on (the optimiser writes them; the packs read them)
    const t2n_field_params* P = &A->params;
    if (do_opt && (!P->basis_weight || !P->mlp_w0 || !P->mlp_b0 || !P->mlp_w1 || !P->mlp_b1 || !P->mlp_w2 || !P->mlp_b2)) { set_error("t2n_train_step: NULL head tensor"); return T2N_ERR_INVALID; }

    char* ws = (char*)A->workspace + (pipe ? (size_t)par * T.total : 0);
    char* fw = ws + T.fwd;
    char* bw = ws + T.bwd;
    const Carve c = carve_workspace(R, N, true, false);
    const KeptRows kr = kept_rows(c.total, T.fwd_bytes);
    if ((int64_t)kr.rows < rows) { set_error("t2n_train_step: internal carve mismatch (%u kept rows < %lld)", kr.rows, (long long)rows); return T2N_ERR_WORKSPACE; }
    const BinGeom ageom = bin_geom(f->dev.app);
    const BinGeom dgeom_t = bin_geom(f->dev.den);
    const BlockGeom bgeom = block_geom(f->dev.den);
    const BwdCarve b = bwd_carve(rows, R, N, dgeom_t.total, bgeom.total * bgeom.copies, 351, true);
    float* rgb = (float*)(ws + T.rgb); float* depth = (float*)(ws + T.depth);
    const uint32_t flags = (A->flags & T2N_FLAG_ADD_BG) | T2N_FLAG_TRAIN | T2N_FLAG_KEEP_CTX;

    if (do_grad) {
        if (!f->train_packed) {   // first step / after an upload: the operands an optimiser phase leaves packed behind every later step
            if ((rc = mlp_bwd_ss_pack(f, gpack, s, false, true))) return rc;
            f->train_packed = true;
        }
        T2N_HIP(hipEventRecord(ev[0], s));
        if (pipe && !chain_prev) T2N_HIP(hipStreamWaitEvent(se, ev[0], 0));   // (the previous call left no density-Adam event on sa: behind `stream`)
        const uint4* up_src = nullptr;
        if (A->host_batch) {
            if (!A->batch_buffer || !A->host_batch_bytes) { set_error("t2n_train_step: host_batch without batch_buffer / host_batch_bytes"); return T2N_ERR_INVALID; }
            // The batch goes through the copy engine. T2N_COPY_KERNEL=1: a pinned, 16-byte-aligned batch is read by the zero-fill launch itself
            // (no engine packet on the compute stream: its once-per-process ~5 ms stall goes away, the steady state is the same) — NOT the
            // default: in the pipelined form, as the first fused work of a fresh process, that kernel died of a GPU memory fault in 7 of 56
            // runs (0 of 44 with the engine copy, 0 of 16 unpipelined; tools/r6_fault_loop.sh, profiles/round6_train_ab.txt); unexplained
            static const bool engine_copy = !(getenv("T2N_COPY_KERNEL") && atoi(getenv("T2N_COPY_KERNEL")) != 0);
            void* dp = nullptr;
            if (!engine_copy && ((uintptr_t)A->host_batch & 15u) == 0 && ((uintptr_t)A->batch_buffer & 15u) == 0 && (A->host_batch_bytes & 15u) == 0 &&
                hipHostGetDevicePointer(&dp, const_cast<void*>((const void*)A->host_batch), 0) == hipSuccess && dp) up_src = (const uint4*)dp;
            else { (void)hipGetLastError(); T2N_HIP(hipMemcpyAsync(A->batch_buffer, A->host_batch, A->host_batch_bytes, hipMemcpyHostToDevice, se)); }
        }
        // ---- every zero fill of the step in one launch
        RenderLaunch L;
        L.rays = A->rays; L.n_rays = R; L.ray_stride = A->ray_stride; L.n_samples = N; L.flags = flags; L.jitter = A->jitter;
        L.rgb = rgb; L.depth = depth; L.weights = nullptr; L.z_vals = nullptr; L.stats = nullptr;   // (weights / sample depths: recomputed where they are needed)
        L.counters = (unsigned*)(fw + c.counters); L.acc = (float*)(fw + c.acc); L.ray_app = (int4*)(fw + c.ray_app);
        L.app_pos = (float4*)(fw + c.app_pos); L.app_rgb = (float4*)(fw + c.app_rgb); L.app_ray = (int*)(fw + c.app_ray);
        L.list_cap = c.list_cap; L.feat = nullptr; L.feat_rows = 0;
        L.sigma_ctx = (float*)(fw + c.sigma); L.rgb_raw = (float4*)(fw + c.rgb_raw);
        float4* go = (float4*)(bw + b.go);
        {
            ZeroOps zo;
            bool ok = zo.add(L.counters, (size_t)kLists * kCounterStride * 4);
            ok = ok && zo.add(go, (size_t)rows * 16);
            ok = ok && zo.add(bw + b.hist, (size_t)bgeom.total * bgeom.copies * 4);
            ok = ok && zo.add(bw + b.a_hist, (size_t)ageom.total * kBinCopies * 4);
            // (head_grads: left zeroed by the previous optimiser phase — its Adam kernel clears what it consumed; a pipelined early part
            // must not clear a buffer the previous step's head Adam has yet to read)
            if (!ok) { set_error("t2n_train_step: zero-fill table overflow"); return T2N_ERR_INVALID; }
            unsigned long long mx = 1;
            for (int r = 0; r < zo.n; ++r) mx = zo.words[r] > mx ? zo.words[r] : mx;
            if (up_src) { zo.up_src = up_src; zo.up_dst = (uint4*)A->batch_buffer; zo.up_n16 = A->host_batch_bytes / 16; mx = zo.up_n16 * 16 > mx ? zo.up_n16 * 16 : mx; }   // (one 16-byte word per thread up to 256 workgroups)
            unsigned bx = (unsigned)((mx + 4095) / 4096);
            bx = bx > 256 ? 256 : (bx < 1 ? 1 : bx);
            hipLaunchKernelGGL(k_zero_regions, dim3(bx, (unsigned)zo.n + (up_src ? 1u : 0u)), dim3(256), 0, se, zo);
        }
        // ---- forward: march
        if ((rc = launch_march(f, L, se))) return rc;
        T2N_HIP(hipEventRecord(ev[2], se));
        if (pipe) T2N_HIP(hipStreamWaitEvent(s, ev[2], 0));
        // ---- forward: shade with the activation rows kept, composite; the driver's loss
        {
            ShadeCtx ctx{(float*)(fw + kr.x144), (float*)(fw + kr.feat32), (float*)(fw + kr.h0), (float*)(fw + kr.h1)};
            if ((rc = launch_shade_list(f, L.app_pos, L.app_ray, L.rays, A->ray_stride, L.counters, L.list_cap, L.app_rgb, &ctx, s, false, kr.rows))) return rc;
        }
        if ((rc = launch_composite(f, L, s))) return rc;
        // (the driver's loss is evaluated by k_bwd_march<.., LOSS>; its partial sums are added up by the step's one reduce launch)
        // ---- sb: plan, then the appearance binning (needs the forward's lists and the plan, not a single gradient)
        T2N_HIP(hipStreamWaitEvent(sb, ev[2], 0));
        BwdPlan* plan = (BwdPlan*)(bw + b.plan);
        hipLaunchKernelGGL(k_train_plan, dim3(1), dim3(64), 0, sb, (const unsigned*)L.counters, c.list_cap, (unsigned)rows, plan, st, A->hyper,
                           A->beta1, A->beta2, vote, f->train_host, slot);
        T2N_HIP(hipEventRecord(ev[3], sb));
        GradSet gapp;
        for (int k = 0; k < 3; ++k) { gapp.plane[k] = f->gbuf_app_plane[k]; gapp.line[k] = f->gbuf_app_line[k]; }
        float* GX = (float*)(bw + b.xpe) + (size_t)rows * 160;
        {
            AppBinArgs ab;
            ab.S = f->dev.app; ab.geom = ageom; ab.app_pos = L.app_pos; ab.counters = L.counters; ab.list_cap = c.list_cap;
            ab.plan = plan; memset(&ab.tp, 0, sizeof(ab.tp)); ab.rows = rows; ab.hist = (unsigned*)(bw + b.a_hist); ab.tile_start = (const unsigned*)(bw + b.a_tile_start); ab.recs = (float4*)(bw + b.a_recs);
            const unsigned nbk = (unsigned)((rows + 255) / 256);
            if (den_early) {
                // + the density scatter's block histogram from the forward's windows (positions only, no gradient needed); its reduce + scan
                // run on sg while the shade kernel runs: behind the backward march only the record pass and the accumulate pass are left
                // on the density chain (k_bin_reduce + k_bin_scan took ~110 us there under the concurrency of the input-gradient chain)
                BinArgs ca;
                ca.F = f->dev; ca.F.ztab = nullptr; ca.geom = bgeom; ca.rays = A->rays; ca.n_rays = R; ca.ray_stride = A->ray_stride; ca.n_samples = N;
                ca.jitter = A->jitter; ca.gfeat = nullptr; ca.ray_app = (const int4*)(fw + c.ray_app); ca.cursor = (unsigned*)(bw + b.hist); ca.tile_start = nullptr; ca.recs = nullptr;
                hipLaunchKernelGGL(k_early_bins, dim3(nbk + (unsigned)((R + 3) / 4)), dim3(256), 0, sb, ab, ca, nbk);
                T2N_HIP(hipEventRecord(ev[10], sb));
                T2N_HIP(hipStreamWaitEvent(sg, ev[10], 0));
                launch_bin_scan((unsigned*)(bw + b.hist), bgeom.total, bgeom.copies, (unsigned*)(bw + b.bin_total), (unsigned*)(bw + b.tile_start), (int4*)(bw + b.segs), (unsigned*)(bw + b.nseg), b.seg_cap, kDenSeg,
                                0u, kDenSeg, sg);
                T2N_HIP(hipEventRecord(ev[11], sg));
            } else
                hipLaunchKernelGGL((k_app_bin<0>), dim3(nbk), dim3(256), 0, sb, ab);
            launch_bin_scan(ab.hist, ab.geom.total, kBinCopies, (unsigned*)(bw + b.a_bin_total), (unsigned*)(bw + b.a_tile_start), (int4*)(bw + b.a_segs), (unsigned*)(bw + b.a_nseg),
                            b.a_seg_cap, 0u, kAccTargetSegsApp, (unsigned)T2N_ACC_SEG_MIN, sb, true);
            hipLaunchKernelGGL((k_app_bin<1>), dim3(nbk), dim3(256), 0, sb, ab);
            T2N_HIP(hipEventRecord(ev[4], sb));
        }
        // ---- sa: the gradient buffer starts as the TV gradient of the current factors (zero where no weight is set): 140 MB of streaming
        // beside the shade kernel (matrix cores + LDS) rather than beside the march (gathers: the two slow each other by a third)
        T2N_HIP(hipStreamWaitEvent(sa, ev[0], 0));     // (behind the previous step's Adam: it read this buffer and wrote the factors)
        if (T2N_SEED_AFTER_MARCH) T2N_HIP(hipStreamWaitEvent(sa, ev[2], 0));
        timing_begin(f, T2N_K_TV_SEED, sa);
        if ((rc = launch_tv_seed_dev(f, A->hyper + 19, sa, sworld, srank))) return rc;
        timing_end(f, T2N_K_TV_SEED, sa);
        T2N_HIP(hipEventRecord(ev[1], sa));
        // ---- backward: per-ray pass
        float* x144 = (float*)(fw + kr.x144); float* feat32 = (float*)(fw + kr.feat32); float* h0 = (float*)(fw + kr.h0); float* h1 = (float*)(fw + kr.h1);
        float* part = (float*)(bw + b.part);
        float* xpe = (float*)(bw + b.xpe);
        float* G0 = xpe; float* GF = xpe + (size_t)rows * 128;
        float* G1 = (float*)(ws + T.g1);
        const unsigned* rows_dev = &plan->rows;
        T2N_HIP(hipStreamWaitEvent(s, ev[3], 0));
        {
            BwdMarchArgs a;
            a.F = f->dev; a.F.ztab = nullptr;
            for (int k = 0; k < 3; ++k) { a.gden.plane[k] = f->gbuf_den_plane[k]; a.gden.line[k] = f->gbuf_den_line[k]; }
            a.rays = A->rays; a.n_rays = R; a.ray_stride = A->ray_stride; a.n_samples = N; a.npad = (N + 63) & ~63;
            a.jitter = A->jitter; a.sigma = (const float*)(fw + c.sigma); a.ray_app = (const int4*)(fw + c.ray_app);
            a.app_rgb = L.app_rgb; a.rgb_raw = (const float4*)(fw + c.rgb_raw);
            a.d_rgb = nullptr; a.d_depth = nullptr; a.d_w = nullptr; a.go = go; a.list_cap = c.list_cap; memset(&a.tp, 0, sizeof(a.tp)); a.plan = plan;
            a.rgb = rgb; a.depth = depth; a.rgb_t = A->rgb_target; a.depth_t = A->depth_target; a.w_depth = A->w_depth; a.w_trans = A->w_trans; a.delta = A->delta;
            a.loss_part = (float*)(ws + T.part);
            a.add_bg = (flags & T2N_FLAG_ADD_BG) ? 1 : 0;
            a.gfeat = (float*)(fw + c.sigma); a.hist = (unsigned*)(bw + b.hist); a.geom = bgeom;
            const size_t lds = (size_t)4 * 4 * a.npad * sizeof(float);
            const unsigned nb = (unsigned)((R + 3) / 4);
            timing_begin(f, T2N_K_BWD_MARCH, s);
            if (den_early) hipLaunchKernelGGL((k_bwd_march<true, true, true, false>), dim3(nb), dim3(256), lds, s, a);
            else hipLaunchKernelGGL((k_bwd_march<true, true, true>), dim3(nb), dim3(256), lds, s, a);
            timing_end(f, T2N_K_BWD_MARCH, s);
            T2N_HIP(hipEventRecord(ev[5], s));
        }
        float* hg = A->head_grads;   // basis | w0 | b0 | w1 | b1 | w2 | b2
        float* g_basis = hg; float* g_w0 = g_basis + 27 * 144; float* g_b0 = g_w0 + 128 * 351; float* g_w1 = g_b0 + 128; float* g_b1 = g_w1 + 128 * 128;
        float* g_w2 = g_b1 + 128; float* g_b2 = g_w2 + 3 * 128;
        // ---- the input-gradient chain
        timing_begin(f, T2N_K_BWD_MLP, s);
        if ((rc = launch_mlp_bwd_ss(f, gpack, (const float4*)go, h1, h0, feat32, G0, GF, GX, rows, s, true, rows_dev, G1))) return rc;
        timing_end(f, T2N_K_BWD_MLP, s);
        T2N_HIP(hipEventRecord(ev[6], s));
        {
            const unsigned nb = (unsigned)((R + 3) / 4);
            // ---- sa: density scatter (behind the TV seed on the same stream)
            T2N_HIP(hipStreamWaitEvent(sa, ev[5], 0));
            timing_begin(f, T2N_K_BWD_DENSITY, sa);
            if (den_early) T2N_HIP(hipStreamWaitEvent(sa, ev[11], 0));     // counted, reduced and scanned on sg beside the shade kernel
            else launch_bin_scan((unsigned*)(bw + b.hist), bgeom.total, bgeom.copies, (unsigned*)(bw + b.bin_total), (unsigned*)(bw + b.tile_start), (int4*)(bw + b.segs), (unsigned*)(bw + b.nseg), b.seg_cap, kDenSeg,
                                 0u, kDenSeg, sa);
            BinArgs ba;
            ba.F = f->dev; ba.F.ztab = nullptr; ba.geom = bgeom; ba.rays = A->rays; ba.n_rays = R; ba.ray_stride = A->ray_stride; ba.n_samples = N;
            ba.jitter = A->jitter; ba.gfeat = (float*)(fw + c.sigma); ba.ray_app = (const int4*)(fw + c.ray_app); ba.cursor = (unsigned*)(bw + b.hist); ba.tile_start = (const unsigned*)(bw + b.tile_start); ba.recs = (float4*)(bw + b.recs);
            if (den_early) hipLaunchKernelGGL((k_bwd_bin<true, true>), dim3(nb), dim3(256), 0, sa, ba);
            else hipLaunchKernelGGL((k_bwd_bin<true>), dim3(nb), dim3(256), 0, sa, ba);
            DenBlockArgs da;
            da.S = f->dev.den; for (int k = 0; k < 3; ++k) { da.G.plane[k] = f->gbuf_den_plane[k]; da.G.line[k] = f->gbuf_den_line[k]; } da.geom = bgeom; da.segs = (const int4*)(bw + b.segs);
            da.nseg = (const unsigned*)(bw + b.nseg); da.recs = (const float4*)(bw + b.recs);
            hipLaunchKernelGGL(k_bwd_den_block, dim3(b.seg_cap < kAccGrid ? b.seg_cap : kAccGrid), dim3(kDenThreads), 0, sa, da);
            timing_end(f, T2N_K_BWD_DENSITY, sa);
            T2N_HIP(hipEventRecord((hipEvent_t)f->ev_den, sa));
        }
        // ---- sg: layer 2's weight gradient needs go and the kept h1 only (the chain writes g1 elsewhere in this form)
        T2N_HIP(hipStreamWaitEvent(sg, ev[5], 0));
        const WgradRegions WR = wgrad_regions(rows, 351);
        launch_bwd_l2((const float4*)go, (const float*)h1, rows, f->params_ref.mlp_w2, nullptr, nullptr, nullptr, (float*)((char*)part + WR.l2), sg, rows_dev);
        T2N_HIP(hipStreamWaitEvent(sg, ev[6], 0));
        timing_begin(f, T2N_K_BWD_WGRAD, sg);
        launch_gemm_tn(4, false, G1, 128, h0, 128, rows, 128, 128, g_w1, 128, (float*)((char*)part + WR.tn[0]), sg, nullptr, g_b1, rows_dev, false);
        launch_gemm_tn(4, false, G0, 128, xpe, 352, rows, 128, 351, g_w0, 351, (float*)((char*)part + WR.tn[1]), sg, feat32, g_b0, rows_dev, false);
        launch_gemm_tn(1, false, GF, 32, x144, 144, rows, 27, 144, g_basis, 144, (float*)((char*)part + WR.tn[2]), sg, nullptr, nullptr, rows_dev, false);
        launch_wgrad_reduce((const char*)part, rows, g_w2, g_b2, g_w1, g_w0, g_basis, sg, (const float*)(ws + T.part), R, A->w_depth, A->w_trans, A->losses);
        timing_end(f, T2N_K_BWD_WGRAD, sg);
        // ---- appearance scatter: records binned on sb, gradient buffer seeded on sa
        T2N_HIP(hipStreamWaitEvent(s, ev[4], 0));
        T2N_HIP(hipStreamWaitEvent(s, ev[1], 0));
        {
            TileAccumArgs ta;
            ta.S = f->dev.app; ta.G = gapp; ta.geom = ageom; ta.segs = (const int4*)(bw + b.a_segs);
            ta.nseg = (const unsigned*)(bw + b.a_nseg); ta.recs = (const float4*)(bw + b.a_recs);
            ta.gx = GX; ta.gx_ld = 144;
            static bool attr_set = false;
            if (!attr_set) {
                T2N_HIP(hipFuncSetAttribute((const void*)k_bwd_tile_accum<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                T2N_HIP(hipFuncSetAttribute((const void*)k_bwd_tile_accum<48, 8, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
                attr_set = true;
            }
            const size_t lds8 = tile_accum_lds(8, Lmax, 256);
            timing_begin(f, T2N_K_BWD_SCATTER, s);
            if (T2N_ACC_CG8 && lds8 <= 80 * 1024)   // two workgroups per CU: one accumulates while the other stages / zeroes / flushes
                hipLaunchKernelGGL((k_bwd_tile_accum<48, 8, 256>), dim3(b.a_seg_cap < kAccGrid ? b.a_seg_cap : kAccGrid, 6), dim3(256), lds8, s, ta);
            else
                hipLaunchKernelGGL((k_bwd_tile_accum<48>), dim3(b.a_seg_cap < kAccGrid ? b.a_seg_cap : kAccGrid, 3), dim3(kAccThreads), lds_acc, s, ta);
            timing_end(f, T2N_K_BWD_SCATTER, s);
        }
        T2N_HIP(hipGetLastError());
        if (!do_opt) {   // gradients final on `s`: join both side chains
            T2N_HIP(hipEventRecord(ev[7], sg));
            T2N_HIP(hipStreamWaitEvent(s, ev[7], 0));
            T2N_HIP(hipStreamWaitEvent(s, (hipEvent_t)f->ev_den, 0));
            return T2N_OK;
        }
    } else {
        // optimiser-only phase (data-parallel): the vote word has been all-reduced with the head gradients
        hipLaunchKernelGGL(k_train_commit, dim3(1), dim3(64), 0, s, st, (const float*)vote, f->train_host, slot);
        T2N_HIP(hipEventRecord(ev[6], s));
        T2N_HIP(hipStreamWaitEvent(sg, ev[6], 0));
    }
    // ---- optimiser. sg: Adam on the seven head tensors, then every packed form of the new head weights (the forward's fp32 and
    // split-f16 operands, the backward chain's transposed ones) beside the factor tensors' Adam on `s`
    timing_begin(f, T2N_K_HEAD_STEP, sg);
    if ((rc = launch_head_adam_dev(P, A->head_grads, A->exp_avg + 12, A->exp_avg_sq + 12, A->beta1, A->beta2, A->eps, sc, sg, true, (unsigned*)mlp_bwd_ss_absmax_words(gpack)))) return rc;
    f->params_ref.basis_weight = P->basis_weight;
    f->params_ref.mlp_w0 = P->mlp_w0; f->params_ref.mlp_b0 = P->mlp_b0; f->params_ref.mlp_w1 = P->mlp_w1; f->params_ref.mlp_b1 = P->mlp_b1;
    f->params_ref.mlp_w2 = P->mlp_w2; f->params_ref.mlp_b2 = P->mlp_b2;
    for (int k = 0; k < 3; ++k) {
        f->params_ref.density_plane[k] = P->density_plane[k]; f->params_ref.density_line[k] = P->density_line[k];
        f->params_ref.app_plane[k] = P->app_plane[k]; f->params_ref.app_line[k] = P->app_line[k];
    }
    if ((rc = train_repack(f, P, sg))) return rc;
    f->ss_dirty = true;
    timing_end(f, T2N_K_HEAD_STEP, sg);
    T2N_HIP(hipEventRecord(ev[7], sg));
    if (do_grad) {
        // the density factors (a quarter of the bytes) step on sa right behind their scatter, beside the appearance scatter on `s`
        if ((rc = launch_factor_adam_dev(f, P, A->exp_avg, A->exp_avg_sq, A->beta1, A->beta2, A->eps, sc, 0, 6, sa, sworld, srank))) return rc;
        T2N_HIP(hipEventRecord(ev[9], sa));
        f->train_chain = true;      // (the next call's early part may start behind this event)
        timing_begin(f, T2N_K_ADAM, s);
        if ((rc = launch_factor_adam_dev(f, P, A->exp_avg, A->exp_avg_sq, A->beta1, A->beta2, A->eps, sc, 6, 6, s, sworld, srank))) return rc;
        timing_end(f, T2N_K_ADAM, s);
        T2N_HIP(hipStreamWaitEvent(s, ev[9], 0));
    } else if ((rc = launch_factor_adam_dev(f, P, A->exp_avg, A->exp_avg_sq, A->beta1, A->beta2, A->eps, sc, 0, 12, s, sworld, srank))) return rc;
    T2N_HIP(hipStreamWaitEvent(s, ev[7], 0));
    T2N_HIP(hipGetLastError());
    return T2N_OK;
}

// ---- the step as a hipGraph: stream capture of the call above (relaxed mode: the library's side streams join the capture through
// their event waits), instantiated once, replayed per step
struct t2n_train_graph { hipGraph_t graph; hipGraphExec_t exec; int nodes; t2n_field* f; };

extern "C" int t2n_train_graph_capture(t2n_field* f, const t2n_train_step_args* a, t2n_stream stream, t2n_train_graph** out) {
    if (!f || !a || !out) { set_error("t2n_train_graph_capture: NULL argument"); return T2N_ERR_INVALID; }
    if (!f->train_dev || !f->train_packed) { set_error("t2n_train_graph_capture: run t2n_train_step eagerly once first (lazy state cannot be captured)"); return T2N_ERR_STATE; }
    // captured on a stream of its own: the caller's may be the legacy default stream, which cannot capture (hipErrorStreamCaptureUnsupported);
    // the instantiated graph is launched on whatever stream the caller passes to t2n_train_graph_launch
    (void)stream;
    hipStream_t s = nullptr;
    T2N_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const hipError_t e0 = hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    if (e0 != hipSuccess) { (void)hipStreamDestroy(s); return hip_fail(e0, "hipStreamBeginCapture"); }
    const int rc = t2n_train_step(f, a, (t2n_stream)s);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &g);
    (void)hipStreamDestroy(s);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess || !g) return hip_fail(e, "hipStreamEndCapture");
    hipGraphExec_t ex = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    if (e2 != hipSuccess) { (void)hipGraphDestroy(g); return hip_fail(e2, "hipGraphInstantiate"); }
    size_t n = 0;
    (void)hipGraphGetNodes(g, nullptr, &n);
    t2n_train_graph* tg = new t2n_train_graph{g, ex, (int)n, f};
    *out = tg;
    return T2N_OK;
}
extern "C" int t2n_train_graph_launch(t2n_train_graph* g, t2n_stream stream) {
    if (!g) { set_error("t2n_train_graph_launch: NULL graph"); return T2N_ERR_INVALID; }
    t2n_field* f = g->f;
    if (!f->train_packed) {   // the head was uploaded since the last step (load_state_dict, a step of another optimiser): the captured step
        // expects the backward chain's operands of the CURRENT weights, which an optimiser phase leaves behind and an upload does not
        const int rc = mlp_bwd_ss_pack(f, (char*)f->train_dev + 256, (hipStream_t)stream, false, true);
        if (rc) return rc;
        f->train_packed = true;
    }
    T2N_HIP(hipGraphLaunch(g->exec, (hipStream_t)stream));
    return T2N_OK;
}
extern "C" int t2n_train_graph_nodes(const t2n_train_graph* g) { return g ? g->nodes : 0; }
extern "C" int t2n_train_graph_destroy(t2n_train_graph* g) {
    if (!g) return T2N_OK;
    (void)hipGraphExecDestroy(g->exec);
    (void)hipGraphDestroy(g->graph);
    delete g;
    return T2N_OK;
}
