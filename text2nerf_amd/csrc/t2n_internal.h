// Internal declarations shared by the translation units of libt2n_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/t2n.h"

namespace t2n {

constexpr int kWave = 64;
constexpr int kMaxSamples = 8192;      // LDS window: 2 floats per sample per wave
constexpr int kAppDimMax = 32;

// matMode / vecMode of models/tensorBase.py:190-191
__host__ __device__ constexpr int mat0(int k) { return k == 2 ? 1 : 0; }
__host__ __device__ constexpr int mat1(int k) { return k == 0 ? 1 : 2; }
__host__ __device__ constexpr int vecm(int k) { return 2 - k; }

// Internal (channel-last) view of one VM factor set. plane k: [H=grid[mat1]][W=grid[mat0]][C]; line k: [L=grid[vec]][C].
struct FactorSet {
    const float* plane[3];
    const float* line[3];
    int W[3], H[3], L[3];
    int C;
    // bf16 factor storage (t2n_field_set_factor_storage): the same channel-last layouts with 2-byte texels, read by the
    // forward march / shade gathers; plane[] / line[] then hold the bf16-ROUNDED values as fp32 (backward, point queries,
    // grid operators). NULL in fp32 mode.
    const void* plane_h[3];
    const void* line_h[3];
};

struct FieldDev {
    FactorSet den, app;
    float aabb0[3], aabb1[3], inv[3];
    float shift, dscale, thres, step, near, far, zgate;
    int act, shading, app_dim;
    // MFMA-packed operands (see t2n_shade.hip for the layout)
    const float* basisA;   // [72][64]
    const float* w0A;      // [196][4][64]
    const float* w1A;      // [65][4][64]
    const float* w2A;      // [65][64]
    // optional AlphaGridMask occupancy volume [D][H][W] (models/tensorBase.py:41-59); NULL when the field has none
    const float* alpha; int aW, aH, aD; float a_min[3], a_inv[3];
    // split-f16 operands (t2n_shade.hip): uint4 = 8 halves per lane per (chunk, block, part)
    const uint4* basisH; const uint4* w0H; const uint4* w1H; const uint4* w2H; const float* biasH;
    // NDC sampling (T2N_FLAG_NDC, models/tensorBase.py:293-302,441-446): per-call table of the n_samples depths shared by all
    // rays; NULL on the regular path (z_i = t_min + step * (i [+ u]))
    const float* ztab;
    // early ray termination (eval launches that materialise neither weights nor z_vals nor a backward context; 0 = off): a ray whose
    // transmittance fell below term_eps evaluates no further sample. The reference never terminates (models/tensorBase.py:19-26,
    // 494-505: every in-box sample is evaluated); what the rest of a ray could add is bounded by T: acc < eps, rgb < eps, depth < eps z_max.
    float term_eps;
};

// Device-side state of the fused training step (t2n_train_step): one 256-B block per field. k_train_plan advances it once per step;
// every optimiser kernel of the step reads its verdict and its scalars from here (nothing about a step is a by-value kernel argument,
// so a captured step can be replayed).
// what the optimiser kernels of ONE step read: two slots, used alternately — the plan of step k + 1 may run (beside step k's tail: the
// pipelined form of the call) while step k's Adam kernels still read theirs
struct TrainScalars {
    float lr_over_bc1[19]; // lr / (1 - beta1^step) per tensor, step = the step being applied
    float inv_bc2_sqrt;    // 1 / sqrt(1 - beta2^step)
    unsigned skip;         // verdict: 1 = the optimiser kernels apply nothing
    unsigned pad[3];
};
struct TrainState {
    unsigned step;         // Adam steps applied so far
    unsigned seq;          // fused steps issued (plan kernels run)
    unsigned skipped;      // steps whose update was withheld (appearance rows beyond the capacity)
    unsigned pad[5];
    TrainScalars sc[2];
};

constexpr int kTimingEvents = 1024;   // timed launches per kernel between two reads; launches beyond are counted and priced at the timed average
struct TimingSlot {
    hipEvent_t start[kTimingEvents], stop[kTimingEvents];
    int used = 0;
    int64_t untimed = 0;
    double ms = 0.0;
    int64_t launches = 0;
};

}  // namespace t2n

struct t2n_field {
    t2n_field_desc desc;
    t2n::FieldDev dev;
    // owned device buffers
    float* buf_den_plane[3] = {nullptr, nullptr, nullptr};
    float* buf_den_line[3] = {nullptr, nullptr, nullptr};
    float* buf_app_plane[3] = {nullptr, nullptr, nullptr};
    float* buf_app_line[3] = {nullptr, nullptr, nullptr};
    // bf16 copies of the factor buffers (factor_bf16 mode only)
    void* hbuf_den_plane[3] = {nullptr, nullptr, nullptr};
    void* hbuf_den_line[3] = {nullptr, nullptr, nullptr};
    void* hbuf_app_plane[3] = {nullptr, nullptr, nullptr};
    void* hbuf_app_line[3] = {nullptr, nullptr, nullptr};
    int factor_bf16 = 0;
    float* buf_mlp = nullptr;  // basisA | w0A | w1A | w2A
    void* buf_mlp_h = nullptr; // split-f16 operands + scaled biases
    const unsigned* split_unsafe = nullptr;   // device word (in buf_mlp_h): a weight x 2^8 left the f16 range at the last upload
    void* buf_ss = nullptr;    // sample-stationary head operands (t2n_mlp_ss.hip), packed lazily from params_ref
    bool ss_dirty = true;
    void* ss_event = nullptr; void* ss_stream = nullptr;   // the pack's stream + an event behind it: a render on ANOTHER stream waits for it
    unsigned* ss_ok_host = nullptr; void* ss_ok_event = nullptr; int ss_variant = -2;   // head instantiation the packed weights allow: -1 copy in flight, -2 unknown (both launched), 0 untracked, 1 tracked
    float* buf_alpha = nullptr; // alpha-mask volume copy
    int mlp_split = 1;         // 1: f16 two-way split products (default), 0: exact fp32 MFMA
    // channel-last gradient accumulators (backward), allocated on first use
    float* gbuf_den_plane[3] = {nullptr, nullptr, nullptr};
    float* gbuf_den_line[3] = {nullptr, nullptr, nullptr};
    float* gbuf_app_plane[3] = {nullptr, nullptr, nullptr};
    float* gbuf_app_line[3] = {nullptr, nullptr, nullptr};
    bool gbuf_external = false;   // caller-owned (t2n_field_set_grad_buffer): never freed, never zeroed by the backward
    float* gbuf_all = nullptr; size_t gbuf_bytes = 0;   // the 12 gradient buffers are slices of ONE allocation (one memset per backward)
    t2n_field_params params_ref;   // reference-layout parameter pointers of the last upload (backward reads W^T operands)
    bool uploaded = false;
    int timing = 0;
    int frame_w = 0;           // image width hint for the tile marcher (0: unknown)
    int head_rows_per_ray = 32;   // general view-dependent heads: activation-scratch rows per ray a sub-launch reserves (t2n_field_set_head_scratch_rows)
    float term_eps = 0.f;      // early ray termination threshold of eval launches (t2n_field_set_early_termination; 0 = off)
    t2n::TimingSlot slots[T2N_K_COUNT];
    // optimistic (budgeted) render launches: the counters travel to pinned host memory behind the march kernels; the entries a
    // ray needed last time size the next call's lists (t2n_render_workspace_bytes_hint)
    static constexpr int kCountSlots = 4;
    struct CountSlot { unsigned* host = nullptr; void* ev = nullptr; bool pending = false; int64_t n_rays = 0; int n_samples = 0; unsigned budget = 0; };
    CountSlot count_slots[kCountSlots]; int count_next = 0;
    void* side_stream = nullptr; void* ev_fork = nullptr; void* ev_join = nullptr;   // (the streams are the process-wide pair of shared_side_streams: not owned)   // backward: the density scatter runs beside the MLP backward
    void* ev_den = nullptr;      // recorded behind the density scatter of the last backward (t2n_field_wait_density_grads)
    unsigned* plan_host = nullptr; unsigned plan_seq = 0;   // T2N_FLAG_DEVICE_ROWS: k_bwd_plan's record in pinned host memory (t2n_field_device_rows_record)
    void* ev_pack = nullptr;   // backward: k_mlp_bwd_ss's operand packing (on gemm_stream) is done
    void* gemm_stream = nullptr; void* ev_fork2 = nullptr; void* ev_join2 = nullptr;  // backward: the weight-gradient GEMMs run beside the appearance scatter
    // fused training step (t2n_train_step): device state + the backward chain's packed operands (one allocation), the pinned host record,
    // a third side stream for the plan + appearance binning and the events of the call's fork / join graph
    void* train_dev = nullptr; unsigned* train_host = nullptr; void* train_ev[12] = {};
    unsigned train_calls = 0; bool train_chain = false;   // pipelined steps: stream of a step's early part; calls so far (workspace / scalar slot parity); the previous call left its density-Adam event
    bool train_packed = false;   // the backward chain's operands in train_dev are those of the current head weights
    unsigned list_hint = 0;          // appearance entries per ray of the last budgeted launch (0: unknown)
    unsigned long long list_retries = 0;
};

namespace t2n {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define T2N_HIP(call)                                        \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return t2n::hip_fail(e__, #call); \
    } while (0)

void counts_poll(t2n_field* f, bool wait);   // budgeted launches' counters that reached the host -> list_hint / list_retries (t2n_api.hip)
// timing helpers (t2n_api.hip)
void timing_begin(t2n_field* f, int k, hipStream_t s);
void timing_end(t2n_field* f, int k, hipStream_t s);

// launchers implemented per translation unit
int launch_relayout(t2n_field* f, const t2n_field_params* p, hipStream_t s);
int launch_pack_mlp(t2n_field* f, const t2n_field_params* p, hipStream_t s, unsigned* absmax_out = nullptr);   // absmax_out: 4 zeroed words, see PackArgs

struct RenderLaunch {
    const float* rays; int64_t n_rays; int ray_stride; int n_samples; uint32_t flags;
    const float* jitter; float* rgb; float* depth; float* weights; float* z_vals; uint64_t* stats;
    // workspace carve (one sub-launch)
    float* acc; int4* ray_app; unsigned* counters; float4* app_pos; int* app_ray; float4* app_rgb; unsigned list_cap;
    float* feat = nullptr; unsigned feat_rows = 0;   // appearance feature rows [feat_rows][32] between the gather + basis kernel and the head
    float* sigma_ctx; float4* rgb_raw;   // KEEP_CTX only, else NULL
};
int launch_march(t2n_field* f, const RenderLaunch& L, hipStream_t s);
int launch_ray_stats(const RenderLaunch& L, hipStream_t s);
// spill: [n_rays][n_samples] scratch rows (used only when L.weights is NULL); scratch: [n_rays][n_samples / 4] staging entries
int launch_march_tiles(t2n_field* f, const RenderLaunch& L, int img_w, int img_h, float* spill, float4* scratch, hipStream_t s);
constexpr int kLists = 8;   // appearance sub-lists per sub-launch
constexpr int kCounterStride = 64;   // unsigned words between sub-list counters: one 256-B line each (same-line atomics serialise)
// list_cap(n_rays, N): worst-case entries of one sub-list = rays of the largest XCD run x samples
inline unsigned list_capacity(long long n_rays, int n_samples) {
    const unsigned nblocks = (unsigned)((n_rays + 3) / 4);
    // + one ray's worth of slack per list: reservations that do not fit a list move on to the next one (compact_ray), and
    // with n_samples spare entries per list a ray that fits nowhere would imply more entries than rays x samples
    return ((nblocks >> 3) + ((nblocks & 7u) ? 1u : 0u)) * 4u * (unsigned)n_samples + (unsigned)n_samples;
}
// Budgeted form (optimistic launches of t2n_render_forward): `budget` appearance entries per ray on average over the launch,
// split evenly over the sub-lists, + the same per-list slack. A launch that does not fit raises word kOverflowWord of the counter
// block (the rays concerned keep no appearance samples) and is redone by the caller with worst-case lists.
inline unsigned list_capacity_budget(long long n_rays, int n_samples, unsigned budget) {
    const unsigned worst = list_capacity(n_rays, n_samples);
    if (budget == 0 || budget >= (unsigned)n_samples) return worst;
    const unsigned long long per = ((unsigned long long)n_rays * budget + kLists - 1) / kLists + (unsigned)n_samples;
    return per < worst ? (unsigned)per : worst;
}
// Activation rows kept by the shade kernel in ctx mode (row = tile * 32 + lane sample; zero rows past a sub-list's end)
struct ShadeCtx { float* x144; float* feat32; float* h0; float* h1; };
int launch_shade_list(t2n_field* f, const float4* app_pos, const int* app_ray, const float* rays, int ray_stride,
                      const unsigned* counters_dev, unsigned list_cap, float4* app_rgb, const ShadeCtx* ctx, hipStream_t s,
                      bool features_only = false, unsigned ctx_rows = 0xffffffffu, float* feat = nullptr, unsigned feat_rows = 0,
                      uint64_t* stats = nullptr, unsigned tile_lo = 0, unsigned tile_hi = 0xffffffffu);   // tile_lo / tile_hi: the one-kernel paths cover these tiles only
// feat / feat_rows: scratch rows for the two-kernel default path (features -> sample-stationary head, t2n_mlp_ss.hip); tiles
// past the capacity take the one-kernel path. Word kRangeFlagWord of the counter block is the head's f16-range flag.
constexpr int kRangeFlagWord = 32;
// Words 33 / 34 of the counter block of a KEEP_CTX forward: the forward states that it keeps the MLP activation rows (magic) and
// with which capacity; the backward takes the kept path only when both match what it derives from ITS workspace size, and clears
// the statement once it has consumed the rows (h0 / h1 are overwritten in place).
constexpr int kKeptMagicWord = 33, kKeptRowsWord = 34;
constexpr int kOverflowWord = 35;   // raised by the march kernels when a ray's appearance entries fit no sub-list (budgeted lists only)
constexpr int kHeadPlanWord = 40;      // general view-dependent heads: the forward's device-side plan (HeadPlanDev, 10 words) lives here
constexpr int kFailEntriesWord = 36;   // appearance entries of the rays that fit no sub-list (finished by k_finish_rays): the next budget counts them
constexpr unsigned kKeptMagic = 0x4b455054u;   // "KEPT"
int launch_mlp_ss(t2n_field* f, const float* feat, const unsigned* counters_dev, unsigned list_cap, unsigned tile_hi, float4* app_rgb,
                  unsigned* range_flag, hipStream_t s);
// features_only: gather + basis stages only (the general heads take over); ctx_rows: capacity of the ctx buffers in rows

// the general (unfused) view-dependent heads (t2n_heads.hip): MLP_Fea / MLP_PE / MLP input layout and launchers
struct HeadDims { int shading, C, fea_pe, view_pe, pos_pe, o_feat, o_view, o_pe_a, n_pe_a, o_pe_v, n_pe_v, K0, K0pad; };
HeadDims head_dims(const t2n_field_desc& d);
inline bool head_is_generic(int shading) { return shading == T2N_SHADE_MLP_FEA || shading == T2N_SHADE_MLP_PE || shading == T2N_SHADE_MLP; }
struct HeadPlanDev { unsigned t[kLists + 1]; unsigned rows; };   // device-side plan of a general-head forward: tile prefix, row count
int launch_head_plan(const unsigned* counters, unsigned list_cap, HeadPlanDev* plan, hipStream_t s);
// OUT[rows, N] = act(IN[rows, K] Wt[N, K]^T + bias) on exact-fp32 MFMA (k_dense); rows clipped to plan->rows - row0 on the device
int launch_dense_rows(const float* IN, int ldin, const float* Wt, int K, int N, const float* bias, int relu, long long rows, float* OUT, int ldo,
                      hipStream_t s, const HeadPlanDev* plan, long long row0);
// plan == NULL: rows / tiles_before from the host (the backward's recompute). plan != NULL: one PASS over rows [row0, row0 + rows) of the
// call, scratch row = row - row0, clipped to the plan's count on the device (tiles_before may be NULL)
int launch_head_forward(t2n_field* f, const unsigned tiles_before[kLists + 1], long long rows, const float* feat32, const float4* app_pos,
                        const int* app_ray, const float* rays, int ray_stride, const unsigned* counters, unsigned list_cap, float* x0,
                        float* h0, float* h1, float4* app_rgb, hipStream_t s, const HeadPlanDev* plan = nullptr, long long row0 = 0);
int launch_head_in_bwd(t2n_field* f, const float* gx, const float* feat32, long long rows, float* gf, hipStream_t s);
// backward of the parameter-free SH / RGB heads: dL/dfeatures rows from the per-sample colour gradients (t2n_heads.hip)
// Several small device-side initialisations as ONE launch (every hipMemsetAsync / hipMemsetD32Async is a 5-us kernel of its own on the
// stream): up to 6 regions zeroed (4-byte words) and up to 4 words set.
struct SetupOps {
    unsigned* zero_ptr[6]; unsigned long long zero_words[6]; int nz = 0;
    unsigned* set_ptr[4]; unsigned set_val[4]; int ns = 0;
    bool overflow = false;   // more entries than slots: launch_setup refuses (never silently drops an initialisation)
    void zero(void* p, size_t bytes) {
        if (!p || !bytes) return;
        if (nz >= 6) { overflow = true; return; }
        zero_ptr[nz] = (unsigned*)p; zero_words[nz] = (bytes + 3) / 4; ++nz;
    }
    void set(void* p, unsigned v) {
        if (ns >= 4) { overflow = true; return; }
        set_ptr[ns] = (unsigned*)p; set_val[ns] = v; ++ns;
    }
};
int launch_setup(const SetupOps& o, hipStream_t s);
int launch_simple_head_bwd(t2n_field* f, const unsigned tiles_before[kLists + 1], long long rows, const float4* go, const float4* app_rgb,
                           const int* app_ray, const float* rays, int ray_stride, const unsigned* counters, unsigned list_cap, float* gf,
                           hipStream_t s);

// input-gradient GEMMs of the MLP backward on the f16 matrix cores (t2n_gemm_h.hip)
size_t gemm_h_pack_bytes(int K0);
int gemm_h_pack(t2n_field* f, void* buf, int K0, hipStream_t s);
int launch_gemm_nn_h(void* packbuf, int which, int K0, const float* IN, int ldin, long long rows, const float* ACT, int ldact, float* OUT,
                     int ldo, hipStream_t s);
int launch_gemm_tn_b(const float* A, int lda, const float* B, int ldb, long long rows, int N, float* part, int ldp, int chunk_rows,
                     int ng, int chunks, bool pe, float* db, hipStream_t s, const unsigned* rows_dev = nullptr);
// MLP part of the render backward: fp32-MFMA GEMMs, layer 2, bias sums, positional encoding (t2n_bwd_mlp.hip)
size_t tn_part_bytes(int64_t rows, int k0);   // partial-sum scratch of the weight-gradient GEMMs (and of layer 2) for `rows` rows
bool gemm_fp32_mode(const t2n_field* f);
void launch_bwd_l2(const float4* go, const float* h1, long long rows, const float* w2, float* g1, float* dw2, float* db2, float* scratch,
                   hipStream_t s, const unsigned* rows_dev = nullptr);   // rows_dev: the row count in device memory (rows = capacity then)
void launch_gemm_tn(int MB, bool fp32, const float* A, int lda, const float* B, int ldb, long long rows, int M, int N, float* C, int ldc,
                    float* part, hipStream_t s, const float* pe_feat = nullptr, float* db = nullptr, const unsigned* rows_dev = nullptr,
                    bool reduce = true);   // reduce = false: the chunk partials stay in `part` (launch_wgrad_reduce adds them up later)
struct WgradRegions { size_t l2, tn[3], total; };   // byte offsets of the partial-sum regions in the backward's `part` buffer
WgradRegions wgrad_regions(int64_t rows, int k0);
void launch_wgrad_reduce(const char* base, long long rows, float* dw2, float* db2, float* dw1, float* dw0, float* dwb, hipStream_t s,
                         const float* loss_part = nullptr, long long n_rays = 0, float w_depth = 0.f, float w_trans = 0.f, float* losses = nullptr);
void launch_gemm_nn(const float* IN, int ldin, const float* W, int ldw, long long rows, int K, int N, const float* ACT, int ldact, float* OUT,
                    int ldo, hipStream_t s);
void launch_colsum(const float* G, int ld, long long rows, int N, float* db, hipStream_t s);
void launch_pe_fwd(const float* feat, long long rows, float* x, hipStream_t s);
void launch_pe_bwd(const float* gx, const float* feat, long long rows, float* gf, hipStream_t s);
// fused input-gradient chain of the MLP_Fea_noview head's backward (t2n_mlp_bwd_ss.hip)
size_t mlp_bwd_ss_pack_bytes();
int launch_mlp_bwd_ss(t2n_field* f, void* packbuf, const float4* go, float* h1, const float* h0, const float* feat, float* g0, float* gf,
                      float* gx, long long rows, hipStream_t s, bool packed = false, const unsigned* rows_dev = nullptr,
                      float* g1_out = nullptr);   // g1_out: g1 written there instead of over h1 (h1 stays intact for a concurrent k_bwd_l2)
int mlp_bwd_ss_pack(t2n_field* f, void* packbuf, hipStream_t s, bool zeroed, bool one_launch = false, bool absmax_done = false);   // one_launch: the pack kernel finds the matrices' largest magnitudes itself
void* mlp_bwd_ss_absmax_words(void* packbuf);
// the fused training step's optimiser launches (t2n_optim.hip): scalars / TV weights / verdict from device memory
int launch_tv_seed_dev(t2n_field* f, const float* tvw_dev, hipStream_t s, int world = 1, int rank = 0);
int launch_factor_adam_dev(t2n_field* f, const t2n_field_params* params, float* const* m, float* const* v, float beta1, float beta2, float eps,
                           const TrainScalars* st, int first, int count, hipStream_t s, int world = 1, int rank = 0, bool relayout_only = false);
void shard_partition(const t2n_field* f, int world, int rank, unsigned lo[12], unsigned hi[12], unsigned body[12]);
int launch_head_adam_dev(const t2n_field_params* params, const float* grads_flat, float* const* m, float* const* v, float beta1, float beta2,
                         float eps, const TrainScalars* st, hipStream_t s, bool zero_grads = false, unsigned* zero_words4 = nullptr);
// the driver's loss (t2n_loss.hip); reduce = false leaves the per-workgroup partial sums [ceil(n_rays / 4)][3] in `part`
int launch_train_loss(const float* rgb, const float* depth, const float* weights, const float* z_vals, const float* rgb_t, const float* depth_t,
                      int64_t n_rays, int n_samples, float w_depth, float w_trans, float delta, float* d_rgb, float* d_depth, float* d_weights,
                      float* losses, float* part, bool reduce, hipStream_t s);
// forward-workspace carve shared by forward and backward (t2n_api.hip)
struct Carve { size_t acc, ray_app, counters, app_pos, app_ray, app_rgb, sigma, rgb_raw, scratch, feat, total; unsigned list_cap, feat_rows; };
Carve carve_workspace(int64_t rays, int n_samples, bool ctx, bool feat = true, unsigned budget = 0);   // budget: appearance entries per ray (0: worst case)   // ctx: also room for sigma [rays,N] and rgb_raw [rays]; feat: feature rows (last region: the other offsets do not depend on it; KEEP_CTX calls carve without)
int launch_composite(t2n_field* f, const RenderLaunch& L, hipStream_t s, int img_w = 0, int img_h = 0);
// Tile-marcher launches: the rays whose appearance entries fitted no sub-list (budgeted lists) are shaded and composited straight from
// their staging slices / spill rows by one more kernel behind k_composite (t2n_shade.hip); a launch without such rays costs one
// empty kernel. Heads the finisher does not evaluate (the general view-dependent MLPs) keep the host-side retry.
inline bool finish_supported(const t2n_field* f) { return !head_is_generic(f->desc.shading); }
int launch_finish_rays(t2n_field* f, const RenderLaunch& L, const float* spill, const float4* scratch, hipStream_t s);
int post_counts(const unsigned* counters_dev, unsigned* host_dst, hipStream_t s);   // counter block -> pinned host memory, by a device store (t2n_api.hip)
int ctx_counts_post(const void* ws, const unsigned* counters_dev, hipStream_t s);   // KEEP_CTX forward: counts -> pinned host copy + event (t2n_backward.hip)
// Activation rows the forward keeps for the backward when the KEEP_CTX workspace is larger than the context itself (the
// caller's guess of the appearance-row count; 1728 B per row): x144 [rows,144], feat32 [rows,32], h0 / h1 [rows,128] behind
// the carved context. rows == 0: nothing kept (the backward re-runs the appearance forward).
struct KeptRows { size_t x144, feat32, h0, h1; unsigned rows; };
inline KeptRows kept_rows(size_t ctx_total, size_t workspace_bytes) {
    KeptRows k{0, 0, 0, 0, 0};
    const size_t base = (ctx_total + 255) / 256 * 256;
    if (workspace_bytes <= base) return k;
    size_t rows = (workspace_bytes - base) / ((144 + 32 + 128 + 128) * 4);
    rows = rows / 32 * 32;
    if (rows > 0x7fffffe0u) rows = 0x7fffffe0u;
    k.rows = (unsigned)rows;
    k.x144 = base; k.feat32 = k.x144 + rows * 144 * 4; k.h0 = k.feat32 + rows * 32 * 4; k.h1 = k.h0 + rows * 128 * 4;
    return k;
}

}  // namespace t2n
