/*
 * t2n.h — C-ABI of libt2n_hip.so: the MI355X (gfx950) TensoRF VM-split ray-marching renderer that replaces the
 * PyTorch op sequence behind Text2NeRF's  OctreeRender_trilinear_fast(rays, tensorf, ...)  /  TensorVMSplit.forward.
 *
 * The reference has no native code on this path (it is ~30 eager ATen ops per chunk), so there is no foreign-function
 * interface to mirror; each entry point below cites the reference Python it replaces (paths relative to the reference
 * repository). INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: pointers + sizes only; every pointer is a DEVICE pointer owned by the caller unless marked "host".
 *   - every call returns 0 on success or a negative T2N_ERR_* code; t2n_last_error() gives a thread-local message.
 *     Nothing throws across the boundary. Unsupported configurations are rejected loudly (T2N_ERR_UNSUPPORTED): there is
 *     no CPU fallback in this library.
 *   - work is enqueued on the caller's HIP stream (pass torch.cuda.current_stream().cuda_stream); calls do not
 *     synchronise unless documented. The library allocates device memory only inside t2n_field_create/_upload (the
 *     channel-last copy of the factor tensors) and, once, in the first t2n_train_step of a field (its training state: a few hundred
 *     KB); per-call scratch comes from the caller's workspace.
 *   - a handle may be used from one host thread at a time. This includes the `const t2n_field*` queries t2n_render_workspace_bytes_hint
 *     and t2n_field_list_retries: they consume the counters that budgeted launches posted (and the latter waits for those still in
 *     flight), i.e. they update the field's hint state.
 */
#ifndef T2N_H_
#define T2N_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct t2n_field t2n_field;      /* opaque */
typedef void* t2n_stream;                /* hipStream_t */

enum {
    T2N_OK = 0,
    T2N_ERR_INVALID = -1,      /* bad argument */
    T2N_ERR_UNSUPPORTED = -2,  /* configuration the HIP path does not implement */
    T2N_ERR_HIP = -3,          /* a HIP runtime call failed */
    T2N_ERR_WORKSPACE = -4,    /* caller workspace too small */
    T2N_ERR_STATE = -5         /* e.g. render before upload, backward without a kept context */
};

/* shading heads: models/tensorBase.py:200-216 */
enum { T2N_SHADE_MLP_FEA_NOVIEW = 0, T2N_SHADE_SH = 1, T2N_SHADE_RGB = 2,
       /* view-dependent heads (models/tensorBase.py:62-86,111-159): general unfused path, csrc/t2n_heads.hip */
       T2N_SHADE_MLP_FEA = 3, T2N_SHADE_MLP_PE = 4, T2N_SHADE_MLP = 5 };
/* fea2denseAct: models/tensorBase.py:406-410 */
enum { T2N_ACT_SOFTPLUS = 0, T2N_ACT_RELU = 1 };

/* render flags */
enum {
    T2N_FLAG_TRAIN = 1u,      /* is_train: per-ray jitter, no z gate (models/tensorBase.py:314-316,459) */
    T2N_FLAG_ADD_BG = 2u,     /* rgb_map += 1-acc  (white_bg, or the train-time coin; models/tensorBase.py:497-498) */
    T2N_FLAG_KEEP_CTX = 4u,   /* keep the per-call context in the workspace for t2n_render_backward */
    T2N_FLAG_NDC = 16u,       /* ndc_ray=True (models/tensorBase.py:293-302,441-446): the sample depths are a table shared by all
                               * rays — pass it through the `jitter` argument: [n_samples] floats = torch.linspace(near, far, N)
                               * (+ the shared train-time jitter, already added); dists are scaled by |d| and the SH head sees
                               * normalised view directions. Not used by the Text2NeRF driver (ndc_ray=0). */
    T2N_FLAG_DEVICE_ROWS = 32u, /* t2n_render_backward only: read NO count on the host. The appearance row capacity is what the two
                               * workspaces hold (the forward's kept activation rows; the backward's row buffers); the actual row count
                               * and tile prefix are derived on the device and every row kernel clips to them. Rows past the capacity
                               * lose their appearance gradient (t2n_field_device_rows_record tells the caller afterwards). ONE backward per
                               * forward in this mode: the kept rows are consumed in place, a second call on the same context finds the
                               * forward's statement cleared and computes no appearance gradients. Needs the fused
                               * MLP_Fea_noview head (27/6/128, 48 comps), the binned scatters and all head gradient tensors. */
    T2N_FLAG_PIPELINE = 64u,  /* t2n_train_step only: the pipelined form is allowed (the caller has not touched the field since its previous
                               * t2n_train_step) */
    T2N_FLAG_COHERENT = 8u    /* hint (eval): rays are a row-major image whose width was given by t2n_field_set_frame_width:
                                 march 8x8-pixel tiles whose rays share one texel x line-row dot-product table per step
                                 (f32 matrix cores). Same samples; the density feature is summed in table order and the
                                 transmittance is a sequential product instead of a wave scan (weights agree to ~2e-6) */
};

/* Scalars of TensorBase.__init__/update_stepSize (models/tensorBase.py:163-231), computed by the host mirror. */
typedef struct t2n_field_desc {
    float aabb_min[3], aabb_max[3];
    float inv_aabb_size[3];        /* 2/(aabb_max-aabb_min), fp32 as the reference computes it (:224) */
    int32_t grid[3];               /* gridSize (x,y,z) */
    int32_t density_n_comp;        /* per plane; the three planes must agree */
    int32_t app_n_comp;
    int32_t app_dim;               /* 27 (MLP/SH heads) or 3 (RGB head) */
    int32_t shading;               /* T2N_SHADE_* */
    int32_t fea_pe;                /* 6 */
    int32_t feature_c;             /* 128 */
    int32_t act;                   /* T2N_ACT_* */
    float density_shift;           /* -10 */
    float distance_scale;          /* 25 */
    float weight_thres;            /* rayMarch_weight_thres 1e-4 */
    float step_size;               /* stepSize */
    float near, far;               /* near_far */
    float z_gate;                  /* 2.0: the eval-only world-z gate (models/tensorBase.py:459-462) */
    int32_t view_pe, pos_pe;       /* octaves of the view-direction / position encodings of the MLP_Fea, MLP_PE, MLP heads */
} t2n_field_desc;

/* Reference-layout parameter pointers (device, fp32), i.e. the tensors of TensorVMSplit.state_dict()
 * (models/tensoRF.py:144-160): plane k is [1,C,grid[matMode[k][1]],grid[matMode[k][0]]], line k is [1,C,grid[vecMode[k]],1],
 * basis_mat.weight [app_dim, 3*app_n_comp], renderModule.mlp.{0,2,4}.{weight,bias} (NULL for SH / RGB heads). */
typedef struct t2n_field_params {
    const float* density_plane[3];
    const float* density_line[3];
    const float* app_plane[3];
    const float* app_line[3];
    const float* basis_weight;
    const float* mlp_w0; const float* mlp_b0;
    const float* mlp_w1; const float* mlp_b1;
    const float* mlp_w2; const float* mlp_b2;
} t2n_field_params;

/* Gradient buffers in the SAME reference layouts (device, fp32, caller-zeroed or accumulated into). */
typedef struct t2n_field_grads {
    float* density_plane[3];
    float* density_line[3];
    float* app_plane[3];
    float* app_line[3];
    float* basis_weight;
    float* mlp_w0; float* mlp_b0;
    float* mlp_w1; float* mlp_b1;
    float* mlp_w2; float* mlp_b2;
} t2n_field_grads;

/* Per-call counters written by the render kernels (device memory, 8 x uint64): the units of the roofline model. */
enum { T2N_STAT_EVALUATED = 0,   /* V: in-box (and z-gated) samples that read the density factors */
       T2N_STAT_APPEARANCE = 1,  /* A: samples with weight > weight_thres that read the appearance factors */
       T2N_STAT_RAYS = 2,
       T2N_STAT_OVERFLOW = 3,    /* must stay 0 */
       T2N_STAT_F16_REDO = 4,    /* sub-launches whose split-f16 appearance stage met a value outside the f16 range and were
                                    redone on the exact fp32 path (results are the exact path's) */
       T2N_STAT_LIST_RETRY = 5,  /* 1: some rays of the call found no room in its budgeted appearance lists and were shaded and
                                    composited by the per-ray finisher (k_finish_rays: exact-fp32 head, no list memory). (The general
                                    view-dependent heads always render with worst-case lists.) */
       T2N_STAT_COUNT = 8 };

const char* t2n_last_error(void);
int t2n_version(void);

/* ---- field lifetime: replaces TensorVMSplit.__init__ containers + the implicit "weights are where ATen reads them" */
int t2n_field_create(const t2n_field_desc* desc, t2n_field** out);
int t2n_field_destroy(t2n_field* f);
/* Re-read the reference-layout parameters into the internal channel-last / MFMA-packed device copy. Call after
 * construction, after load_state_dict, and after every optimiser step. Asynchronous on `stream`. */
int t2n_field_upload(t2n_field* f, const t2n_field_params* p, t2n_stream stream);
/* Update scalars only (step size, near/far, ...): TensorBase.update_stepSize (models/tensorBase.py:220-231). */
int t2n_field_set_desc(t2n_field* f, const t2n_field_desc* desc);
/* Factor storage for the forward render: 0 = fp32 (default), 1 = bf16 (BASELINE.json configs[4]). In bf16 mode
 * t2n_field_upload rounds the 12 plane / line tensors to bf16 (nearest-even) and keeps two device copies: 2-byte texels for
 * the march / shade gathers (half the bytes through L1) and the rounded values as fp32 for every other entry point, so a
 * render equals the fp32 render of the bf16-rounded tensors BIT FOR BIT; gradients are taken at the rounded values and
 * returned in fp32 for the caller's fp32 master copy. basis_mat / MLP stay fp32. Call before t2n_field_upload. */
int t2n_field_set_factor_storage(t2n_field* f, int bf16);

/* Arithmetic of the basis/MLP contractions (nn.Linear in the reference, models/tensoRF.py:239, tensorBase.py:94-106):
 * exact_fp32 = 0 (default): every fp32 product as three f16 MFMA products of hi/lo splits (22-bit mantissa), fp32
 * accumulation; exact_fp32 = 1: v_mfma_f32_32x32x2_f32, bit-exact fp32 FMA chains (5x the matrix-core time). The backward
 * pass always recomputes activations with the exact path. */
int t2n_field_set_mlp_precision(t2n_field* f, int exact_fp32);
/* a-7 (optional): AlphaGridMask (models/tensorBase.py:41-59). volume = device fp32 [D(z),H(y),W(x)] occupancy (copied);
 * aabb_min / inv_size (= 1/aabbSize*2, fp32 as the reference computes it) are HOST float[3]. Samples whose trilinear
 * mask value is not > 0 are dropped from ray_valid (:451-456). volume = NULL removes the mask. The tile marcher is
 * bypassed while a mask is set. t2n_alpha_at = AlphaGridMask.sample_alpha at world points. */
int t2n_field_set_alpha_mask(t2n_field* f, const float* volume, int D, int H, int W, const float* aabb_min_host,
                             const float* inv_size_host, t2n_stream stream);
int t2n_alpha_at(const t2n_field* f, const float* xyz_world, int64_t n, float* alpha, t2n_stream stream);
/* Early ray termination for EVAL launches that materialise neither weights nor z_vals (and keep no backward context): once a ray's
 * transmittance T fell below eps it evaluates no further sample — per 8x8-pixel tile in the tile marcher (the wave leaves its step
 * loop when all 64 rays are done), per 64-sample block in the per-ray marcher. The reference has no such mode: it evaluates every
 * in-box sample whatever T is (models/tensorBase.py:19-26 raw2alpha, :494-505 compositing), so this is a bounded deviation, not
 * parity: what the skipped samples could add is < eps to acc, < eps to each colour channel, < eps * (z_max - d_z) to depth; the
 * evaluated-sample count (T2N_STAT_EVALUATED) becomes a lower bound of the reference's. eps in [0, weight_thres]; 0 (default of a new
 * handle) = off = the reference's arithmetic sample for sample. Train launches, launches with weights / z_vals outputs and
 * T2N_FLAG_KEEP_CTX launches ignore it. */
int t2n_field_set_early_termination(t2n_field* f, float eps);
/* Image width of the row-major frames passed with T2N_FLAG_COHERENT (0 = unknown: the flag is ignored). */
int t2n_field_set_frame_width(t2n_field* f, int width);

/* ---- a-1/a-2: get_ray_directions (dataLoader/ray_utils.py:24-42), normalisation (dataLoader/scene_gen.py:45),
 *      get_rays (dataLoader/ray_utils.py:66-87) */
int t2n_ray_directions(int H, int W, float fx, float fy, float cx, float cy, int normalize, float* dirs /*[H*W,3]*/,
                       t2n_stream stream);
int t2n_get_rays(const float* dirs /*[n,3]*/, int64_t n, const float* c2w_host /*[3 rows x 4] row-major, host*/,
                 float* rays_o /*[n,3] or NULL*/, float* rays_d /*[n,3] or NULL*/, float* rays6 /*[n,6] or NULL*/,
                 t2n_stream stream);
/* fused: pixel -> [H*W,6] ray rows (ox,oy,oz,dx,dy,dz), SceneGen recipe (scene_gen.py:44-45,92-94) */
int t2n_generate_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w_host, float* rays6,
                      t2n_stream stream);

/* ---- a-5 as a stage of its own: TensorBase.sample_ray (models/tensorBase.py:304-323; ndc = 0) and sample_ray_ndc (:293-302; ndc = 1).
 * rays_o / rays_d [n,3] -> pts [n,N,3] = o + d z, valid [n,N] = 1 inside the box (the complement of mask_outbbox; no z gate: forward
 * applies that one itself, :459-462). ndc = 0: z_i = t_min + step (i [+ jitter[ray]]) goes to z_vals [n,N]; jitter = the per-ray
 * U[0,1) draws of train mode or NULL. ndc = 1: `jitter` is the caller's table of the N depths shared by all rays (linspace(near, far, N)
 * [+ its jitter row]); z_vals is not written (the reference returns the [1,N] table itself). */
int t2n_sample_ray(const t2n_field* f, const float* rays_o, const float* rays_d, int64_t n, int n_samples, const float* jitter, int ndc,
                   float* pts /*[n,N,3]*/, float* z_vals /*[n,N] or NULL*/, uint8_t* valid /*[n,N]*/, t2n_stream stream);

/* ---- a-16: TensorBase.filtering_rays(bbox_only=True) slab test (models/tensorBase.py:385-391) */
int t2n_filter_rays_bbox(const t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, uint8_t* mask,
                         t2n_stream stream);

/* ---- stage entry points (mirrors of TensorVMSplit methods, used by the Python mirror and the stage parity tests)
 * a-9 + a-10: compute_densityfeature (models/tensoRF.py:205-220) [+ feature2density (tensorBase.py:406-410)].
 *   xyz_norm [n,3] normalised to [-1,1]; feat/sigma [n] (either may be NULL). */
int t2n_density_at(const t2n_field* f, const float* xyz_norm, int64_t n, float* feat, float* sigma, t2n_stream stream);
/* a-12 + a-13: compute_appfeature (models/tensoRF.py:223-239) and renderModule (models/tensorBase.py:29-33,88-109).
 *   viewdirs [n,3] (SH head only, else NULL); app_feat [n,app_dim] and rgb [n,3] (either may be NULL).
 *   workspace: t2n_shade_workspace_bytes(n) bytes of device memory (holds the launch's f16-range flag; T2N_ERR_INVALID when
 *   missing on a field whose head runs as split-f16 products). */
size_t t2n_shade_workspace_bytes(int64_t n);
int t2n_shade_at(const t2n_field* f, const float* xyz_norm, const float* viewdirs, int64_t n, float* app_feat, float* rgb,
                 void* workspace, size_t workspace_bytes, t2n_stream stream);
/* a-11: raw2alpha (models/tensorBase.py:19-26): sigma,dist [R,N] -> alpha, weights [R,N], bg [R] (any output NULL). */
int t2n_raw2alpha(const float* sigma, const float* dist, int64_t n_rays, int n_samples, float* alpha, float* weights,
                  float* bg, t2n_stream stream);

/* ---- a-3/a-4: the render call. Replaces the chunk loop of OctreeRender_trilinear_fast (renderer.py:28-42) and
 * TensorBase.forward (models/tensorBase.py:436-507) for ndc_ray=False, alphaMask=None.
 *   rays      [n_rays, ray_stride] fp32 rows, first 6 = (o, d); the LAST channel feeds depth's background term (:505)
 *   jitter    [n_rays] U[0,1) draws (required iff T2N_FLAG_TRAIN; the reference draws them on the CPU generator, :313-316)
 *   rgb [n_rays,3], depth [n_rays]   required
 *   weights, z_vals [n_rays, n_samples]   optional (NULL = not materialised; the reference always returns them)
 *   stats     optional device uint64[T2N_STAT_COUNT], zeroed by the call
 *   workspace caller scratch; the call splits n_rays into sub-launches that fit (see t2n_render_workspace_bytes).
 *             t2n_render_workspace_bytes sizes the appearance lists for the worst case (every sample of every ray an appearance
 *             sample: 16 GiB for an 800x800 x 518 frame that uses ~0.2 GB of it). An image-ordered eval frame (T2N_FLAG_COHERENT
 *             with t2n_field_set_frame_width) of >= 65536 rays given LESS than that runs as ONE launch with lists budgeted to
 *             what the workspace holds. The call never waits for the device: rays whose entries find no room are finished on the
 *             device (stats[T2N_STAT_LIST_RETRY] = 1; their colours come from the exact-fp32 head, within ~1e-6 of the split-f16
 *             one; depth and every other ray are unaffected); the sub-list counters travel to pinned host memory behind an event
 *             and a LATER call (or t2n_render_workspace_bytes_hint / t2n_field_list_retries) turns them into the next budget.
 *             t2n_render_workspace_bytes_hint returns a workspace size for that mode: lists for 32 entries per ray while nothing is
 *             known, then twice what the field's last completed budgeted launch needed + 16 (~4.3 GB for the frame above).
 */
size_t t2n_render_workspace_bytes(int64_t rays_per_launch, int n_samples);
/* The general view-dependent heads (T2N_SHADE_MLP_FEA / _MLP_PE / _MLP; models/tensorBase.py:62-86,111-159) evaluate their unfused MLP
 * through activation scratch behind the sizes above: add this many bytes (0 for the other heads) and one pass covers 32 appearance rows
 * per ray; rows beyond that are taken in further passes over the same scratch. The call reads no count on the host and allocates
 * nothing: tile prefix and row count are derived on the device, every kernel clips to them, and the host issues the passes the worst
 * case needs (a pass past the count costs its empty launches). */
size_t t2n_render_head_scratch_bytes(const t2n_field* f, int64_t n_rays);
/* rows per ray that sizing reserves (default 32; the driver's scenes need ~7): fewer rows = less memory per ray, more passes */
int t2n_field_set_head_scratch_rows(t2n_field* f, int rows_per_ray);
size_t t2n_render_workspace_bytes_hint(const t2n_field* f, int64_t n_rays, int n_samples);
size_t t2n_render_workspace_bytes_budget(int64_t n_rays, int n_samples, int entries_per_ray);   /* one launch, lists for this many entries per ray */
uint64_t t2n_field_list_retries(const t2n_field* f);   /* budgeted calls that overflowed their lists since the field was created (waits for counters still in flight) */
int t2n_render_forward(t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags,
                       const float* jitter, float* rgb, float* depth, float* weights, float* z_vals, uint64_t* stats,
                       void* workspace, size_t workspace_bytes, t2n_stream stream);

/* ---- general-shape path (csrc/t2n_generic.hip): field / head shapes beyond the tuned kernels' — more than 16 density or 48 appearance
 * components per plane (different counts per plane allowed), app_dim up to 64, any fea_pe / view_pe up to 16, featureC up to 256 — the
 * shapes the reference accepts through n_lamb_sigma / n_lamb_sh / data_dim_color / fea_pe / featureC (models/tensoRF.py:144-160,
 * models/tensorBase.py:62-159, e_opt.py:83-107). No field handle: the parameters are read in place in the reference's layouts and the
 * gradients are ACCUMULATED (atomics) into tensors of the same layouts. One thread per ray / per appearance sample, no MFMA: a slow
 * path. Heads: MLP_Fea_noview, MLP_Fea, MLP, SH, RGB. No NDC sampling, no AlphaGridMask. `weights` / `z_vals` may be NULL in the forward
 * (kept in the workspace then); the backward takes the forward's workspace (and the same weights / z_vals pointers) unchanged. */
typedef struct t2n_generic_desc {
    float aabb_min[3], aabb_max[3], inv_aabb_size[3];
    int32_t grid[3];
    int32_t density_n_comp[3], app_n_comp[3];
    int32_t app_dim, shading, fea_pe, view_pe, feature_c, act;
    float density_shift, distance_scale, weight_thres, step_size, near, far, z_gate;
} t2n_generic_desc;
size_t t2n_generic_workspace_bytes(int64_t n_rays, int n_samples);
/* the same + room for channel-last staging copies of the factor tensors and the appearance-sample list: with this much workspace the
 * FORWARD runs as cooperative kernels (thread per sample for the density, a workgroup per 64 appearance samples for features + head)
 * instead of the plain form — 1 500 x per sample apart on a wide 128^3 field. Same outputs and context; the backward is unchanged. */
size_t t2n_generic_workspace_bytes_desc(const t2n_generic_desc* desc, int64_t n_rays, int n_samples);
int t2n_generic_forward(const t2n_generic_desc* desc, const t2n_field_params* params, const float* rays, int64_t n_rays, int ray_stride,
                        int n_samples, uint32_t flags, const float* jitter, float* rgb, float* depth, float* weights, float* z_vals,
                        uint64_t* stats, void* workspace, size_t workspace_bytes, t2n_stream stream);
int t2n_generic_backward(const t2n_generic_desc* desc, const t2n_field_params* params, const float* rays, int64_t n_rays, int ray_stride,
                         int n_samples, uint32_t flags, const float* jitter, const float* weights, const float* z_vals,
                         const float* d_rgb, const float* d_depth, const float* d_weights, const t2n_field_grads* grads,
                         void* workspace, size_t workspace_bytes, t2n_stream stream);

/* ndc_rays_blender (blender = 1) / ndc_rays (0) (dataLoader/ray_utils.py:88-124): [n,3] origins + directions -> NDC. */
int t2n_ndc_rays(int H, int W, float focal, float near, int blender, const float* rays_o, const float* rays_d, int64_t n,
                 float* o_out, float* d_out, t2n_stream stream);

/* eval_sh_bases (models/sh.py:87-133): (deg+1)^2 real SH basis values per unit direction, deg 0..4. dirs [n,3] -> out [n,(deg+1)^2]. */
int t2n_eval_sh_bases(int deg, const float* dirs, int64_t n, float* out, t2n_stream stream);

/* dda + ray_marcher (dataLoader/ray_utils.py:174-228): the AABB-clipped linspace sampler (no call sites in this driver; the
 * live sampler is inside t2n_render_forward). bbox_host = {min xyz, max xyz} or NULL (then near/far = ray columns 6, 7);
 * steps [n_samples] = torch.linspace(0, 1, n_samples) on the device; perturb [n, n_samples] = perturb * U[0,1) draws or NULL.
 * Outputs (any may be NULL): xyz [n, n_samples, 3], z_vals [n, n_samples], near_far [n, 2] (dda's t_min, t_max). */
int t2n_ray_marcher(const float* rays, int64_t n, int ray_stride, int n_samples, int lindisp, const float* bbox_host,
                    const float* steps, const float* perturb, float* xyz, float* z_vals, float* near_far, t2n_stream stream);

/* ---- f-2: frame post-processing of `evaluation` / `evaluation_path` (renderer.py:91-113,168-176; utils.py:241-257) on the
 * device: rgb8 [n,3] = uint8(255 * clamp(rgb,0,1)) (truncation); depth8 [n,3] = OpenCV COLORMAP_JET (B,G,R) of
 * uint8(255 * max((nan_to_num(d) - mi) / (ma - mi + 1e-8), 0)) with d = max((depth - depth_sub) + depth_add, 0) when
 * shift_clamp (the `evaluation` form: push_depth, 0.8) else d = depth (`evaluation_path`); when gt_rgb is given, *sq_err_sum += sum((clamp(rgb) - gt)^2) (double; PSNR = -10 log10(sum / (3 n))). Any output may be NULL. */
int t2n_frame_postprocess(const float* rgb /*[n,3]*/, const float* depth /*[n]*/, int64_t n, float depth_sub, float depth_add,
                          int shift_clamp, float mi, float ma, uint8_t* rgb8, uint8_t* depth8, const float* gt_rgb, double* sq_err_sum,
                          t2n_stream stream);

/* ---- f-3: the image-space steps either side of the renderer in render_warping_inapinting (text2nerf_main.py:102-141).
 * t2n_sparse_bilateral_filtering = dataLoader/bilateral_filtering.py:5-35 with mask=None, HR=False: per pass the depth
 * discontinuity map (:64-136; 4-neighbour disparity jumps > depth_threshold, or original depth == 0) and the median of the
 * window's non-discontinuity pixels (:138-196) on depth and the three colour channels. Outputs what the driver keeps
 * (:119-120): photo_out [H,W,3] after ALL num_iter passes and depth_out [num_iter,H,W] = the depth BEFORE each pass, i.e.
 * save_depths; its last entry has seen num_iter - 1 passes (reference quirk: the image list aliases one in-place array,
 * the depth list does not). filter_sizes_host: num_iter odd windows <= 7. Bit-exact. */
size_t t2n_image_filter_workspace_bytes(int H, int W);
int t2n_sparse_bilateral_filtering(const float* depth /*[H,W]*/, const float* image /*[H,W,3]*/, int H, int W,
                                   const int* filter_sizes_host, int num_iter, float depth_threshold, float* photo_out,
                                   float* depth_out, void* workspace, size_t workspace_bytes, t2n_stream stream);
/* DIBR: one source view of bilinear_splat_warping_multiview (utils.py:83-119) = Warper.forward_warp (scripts/Warper.py:21-186:
 * per-pixel reprojection in fp64, inverse-bilinear splat with depth-softmax weights into an (H+2, W+2) canvas with fp64
 * atomics) merged into the running target (filled [H,W] u8, image_u8 [H,W,3], depth_out [H,W] f64; caller-zeroed before
 * the first view; earlier views win). Ki9 = inv(K1), T12 = first three rows of T2 inv(T1), K9 = K2 (row-major doubles,
 * host). t2n_warp_finish: white where nothing landed, image / 255 -> fp32, int64 mask. */
size_t t2n_warp_workspace_bytes(int H, int W);
int t2n_warp_view(const float* rgb /*[H,W,3] in [0,1]*/, const float* depth /*[H,W]*/, const uint8_t* mask1 /*[H,W] or NULL*/, int H,
                  int W, const double* Ki9_host, const double* T12_host, const double* K9_host, uint8_t* filled, uint8_t* image_u8,
                  double* depth_out, void* workspace, size_t workspace_bytes, t2n_stream stream);
int t2n_warp_finish(const uint8_t* filled, const uint8_t* image_u8, int H, int W, float* image_out /*[H,W,3]*/,
                    int64_t* mask_out /*[H,W] or NULL*/, t2n_stream stream);

/* dibr_filter_mask2 (utils.py:393-409): raster-order in-place hole filling of the merged warp — unknown pixels whose 5x5
 * neighbourhood is known to more than `threshold` (0.65) take the mean of their known 3x3 neighbours and count as known for
 * the rest of the scan. Runs as skewed wavefronts (t = col + 3 row) in one workgroup. image [H,W,3] fp32, known [H,W]
 * int32 (0 / non-zero), depth [H,W] fp64 or NULL; all updated in place. */
int t2n_dibr_filter_mask2(float* image, int32_t* known, double* depth, int H, int W, float threshold, t2n_stream stream);
/* dibr_filter_mask (utils.py:345-392; the reference's driver does not call it): the scan above with threshold 0.6 and no depth, then a
 * 3x3 fill scan (unknown pixels whose 3x3 neighbourhood is known to more than 0.5), the four border lines (copy the inner neighbour
 * where it is known) and a 3x3 erase scan (known pixels whose neighbourhood is known to less than 0.45 become 255 / unknown), all in
 * place and in raster order (fronts t = col + 2 row). image [H,W,3] fp32, known [H,W] int32. */
int t2n_dibr_filter_mask(float* image, int32_t* known, int H, int W, t2n_stream stream);

/* ---- a-15: backward of the render call w.r.t. all field parameters (what autograd derives in the reference,
 * text2nerf_main.py:589; coordinates are detached there, models/tensoRF.py:208-210,226-228, so no ray gradients exist).
 * Protocol: (1) forward with T2N_FLAG_KEEP_CTX as ONE launch (workspace >= t2n_render_workspace_bytes_ctx), weights and
 * z_vals materialised; (2) t2n_render_ctx_rows returns the padded activation row count: the forward posted the
 * appearance-sample counts to pinned host memory behind its march kernel, so this waits for THAT copy's event (the stream keeps
 * running; a forward whose slot was evicted falls back to a stream-draining read); (3) t2n_render_backward with a second workspace of
 * t2n_backward_workspace_bytes(field, rows, n_rays, n_samples). Differentiable heads: MLP_Fea_noview (fused chain), MLP_Fea and
 * MLP (general path), SH, RGB — tests/test_hip_parity.py::test_g8_*, tests/test_heads.py, tests/test_shapes.py.
 * d_weights may be NULL. Gradients are ACCUMULATED into `g` (reference layouts; NULL members are skipped: the channel-last
 * plane / line gradients then stay in the field for t2n_field_tv_adam_step).
 * Optional: a forward workspace LARGER than t2n_render_workspace_bytes_ctx by 256 + rows * 1728 bytes lets the forward keep
 * the MLP activations of up to `rows` appearance samples (a guess, e.g. 1.25 x the previous call's row count); when the
 * actual count fits — and the same byte count is passed to t2n_render_backward — the backward skips its re-run of the
 * appearance forward. Too small a guess only costs that re-run.
 * Streams: all work is ordered on `stream` as seen by the caller. Internally the density scatter of a call runs on a library-owned
 * side stream that forks from `stream` behind the per-ray backward kernel and is joined into `stream` before the call's gradient
 * writes end (so both workspaces must stay alive until `stream` has passed the call, as for any asynchronous call). */
size_t t2n_render_workspace_bytes_ctx(int64_t n_rays, int n_samples);
int t2n_render_ctx_rows(const void* fwd_workspace, int64_t n_rays, int n_samples, t2n_stream stream, int64_t* rows);
/* What the T2N_FLAG_DEVICE_ROWS backwards of this field recorded, read from pinned host memory (never waits, never touches a
 * stream): out[0..7] = appearance rows the last eight such calls NEEDED (before clipping to their capacity; slot = sequence & 7),
 * out[8] = how many calls overflowed their capacity so far, out[9] = records written. A caller sizes its next capacity from it. */
int t2n_field_device_rows_record(const t2n_field* f, uint32_t out[10]);
size_t t2n_backward_workspace_bytes(const t2n_field* f, int64_t rows, int64_t n_rays, int n_samples);
int t2n_render_backward(t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags,
                        const float* jitter, const float* d_rgb, const float* d_depth, const float* d_weights,
                        const t2n_field_grads* g, void* fwd_workspace, size_t fwd_workspace_bytes, void* bwd_workspace,
                        size_t bwd_workspace_bytes, t2n_stream stream);

/* The channel-last gradients of the 12 factor tensors live in ONE buffer of t2n_field_grad_buffer_bytes (order: density planes,
 * density lines, appearance planes, appearance lines; 256-B aligned slices). By default the library owns it and every
 * t2n_render_backward zeroes it first (it then holds the gradients of that call). A caller that hands in its own device buffer
 * (t2n_field_set_grad_buffer; NULL returns to the library-owned one) owns the zeroing: backward calls ACCUMULATE into it (a batch
 * split into chunks, gradient accumulation), a data-parallel all-reduce can run on it in place, and t2n_field_tv_adam_step reads
 * it. Replaces the autograd accumulation of text2nerf_main.py:588-590 for those tensors. */
size_t t2n_field_grad_buffer_bytes(const t2n_field* f);
int t2n_field_set_grad_buffer(t2n_field* f, void* buf, size_t bytes);
/* Layout of that buffer: [density planes 0..2 | density lines 0..2 | appearance planes 0..2 | appearance lines 0..2], every tensor
 * channel-last and 256-B aligned. The density part (the first t2n_field_grad_buffer_density_bytes bytes) is final when the backward's
 * density scatter is done — about half a millisecond of a C3 step before the appearance scatter: t2n_field_wait_density_grads makes
 * `waiter` wait for that point of the LAST t2n_render_backward call (no-op before the first one), so that a data-parallel caller
 * can start all-reducing the density bucket on `waiter` while the rest of the backward still runs on its own stream
 * (text2nerf_amd/parallel.py; the reference trains on one GPU: text2nerf_main.py:547-601). */
size_t t2n_field_grad_buffer_density_bytes(const t2n_field* f);
int t2n_field_wait_density_grads(const t2n_field* f, t2n_stream waiter);

/* The driver's loss of one training batch (text2nerf_main.py:559-575; TransMittanceLoss_mask, utils.py:67-80) in one pass over the
 * render outputs: loss = mean((rgb - rgb_t)^2) + w_depth * mean((depth - depth_t)^2) + w_trans * mean_r(m_r^2) with
 * m_r = mean_n(weights[r,n] * [z_vals[r,n] - depth_t[r] + delta < 0]); NaN depths count as 0 and get no gradient (:559-560).
 * Writes the upstream gradients t2n_render_backward takes (d_rgb [R,3], d_depth [R], d_weights [R,N]) and
 * losses[4] = {mse, depth loss, transmittance loss, total} (device floats, deterministic sums). The driver's values:
 * w_depth 0.005, w_trans 1e3, delta 0.1. workspace: t2n_train_loss_workspace_bytes(n_rays). */
size_t t2n_train_loss_workspace_bytes(int64_t n_rays);
int t2n_train_loss(const float* rgb, const float* depth, const float* weights, const float* z_vals, const float* rgb_t,
                   const float* depth_t, int64_t n_rays, int n_samples, float w_depth, float w_trans, float delta, float* d_rgb,
                   float* d_depth, float* d_weights, float* losses, void* workspace, size_t workspace_bytes, t2n_stream stream);

/* ---- SURVEY.md 8(f-1), optional: the dense tail of the training step as streaming kernels.
 * t2n_tv_grad_add: grad += d/dparam [ weight * TVLoss(param) ] for one reference-layout plane [1,C,H,W]
 *   (utils.py:488-504; the driver's term is TV_loss_*(tvreg) * TV_weight with TV_loss_* = sum_planes 1e-2 * TVLoss,
 *   models/tensoRF.py:193-203 — pass weight = 1e-2 * TV_weight).
 * t2n_adam_step: torch.optim.Adam's update (betas, eps; no weight decay, no amsgrad) on one tensor, `step` = 1-based
 *   step count of that tensor (text2nerf_main.py:453-454,590). */
int t2n_tv_grad_add(const float* param, float* grad, int C, int H, int W, float weight, t2n_stream stream);
/* the same gradient WRITTEN (not added) to `grad`, times the device scalar *upstream when that is non-NULL: the backward of an
 * autograd node whose upstream gradient lives on the device (no host read, no zero fill, no extra scaling pass). */
int t2n_tv_grad_set(const float* param, float* grad, int C, int H, int W, float weight, const float* upstream, t2n_stream stream);
/* TVLoss's two sums of one plane (utils.py:497-498), spread over T2N_TV_SLOTS slot pairs the caller zeroes and adds up (device
 * doubles): sum_s sums[2 s] = sum of squared differences along H, sum_s sums[2 s + 1] = along W;
 * TVLoss(param) = weight * 2 * (h_tv / (C (H-1) W) + w_tv / (C H (W-1))) / batch. */
#define T2N_TV_SLOTS 32
int t2n_tv_value(const float* param, int C, int H, int W, double* sums, t2n_stream stream);
int t2n_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t step, t2n_stream stream);
/* the same update for `count` tensors in one launch (host arrays of device pointers / sizes / learning rates / step numbers) */
int t2n_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                        const int64_t* sizes, const float* lrs, float beta1, float beta2, float eps, const int64_t* steps,
                        t2n_stream stream);

/* The same step where the data already is: on the field's channel-last device copies, with the channel-last gradients the last
 * t2n_render_backward left in the field (call that with NULL plane / line pointers in t2n_field_grads: the layout pass back
 * to [1,C,H,W] and the caller's zero-filled gradient tensors are then skipped). For each of the 12 factor tensors (order:
 * density_plane[0..2], density_line[0..2], app_plane[0..2], app_line[0..2]): TV gradient (planes; weights as for
 * t2n_tv_grad_add, 0 = none), Adam update with lrs[i] / steps[i], new values written to the device copy AND to the
 * reference-layout tensor params-> (so the caller's nn.Parameters stay current and no t2n_field_upload of the factors is
 * needed — follow with t2n_field_upload_head when the head's tensors changed). exp_avg / exp_avg_sq: caller-owned
 * CHANNEL-LAST state buffers [pos][C] of the tensors' sizes. Per-element arithmetic identical to t2n_tv_grad_add +
 * t2n_adam_step. Not available with bf16 factor storage (T2N_ERR_UNSUPPORTED). */
int t2n_field_tv_adam_step(t2n_field* f, const t2n_field_params* params, float* const* exp_avg, float* const* exp_avg_sq,
                           const float* lrs, const int64_t* steps, float beta1, float beta2, float eps,
                           float tv_weight_density, float tv_weight_app, t2n_stream stream);
/* The TV terms the other way round: the factor gradient buffer (t2n_field_set_grad_buffer, or the library's own) is INITIALISED to the
 * TV gradient of the current device copies — zero for the lines and for a zero weight — before the backward accumulates into it
 * (replaces a zero fill + the TV pass of t2n_field_tv_adam_step, which is then called with both weights 0). One write-only pass that
 * depends on nothing but the parameters: it may run on another stream beside the forward; the backward must be ordered behind it.
 * Replaces (reference): the autograd of TV_loss_density / TV_loss_app (models/tensoRF.py:193-203, text2nerf_main.py:577-586). */
int t2n_field_tv_seed(t2n_field* f, float tv_weight_density, float tv_weight_app, t2n_stream stream);
/* re-pack basis_mat / renderModule only (the factor copies of an uploaded field are left as they are) */
int t2n_field_upload_head(t2n_field* f, const t2n_field_params* p, t2n_stream stream);

/* ---- coarse-to-fine / occupancy-mask maintenance between training stages.
 * t2n_compute_alpha: models/tensorBase.py:412-434 — alpha = 1 - exp(-sigma * length) at world-space points [n,3]; sigma = 0
 *   where the field's AlphaGridMask (if one is set) samples <= 0.
 * t2n_dense_alpha: models/tensorBase.py:328-344 (getDenseAlpha) — the same at the nodes aabb0*(1-s) + aabb1*s of a dense
 *   grid, s = (lin_x[i], lin_y[j], lin_z[k]) (device arrays holding torch.linspace(0,1,g)); alpha is [gx][gy][gz].
 * t2n_alpha_volume: models/tensorBase.py:349-363 (updateAlphaMask) — clamp, transpose to [gz][gy][gx], 3x3x3 max pool,
 *   binarise at `thres`; bbox_idx (device int[6]) receives the index box of the kept voxels (min x,y,z, max x,y,z;
 *   INT_MAX / -1 when nothing is kept).
 * t2n_upsample_bilinear: models/tensoRF.py:258-272 (up_sampling_VM) — F.interpolate(mode='bilinear', align_corners=True)
 *   of one [C,Hin,Win] factor to [C,Hout,Wout] (lines are [C,L,1]). */
int t2n_compute_alpha(const t2n_field* f, const float* xyz_world, int64_t n, float length, float* alpha, t2n_stream stream);
int t2n_dense_alpha(const t2n_field* f, const float* lin_x, const float* lin_y, const float* lin_z, int gx, int gy, int gz,
                    float length, float* alpha, t2n_stream stream);
int t2n_alpha_volume(const float* dense_alpha, int gx, int gy, int gz, float thres, float* volume, int* bbox_idx,
                     t2n_stream stream);
int t2n_upsample_bilinear(const float* src, int C, int Hin, int Win, float* dst, int Hout, int Wout, t2n_stream stream);
/* filtering_rays(bbox_only=False), models/tensorBase.py:393-395: mask[r] = any of the ray's first n_samples eval samples
 * reads the field's AlphaGridMask > 0 (needs a mask set with t2n_field_set_alpha_mask). */
int t2n_filter_rays_alpha(const t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint8_t* mask,
                          t2n_stream stream);

/* Global depth alignment of the inpainting loop, text2nerf_main.py:241-270: scale of the monocular estimate from the ratios of depth
 * differences between consecutive pixels of the caller's sample list ([n_samples][2] int32 (row, col): random.sample of the filled
 * pixels, :234-240), kept when finite, >= 0 and within 5 |thresh - 1| of 1, thresh = (max rendered - push_depth) / (max estimate -
 * push_depth), fallback thresh; then the mean shift against the rendered depth over the samples within 2 |max scaled - max rendered|,
 * fallback that difference. depth_shift [H,W] = scale * depth_est - shift; scale_shift: device double[4] = {scale, shift, pairs kept,
 * samples kept}. Double precision, deterministic. */
int t2n_depth_align_global(const float* depth_rendered, const float* depth_est, int H, int W, const int32_t* pixel_sample_yx,
                           int n_samples, double push_depth, float* depth_shift, double* scale_shift, t2n_stream stream);

/* ---- One optimisation step of the reference's loop (text2nerf_main.py:547-601) as ONE submission: TV gradient -> train-mode render
 * (KEEP_CTX) -> t2n_train_loss -> t2n_render_backward (device-side row plan, as T2N_FLAG_DEVICE_ROWS) -> Adam on all 19 tensors ->
 * re-packed head operands, on the caller's stream and three library-owned side streams (forked from / joined into `stream` by events, so
 * the whole call can be captured into a hipGraph: t2n_train_graph_*). Nothing in it is read on the host and no kernel argument changes
 * from step to step: the batch comes through the caller's fixed device buffers, the per-step scalars (learning rates, TV weights) through
 * `hyper` in device memory, Adam's step count lives in the field (device memory; bias corrections are derived from it on the device).
 * Fused MLP_Fea_noview head (27 / 6 / 128), 16 + 48 components, fp32 factor storage, split-f16 head arithmetic, no NDC, no alpha mask:
 * anything else is T2N_ERR_UNSUPPORTED (use the separate calls).
 *   hyper          device float[T2N_TRAIN_HYPER_FLOATS]: [0..18] learning rates of the 19 tensors in t2n_field_params order, [19] / [20]
 *                  TV weights of the density / appearance planes (already x 1e-2: models/tensoRF.py:193-203), rest reserved
 *   params         the caller's reference-layout tensors: updated in place (like t2n_field_tv_adam_step / t2n_adam_step_multi)
 *   exp_avg(_sq)   Adam moments: [0..11] CHANNEL-LAST buffers of the factor tensors, [12..18] reference layout
 *   head_grads     device float[T2N_TRAIN_HEAD_GRAD_FLOATS]: the gradients of basis_mat and the six MLP tensors, contiguous, in
 *                  parameter order, followed by ONE vote word (a data-parallel caller all-reduces the whole buffer with the gradients;
 *                  the factor gradients are in the field's gradient buffer: t2n_field_set_grad_buffer)
 *   rows_capacity  appearance rows (multiple of 32) the workspace is sized for. A step that needs MORE applies NO update at all (every
 *                  optimiser kernel reads the verdict from device memory and returns; Adam's step count does not advance): the caller
 *                  learns it from t2n_field_train_record — without waiting — and submits the same batch again with a capacity >= the
 *                  recorded need.
 *   shard_world    > 1 selects the SHARDED OPTIMISER of a data-parallel group: of every plane's n / 64 whole 64-position
 *                  blocks, rank r owns blocks [r c, (r + 1) c), c = (n / 64) / world (the "body", world c blocks); the blocks behind the
 *                  body and the six lines are replicated (t2n_field_shard_layout gives the offsets). phases = 1 then seeds the gradient
 *                  buffer with world x the TV gradient on the blocks this rank owns, zero on the rest of the body and the plain TV
 *                  gradient on the replicated part: AVERAGING the ranks' buffers — a reduce-scatter of each body, one small all-reduce of
 *                  the replicated parts and head_grads — leaves every owner the full-batch gradient of its slice. phases = 2 steps the
 *                  owned and the replicated blocks only (1 / world of the TV + Adam traffic); the caller then all-gathers each body of
 *                  the channel-last parameter copies (t2n_field_factor_buffer) and a phases = 4 call writes the gathered blocks into its
 *                  reference-layout tensors. Moments of blocks a rank does not own are never touched.
 *   phases         1: render + loss + backward only (gradients left in the field's buffer and head_grads), 2: optimiser only (after a
 *                  data-parallel all-reduce of both; a non-zero vote word withholds the update on every rank), 3: both.
 *   losses         device float[4] = {mse, depth loss, transmittance loss, total} of the batch
 * host_batch: the call itself copies host_batch_bytes from pinned host memory to batch_buffer (asynchronously, ahead of everything that
 * reads the batch, with an engine copy; T2N_COPY_KERNEL=1: a 16-byte-aligned pinned batch is read by the step's zero-fill launch itself). Pipelined form — an eager call with T2N_FLAG_PIPELINE, phases = 3, host_batch set and a workspace of TWICE
 * t2n_train_step_workspace_bytes: the copy and the step's early part (zero fills, the march — it reads the density factors only —, the
 * plan, the appearance binning) are enqueued on the library's side stream right behind the PREVIOUS step's density Adam instead of
 * behind `stream`, i.e. they run beside the previous step's appearance scatter / weight-gradient GEMMs / Adam. The two halves of the
 * workspace and two slots of per-step scalars alternate. The caller must not have touched the field (uploads, other optimiser steps)
 * between the two calls and must not reuse host_batch / batch_buffer before the call after next has been submitted; everything else
 * about the call (ordering of its results on `stream`, the record) is unchanged. head_grads is left ZEROED by the optimiser phase (and
 * must be zero when a phases = 1 | 3 call starts). The whole call uses three side streams (early part + TV seed + density scatter; layer
 * 2 + weight-gradient GEMMs + head step; none for the rest): with the caller's, one per hardware queue.
 * Replaces (reference): text2nerf_main.py:553-590 (renderer call, losses, TV terms, zero_grad / backward / step). */
#define T2N_TRAIN_HYPER_FLOATS 32
#define T2N_TRAIN_HEAD_GRAD_FLOATS (27 * 144 + 128 * 351 + 128 + 128 * 128 + 128 + 3 * 128 + 3 + 1)
typedef struct t2n_train_step_args {
    const float* rays; int64_t n_rays; int32_t ray_stride; int32_t n_samples;
    uint32_t flags;            /* T2N_FLAG_ADD_BG, T2N_FLAG_PIPELINE or 0 (T2N_FLAG_TRAIN is implied) */
    uint32_t phases;           /* 1 | 2, or 4 alone (sharded optimiser: see shard_world) */
    const float* jitter; const float* rgb_target; const float* depth_target;
    float w_depth, w_trans, delta;
    float beta1, beta2, eps;
    const float* hyper;
    t2n_field_params params;
    float* exp_avg[19]; float* exp_avg_sq[19];
    float* head_grads;
    int64_t rows_capacity;
    void* workspace; size_t workspace_bytes;
    float* losses;
    const void* host_batch;    /* optional: the batch in PINNED host memory (NULL: the device buffers are already filled), see below */
    size_t host_batch_bytes;
    void* batch_buffer;        /* device destination of host_batch: the start of the ONE allocation rays / jitter / targets / hyper point into */
    int32_t shard_world, shard_rank;   /* sharded optimiser (data-parallel): ranks and this rank; 0 / 1 = every rank steps every factor */
} t2n_train_step_args;
size_t t2n_train_step_workspace_bytes(const t2n_field* f, int64_t n_rays, int n_samples, int64_t rows_capacity);
int t2n_train_step(t2n_field* f, const t2n_train_step_args* a, t2n_stream stream);
/* The sharded optimiser's partition for `world` ranks: out[3 t + 0] = offset of factor tensor t in the field's gradient buffer (floats),
 * out[3 t + 1] = floats of ONE rank's slice of its body (0 for the lines), out[3 t + 2] = its floats in all; t = 0..11 in the order density
 * planes, density lines, appearance planes, appearance lines (channel-last: position-major, C = 16 / 48 floats per position). */
int t2n_field_shard_layout(const t2n_field* f, int world, int64_t out[36]);
/* Device pointer of the channel-last fp32 master copy of factor tensor t (the array the kernels read and the fused step's Adam updates). */
int t2n_field_factor_buffer(const t2n_field* f, int t, void** ptr);
/* Adam's step count of the field's fused steps (device memory): set when an optimiser state is loaded / reset. Asynchronous on `stream`. */
int t2n_field_train_set_step(t2n_field* f, uint32_t step, t2n_stream stream);
/* What the fused steps recorded, from pinned host memory (never waits, never touches a stream): out[0] = newest sequence number + 1 in
 * the ring (steps of the field are numbered from 0), out[1] = Adam steps applied, out[2] = steps withheld (both informational: written
 * without ordering); out[4 + 2 k], out[5 + 2 k], k = 0..15: the record of the step with (sequence number & 15) == k — the appearance rows
 * it NEEDED and its tag = (sequence number + 1) << 1 | withheld. A record is ONE 8-byte store: a slot whose tag carries another sequence
 * number has not been written for this step yet. */
int t2n_field_train_record(const t2n_field* f, uint32_t out[36]);
/* The same call captured ONCE into a hipGraph (stream capture of t2n_train_step on `stream`, the side streams included) and replayed:
 * every pointer and size in `a` is frozen — the caller refills the buffers behind them (batch, hyper) before each launch and
 * re-captures when a size (n_rays, n_samples, rows_capacity, workspace) or a pointer changes. t2n_train_step must have run eagerly once
 * with the same field (lazy stream / event / buffer creation cannot be captured). */
typedef struct t2n_train_graph t2n_train_graph;
int t2n_train_graph_capture(t2n_field* f, const t2n_train_step_args* a, t2n_stream stream, t2n_train_graph** out);
int t2n_train_graph_launch(t2n_train_graph* g, t2n_stream stream);
int t2n_train_graph_nodes(const t2n_train_graph* g);   /* kernel nodes of the captured step */
int t2n_train_graph_destroy(t2n_train_graph* g);

/* ---- measurement hooks (bench.py): when enabled, each kernel launch of the render call is bracketed by HIP events on
 * the launch stream. t2n_timing_read synchronises those events and returns accumulated milliseconds and launch counts
 * per kernel since the last reset (up to 1024 launches per kernel are timed between two reads; launches beyond that are counted and
 * priced at the timed average). Kernel ids: */
enum { T2N_K_MARCH = 0, T2N_K_SHADE = 1, T2N_K_COMPOSITE = 2, T2N_K_UPLOAD = 3, T2N_K_BWD_MARCH = 4, T2N_K_BWD_MLP = 5,
       T2N_K_BWD_SCATTER = 6, T2N_K_DENSITY = 7, T2N_K_APPFEAT = 8 /* appearance gather + basis_mat */,
       /* t2n_train_step only (its groups run side by side on four streams: each figure is the group's own start-to-end time under that
        * concurrency): weight-gradient GEMMs + their reduce; density binning + scatter; TV seed; Adam on the factors; head Adam + re-packs */
       T2N_K_BWD_WGRAD = 9, T2N_K_BWD_DENSITY = 10, T2N_K_TV_SEED = 11, T2N_K_ADAM = 12, T2N_K_HEAD_STEP = 13, T2N_K_COUNT = 14 };
int t2n_timing_enable(t2n_field* f, int on);   /* on: 0 off, 1 every kernel, else a mask: bit (k + 1) brackets kernel id k only */
int t2n_timing_read(t2n_field* f, double* ms /*[T2N_K_COUNT]*/, int64_t* launches /*[T2N_K_COUNT]*/, int reset);

#ifdef __cplusplus
}
#endif
#endif /* T2N_H_ */
