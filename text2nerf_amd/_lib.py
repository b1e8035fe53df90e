"""ctypes binding of libt2n_hip.so (include/t2n.h). No CPU fallback: a missing library or a failing call raises."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# T2N_LIB selects another build of the library (experiment / instrumented variants) without touching the shipped artefact
LIB_PATH = os.environ.get("T2N_LIB") or os.path.join(_HERE, "libt2n_hip.so")

T2N_STAT_COUNT = 8
T2N_K_COUNT = 14
KERNEL_NAMES = ("march", "shade", "composite", "upload", "bwd_march", "bwd_mlp", "bwd_scatter", "density", "app_features",
                "bwd_wgrad", "bwd_density", "tv_seed", "adam", "head_step")
STAT_EVALUATED, STAT_APPEARANCE, STAT_RAYS, STAT_OVERFLOW, STAT_F16_REDO, STAT_LIST_RETRY = 0, 1, 2, 3, 4, 5

FLAG_TRAIN, FLAG_ADD_BG, FLAG_KEEP_CTX, FLAG_COHERENT, FLAG_NDC = 1, 2, 4, 8, 16
FLAG_DEVICE_ROWS = 32
FLAG_PIPELINE = 64
SHADE_IDS = {"MLP_Fea_noview": 0, "SH": 1, "RGB": 2, "MLP_Fea": 3, "MLP_PE": 4, "MLP": 5}   # MLP_PE: rejected in tensorf.py (broken upstream)
ACT_IDS = {"softplus": 0, "relu": 1}


class T2NError(RuntimeError):
    pass


class FieldDesc(C.Structure):
    _fields_ = [("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("inv_aabb_size", C.c_float * 3),
                ("grid", C.c_int32 * 3), ("density_n_comp", C.c_int32), ("app_n_comp", C.c_int32),
                ("app_dim", C.c_int32), ("shading", C.c_int32), ("fea_pe", C.c_int32), ("feature_c", C.c_int32),
                ("act", C.c_int32), ("density_shift", C.c_float), ("distance_scale", C.c_float),
                ("weight_thres", C.c_float), ("step_size", C.c_float), ("near", C.c_float), ("far", C.c_float),
                ("z_gate", C.c_float), ("view_pe", C.c_int32), ("pos_pe", C.c_int32)]


class FieldParams(C.Structure):
    _fields_ = [("density_plane", C.c_void_p * 3), ("density_line", C.c_void_p * 3), ("app_plane", C.c_void_p * 3),
                ("app_line", C.c_void_p * 3), ("basis_weight", C.c_void_p),
                ("mlp_w0", C.c_void_p), ("mlp_b0", C.c_void_p), ("mlp_w1", C.c_void_p), ("mlp_b1", C.c_void_p),
                ("mlp_w2", C.c_void_p), ("mlp_b2", C.c_void_p)]


FieldGrads = FieldParams  # same shape: pointers in reference layouts


class GenericDesc(C.Structure):   # t2n_generic_desc: the general-shape path (csrc/t2n_generic.hip)
    _fields_ = [("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3), ("inv_aabb_size", C.c_float * 3), ("grid", C.c_int32 * 3),
                ("density_n_comp", C.c_int32 * 3), ("app_n_comp", C.c_int32 * 3), ("app_dim", C.c_int32), ("shading", C.c_int32),
                ("fea_pe", C.c_int32), ("view_pe", C.c_int32), ("feature_c", C.c_int32), ("act", C.c_int32),
                ("density_shift", C.c_float), ("distance_scale", C.c_float), ("weight_thres", C.c_float), ("step_size", C.c_float),
                ("near", C.c_float), ("far", C.c_float), ("z_gate", C.c_float)]

TRAIN_HYPER_FLOATS = 32
TRAIN_HEAD_GRAD_FLOATS = 27 * 144 + 128 * 351 + 128 + 128 * 128 + 128 + 3 * 128 + 3 + 1   # + the vote word
TRAIN_HEAD_SIZES = (27 * 144, 128 * 351, 128, 128 * 128, 128, 3 * 128, 3)


class TrainStepArgs(C.Structure):   # t2n_train_step_args
    _fields_ = [("rays", C.c_void_p), ("n_rays", C.c_int64), ("ray_stride", C.c_int32), ("n_samples", C.c_int32),
                ("flags", C.c_uint32), ("phases", C.c_uint32),
                ("jitter", C.c_void_p), ("rgb_target", C.c_void_p), ("depth_target", C.c_void_p),
                ("w_depth", C.c_float), ("w_trans", C.c_float), ("delta", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("hyper", C.c_void_p), ("params", FieldParams),
                ("exp_avg", C.c_void_p * 19), ("exp_avg_sq", C.c_void_p * 19),
                ("head_grads", C.c_void_p), ("rows_capacity", C.c_int64),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("losses", C.c_void_p), ("host_batch", C.c_void_p), ("host_batch_bytes", C.c_size_t), ("batch_buffer", C.c_void_p),
                ("shard_world", C.c_int32), ("shard_rank", C.c_int32)]


_lib = None
_lock = threading.Lock()

# name -> (restype, argtypes); also the list the symbol test checks against include/t2n.h
SIGNATURES = {
    "t2n_last_error": (C.c_char_p, []),
    "t2n_version": (C.c_int, []),
    "t2n_field_create": (C.c_int, [C.POINTER(FieldDesc), C.POINTER(C.c_void_p)]),
    "t2n_field_destroy": (C.c_int, [C.c_void_p]),
    "t2n_field_upload": (C.c_int, [C.c_void_p, C.POINTER(FieldParams), C.c_void_p]),
    "t2n_field_set_desc": (C.c_int, [C.c_void_p, C.POINTER(FieldDesc)]),
    "t2n_field_set_mlp_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "t2n_field_set_early_termination": (C.c_int, [C.c_void_p, C.c_float]),
    "t2n_sample_ray": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "t2n_field_set_frame_width": (C.c_int, [C.c_void_p, C.c_int]),
    "t2n_field_set_alpha_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                           C.POINTER(C.c_float), C.c_void_p]),
    "t2n_alpha_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "t2n_ray_directions": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "t2n_get_rays": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p]),
    "t2n_generate_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.POINTER(C.c_float), C.c_void_p, C.c_void_p]),
    "t2n_filter_rays_bbox": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    "t2n_density_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_shade_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "t2n_shade_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_size_t, C.c_void_p]),
    "t2n_raw2alpha": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "t2n_render_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "t2n_render_head_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int64]),
    "t2n_field_set_head_scratch_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "t2n_render_workspace_bytes_hint": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int]),
    "t2n_field_list_retries": (C.c_uint64, [C.c_void_p]),
    "t2n_render_workspace_bytes_budget": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "t2n_render_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.c_void_p]),
    "t2n_render_workspace_bytes_ctx": (C.c_size_t, [C.c_int64, C.c_int]),
    "t2n_depth_align_global": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_void_p]),
    "t2n_field_grad_buffer_bytes": (C.c_size_t, [C.c_void_p]),
    "t2n_field_set_grad_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "t2n_field_grad_buffer_density_bytes": (C.c_size_t, [C.c_void_p]),
    "t2n_field_wait_density_grads": (C.c_int, [C.c_void_p, C.c_void_p]),
    "t2n_train_loss_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "t2n_train_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                 C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "t2n_field_set_factor_storage": (C.c_int, [C.c_void_p, C.c_int]),
    "t2n_frame_postprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_image_filter_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "t2n_sparse_bilateral_filtering": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "t2n_warp_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "t2n_warp_view": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_double), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "t2n_warp_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_adam_step_multi": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                            C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "t2n_ray_marcher": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_dibr_filter_mask2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "t2n_dibr_filter_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "t2n_eval_sh_bases": (C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "t2n_ndc_rays": (C.c_int, [C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                     C.c_void_p, C.c_void_p]),
    "t2n_render_ctx_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.POINTER(C.c_int64)]),
    "t2n_field_device_rows_record": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "t2n_backward_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int64, C.c_int]),
    "t2n_render_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FieldGrads), C.c_void_p, C.c_size_t,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "t2n_tv_grad_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "t2n_tv_value": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "t2n_tv_grad_set": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p]),
    "t2n_field_tv_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "t2n_field_tv_seed": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "t2n_field_upload_head": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_int64, C.c_void_p]),
    "t2n_compute_alpha": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "t2n_dense_alpha": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_void_p, C.c_void_p]),
    "t2n_alpha_volume": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "t2n_upsample_bilinear": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "t2n_filter_rays_alpha": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "t2n_generic_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "t2n_generic_workspace_bytes_desc": (C.c_size_t, [C.POINTER(GenericDesc), C.c_int64, C.c_int]),
    "t2n_generic_forward": (C.c_int, [C.POINTER(GenericDesc), C.POINTER(FieldParams), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                      C.c_void_p]),
    "t2n_generic_backward": (C.c_int, [C.POINTER(GenericDesc), C.POINTER(FieldParams), C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_uint32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(FieldGrads),
                                       C.c_void_p, C.c_size_t, C.c_void_p]),
    "t2n_train_step_workspace_bytes": (C.c_size_t, [C.c_void_p, C.c_int64, C.c_int, C.c_int64]),
    "t2n_train_step": (C.c_int, [C.c_void_p, C.POINTER(TrainStepArgs), C.c_void_p]),
    "t2n_field_train_set_step": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    "t2n_field_train_record": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "t2n_field_shard_layout": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]),
    "t2n_field_factor_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "t2n_train_graph_capture": (C.c_int, [C.c_void_p, C.POINTER(TrainStepArgs), C.c_void_p, C.POINTER(C.c_void_p)]),
    "t2n_train_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "t2n_train_graph_nodes": (C.c_int, [C.c_void_p]),
    "t2n_train_graph_destroy": (C.c_int, [C.c_void_p]),
    "t2n_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "t2n_timing_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
}


def load(path: str = LIB_PATH):
    """dlopen the C-ABI library (after torch, so both share one HIP runtime) and bind every declared symbol."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(path):
            raise T2NError(f"{path} is missing: build it with `python -m text2nerf_amd.build` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64 first; SONAME-matched by ours)
        except Exception:  # pragma: no cover
            pass
        lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().t2n_last_error()
        raise T2NError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream_ptr(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
