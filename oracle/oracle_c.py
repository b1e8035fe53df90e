"""ctypes wrapper of oracle/liboracle_c.so (the plain-C restatement; test infrastructure only — see oracle_c.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle_c.so")


class Field(C.Structure):
    _fields_ = [("aabb0", C.c_float * 3), ("aabb1", C.c_float * 3), ("inv", C.c_float * 3), ("grid", C.c_int * 3),
                ("density_c", C.c_int), ("app_c", C.c_int), ("app_dim", C.c_int), ("shading", C.c_int), ("fea_pe", C.c_int),
                ("feature_c", C.c_int), ("act", C.c_int), ("density_shift", C.c_float), ("distance_scale", C.c_float),
                ("weight_thres", C.c_float), ("step", C.c_float), ("near_", C.c_float), ("far_", C.c_float),
                ("z_gate", C.c_float), ("density_plane", C.c_void_p * 3), ("density_line", C.c_void_p * 3),
                ("app_plane", C.c_void_p * 3), ("app_line", C.c_void_p * 3), ("basis", C.c_void_p), ("w0", C.c_void_p),
                ("b0", C.c_void_p), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p)]


def load():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(HERE, "oracle_c.c")):
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    lib = C.CDLL(LIB)
    lib.t2n_oracle_render.restype = C.c_int
    lib.t2n_oracle_render.argtypes = [C.POINTER(Field), C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_long)]
    lib.t2n_oracle_threads.restype = C.c_int
    return lib


class COracle:
    """params: {state_dict key: float32 ndarray} in the reference layouts; cfg: oracle_torch.FieldConfig."""

    def __init__(self, cfg, params):
        self.lib = load()
        self.keep = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in params.items()}
        f = Field()
        a = np.asarray(cfg.aabb, np.float32)
        inv = (np.float32(2.0) / (a[1] - a[0])).astype(np.float32)
        for k in range(3):
            f.aabb0[k], f.aabb1[k], f.inv[k], f.grid[k] = float(a[0, k]), float(a[1, k]), float(inv[k]), int(cfg.grid_size[k])
            f.density_plane[k] = self.keep[f"density_plane.{k}"].ctypes.data
            f.density_line[k] = self.keep[f"density_line.{k}"].ctypes.data
            f.app_plane[k] = self.keep[f"app_plane.{k}"].ctypes.data
            f.app_line[k] = self.keep[f"app_line.{k}"].ctypes.data
        f.density_c = self.keep["density_plane.0"].shape[1]
        f.app_c = self.keep["app_plane.0"].shape[1]
        f.app_dim = self.keep["basis_mat.weight"].shape[0]
        f.shading = {"MLP_Fea_noview": 0, "SH": 1, "RGB": 2}[cfg.shading_mode]
        f.fea_pe, f.feature_c = cfg.fea_pe, 128
        f.act = 0 if cfg.fea2dense_act == "softplus" else 1
        f.density_shift, f.distance_scale = cfg.density_shift, cfg.distance_scale
        f.weight_thres, f.step = cfg.ray_march_weight_thres, cfg.step_size
        f.near_, f.far_, f.z_gate = cfg.near_far[0], cfg.near_far[1], cfg.z_gate
        f.basis = self.keep["basis_mat.weight"].ctypes.data
        if f.shading == 0:
            f.w0, f.b0 = self.keep["renderModule.mlp.0.weight"].ctypes.data, self.keep["renderModule.mlp.0.bias"].ctypes.data
            f.w1, f.b1 = self.keep["renderModule.mlp.2.weight"].ctypes.data, self.keep["renderModule.mlp.2.bias"].ctypes.data
            f.w2, f.b2 = self.keep["renderModule.mlp.4.weight"].ctypes.data, self.keep["renderModule.mlp.4.bias"].ctypes.data
            f.feature_c = self.keep["renderModule.mlp.0.weight"].shape[0]
        self.f, self.cfg = f, cfg

    def threads(self):
        return int(self.lib.t2n_oracle_threads())

    def render(self, rays, n_samples=-1, is_train=False, white_bg=True, jitter=None, want_weights=True):
        rays = np.ascontiguousarray(rays, np.float32)
        R, stride = rays.shape
        N = n_samples if n_samples > 0 else self.cfg.n_samples
        rgb, depth = np.empty((R, 3), np.float32), np.empty(R, np.float32)
        w = np.empty((R, N), np.float32) if want_weights else None
        z = np.empty((R, N), np.float32) if want_weights else None
        j = None if jitter is None else np.ascontiguousarray(jitter, np.float32).reshape(-1)
        stats = (C.c_long * 2)(0, 0)
        p = lambda x: None if x is None else x.ctypes.data   # noqa: E731
        rc = self.lib.t2n_oracle_render(C.byref(self.f), p(rays), R, stride, N, 1 if is_train else 0, 1 if white_bg else 0,
                                        p(j), p(rgb), p(depth), p(w), p(z), stats)
        assert rc == 0
        self.last_stats = {"evaluated": int(stats[0]), "appearance": int(stats[1])}
        return rgb, depth, z, w
