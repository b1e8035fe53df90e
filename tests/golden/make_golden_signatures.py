#!/usr/bin/env python3
"""Dump the call signatures of every name of the reference's hot-path surface (SURVEY.md 8(b) + the module-level helpers beside it)
to tests/golden/signatures.json, by IMPORTING the reference on CPU in the build container (same stubs as make_golden.py).
tests/test_host_cpu.py::test_signatures_match_the_reference compares the mirror with it: parameter names, order, kinds and defaults.

    python tests/golden/make_golden_signatures.py
"""
import contextlib
import inspect
import io
import json
import os
import sys
import types
from unittest.mock import MagicMock

import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
for name in ["cv2", "imageio", "imageio.v2", "configargparse", "torchvision", "torchvision.transforms", "statsmodels",
             "statsmodels.api", "lpips", "plyfile", "skimage", "skimage.metrics", "skimage.measure", "scripts.Warper"]:
    sys.modules.setdefault(name, MagicMock())
kornia = types.ModuleType("kornia")
kornia.create_meshgrid = lambda *a, **k: None      # imported by name only; never called here
sys.modules["kornia"] = kornia
torch.set_num_threads(2)

with contextlib.redirect_stdout(io.StringIO()):
    import models.tensorBase as tb
    import models.tensoRF as tr
    import models.sh as sh
    import dataLoader.ray_utils as ru
    import renderer as rr

FUNCS = {
    "renderer": (rr, ["OctreeRender_trilinear_fast", "evaluation", "evaluation_path"]),
    "tensorBase": (tb, ["positional_encoding", "raw2alpha", "SHRender", "RGBRender"]),
    "sh": (sh, ["eval_sh_bases"]),
    "ray_utils": (ru, ["get_ray_directions", "get_ray_directions_blender", "get_rays", "ndc_rays_blender", "ndc_rays", "sample_pdf",
                       "depth2dist", "ndc2dist", "dda", "ray_marcher"]),
}
CLASSES = {
    "SimpleSampler": (rr.SimpleSampler, ["__init__", "nextids"]),
    "AlphaGridMask": (tb.AlphaGridMask, ["__init__", "sample_alpha", "normalize_coord"]),
    "MLPRender_Fea": (tb.MLPRender_Fea, ["__init__", "forward"]),
    "MLPRender_Fea_noview": (tb.MLPRender_Fea_noview, ["__init__", "forward"]),
    "MLPRender_PE": (tb.MLPRender_PE, ["__init__", "forward"]),
    "MLPRender": (tb.MLPRender, ["__init__", "forward"]),
    "TensorBase": (tb.TensorBase, ["__init__", "init_render_func", "update_stepSize", "normalize_coord", "get_kwargs", "save", "load",
                                   "sample_ray_ndc", "sample_ray", "getDenseAlpha", "updateAlphaMask", "filtering_rays",
                                   "feature2density", "compute_alpha", "forward"]),
    "TensorVMSplit": (tr.TensorVMSplit, ["__init__", "init_svd_volume", "init_one_svd", "get_optparam_groups", "vectorDiffs",
                                         "vector_comp_diffs", "density_L1", "TV_loss_density", "TV_loss_app", "compute_densityfeature",
                                         "compute_appfeature", "up_sampling_VM", "upsample_volume_grid", "shrink"]),
    "TensorVM": (tr.TensorVM, ["__init__", "init_svd_volume", "get_optparam_groups", "compute_features", "compute_densityfeature",
                               "compute_appfeature", "vectorDiffs", "vector_comp_diffs", "upsample_volume_grid"]),
    "TensorCP": (tr.TensorCP, ["__init__", "init_svd_volume", "init_one_svd", "get_optparam_groups", "compute_densityfeature",
                               "compute_appfeature", "upsample_volume_grid", "shrink", "density_L1", "TV_loss_density", "TV_loss_app"]),
}


def describe(fn):
    out = []
    for p in inspect.signature(fn).parameters.values():
        d = None if p.default is inspect.Parameter.empty else repr(p.default)
        out.append([p.name, p.kind.name, d])
    return out


def main():
    sig = {}
    for mod, (m, names) in FUNCS.items():
        for n in names:
            sig[f"{mod}.{n}"] = describe(getattr(m, n))
    for cname, (cls, names) in CLASSES.items():
        for n in names:
            sig[f"{cname}.{n}"] = describe(getattr(cls, n))
    sig["_bases"] = {c: [b.__name__ for b in cls.__mro__[1:-1]] for c, (cls, _) in CLASSES.items()}
    with open(os.path.join(HERE, "signatures.json"), "w") as fh:
        json.dump(sig, fh, indent=1, sort_keys=True)
    print(f"wrote {len(sig) - 1} signatures")


if __name__ == "__main__":
    main()
