#!/usr/bin/env python3
"""Golden vectors for the global depth alignment of `render_warping_inapinting` (text2nerf_main.py:233-270: scale from ratios of depth
differences between consecutive sampled pixel pairs, then the mean shift), the last numpy / Python-loop stage between two renders of the
inpainting loop. The stage is inline code of a 300-line function that cannot run here (Stable Diffusion, CLIP, LeReS),
so this script EXECUTES lines 233-270 of the reference source, read from /root/reference at run time, against synthetic inputs bound to
the names the excerpt uses (H, W, myMap_filt, depth_rendered, depth_est, push_depth; `random` seeded). Nothing of the excerpt is stored:
the fixture tests/golden/align.npz holds inputs that cannot be rebuilt from a seed (the sampled pixel list) and the outputs.

    python tests/golden/make_golden_align.py
"""
import os
import random
import textwrap

import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from text2nerf_amd.synth import align_inputs as inputs  # noqa: E402

REF = "/root/reference/text2nerf_main.py"
HERE = os.path.dirname(os.path.abspath(__file__))
LO, HI = 233, 270     # 1-based, inclusive: "## depth alignment stage 1" .. "depth_shift = depth_scaled - shift"


def main():
    src = open(REF).read().splitlines()
    excerpt = textwrap.dedent("\n".join(src[LO - 1:HI]))
    assert "pixel_filled = []" in excerpt and excerpt.rstrip().endswith("depth_shift = depth_scaled - shift"), "reference lines moved"
    out = {}
    for case, (seed, H) in enumerate([(0, 48), (1, 64), (2, 40)]):
        dr, de, mm = inputs(seed, H)
        if case == 2:
            de = de * 0 + 2.5 + np.linspace(0, 1e-9, H * H).reshape(H, H)     # degenerate estimate: every ratio is rejected -> the fallbacks
        env = {"np": np, "random": random, "H": H, "W": H, "myMap_filt": mm, "depth_rendered": dr, "depth_est": de, "push_depth": 2.0,
               "print": lambda *a, **k: None}
        random.seed(100 + case)
        exec(compile(excerpt, "<text2nerf_main.py:%d-%d>" % (LO, HI), "exec"), env)
        out[f"c{case}_seed_H"] = np.array([seed, H], np.int64)
        out[f"c{case}_depth_est"] = de if case == 2 else np.zeros(0)
        out[f"c{case}_pixel_sample"] = np.asarray(env["pixel_sample"], np.int32)
        out[f"c{case}_scale_shift"] = np.array([env["scale"], env["shift"]], np.float64)
        out[f"c{case}_depth_shift"] = np.asarray(env["depth_shift"], np.float64)
        print(case, "pixels", len(env["pixel_sample"]), "scale %.6f shift %.6f" % (env["scale"], env["shift"]), "kept", len(env["scales"]), len(env["shifts"]))
    np.savez_compressed(os.path.join(HERE, "align.npz"), **out)


if __name__ == "__main__":
    main()
