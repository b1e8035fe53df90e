#!/usr/bin/env python3
"""Golden vectors for eval_sh_bases (models/sh.py:87-133), degrees 0..4, produced by IMPORTING the reference on CPU. Directions
are rebuilt in the tests from numpy PCG64(42). Writes tests/golden/sh.npz.    python tests/golden/make_golden_sh.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402,F401  (seeds sys.path / module stubs)
from models.sh import eval_sh_bases  # noqa: E402


def dirs():
    g = np.random.Generator(np.random.PCG64(42))
    d = g.standard_normal((500, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d[:4] = [[0, 0, 1], [1, 0, 0], [0, -1, 0], [0.6, 0.0, 0.8]]
    return d.astype(np.float32)


if __name__ == "__main__":
    out = {f"sh{deg}": eval_sh_bases(deg, torch.from_numpy(dirs())).numpy() for deg in range(5)}
    np.savez_compressed(os.path.join(HERE, "sh.npz"), **out)
    print({k: v.shape for k, v in out.items()})
