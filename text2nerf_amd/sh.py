"""models/sh.py:87-133 with the reference's function name: real spherical-harmonics basis values at unit directions (HIP kernel)."""
from __future__ import annotations

import torch

from . import _lib


def eval_sh_bases(deg, dirs):
    """``dirs (..., 3)`` unit directions -> ``(..., (deg + 1) ** 2)`` basis values, 0 <= deg <= 4."""
    assert deg <= 4 and deg >= 0
    lib = _lib.load()
    if dirs.device.type != "cuda":
        raise _lib.T2NError("eval_sh_bases runs on the MI355X only (no CPU fallback)")
    d = dirs.reshape(-1, 3).contiguous().float()
    out = torch.empty(d.shape[0], (deg + 1) ** 2, device=d.device, dtype=torch.float32)
    with torch.cuda.device(d.device):
        _lib.check(lib.t2n_eval_sh_bases(int(deg), _lib.ptr(d), d.shape[0], _lib.ptr(out), _lib.current_stream_ptr(d.device)),
                   "t2n_eval_sh_bases")
    return out.reshape(tuple(dirs.shape[:-1]) + ((deg + 1) ** 2,))
