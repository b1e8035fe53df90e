"""Global depth alignment of the inpainting loop (text2nerf_main.py:233-270): the oracle restatement (oracle/oracle_warp.py) against
goldens produced by EXECUTING those reference lines (tests/golden/make_golden_align.py), and the HIP path (t2n_depth_align_global)
against both. Tolerances: scale / shift 1e-9 relative for the oracle (same numpy arithmetic), 2e-6 for the HIP path (it takes the
estimate as float32; the reference holds it in float64), depth maps accordingly."""
import os

import numpy as np
import pytest

from tests.conftest import GOLDEN
from text2nerf_amd import synth


@pytest.fixture(scope="module")
def gold():
    return dict(np.load(os.path.join(GOLDEN, "align.npz")))


def case_inputs(gold, c):
    seed, H = (int(v) for v in gold[f"c{c}_seed_H"])
    dr, de, mm = synth.align_inputs(seed, H)
    if gold[f"c{c}_depth_est"].size:
        de = gold[f"c{c}_depth_est"]
    return dr, de, mm, gold[f"c{c}_pixel_sample"]


@pytest.mark.parametrize("c", [0, 1, 2])
def test_oracle_matches_reference_excerpt(gold, c):
    from oracle import oracle_warp as OW
    dr, de, mm, ps = case_inputs(gold, c)
    assert all(mm[y, x] > 0 for y, x in ps)            # the samples are filled pixels (:234-240)
    scale, shift, ds = OW.align_depth_global(dr, de, ps, 2.0)
    g_scale, g_shift = gold[f"c{c}_scale_shift"]
    assert abs(scale - g_scale) <= 1e-9 * abs(g_scale) and abs(shift - g_shift) <= 1e-9 * max(abs(g_shift), 1.0)
    np.testing.assert_allclose(ds, gold[f"c{c}_depth_shift"], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [0, 1, 2])
def test_hip_alignment_matches_golden_and_oracle(gold, c):
    from oracle import oracle_warp as OW
    from text2nerf_amd.warp import align_depth_global
    dr, de, mm, ps = case_inputs(gold, c)
    scale, shift, ds = align_depth_global(dr, de, ps, push_depth=2.0)
    g_scale, g_shift = gold[f"c{c}_scale_shift"]
    if c < 2:
        assert abs(scale - g_scale) <= 2e-6 * abs(g_scale) and abs(shift - g_shift) <= 2e-6
        np.testing.assert_allclose(ds.cpu().numpy(), gold[f"c{c}_depth_shift"], rtol=0, atol=5e-6)
    # the degenerate case (every ratio rejected -> fallbacks) is ill-conditioned in the estimate's precision: compare with the oracle
    # on the float32-rounded estimate the device path actually reads
    o_scale, o_shift, o_ds = OW.align_depth_global(dr, de.astype(np.float32).astype(np.float64), ps, 2.0)
    assert abs(scale - o_scale) <= 1e-9 * abs(o_scale) + 1e-12 and abs(shift - o_shift) <= 1e-9 * max(abs(o_shift), 1.0)
    np.testing.assert_allclose(ds.cpu().numpy(), o_ds, rtol=0, atol=2e-6 * max(1.0, float(np.abs(o_ds).max())))
