"""Inputs of the f-3 goldens (shared by make_golden_warp.py and the tests): frame size, filter schedules, poses."""
import numpy as np

from text2nerf_amd import synth

H, W = 40, 56
FILTER_CASES = {"a": dict(seed=21, holes=0, filter_size=[7, 5, 5, 3, 3], num_iter=5),      # text2nerf_main.py:116-118
                "b": dict(seed=22, holes=6, filter_size=[5, 5, 3, 3], num_iter=4)}          # text2nerf_main.py:286-288


def warp_poses():
    """Source views 0..2 and a target view: small-baseline camera-to-world matrices like the driver's local poses."""
    return [synth.look_pose(0.0, 0.0, (0.0, 0.0, 0.0)), synth.look_pose(0.08, -0.03, (0.15, 0.02, 0.05)),
            synth.look_pose(-0.06, 0.04, (-0.12, -0.05, 0.0)), synth.look_pose(0.05, 0.02, (0.06, -0.04, 0.1))]


def pose44(p):
    m = np.eye(4, dtype=np.float32)
    m[:3, :4] = np.asarray(p, np.float32)[:3, :4]
    return m


