#!/usr/bin/env python3
"""Camera poses of the reference's own trajectory generators, captured as a fixture (tests/golden/poses.npz) by IMPORTING
/root/reference/dataLoader/scene_util.py on CPU in the build container (SURVEY.md §8 d: C3 = the 9 `local_fixed` poses, C5 = the 48
training poses of the 360-degree circle, plus the 120-view evaluation spiral). The GPU box and the tests read only the .npz.

    python tests/golden/make_golden_poses.py

Calls restated from dataLoader/scene_gen.py:240-250,268-278 with the option defaults of e_opt.py:21-22 (angle 0.2, trans_range 0.2).
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
for name in ["cv2", "imageio", "imageio.v2", "configargparse", "torchvision", "torchvision.transforms", "kornia", "lpips", "plyfile",
             "skimage", "skimage.metrics", "skimage.measure", "statsmodels", "statsmodels.api", "scipy.spatial.transform"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = MagicMock()

import importlib.util  # noqa: E402

# the module file itself: dataLoader/__init__.py would pull in the dataset classes and their image stack
spec = importlib.util.spec_from_file_location("ref_scene_util", os.path.join(REF, "dataLoader", "scene_util.py"))
su = importlib.util.module_from_spec(spec)
spec.loader.exec_module(su)

pose_ref = np.eye(4)
out = {}
# C3: scene_gen.py:242 (training poses of the local_fixed trajectory) — 9 poses
out["local_fixed"] = np.asarray(su.get_local_fixed_poses2(pose_ref, angle=0.2, range_center=0.2, range_yaw=0.6, range_pitch=0.2), dtype=np.float32)
# the support poses the inpainting loop renders around a known view: text2nerf_main.py:356,381 (angle=0)
out["local_fixed_support_angle0"] = np.asarray(su.get_local_fixed_poses2(pose_ref, angle=0, range_center=0.2, range_yaw=0.6, range_pitch=0.2),
                                               dtype=np.float32)
# C5: scene_gen.py:250 with --pose_traj circle --num_training 96 (SURVEY.md §8 d) -> 48 training poses
out["circle_train_96"] = np.asarray(su.cam_traj_gen(96, traj_type="circle", random_sample=False, radius=0.2, pose_ref=pose_ref, for_training=True),
                                    dtype=np.float32)
# evaluation paths: scene_gen.py:269 (local trajectories) and :278 (global ones)
out["eval_spiral_120"] = np.asarray(su.get_circle_spiral_poses_from_pose(out["local_fixed"][0], N_views=120, n_r=1, angle_h_start=0.2 - 0.03,
                                                                         trans_start=0.2, use_rand=False), dtype=np.float32)
out["circle_eval_360"] = np.asarray(su.cam_traj_gen(360, traj_type="circle", random_sample=False, radius=0.2, pose_ref=out["circle_train_96"][0]),
                                    dtype=np.float32)
for k, v in out.items():
    print(k, v.shape, "det(R0) = %.6f" % np.linalg.det(v[0][:3, :3]))
np.savez_compressed(os.path.join(HERE, "poses.npz"), **out)
