"""Full-size check of the materialised outputs: the C2 frame's weights / z_vals from the tile marcher (fill kernel + window writes)
against the per-ray marcher's rows — z bit-exact, weights within the parity tolerance, and every row's weight sum against acc."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from text2nerf_amd import generate_rays  # noqa: E402
torch.set_num_threads(8)
dev = torch.device("cuda", 0)
field, params, aabb = bench.build_field(dev, scene="S1-soft", seed=0)
field.materialize_weights = True
rays = generate_rays(800, 800, [800.0, 800.0, 400, 400], torch.eye(4).numpy(), device=dev)
with torch.no_grad():
    field.frame_width = 800
    rgb_t, depth_t, z_t, w_t = field(rays, white_bg=True, is_train=False, N_samples=-1)
    st_t = field.stats()
    field.frame_width = 0
    rgb_r, depth_r, z_r, w_r = field(rays, white_bg=True, is_train=False, N_samples=-1)
    st_r = field.stats()
print("z bit-exact:", bool(torch.equal(z_t, z_r)), " max |w_tile - w_ray|:", float((w_t - w_r).abs().max()),
      " max |rgb|:", float((rgb_t - rgb_r).abs().max()), " max |depth|:", float((depth_t - depth_r).abs().max()))
print("evaluated equal:", st_t["evaluated"] == st_r["evaluated"], st_t["evaluated"], " appearance:", st_t["appearance"], st_r["appearance"])
nz_t, nz_r = int((w_t != 0).sum()), int((w_r != 0).sum())
print("non-zero weights:", nz_t, nz_r)
