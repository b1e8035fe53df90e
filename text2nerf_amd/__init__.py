"""MI355X-native TensoRF VM-split ray-marching renderer (the Text2NeRF hot path).

Host-side mirror of the reference interface for this path; the arithmetic runs in libt2n_hip.so (include/t2n.h).
"""
from .renderer import (OctreeRender_trilinear_fast, SimpleSampler, BatchPrefetcher, render_views, postprocess_frame,  # noqa: F401
                       evaluation_frames, evaluation, evaluation_path)
from .tensorf import (AlphaGridMask, MLPRender, MLPRender_Fea, MLPRender_Fea_noview, MLPRender_PE, RGBRender, SHRender, TensorBase,  # noqa: F401
                      TensorCP, TensorVM, TensorVMSplit, positional_encoding, raw2alpha, release_workspaces, to_device_async,
                      workspace_reserved)
from .ray_utils import (get_ray_directions, get_ray_directions_blender, get_rays, generate_rays, dda, ray_marcher,  # noqa: F401
                        ndc_rays_blender, ndc_rays, sample_pdf, depth2dist, ndc2dist)
from .sh import eval_sh_bases  # noqa: F401
