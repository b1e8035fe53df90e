"""MI355X-native TensoRF VM-split ray-marching renderer (the Text2NeRF hot path).

Host-side mirror of the reference interface for this path; the arithmetic runs in libt2n_hip.so (include/t2n.h).
"""
from .renderer import (OctreeRender_trilinear_fast, SimpleSampler, render_views, postprocess_frame,  # noqa: F401
                       evaluation_frames, evaluation, evaluation_path)
from .tensorf import AlphaGridMask, TensorCP, TensorVM, TensorVMSplit, raw2alpha, to_device_async  # noqa: F401
from .ray_utils import (get_ray_directions, get_rays, generate_rays, dda, ray_marcher, ndc_rays_blender,  # noqa: F401
                        ndc_rays)
