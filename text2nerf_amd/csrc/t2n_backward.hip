// Backward of the render call (a-15). Implemented in a later milestone; until then the entry point fails loudly.
#include "t2n_device.h"

using namespace t2n;

extern "C" int t2n_render_backward(t2n_field* f, const float* rays, int64_t n_rays, int ray_stride, int n_samples, uint32_t flags,
                                   const float* jitter, const float* weights, const float* z_vals, const float* d_rgb,
                                   const float* d_depth, const float* d_weights, const t2n_field_grads* g, void* workspace,
                                   size_t workspace_bytes, t2n_stream stream) {
    (void)f; (void)rays; (void)n_rays; (void)ray_stride; (void)n_samples; (void)flags; (void)jitter; (void)weights; (void)z_vals;
    (void)d_rgb; (void)d_depth; (void)d_weights; (void)g; (void)workspace; (void)workspace_bytes; (void)stream;
    set_error("t2n_render_backward: not implemented in this build");
    return T2N_ERR_UNSUPPORTED;
}
