// placeholder until the CNN kernels land (next milestone)
#include "vpk_internal.hpp"
void vpk_cnn_free(vpk_handle*) {}
extern "C" {
int vpk_cnn_load(vpk_handle* h, const float* const*, const float*) { return vpk_fail(h, VPK_ERR_STATE, "CNN not built yet"); }
int vpk_cnn_forward(vpk_handle* h, const uint8_t*, int, float*) { return vpk_fail(h, VPK_ERR_STATE, "CNN not built yet"); }
int vpk_cnn_forward_tap(vpk_handle* h, const uint8_t*, int, float*, int, float*) { return vpk_fail(h, VPK_ERR_STATE, "CNN not built yet"); }
}
