// placeholder until the raster kernel lands
#include "vpk_internal.hpp"
extern "C" {
int vpk_sphere_raster(vpk_handle* h, const double*, const int64_t*, int, int, double, uint8_t*) { return vpk_fail(h, VPK_ERR_STATE, "raster not built yet"); }
}
