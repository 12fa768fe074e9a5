// vpk_raster.hip -- inverse-gnomonic sphere rasteriser (sphere_mapping.py:36-72) for gfx950.
//
// The reference draws, for every line (a,b,c), the curve beta(alpha) = atan((-a sin alpha - c cos alpha)/b)
// over alpha in [-pi/2, pi/2] (:61-63) as an anti-aliased 1-pt polyline (matplotlib Agg, 100 dpi ->
// 1.389 px wide), white with alpha 0.1 over black, one draw call per line (:65), and returns the grey
// mean as uint8 (:67-68); image row 0 is beta = +pi/2.  Bit-exact parity with Agg's scanline
// rasteriser is not attainable (and the reference itself depends on the matplotlib version, SURVEY 8a
// R1), so this kernel reproduces the geometry and the compositing model: per line, pixel coverage
// from the perpendicular distance to the curve (box filter of the stroke width), composited "over" in
// line order with 8-bit rounding after every line, alpha = floor(0.1 * 255) / 255.
//
// Decomposition: a thread owns one pixel COLUMN (fixed alpha, so sin/cos are computed once) and
// walks the lines in order; for each line it touches only the rows the curve covers in that column
// (work ~ curve length, not pixels x lines).  The column's pixels live in LDS ([row][col] so a wave
// touches consecutive bytes), and are written out as coalesced rows at the end.
#include "vpk_internal.hpp"

namespace {

constexpr int RCOLS = 128;            // columns per workgroup
constexpr float LINE_WIDTH_PX = 100.0f / 72.0f;   // 1 pt at 100 dpi (matplotlib 1.5.1 default width)

__global__ __launch_bounds__(RCOLS) void raster_kernel(const double* __restrict__ l, const long long* __restrict__ offsets,
                                                        int size, float alpha, unsigned char* __restrict__ out) {
    extern __shared__ unsigned char px[];            // [size][RCOLS]
    const int img = blockIdx.y;
    const int col0 = blockIdx.x * RCOLS;
    const int x = col0 + threadIdx.x;
    const long long lo = offsets[img], hi = offsets[img + 1];
    for (int r = 0; r < size; ++r) px[r * RCOLS + threadIdx.x] = 0;
    const float PI = 3.14159265358979323846f;
    const float px_per_rad = size / PI;
    const float a_x = -PI / 2 + (x + 0.5f) * (PI / size);
    float sa, ca;
    sincosf(a_x, &sa, &ca);
    const float a8 = floorf(alpha * 255.0f) / 255.0f;   // 8-bit alpha (0.1 -> 25/255)
    const float reach = 0.5f * LINE_WIDTH_PX + 0.5f;    // coverage falls to 0 at this distance
    if (x < size) {
        for (long long n = lo; n < hi; ++n) {
            const float la = (float)l[3 * n], lb = (float)l[3 * n + 1], lc = (float)l[3 * n + 2];
            const float g = -la * sa - lc * ca;
            const float u = g / lb;
            const float beta = atanf(u);                                  // sphere_mapping.py:63
            const float slope = ((-la * ca + lc * sa) / lb) / (1.0f + u * u);   // d beta / d alpha
            const float inv = rsqrtf(1.0f + slope * slope);
            if (!(beta == beta)) continue;
            const float yc = (PI / 2 - beta) * px_per_rad - 0.5f;          // row coordinate of the curve
            const float ext = reach / inv;                                 // vertical extent of the stroke
            int r0 = (int)floorf(yc - ext), r1 = (int)ceilf(yc + ext);
            r0 = r0 < 0 ? 0 : r0;
            r1 = r1 > size - 1 ? size - 1 : r1;
            for (int r = r0; r <= r1; ++r) {
                const float d = fabsf((float)r - yc) * inv;               // perpendicular distance (px)
                float cov = reach - d;
                cov = cov < 0.f ? 0.f : (cov > 1.f ? 1.f : cov);
                if (cov > 0.f) {
                    const float v = (float)px[r * RCOLS + threadIdx.x];
                    px[r * RCOLS + threadIdx.x] = (unsigned char)floorf(v + (255.0f - v) * a8 * cov + 0.5f);
                }
            }
        }
    }
    __syncthreads();
    unsigned char* o = out + (size_t)img * size * size;
    for (int p = threadIdx.x; p < size * RCOLS; p += RCOLS) {
        int r = p / RCOLS, c = p % RCOLS;
        if (col0 + c < size) o[(size_t)r * size + col0 + c] = px[r * RCOLS + c];
    }
}

}  // namespace

extern "C" {

int vpk_sphere_raster(vpk_handle* h, const double* l, const int64_t* offsets, int batch, int size, double alpha,
                      uint8_t* out) {
    if (!h || !l || !offsets || !out || batch < 1 || size < 8 || size > 1024)
        return vpk_fail(h, VPK_ERR_ARG, "vpk_sphere_raster: bad argument (size must be 8..1024)");
    VPK_HIP(h, hipSetDevice(h->device));
    const size_t lds = (size_t)size * RCOLS;
    if (!h->raster_ready) {
        VPK_HIP(h, hipFuncSetAttribute(reinterpret_cast<const void*>(raster_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 1024 * RCOLS));
        h->raster_ready = true;
    }
    // offsets [host] -> device (stream-ordered staging through the handle's pinned buffer)
    VPK_HIP(h, hipStreamSynchronize(h->stream));
    const size_t ob = (size_t)(batch + 1) * 8;
    int rc = vpk_reserve(h, &h->raster_hdr, &h->raster_hdr_bytes, ob, "hipMalloc(raster offsets)");
    if (rc) return rc;
    VPK_HIP(h, hipMemcpyAsync(h->raster_hdr, offsets, ob, hipMemcpyHostToDevice, h->stream));
    VPK_HIP(h, hipStreamSynchronize(h->stream));       // offsets is caller-owned pageable memory
    dim3 grid((size + RCOLS - 1) / RCOLS, batch);
    hipLaunchKernelGGL(raster_kernel, grid, dim3(RCOLS), lds, h->stream, l, (const long long*)h->raster_hdr, size,
                       (float)alpha, out);
    VPK_HIP(h, hipGetLastError());
    return VPK_OK;
}

}  // extern "C"
