// Occupy a few CUs for a while on a given stream: used to find out what slows a co-running kernel (dev tool).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(512) void spin_kernel(double* buf, long long iters, int mode, size_t per_wg) {
    double a = threadIdx.x * 1e-3, b = 1.000001;
    double* p = buf + (size_t)blockIdx.x * per_wg;
    if (mode == 0) {
        for (long long i = 0; i < iters; ++i) a = fma(a, b, 1e-9);
    } else if (mode == 1) {                                   // read-modify-write over this workgroup's slice
        for (long long i = 0; i < iters; ++i) {
            size_t k = ((size_t)i * 512 + threadIdx.x) % per_wg;
            a += p[k];
            p[k] = a * 1e-9;
        }
    } else {                                                  // read only
        for (long long i = 0; i < iters; ++i) a += p[((size_t)i * 512 + threadIdx.x) % per_wg];
    }
    if (a == 12345.678) buf[0] = a;
}
extern "C" int spin(void* stream, int wgs, long long iters, int mode, double* buf, size_t per_wg, int lds_bytes) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(512), lds_bytes, (hipStream_t)stream, buf, iters, mode, per_wg);
    return (int)hipGetLastError();
}
