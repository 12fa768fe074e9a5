// Achievable fp32 / fp64 MFMA rate on this chip (no memory traffic): the practical ceiling for the conv
// kernels (power-managed clocks make it lower than the 157.3 TF datasheet peak).  Dev tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters, float a, float b) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(256) void k_mfma64(double* out, int iters, double a, double b) {
    f64x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
    float* o; double* o2;
    hipMalloc(&o, 4 * 256 * 4096); hipMalloc(&o2, 8 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 768, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma32, dim3(wgs), dim3(256), 0, 0, o, iters, 1.0f, 0.5f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)wgs * 4 * iters * 4 * 2.0 * 32 * 32 * 2;
            if (rep) printf("f32 32x32x2  wgs=%4d  %.2f ms  %.1f TF\n", wgs, ms, flop / ms / 1e9);
        }
    }
    for (int wgs : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma64, dim3(wgs), dim3(256), 0, 0, o2, iters, 1.0, 0.5);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)wgs * 4 * iters * 4 * 2.0 * 16 * 16 * 4;
            if (rep) printf("f64 16x16x4  wgs=%4d  %.2f ms  %.1f TF\n", wgs, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
