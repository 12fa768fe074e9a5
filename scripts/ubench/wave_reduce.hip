// cycles per call of the wave reductions used by the one-wave clustering (dev micro-benchmark)
//   hipcc --offload-arch=gfx950 -O3 -I. scripts/ubench/wave_reduce.hip -o /tmp/wave_reduce && /tmp/wave_reduce
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../vanishing_points_2017_amd/csrc/wave_prims.hpp"
using namespace vpk;

__device__ void argmin_shfl(double& v, int& idx) {
    for (int o = 32; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o);
        int j = __shfl_xor(idx, o);
        bool take = (u < v) || (u == v && j < idx) || (v != v && u == u);
        v = take ? u : v; idx = take ? j : idx;
    }
}
__global__ void k(long long* out, double* sink, int reps) {
    double v = (double)((threadIdx.x * 2654435761u) % 1000) * 0.001;
    int idx = threadIdx.x;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) { double a = v + r; int b = idx; argmin_shfl(a, b); v += a * 1e-9 + b * 1e-12; }
    long long t1 = clock64();
    for (int r = 0; r < reps; ++r) { double a = v + r; int b = idx; wave_argmin(a, b); v += a * 1e-9 + b * 1e-12; }
    long long t2 = clock64();
    int s = idx;
    for (int r = 0; r < reps; ++r) { s = wave_sum_int(s + r) & 1023; }
    long long t3 = clock64();
    for (int r = 0; r < reps; ++r) { wave_sync(); }
    long long t4 = clock64();
    unsigned long long m = 0;
    for (int r = 0; r < reps; ++r) { m += wave_ballot((idx + r) & 1); }
    long long t5 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; }
    sink[threadIdx.x] = v + s + (double)m;
}
int main() {
    long long* o; double* s;
    hipMalloc(&o, 64); hipMalloc(&s, 64 * 8);
    const int reps = 1000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, reps);
    long long h[5];
    hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
    printf("cycles per call: argmin(shfl) %.0f  argmin(dpp) %.0f  sum_int(dpp) %.0f  wave_sync %.0f  ballot %.0f\n",
           h[0] / (double)reps, h[1] / (double)reps, h[2] / (double)reps, h[3] / (double)reps, h[4] / (double)reps);
    return 0;
}
