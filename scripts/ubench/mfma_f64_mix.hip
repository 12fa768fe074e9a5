// micro-benchmark (dev tool, round 6): can the EM's smoother gain from v_mfma_f64_16x16x4_f64?  (VERDICT r5 item 2b priced it at
// "17 us at the f64 matrix rate against 50-57".)  One EM-shaped workgroup per CU -- 512 threads, two waves per SIMD -- runs
//   mode 0   48 independent v_fma_f64 per step and wave (the smoother's inner step without its operand traffic)
//   mode 1   12 v_mfma_f64_16x16x4_f64 per step and wave (four independent accumulator chains)
//   mode 2   both, interleaved 4 : 1 -- do the vector ALU and the matrix pipe run side by side, or do they share the issue / the
//            f64 datapath?
// and prints f64 FMAs per cycle and CU (clock from wall time / s_memtime is avoided: cycles are derived at the measured 2.4 GHz only
// for the per-step figure; the FMA / s number is clock-free).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f64_mix.hip -o scripts/ubench/mfma_f64_mix
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k_mix(double* out, int iters, double seed) {
    double acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.0;
    f64x4 c[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = f64x4{0.0, 0.0, 0.0, 0.0};
    double a = seed + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 48; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 12; ++i) c[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i & 3], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                c[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[4 * i + q]) : "v"(a), "v"(b));
            }
        }
        a += 1e-12;                                  // (operands change between steps)
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> float timeit(F f, int n = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < n; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main() {
    double* out;
    if (hipMalloc(&out, 1 << 24) != hipSuccess) return 1;
    const int iters = 20000;
    for (int blocks : {1, 102, 256}) {
        const double valu = 48.0 * 64, mfma = 12.0 * 16 * 16 * 4;           // FMAs per step and wave
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_mix<0>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0); });
        printf("blocks %3d  48 v_fma_f64            : %.3f ms  %.1f FMA/ns/CU  (%.0f cycles@2.4GHz per step)\n", blocks, ms, valu * 8 * iters / (ms * 1e6), ms * 1e-3 * 2.4e9 / iters);
        ms = timeit([&] { hipLaunchKernelGGL(k_mix<1>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0); });
        printf("blocks %3d  12 v_mfma_f64_16x16x4   : %.3f ms  %.1f FMA/ns/CU  (%.0f cycles per step)\n", blocks, ms, mfma * 8 * iters / (ms * 1e6), ms * 1e-3 * 2.4e9 / iters);
        ms = timeit([&] { hipLaunchKernelGGL(k_mix<2>, dim3(blocks), dim3(512), 0, 0, out, iters, 1.0); });
        printf("blocks %3d  12 mfma + 48 fma mixed  : %.3f ms  %.1f FMA/ns/CU  (%.0f cycles per step)\n", blocks, ms, (valu + mfma) * 8 * iters / (ms * 1e6), ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
