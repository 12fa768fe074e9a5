// micro-benchmark (dev tool): fp64 FMA with a DPP row_newbcast operand against the smoother's current
// inner step (wave-uniform ds_read_b128 operand reads + v_fma_f64)
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/dpp_fma.hip -o scripts/ubench/dpp_fma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <math.h>

#define FMAC_DPP(ACC, OP, B, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(OP), "v"(B))
#define FMA_PLAIN(ACC, OP, B) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(OP), "v"(B))

// correctness: out[lane][n] = value of lane (row base + n) times b + 0
__global__ void k_check(double* out, const double* in) {
    const double x = in[threadIdx.x];
    const double one = 1.0;
    double r[16];
    for (int i = 0; i < 16; ++i) r[i] = 0.0;
    FMAC_DPP(r[0], x, one, 0); FMAC_DPP(r[1], x, one, 1); FMAC_DPP(r[2], x, one, 2); FMAC_DPP(r[3], x, one, 3);
    FMAC_DPP(r[4], x, one, 4); FMAC_DPP(r[5], x, one, 5); FMAC_DPP(r[6], x, one, 6); FMAC_DPP(r[7], x, one, 7);
    FMAC_DPP(r[8], x, one, 8); FMAC_DPP(r[9], x, one, 9); FMAC_DPP(r[10], x, one, 10); FMAC_DPP(r[11], x, one, 11);
    FMAC_DPP(r[12], x, one, 12); FMAC_DPP(r[13], x, one, 13); FMAC_DPP(r[14], x, one, 14); FMAC_DPP(r[15], x, one, 15);
    for (int i = 0; i < 16; ++i) out[threadIdx.x * 16 + i] = r[i];
}

// 48 accumulators; per step 48 DPP FMAs whose operands come from 3 registers (16 + 16 + 16 VPs)
template <int MODE>   // 0: pure DPP fmac, 1: + 4 ds_read_b64 per step (the new smoother step), 2: pure v_fma_f64,
                      // 3: 12 broadcast ds_read_b128 + 48 v_fma_f64 (the current step)
__global__ __launch_bounds__(512) void k_step(double* out, int iters, const double* src) {
    extern __shared__ double w[];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) w[i] = src[i];
    __syncthreads();
    double acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.0;
    const int lane = threadIdx.x & 63;
    double b0 = 1.0 + threadIdx.x * 1e-6, b1 = 2.0 + threadIdx.x * 1e-6;
    double o0 = w[lane], o1 = w[lane + 64], o2 = w[lane + 128];
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
            const double* p = w + ((it * 67) & 4095) + (lane >> 4) * 16 * 33 + (lane & 15);
            o0 = p[0]; o1 = p[16 * 4]; o2 = p[16 * 8];
            b1 = p[16 * 12];
        }
        if (MODE == 0 || MODE == 1) {
#define ROW3(N) FMAC_DPP(acc[N], o0, b0, N); FMAC_DPP(acc[16 + N], o1, b0, N); FMAC_DPP(acc[32 + N], o2, b1, N);
            ROW3(0) ROW3(1) ROW3(2) ROW3(3) ROW3(4) ROW3(5) ROW3(6) ROW3(7)
            ROW3(8) ROW3(9) ROW3(10) ROW3(11) ROW3(12) ROW3(13) ROW3(14) ROW3(15)
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 48; ++i) FMA_PLAIN(acc[i], o0, b0);
        } else {
            const double* p = w + ((it * 24) & 4095);
#pragma unroll
            for (int t = 0; t < 24; ++t) {
                const double wv = p[t];
                acc[2 * t] = fma(wv, b0, acc[2 * t]);
                acc[2 * t + 1] = fma(wv, b1, acc[2 * t + 1]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> float timeit(F f, int n = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < n; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main() {
    double *out, *src;
    CK(hipMalloc(&out, 1 << 24));
    CK(hipMalloc(&src, 8192 * 8));
    std::vector<double> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = 1.0 + i * 1e-3;
    CK(hipMemcpy(src, h.data(), 8192 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, out, src);
    std::vector<double> r(64 * 16);
    CK(hipMemcpy(r.data(), out, 64 * 16 * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int n = 0; n < 16; ++n)
            if (r[l * 16 + n] != h[(l & ~15) + n]) ++bad;
    printf("row_newbcast check: %d mismatches (lane l, n -> value of lane (l & ~15) + n)\n", bad);
    const int iters = 4000;
    for (int blocks : {1, 256}) {
        float ms;
        ms = timeit([&] { hipLaunchKernelGGL(k_step<0>, dim3(blocks), dim3(512), 65536, 0, out, iters, src); });
        printf("blocks %3d  dpp fmac only          : %.3f ms -> %.1f cycles(2.4GHz)/step of 48 FMA per wave\n", blocks, ms, ms * 1e-3 * 2.4e9 / iters);
        ms = timeit([&] { hipLaunchKernelGGL(k_step<1>, dim3(blocks), dim3(512), 65536, 0, out, iters, src); });
        printf("blocks %3d  dpp fmac + 4 ds_read_b64: %.3f ms -> %.1f cycles/step\n", blocks, ms, ms * 1e-3 * 2.4e9 / iters);
        ms = timeit([&] { hipLaunchKernelGGL(k_step<2>, dim3(blocks), dim3(512), 65536, 0, out, iters, src); });
        printf("blocks %3d  v_fma_f64 only          : %.3f ms -> %.1f cycles/step\n", blocks, ms, ms * 1e-3 * 2.4e9 / iters);
        ms = timeit([&] { hipLaunchKernelGGL(k_step<3>, dim3(blocks), dim3(512), 65536, 0, out, iters, src); });
        printf("blocks %3d  12 bcast b128 + 48 fma  : %.3f ms -> %.1f cycles/step\n", blocks, ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
