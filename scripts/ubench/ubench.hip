// micro-benchmarks of the primitives the EM smoother is built from (dev tool, not product code)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k_fma(double* out, int iters, double a, double b) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_lds_fma(double* out, int iters, const double* src) {
    __shared__ double w[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) w[i] = src[i];
    __syncthreads();
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    double x0 = threadIdx.x * 1e-3, x1 = x0 + 1;
    for (int it = 0; it < iters; ++it) {
        const double* wr = w + (it & 511) * 8;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            double wv = wr[t];
            acc[2 * t] = fma(wv, x0, acc[2 * t]);
            acc[2 * t + 1] = fma(wv, x1, acc[2 * t + 1]);
        }
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the smoother's shape: rows of 128 doubles per wave (16 B per lane), 8 rows in flight
__global__ void k_stream(double* out, const double* mat, int ld, int rows, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* base = mat + (size_t)blockIdx.x * rows * ld + wave * 128 + lane * 2;
    double s0 = 0, s1 = 0;
    for (int r = 0; r < reps; ++r)
        for (int j = 0; j + 8 <= rows; j += 8) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(base + (size_t)(j + u) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s0 += v[u].x; s1 += v[u].y; }
        }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1;
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
    double *out, *src, *mat;
    CK(hipMalloc(&out, 1 << 24));
    CK(hipMalloc(&src, 4096 * 8));
    CK(hipMemset(src, 0, 4096 * 8));
    const int ld = 1024, rows = 1000;
    size_t matbytes = (size_t)512 * rows * ld * 8;
    CK(hipMalloc(&mat, matbytes));
    CK(hipMemset(mat, 0, matbytes));
    for (int blocks : {1, 102, 256, 512}) {
        for (int threads : {256, 512, 1024}) {
            const int iters = 20000;
            float ms = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0000001, 1e-9); });
            double cyc_per_fma = ms * 1e-3 * 2.4e9 / (iters * 16.0);
            printf("fma      blocks %3d threads %4d: %.3f ms  -> %.2f cycles(2.4GHz)/wave-FMA per wave, %.1f GFMA/s per block\n", blocks, threads, ms, cyc_per_fma,
                   (double)iters * 16 * threads / (ms * 1e-3) / 1e9);
            ms = timeit([&] { hipLaunchKernelGGL(k_lds_fma, dim3(blocks), dim3(threads), 0, 0, out, iters, src); });
            printf("lds+fma  blocks %3d threads %4d: %.3f ms  -> %.2f cycles/row(16 FMA + 4 b128) per wave\n", blocks, threads, ms, ms * 1e-3 * 2.4e9 / iters);
        }
    }
    for (int blocks : {1, 102, 256, 512}) {
        // L2/MALL-resident (rows=250 -> 2 MB per block) and HBM-streaming (rows=1000 -> 8 MB per block)
        for (int r : {250, 1000}) {
            int reps = r == 250 ? 40 : 10;
            float ms = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(512), 0, 0, out, mat, ld, r, reps); });
            double bytes = (double)blocks * r * 1024 * 8 * reps;
            printf("stream   blocks %3d rows %4d: %.3f ms  -> %.1f GB/s per block, %.2f TB/s total, %.0f cycles per row-step\n", blocks, r, ms,
                   bytes / blocks / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (r * reps));
        }
    }
    return 0;
}
