// Cycles per v_mfma_f32_32x32x16_bf16 as a function of how many independent accumulator chains a wave interleaves and how many
// waves share a SIMD (dev tool): a chain's next instruction depends on its previous one (C = D).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_bf16_chain.hip -o scripts/ubench/mfma_bf16_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters, float seed) {
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(seed + e); b[e] = (__bf16)(seed - e); }
    f32x16 c[CH];
    for (int i = 0; i < CH; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
            for (int i = 0; i < CH; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < CH; ++i) s += c[i][0] + c[i][15];
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CH>
void run(int threads, float* out, long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * (threads / 64) * iters * 8.0 * 2.0 * 32 * 32 * 16;
    printf("chains %d, waves/SIMD %d: %.1f ticks per MFMA per wave, %.1f per SIMD; %.3f ms -> %.0f TF, tick rate %.2f GHz\n", CH, threads / 256,
           (double)h / (iters * 8.0), (double)h / (iters * 8.0) / (threads / 256), ms, flop / ms / 1e9, (double)h / ms / 1e6);
}
int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 4 * 512 * 256); hipMalloc(&cyc, 8);
    for (int threads : {256, 512}) { run<1>(threads, out, cyc); run<2>(threads, out, cyc); run<4>(threads, out, cyc); run<8>(threads, out, cyc); }
    return 0;
}
