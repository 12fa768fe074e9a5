// Two questions about v_mfma_f32_32x32x16_f16 on gfx950 (dev tool):
//   1. are fp16 DENORMAL inputs multiplied or flushed?  (A = 2^-20, B = 1: D = 16 * 2^-20 if multiplied)
//   2. what rate and clock does the chip sustain with operands that CHANGE between instructions (random bits, four A and four B
//      sets in rotation) -- the constant-operand peak of mfma_bf16_chain.hip draws less power than a real kernel does
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_f16_pairs.hip -o scripts/ubench/mfma_f16_pairs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void denorm(float* out) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)9.5367431640625e-07f; b[e] = (_Float16)1.0f; }   // 2^-20: an fp16 denormal
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}

template <bool F16, bool RANDOM>
__global__ __launch_bounds__(512) void rate(float* out, int iters, unsigned seed) {
    u32x4 ra[4], rb[4];
    unsigned x = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 4; ++e) {
            x = x * 1664525u + 1013904223u;
            const unsigned v = RANDOM ? x : 0x3c003c00u;
            // keep exponents moderate (no inf / nan): clear the top exponent bit of both halves
            ra[i][e] = F16 ? (v & 0xbfffbfffu) : (v & 0xbfffbfffu);
            x = x * 1664525u + 1013904223u;
            const unsigned w = RANDOM ? x : 0x3c003c00u;
            rb[i][e] = w & 0xbfffbfffu;
        }
    f32x16 c[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) c[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (F16) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[i]), __builtin_bit_cast(f16x8, rb[(i + j) & 3]), c[j], 0, 0, 0);
                else c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[(i + j) & 3]), c[j], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool F16, bool RANDOM>
void run(float* out, const char* name) {
    const int iters = 40000;                        // 16 instructions each: ~10 ms
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rate<F16, RANDOM>), dim3(256), dim3(512), 0, 0, out, iters, 12345u + rep);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const double n = 256.0 * 8 * iters * 16.0;
        printf("%s rep %d: %.2f ms -> %.0f TF; implied clock %.2f GHz (32 cycles per instruction, 2 waves per SIMD)\n", name, rep, ms,
               n * 2.0 * 32 * 32 * 16 / ms / 1e9, n * 32.0 / 1024 / ms / 1e6);
    }
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipLaunchKernelGGL(denorm, dim3(1), dim3(64), 0, 0, out);
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    printf("denormal input 2^-20 (as fp16: %g) x 1 over K = 16: D = %g (multiplied: %g; flushed: 0)\n", h[1], h[0], 16 * 9.5367431640625e-07);
    run<false, false>(out, "bf16 constant operands");
    run<false, true>(out, "bf16 random operands  ");
    run<true, true>(out, "f16  random operands  ");
    return 0;
}
