// How well do operand fetches hide under another wave's matrix instructions? (dev tool; the conv2 / fc6 piece kernels lose 0.3 of their
// time to operand supply that does not overlap)  A wave repeats: 24 x v_mfma_f32_32x32x16_bf16 (four accumulator
// chains, the kernel's pattern) fed by 12 x ds_read_b128 (+ 3 x global_load_dwordx4 of an L2-resident stream) in several schedules;
// two waves per SIMD (two 4-wave workgroups per CU).  Prints TF/s per variant.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/mfma_feed.hip -o scripts/ubench/mfma_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) const bf16x8 lds_cbf8;

// MODE 0: matrix instructions only (operands loaded once)
// MODE 1: B operands (12 ds_read_b128) fetched AFTER the step's matrix instructions, waited for at the top of the next step
// MODE 2: MODE 1 + A operands (3 global loads, two steps ahead, two register sets)
// MODE 3: B operands fetched BEFORE the step's matrix instructions into a second register set
// MODE 4: MODE 2 with a workgroup barrier every step
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const bf16x8* __restrict__ wsrc, float* out, int iters, int wstride) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[49152];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 49152 / 4; i += 256) ((unsigned*)lds)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + lane * 16;
    bf16x8 a[2][3], b[2][4][3];
    const bf16x8* wp = wsrc + (size_t)(blockIdx.x % 64) * wstride + wave * 192 + lane;
    for (int p = 0; p < 3; ++p) { a[0][p] = wp[p * 64]; a[1][p] = wp[p * 64 + 768]; }
    for (int s = 0; s < 2; ++s) for (int j = 0; j < 4; ++j) for (int p = 0; p < 3; ++p) b[s][j][p] = *(lds_cbf8*)(base + ((j * 3 + p) * 1024));
    f32x16 t[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) t[j][e] = 0.f;
    auto fetch_b = [&](int set, int it) {
        const unsigned o = base + (unsigned)((it & 3) * 12288);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) b[set][j][p] = *(lds_cbf8*)(o + ((j * 3 + p) * 1024));
    };
    auto mm = [&](int sa, int sb) {
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][2], b[sb][j][0], t[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][1], b[sb][j][1], t[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][0], b[sb][j][2], t[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][1], b[sb][j][0], t[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][0], b[sb][j][1], t[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[sa][0], b[sb][j][0], t[j], 0, 0, 0);
    };
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (MODE == 4) __builtin_amdgcn_s_barrier();
            if (MODE != 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (MODE == 3) { fetch_b(h ^ 1, it + h + 1); __builtin_amdgcn_sched_barrier(0); }
            mm((MODE == 2 || MODE == 4) ? h : 0, MODE == 3 ? h : 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2 || MODE == 4) {
                const bf16x8* q = wp + (size_t)(((it + h + 2) * 2304) % wstride);
#pragma unroll
                for (int p = 0; p < 3; ++p) a[h][p] = q[p * 64];
            }
            if (MODE == 1 || MODE == 2 || MODE == 4) fetch_b(0, it + h + 1);
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) s += t[j][0] + t[j][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(const char* what, const bf16x8* w, float* out, int wstride) {
    const int iters = 4000, wgs = 512;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(wgs), dim3(256), 0, 0, w, out, iters, wstride);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = (double)wgs * 4 * iters * 24.0 * 2.0 * 32 * 32 * 16;
    printf("%-86s %.3f ms  %.0f TF\n", what, ms, flop / ms / 1e9);
}
int main() {
    const int wstride = 2304 * 64;                       // 16-byte words per workgroup slice: 64 slices x 2.25 MB... (L2 / MALL resident)
    bf16x8* w; float* out;
    hipMalloc(&w, (size_t)64 * wstride * 16); hipMemset(w, 0x3f, (size_t)64 * wstride * 16);
    hipMalloc(&out, 4 * 256 * 512);
    run<0>("matrix instructions only", w, out, wstride);
    run<1>("+ 12 ds_read_b128 after the products, waited for at the next step's top", w, out, wstride);
    run<2>("+ 3 global_load_dwordx4 two steps ahead", w, out, wstride);
    run<3>("12 ds_read_b128 BEFORE the products into a second register set", w, out, wstride);
    run<4>("as the third line, with a workgroup barrier every step", w, out, wstride);
    return 0;
}
