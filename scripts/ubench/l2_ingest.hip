// How many bytes per second a CU can pull from L2 / Infinity Cache (dev tool; decides whether the convolution kernels are
// bound by what they ingest): every workgroup re-reads a footprint that is shared by all workgroups (weights-like: F bytes
// in total, each workgroup streams all of it, starting at its own phase) with
//   (a) global_load_dwordx4 into registers (16 loads in flight per lane),
//   (b) global_load_lds_dwordx4 (LDS-DMA, 1 KB per wave-instruction, counted vmcnt),
// for 1, 2 and 3 workgroups of 256 threads per CU.  Prints GB/s per CU and in total.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/l2_ingest.hip -o scripts/ubench/l2_ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__global__ __launch_bounds__(256) void k_regs(const f32x4* __restrict__ src, size_t n16, int rounds, float* out) {
    // n16 = footprint in 16-byte words (a multiple of 256 * 16)
    f32x4 acc = {0, 0, 0, 0};
    const size_t chunk = 256 * 16;                       // words one workgroup reads per step
    size_t pos = ((size_t)blockIdx.x * 7919u * chunk) % n16;
    for (int r = 0; r < rounds; ++r) {
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = src[pos + (size_t)u * 256 + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
        pos += chunk;
        if (pos >= n16) pos -= n16;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(256) void k_dma(const f32x4* __restrict__ src, size_t n16, int rounds, float* out) {
    __shared__ __attribute__((aligned(16))) f32x4 buf[2][16 * 256];        // 2 x 64 KB? no: 16 * 256 * 16 B = 64 KB each
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const size_t chunk = 256 * 16;
    size_t pos = ((size_t)blockIdx.x * 7919u * chunk) % n16;
    const unsigned base = (unsigned)(size_t)(lds_ptr_t)&buf[0][0];
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        const unsigned dst0 = base + (unsigned)((r & 1) * 16 * 256 * 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {                   // wave w moves pieces 4 u + w (1 KB each)
            const f32x4* s = src + pos + (size_t)(4 * u + wave) * 64;
            const unsigned dst = __builtin_amdgcn_readfirstlane(dst0 + (unsigned)((4 * u + wave) * 1024));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"((unsigned)lane * 16u), "s"(s), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");   // the previous round's pieces have landed
        if (r) acc += ((const float*)&buf[(r - 1) & 1][0])[threadIdx.x];
        pos += chunk;
        if (pos >= n16) pos -= n16;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    float* out; hipMalloc(&out, 4 * 256 * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {1, 4, 16, 64, 1024}) {
        const size_t bytes = mb << 20, n16 = bytes / 16;
        f32x4* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
        for (int per_cu : {1, 2, 3}) {
            for (int which = 0; which < 2; ++which) {
                if (which == 1 && per_cu > 1) continue;    // the DMA kernel holds 128 KB of LDS
                const int wgs = cus * per_cu, rounds = 2000;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (which == 0) hipLaunchKernelGGL(k_regs, dim3(wgs), dim3(256), 0, 0, src, n16, rounds, out);
                    else hipLaunchKernelGGL(k_dma, dim3(wgs), dim3(256), 0, 0, src, n16, rounds, out);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double gb = (double)wgs * rounds * 65536.0 / 1e9;
                printf("footprint %5zu MB  %s  %d wg/CU : %7.1f GB/s total, %6.1f GB/s per CU\n", mb, which ? "lds-dma" : "regs   ",
                       per_cu, gb / (ms * 1e-3), gb / (ms * 1e-3) / cus);
            }
        }
        hipFree(src);
    }
    return 0;
}
