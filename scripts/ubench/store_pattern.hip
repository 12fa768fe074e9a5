// Store throughput of the conv epilogue's access shape (dev tool).  A wave writes a [rows][128 positions]
// tile of a plane-major output (row stride = plane floats, unaligned): (A) 4 bytes per lane, 32 positions of
// two rows per instruction (what the epilogue does), (B) 16 bytes per lane, 4 positions x 8 rows x ... per
// instruction (what a transpose through LDS would allow).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_a(float* out, int plane, int tiles_per_img, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles_per_img * 102; tile += gridDim.x) {
        const int img = tile / tiles_per_img, t = tile % tiles_per_img;
        float* base = out + (size_t)img * rows * plane + t * 128 + wave * 32 + (lane & 31);
        for (int m = (lane >> 5) * 4; m < rows; m += 8)
#pragma unroll
            for (int e = 0; e < 4; ++e) base[(size_t)(m + e) * plane] = (float)(m + e);
    }
}
__global__ __launch_bounds__(256) void k_b(float* out, int plane, int tiles_per_img, int rows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int tile = blockIdx.x; tile < tiles_per_img * 102; tile += gridDim.x) {
        const int img = tile / tiles_per_img, t = tile % tiles_per_img;
        // lane -> (row offset lane / 8, 4 positions (lane % 8) * 4) of this wave's 32 positions
        float* base = out + (size_t)img * rows * plane + t * 128 + wave * 32 + (lane & 7) * 4;
        for (int m = lane >> 3; m < rows; m += 8) {
            float* p = base + (size_t)m * plane;
            float4 v = make_float4((float)m, 1.f, 2.f, 3.f);
            __builtin_memcpy(p, &v, 16);      // possibly unaligned 16-byte store
        }
    }
}
int main() {
    const int plane = 15129, rows = 96, tiles = 15129 / 128;
    float* out; hipMalloc(&out, (size_t)102 * rows * plane * 4 + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        hipEventRecord(a); hipLaunchKernelGGL(k_a, dim3(768), dim3(256), 0, 0, out, plane, tiles, rows); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        const double gb = 102.0 * rows * tiles * 128 * 4 / 1e9;
        printf("A (dword, 2 x 128 B per instr):  %.3f ms  %.2f TB/s\n", ms, gb / ms);
        hipEventRecord(a); hipLaunchKernelGGL(k_b, dim3(768), dim3(256), 0, 0, out, plane, tiles, rows); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("B (dwordx4, 8 x 128 B per instr): %.3f ms  %.2f TB/s\n", ms, gb / ms);
    }
    return 0;
}
