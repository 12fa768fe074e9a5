// How a HIP CU mask (hipExtStreamCreateWithCUMask) maps to physical CUs on MI355X (dev tool).
// Finding (r1): bit i selects XCC i % 8, slot i / 8 inside it (shader engine = slot % 4), so the low 8k bits
// are k CUs on every XCC.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xff00);   // se | sh | cu
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}
static void probe(const char* name, int lo, int hi, unsigned* d) {
    uint32_t mask[8] = {0};
    for (int b = lo; b < hi; ++b) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: create failed\n", name); return; }
    (void)hipMemsetAsync(d, 0xff, 8192 * 4, s);
    hipLaunchKernelGGL(where, dim3(8192), dim3(64), 0, s, d);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned> h(8192);
    (void)hipMemcpy(h.data(), d, 8192 * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> u(h.begin(), h.end());
    int per[8] = {0};
    for (unsigned v : u) per[(v >> 16) & 7]++;
    printf("%-12s bits [%3d,%3d): %3zu CUs; per XCC:", name, lo, hi, u.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("\n");
    (void)hipStreamDestroy(s);
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 8192 * 4);
    probe("all", 0, 256, d);
    probe("low 48", 0, 48, d);
    probe("high 208", 48, 256, d);
    probe("low 64", 0, 64, d);
    probe("high 192", 64, 256, d);
    probe("low 8", 0, 8, d);
    probe("one bit", 5, 6, d);
    return 0;
}
