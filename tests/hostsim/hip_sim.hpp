// hip_sim.hpp -- TEST-ONLY single-lane stand-in for csrc/wave_prims.hpp.
//
// The EM device code (csrc/em_device.hpp) is written against the small vocabulary of
// wave_prims.hpp.  This header defines the same vocabulary for ONE thread per workgroup and a
// wave width of 1, so the *unmodified* device source can be compiled with g++ and its control
// flow / indexing checked against the oracle on a machine without a GPU (pytest -m "not gpu").
// It is not a product path: nothing under vanishing_points_2017_amd/ builds, loads or links it,
// and it says nothing about races or barriers -- those are covered by the -m gpu tests.
#ifndef VPK_WAVE_PRIMS_HPP_
#define VPK_WAVE_PRIMS_HPP_

#include <math.h>
#include <stdint.h>
#include <string.h>

namespace vpk {

#define VPK_GLOBAL
typedef double* gdp;
typedef const double* cgdp;
typedef int* gip;
typedef const float* cgfp;
typedef const unsigned char* cgbp;

constexpr int WAVE = 1;

#define VPK_DEV static inline
#define VPK_DEVFN static
#define VPK_LDS static

alignas(16) static unsigned char g_sim_lds[163840];
VPK_DEV unsigned char* lds_base() { return g_sim_lds; }
VPK_DEV int tid() { return 0; }
VPK_DEV int nthreads() { return 1; }
VPK_DEV int lane() { return 0; }
VPK_DEV int wave_id() { return 0; }
VPK_DEV int nwaves() { return 1; }
VPK_DEV int block_id() { return 0; }
VPK_DEV int nblocks() { return 1; }
VPK_DEV void block_sync() {}
VPK_DEV void wave_sync() {}
VPK_DEV double wave_sum(double v) { return v; }
VPK_DEV int wave_sum_int(int v) { return v; }
VPK_DEV double nanmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
VPK_DEV double wave_max(double v) { return v; }
VPK_DEV int wave_max_int(int v) { return v; }
constexpr int VPG = 1;
template <int G> VPK_DEV double group_sum(double v) { return v; }
template <int G> VPK_DEV int group_sum_int(int v) { return v; }
template <int G> VPK_DEV double group_max(double v) { return v; }
template <int G> VPK_DEV int group_max_int(int v) { return v; }
VPK_DEV void wave_argmin(double&, int&) {}
constexpr int ROWG = 1;
VPK_DEV void row16_argmin(double&, int&) {}
VPK_DEV unsigned long long wave_ballot(bool pred) { return pred ? 1ull : 0ull; }
VPK_DEV unsigned long long lanes_below() { return 0ull; }
VPK_DEV int popcount64(unsigned long long m) { return __builtin_popcountll(m); }
VPK_DEV double wave_bcast(double v, int) { return v; }
VPK_DEV int wave_bcast_int(int v, int) { return v; }
template <int C> VPK_DEV void load_cols(cgdp p, double (&out)[C]) {
    for (int q = 0; q < C; ++q) out[q] = p[q];
}
VPK_DEV void store_cols2(gdp p, double a, double b) { p[0] = a; p[1] = b; }
VPK_DEV void sched_fence() {}
// never executed with one lane (the row-sliced smoother needs a 64-lane wave); present so that the source compiles
template <int BASE> VPK_DEV void fmac8_row_bcast(double* a, double op, double b) { for (int q = 0; q < 8; ++q) a[q] = fma(op, b, a[q]); }
VPK_DEV void wave_lds_order() {}
VPK_DEV double readlane_f64(double v, int) { return v; }
VPK_DEV unsigned lds_addr_of(const void*) { return 0; }
VPK_DEV void lds_dma16(unsigned, const void*, unsigned) {}
template <int N> VPK_DEV void wait_vm() {}
VPK_DEV void raw_barrier() {}
VPK_DEV void pin1(double&) {}   // (the sparse smoother: never executed with one lane either)
VPK_DEV int uniform_int(int v) { return v; }
VPK_DEV cgdp uniform_ptr(cgdp p) { return p; }
VPK_DEV double load_at(cgdp base, unsigned byte_off) { return *reinterpret_cast<cgdp>(reinterpret_cast<const char*>(base) + byte_off); }
VPK_DEV void pin8(double&, double&, double&, double&, double&, double&, double&, double&) {}
VPK_DEV long long __double_as_longlong(double v) { long long r; memcpy(&r, &v, 8); return r; }
VPK_DEV double __longlong_as_double(long long v) { double r; memcpy(&r, &v, 8); return r; }
VPK_DEV long long clock_ticks() { return 0; }
constexpr double CLOCK_US = 0.01;
VPK_DEV int atomic_add_int(int* p, int v) { int o = *p; *p += v; return o; }
VPK_DEV unsigned atomic_or_u32(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }

}  // namespace vpk
#endif
