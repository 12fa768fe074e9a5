// wave_prims.hpp -- gfx950 wavefront / workgroup primitives used by the EM device code.
//
// Everything that touches a hardware intrinsic lives here: 64-lane butterfly reductions
// (ds_swizzle/DPP via __shfl_xor), workgroup barriers and LDS-visible wave fences.  The EM
// device code (em_device.hpp) is written against this small vocabulary only.
#ifndef VPK_WAVE_PRIMS_HPP_
#define VPK_WAVE_PRIMS_HPP_

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace vpk {

constexpr int WAVE = 64;  // CDNA4 wavefront width (hard-coded on purpose)

#define VPK_DEV __device__ __forceinline__
#define VPK_DEVFN __device__
#define VPK_LDS __shared__

VPK_DEV int tid() { return (int)threadIdx.x; }
VPK_DEV int nthreads() { return (int)blockDim.x; }
VPK_DEV int lane() { return (int)(threadIdx.x & 63u); }
VPK_DEV int wave_id() { return (int)(threadIdx.x >> 6); }
VPK_DEV int nwaves() { return (int)(blockDim.x >> 6); }
VPK_DEV int block_id() { return (int)blockIdx.x; }
VPK_DEV int nblocks() { return (int)gridDim.x; }

// workgroup barrier; HIP's __syncthreads also orders global + LDS accesses at workgroup scope
VPK_DEV void block_sync() { __syncthreads(); }

// make this wave's LDS/global writes visible to its own other lanes (no cross-wave meaning)
VPK_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // drains vmcnt/lgkmcnt for this wave
    __builtin_amdgcn_wave_barrier();
}

// 64-lane butterfly reductions: every lane receives the (bitwise identical) result
VPK_DEV double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
VPK_DEV int wave_sum_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// NaN-propagating max (numpy.max semantics): any NaN lane makes the result NaN
VPK_DEV double nanmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
VPK_DEV double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o));
    return v;
}
VPK_DEV int wave_max_int(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        int u = __shfl_xor(v, o);
        v = u > v ? u : v;
    }
    return v;
}
// lexicographic (value, index) minimum; NaN values never win
VPK_DEV void wave_argmin(double& v, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double u = __shfl_xor(v, o);
        int j = __shfl_xor(idx, o);
        bool take = (u < v) || (u == v && j < idx) || (v != v && u == u);
        v = take ? u : v;
        idx = take ? j : idx;
    }
}
VPK_DEV double wave_bcast(double v, int src_lane) { return __shfl(v, src_lane); }
VPK_DEV int wave_bcast_int(int v, int src_lane) { return __shfl(v, src_lane); }

// C adjacent doubles as one load: 16-byte global_load_dwordx4 when C == 2 (p must be 16-byte aligned)
template <int C> VPK_DEV void load_cols(const double* p, double (&out)[C]);
template <> VPK_DEV void load_cols<1>(const double* p, double (&out)[1]) { out[0] = p[0]; }
template <> VPK_DEV void load_cols<2>(const double* p, double (&out)[2]) {
    const double2 v = *reinterpret_cast<const double2*>(p);
    out[0] = v.x;
    out[1] = v.y;
}

// constant-rate (100 MHz) device clock for the optional phase timing in the EM trace
VPK_DEV long long clock_ticks() { return (long long)wall_clock64(); }
constexpr double CLOCK_US = 0.01;

VPK_DEV int atomic_add_int(int* p, int v) { return atomicAdd(p, v); }
VPK_DEV unsigned atomic_or_u32(unsigned* p, unsigned v) { return atomicOr(p, v); }

}  // namespace vpk
#endif
