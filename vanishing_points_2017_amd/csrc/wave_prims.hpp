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

// The one dynamic-LDS symbol of every EM kernel.  Device functions derive their LDS pointers from
// this symbol (not from pointers stored in structs) so the compiler can prove address space 3 and
// emit ds_read/ds_write instead of flat accesses, also in non-inlined functions.
extern __shared__ __attribute__((aligned(16))) unsigned char vpk_smem[];

namespace vpk {

// HBM pointers carry the global address space explicitly: device functions that are not inlined into
// the kernel would otherwise see generic pointers and emit flat_* accesses, which also count on
// lgkmcnt and so serialise against the LDS operand reads of the smoother.
#define VPK_GLOBAL __attribute__((address_space(1)))
typedef VPK_GLOBAL double* gdp;
typedef const VPK_GLOBAL double* cgdp;
typedef VPK_GLOBAL int* gip;
typedef const VPK_GLOBAL float* cgfp;
typedef const VPK_GLOBAL unsigned char* cgbp;

constexpr int WAVE = 64;  // CDNA4 wavefront width (hard-coded on purpose)

#define VPK_DEV __device__ __forceinline__
#define VPK_DEVFN __device__ __attribute__((noinline))
#define VPK_LDS __shared__

VPK_DEV unsigned char* lds_base() { return vpk_smem; }
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
// Cross-lane moves inside a row of 16 lanes by DPP (row_ror:n, one VALU instruction, no LDS crossbar); a rotation
// serves as well as a butterfly for associative + commutative reductions.  The two cross-row steps (16, 32) stay on
// __shfl_xor.  Used for the exact reductions only (integer sums, lexicographic minima): their result does not
// depend on the order of the operations.  (__shfl_xor = ds_bpermute costs ~100 cycles per dword and step: a 64-lane
// arg-min of a double + index took ~2000 cycles, which made the one-wave clustering reduction-bound.)
template <int N> VPK_DEV int row_ror_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false);
}
template <int N> VPK_DEV double row_ror_f64(double v) {
    const int lo = row_ror_i32<N>(__double2loint(v)), hi = row_ror_i32<N>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
// After the four in-row steps every lane of a row holds its row's result; the four rows are combined through
// v_readlane (SGPR reads of lanes 0 / 16 / 32 / 48): the result is wave-uniform and costs no LDS-crossbar trip.
VPK_DEV int wave_sum_int(int v) {
    v += row_ror_i32<8>(v);
    v += row_ror_i32<4>(v);
    v += row_ror_i32<2>(v);
    v += row_ror_i32<1>(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
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
// The same butterflies over aligned groups of G lanes (G = 16: four independent problems per wave, used
// where a phase has more small problems than waves, e.g. one M-step per VP).  G = 64 == the wave forms.
constexpr int VPG = 16;   // lanes per VP in the M-step
template <int G> VPK_DEV double group_sum(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int G> VPK_DEV int group_sum_int(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int G> VPK_DEV double group_max(double v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v = nanmax(v, __shfl_xor(v, o));
    return v;
}
template <int G> VPK_DEV int group_max_int(int v) {
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
        int u = __shfl_xor(v, o);
        v = u > v ? u : v;
    }
    return v;
}
// lexicographic (value, index) minimum; NaN values never win
VPK_DEV void argmin_take(double& v, int& idx, double u, int j) {
    const bool take = (u < v) || (u == v && j < idx) || (v != v && u == u);
    v = take ? u : v;
    idx = take ? j : idx;
}
VPK_DEV double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// the same over one DPP row of 16 lanes (ROWG): four rotate steps, result in every lane of the row
constexpr int ROWG = 16;
VPK_DEV void row16_argmin(double& v, int& idx) {
    argmin_take(v, idx, row_ror_f64<8>(v), row_ror_i32<8>(idx));
    argmin_take(v, idx, row_ror_f64<4>(v), row_ror_i32<4>(idx));
    argmin_take(v, idx, row_ror_f64<2>(v), row_ror_i32<2>(idx));
    argmin_take(v, idx, row_ror_f64<1>(v), row_ror_i32<1>(idx));
}
VPK_DEV void wave_argmin(double& v, int& idx) {
    argmin_take(v, idx, row_ror_f64<8>(v), row_ror_i32<8>(idx));
    argmin_take(v, idx, row_ror_f64<4>(v), row_ror_i32<4>(idx));
    argmin_take(v, idx, row_ror_f64<2>(v), row_ror_i32<2>(idx));
    argmin_take(v, idx, row_ror_f64<1>(v), row_ror_i32<1>(idx));
    double bv = readlane_f64(v, 0);
    int bi = __builtin_amdgcn_readlane(idx, 0);
    argmin_take(bv, bi, readlane_f64(v, 16), __builtin_amdgcn_readlane(idx, 16));
    argmin_take(bv, bi, readlane_f64(v, 32), __builtin_amdgcn_readlane(idx, 32));
    argmin_take(bv, bi, readlane_f64(v, 48), __builtin_amdgcn_readlane(idx, 48));
    v = bv;
    idx = bi;
}
// lanes of this wave for which pred holds (bit i = lane i), and helpers for ordered compaction
VPK_DEV unsigned long long wave_ballot(bool pred) { return __ballot(pred); }
VPK_DEV unsigned long long lanes_below() { return (1ull << lane()) - 1ull; }
VPK_DEV int popcount64(unsigned long long m) { return __popcll(m); }
VPK_DEV double wave_bcast(double v, int src_lane) { return __shfl(v, src_lane); }
VPK_DEV int wave_bcast_int(int v, int src_lane) { return __shfl(v, src_lane); }

// C adjacent doubles as one load: 16-byte global_load_dwordx4 when C == 2 (p must be 16-byte aligned)
template <int C> VPK_DEV void load_cols(cgdp p, double (&out)[C]);
template <> VPK_DEV void load_cols<1>(cgdp p, double (&out)[1]) { out[0] = p[0]; }
template <> VPK_DEV void load_cols<2>(cgdp p, double (&out)[2]) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t v = *reinterpret_cast<const VPK_GLOBAL d2_t*>(p);   // one global_load_dwordx4
    out[0] = v.x;
    out[1] = v.y;
}

// two adjacent doubles as one 16-byte store (p must be 16-byte aligned)
VPK_DEV void store_cols2(gdp p, double a, double b) {
    typedef double d2_t __attribute__((ext_vector_type(2)));
    const d2_t v = {a, b};
    *reinterpret_cast<VPK_GLOBAL d2_t*>(p) = v;                     // one global_store_dwordx4
}

// scheduling fence: the compiler may not move instructions across it (keeps the unrolled rows of the
// smoother from hoisting all their LDS operand reads to the top and spilling)
VPK_DEV void sched_fence() {
    asm volatile("" ::: "memory");          // IR level: no load/store motion across this point
    __builtin_amdgcn_sched_barrier(0);     // machine scheduler: nothing moves across
}

// pin eight accumulators: everything that produces them is complete before this point and no memory
// access moves across it (an empty asm the optimiser must treat as reading+writing the values)
VPK_DEV void pin8(double& a0, double& a1, double& a2, double& a3, double& a4, double& a5, double& a6, double& a7) {
    asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "memory");
}

// Eight fp64 FMAs whose first factor comes from ANOTHER lane of the same row of 16 lanes: acc[q] += (the value `op`
// has in lane BASE + q of this lane's row) * b, q = 0..7 in this order.  DPP row_newbcast is the one DPP control the
// double-precision ALU has on gfx90a+/gfx950 and it costs nothing on top of the FMA (scripts/ubench/dpp_fma.hip), so 16
// lanes that hold 16 different operands replace 16 wave-uniform LDS broadcast reads by ONE 8-byte read per lane.  All
// lanes of the row must be active (they are the operand sources).  The leading s_nop covers the wait states a DPP read
// needs after a VALU write of `op`: the hazard recogniser does not look inside an asm statement.
template <int BASE> VPK_DEV void fmac8_row_bcast(double* a, double op, double b) {
    asm("s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %8, %9 row_newbcast:%10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %8, %9 row_newbcast:%11 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %8, %9 row_newbcast:%12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %8, %9 row_newbcast:%13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %4, %8, %9 row_newbcast:%14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %5, %8, %9 row_newbcast:%15 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %6, %8, %9 row_newbcast:%16 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %7, %8, %9 row_newbcast:%17 row_mask:0xf bank_mask:0xf"
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
        : "v"(op), "v"(b), "n"(BASE), "n"(BASE + 1), "n"(BASE + 2), "n"(BASE + 3), "n"(BASE + 4), "n"(BASE + 5),
          "n"(BASE + 6), "n"(BASE + 7));
}
// one value: whatever produces it (a load) is complete before this point
VPK_DEV void pin1(double& a) { asm volatile("" : "+v"(a)); }
// a value every lane holds identically, moved to a scalar register: loops and branches on it are scalar (values read through
// a context pointer arrive in vector registers and would otherwise be treated as divergent)
VPK_DEV int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
VPK_DEV cgdp uniform_ptr(cgdp p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (cgdp)(((unsigned long long)hi << 32) | lo);
}
// one double at (scalar base) + (per-lane byte offset): the saddr form of global_load, no per-load address arithmetic
VPK_DEV double load_at(cgdp base, unsigned byte_off) {
    return *reinterpret_cast<cgdp>(reinterpret_cast<const VPK_GLOBAL char*>(base) + byte_off);
}
// Order this wave's LDS accesses as written.  The LDS serves one wave's instructions in issue order, so a wave that
// writes LDS and reads the words back (other lanes' words included: a wave's lanes run in lockstep) needs no wait and
// no barrier instruction -- only the compiler must keep the order.
VPK_DEV void wave_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// LDS-DMA (global_load_lds_dwordx4): the 64 lanes of a wave move 16 bytes each from (sbase + voff), sbase wave-uniform, to
// LDS at lds_dst + 16 * lane -- no staging registers, no ds_write.  Issued from inline asm: the compiler does not know that
// the instruction writes LDS and is not asked to; completion is counted by hand (wait_vm<N>: at most N vector-memory
// operations of this wave still outstanding; they retire in order) and published to the other waves by a barrier.  M0
// (compiler-reserved) carries the LDS address and is saved / restored inside the statement; the s_nop covers the wait
// states after a VALU write of the SGPR operands, which the hazard recogniser cannot see inside an asm statement.
VPK_DEV unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
VPK_DEV void lds_dma16(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int N> VPK_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
VPK_DEV void raw_barrier() {                        // LDS traffic of this wave done, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// constant-rate (100 MHz) device clock for the optional phase timing in the EM trace
VPK_DEV long long clock_ticks() { return (long long)wall_clock64(); }
constexpr double CLOCK_US = 0.01;

VPK_DEV int atomic_add_int(int* p, int v) { return atomicAdd(p, v); }
VPK_DEV unsigned atomic_or_u32(unsigned* p, unsigned v) { return atomicOr(p, v); }

}  // namespace vpk
#endif
