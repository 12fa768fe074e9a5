// cnn_conv_pieces.hpp -- conv2 of cnn/deploy.prototxt (:56-75: 5 x 5, stride 1, pad 2, two groups of 48 -> 128 channels) as a DIRECT
// convolution on the bf16 matrix cores with exact operands.  Included by vpk_cnn.hip (after cnn_split_gemm.hpp: bf16x8, split3, dma16).
//
// Arithmetic (as cnn_split_gemm.hpp): every f32 operand is exactly the sum of three bf16 pieces, a product of two pieces is
// exact in f32, and of the nine partial products the six with i + j <= 4 carry everything above 2^-24 of the product.  What
// is new here is where the sums are rounded: the products of FP consecutive K16 steps (a kernel row of conv2: 5 taps x 16
// channels) accumulate in a BLOCK SUM that starts from zero -- per step the five small products first, the large one last --
// and the block sum joins the running accumulator with one f32 addition.  The accumulator is rounded once per kernel row and
// 16 channels (15 times per output) instead of six times per step (450 times), and what is rounded inside a block is a
// twentieth of the final magnitude.  Measured against the float64 net (round 5, scripts/cnn_accuracy.py, error of conv2's blob
// relative to its largest value): 0.21e-6 -- f32-input direct kernel 1.02e-6, Winograd F(2 x 2, 5 x 5) on the f32 cores 0.59e-6,
// six chained roundings per step (cnn_split_gemm.hpp as it was) 1.06e-6.
//
// Data movement:
//   * activations as "P6" planes: [image][channel group of 16][piece x k half = 6][y][x] -> 16 bytes (8 bf16 = the B operand of
//     one lane for one pixel), with the convolution's zero border (to_p6_kernel).  A tile = 128 output channels x 4 rows x 32
//     columns; per channel group its RAW input patch ((4 + KH - 1) x (32 + KW - 1) pixels x 6 planes, 30 KB) comes ONCE by
//     LDS-DMA, a channel group ahead, and every tap reads it at a shifted address: a lane's operand for tap (kh, kw) is the 16
//     bytes at (pixel + kh * row + kw) -- consecutive lanes, consecutive words, no bank conflicts, no im2col (the gather of
//     cnn_split_gemm.hpp fetches every pixel 25 times; its waves spent as long issuing it as multiplying);
//   * weights: the A-fragment stream of cnn_split_gemm.hpp ([group][K16 step][32-row block][piece][lane][8]; 1.8 MB, L2-resident)
//     goes per wave straight from L2 into registers, two steps ahead (two register sets): the waves of a workgroup share
//     nothing but the patch, so the only barrier is at a channel group's first step (every 25 steps);
//   * four waves per workgroup (one 32-row block of the output channels each, four 32 x 32 accumulator blocks), two workgroups
//     per CU: one's prologue / epilogue runs under the other's matrix instructions.
// Where it stands (round 5, B = 102): 1.2 ms against 1.46 ms for the Winograd kernel; the matrix pipes are 60 % busy (0.72 ms of
// matrix instructions at the 2.05 GHz the kernel holds).  Ablations on the GPU (scripts/cp_experiments.sh: the loop with the
// weight loads, the LDS operand reads, the patch DMA, the barrier, the block-sum additions and the stores removed one by one):
// matrix instructions alone 0.83 ms; + operand reads 0.11, + weight loads 0.15, + additions / stores 0.05 -- these do NOT hide
// under the other wave's matrix instructions, in any of the four schedules tried (operands at the top of a lockstep step: 1.22;
// a ping-pong of the SIMD's two waves over two barriers per step: 1.48; lockstep with the next step's operands requested before
// the products: 1.33; independent waves, this version: 1.21).
#ifndef VPK_CNN_CONV_PIECES_HPP_
#define VPK_CNN_CONV_PIECES_HPP_

namespace {

constexpr int CP_THREADS = 256;            // four waves; TWO workgroups per CU (one's prologue / epilogue under the other's products)
constexpr int CP_TC = 32;                       // columns of a tile (its rows: the kernel's NB)
constexpr int CP_NST = 4;                       // weight stages in LDS (stage s + 4 takes the slot of stage s, which is in registers by then)

struct PieceDims {
    int B;                                      // images
    int Cg16, CGtot, Hp, Wp;                    // input: channel groups (of 16) per conv group / in the tensor; padded plane
    int OC, OH, OW, groups;                     // OC = output channels per conv group
    int KW, ntaps, ksteps;                      // kernel width; KH * KW; K16 steps per conv group (Cg16 * ntaps)
    int mblocks;                                // 32-row blocks per conv group in the packed weights
    int mtiles, rtiles, ctiles;                 // tiles per (image, conv group): output-channel tiles, row tiles, column tiles
    int relu;
    int OHp, OWp, opad;                         // f32 NCHW output planes
    long long in_image;                         // bytes per image of the P6 input tensor
};

// f32 NCHW planes (with their zero border) -> P6 planes.  One workgroup per (image, channel group, row): 16 channels x Wp
// values in, 6 x Wp 16-byte words out.
__global__ __launch_bounds__(256) void to_p6_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C, int Hp,
                                                    int Wp) {
    const int y = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
    const float* src = in + (((size_t)b * C + cg * 16) * Hp + y) * Wp;
    u32x4* dst = reinterpret_cast<u32x4*>(out) + (((size_t)b * (C >> 4) + cg) * 6 * Hp + y) * Wp;
    for (int idx = threadIdx.x; idx < 2 * Wp; idx += 256) {      // (k half, x)
        const int h = idx / Wp, x = idx - h * Wp;
        unsigned short p[3][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(src[(size_t)(8 * h + e) * Hp * Wp + x], p[0][e], p[1][e], p[2][e]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 w4;
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = (unsigned)p[q][2 * e] | ((unsigned)p[q][2 * e + 1] << 16);
            dst[(size_t)(2 * q + h) * Hp * Wp + x] = w4;
        }
    }
}

#ifdef CP_TIME
// development: shader-clock laps of the kernel's phases per wave (scripts/cp_phase_times.py; a build with -DCP_TIME)
__device__ long long cp_dbg[2 * 256 * 8 * 8];
#define CP_LAP(slot) do { const long long now_ = __builtin_readcyclecounter(); tacc[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define CP_LAP(slot) do { } while (0)
#endif
template <int N>
__device__ __forceinline__ void cp_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// A 16-byte global load the COMPILER DOES NOT KNOW TO BE A LOAD (round 5).  hipcc keeps a scoreboard of the vector-memory operations it
// has emitted and puts its own s_waitcnt in front of the first use of a loaded register; with LDS-DMA issued from inline asm in between
// (which it cannot count) it falls back to `s_waitcnt vmcnt(0)` -- in front of every K16 step's first matrix instruction here, which turned
// "weights two steps ahead" into "everything in flight must land now" (the ISA of the first version: `s_waitcnt vmcnt(0) lgkmcnt(11)`).
// Issued from asm, completion is counted by hand (cp_wait_for below ties the wait to the registers it releases, so that no use can be
// scheduled above it).  s_nop: the scalar base may have been written by v_readlane just before (VALU-writes-SGPR -> VMEM hazard).
// The destination is a "+v" operand -- the load overwrites the register the variable already lives in -- so that the compiler has no
// fresh value to copy into place (a copy issued before the data has landed would copy the old contents).
__device__ __forceinline__ void cp_gload16(bf16x8& dst, const void* sbase, unsigned voff) {
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "+v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// (The wait must NOT take the registers as operands: hipcc then copies them into the asm's operand registers BEFORE the wait -- stale data.
// A scheduling barrier behind the wait keeps the matrix instructions, the only readers, below it.)
template <int N>
__device__ __forceinline__ void cp_wait_for(bf16x8&, bf16x8&, bf16x8&) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// KH: kernel size (5: conv2; 3 is kept for measurements on the 3 x 3 layers).  A tile = 128 output channels x (4 rows x 32 columns).
// Four waves, one per 32-row block of the output channels, each all 4 x 32 pixels (four 32 x 32 blocks).
template <int KH, int NB>
__global__ __launch_bounds__(CP_THREADS, NB == 4 ? 2 : 3) void conv_pieces_kernel(PieceDims d, const unsigned short* __restrict__ act,
                                                                     const unsigned short* __restrict__ wfrag,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     int* __restrict__ tile_counter, int total_tiles) {
    constexpr int MB = 4;                                    // 32-row blocks per tile
    constexpr int CP_TR = NB;                                // rows of a tile = 32 x 32 blocks per wave
    constexpr int PR = CP_TR + KH - 1, PC = CP_TC + KH - 1;  // patch rows / columns
    constexpr int PRW = PR * PC;                             // 16-byte words per plane
    constexpr int KPP = (PRW + 63) / 64;                     // DMA instructions per plane
    constexpr int PLANE = KPP * 64 * 16;                     // bytes per plane in LDS (tail lanes write into the padding)
    constexpr int PBUF = 6 * PLANE;                          // bytes per patch buffer
    constexpr int NPW = (6 * KPP + 3) / 4;                   // patch DMA instructions per wave and channel group
    constexpr int FP = KH == 5 ? 5 : 9;                      // K16 steps per block sum (a kernel row of conv2, all taps of a 3 x 3 layer)
    __shared__ __attribute__((aligned(16))) unsigned char cp_lds[2 * PBUF];
    __shared__ int s_next[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wmq = wave;                                    // this wave: block wmq of the tile's output channels, all four rows
    const int n31 = lane & 31, kh_ = lane >> 5;
    const unsigned lds0 = lds_addr(cp_lds);
    typedef __attribute__((address_space(3))) const bf16x8 lds_cbf8;
    int parity = 0;
#ifdef CP_TIME
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#endif
    for (int tile = blockIdx.x; tile < total_tiles;) {
        int nx = 0;
        if (tid == 0)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(nx) : "v"(0), "v"(1), "s"(tile_counter) : "memory");
        int t = tile;
        const int mt = t % d.mtiles; t /= d.mtiles;
        const int ct = t % d.ctiles; t /= d.ctiles;
        const int rt = t % d.rtiles; t /= d.rtiles;
        const int b = t % d.B;
        const int g = t / d.B;
        const int y0 = rt * CP_TR, x0 = ct * CP_TC;
        // ---- patch loader: instruction q = wave * NPW + k of a channel group brings 64 words of plane q / KPP; a lane's word
        //      (row, column) of the patch is recomputed per channel group (a few integer operations per 25 / 9 steps) rather than
        //      kept in registers ----
        const unsigned char* in_g = reinterpret_cast<const unsigned char*>(act) + (size_t)b * d.in_image +
                                    (size_t)g * d.Cg16 * 6 * d.Hp * d.Wp * 16;
        const size_t plane_bytes = (size_t)d.Hp * d.Wp * 16;
        auto issue_patch = [&](int cg, int buf) {
            const unsigned char* src = in_g + (size_t)cg * 6 * plane_bytes;
#pragma unroll
            for (int k = 0; k < NPW; ++k) {
                int q = wave * NPW + k;
                q = q < 6 * KPP ? q : 6 * KPP - 1;           // (surplus instructions repeat the last one)
                const int pl = q / KPP, kk = q - pl * KPP;   // wave-uniform
                int i = kk * 64 + lane;
                i = i < PRW ? i : PRW - 1;
                const int r = i / PC, c = i - r * PC;
                const int yy = y0 + r < d.Hp ? y0 + r : d.Hp - 1, xx = x0 + c < d.Wp ? x0 + c : d.Wp - 1;   // overhang: clamped (finite data)
                dma16((unsigned)(yy * d.Wp + xx) * 16u, src + (size_t)pl * plane_bytes,
                      __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * PBUF + pl * PLANE + kk * 1024)));
            }
        };
        // ---- weights: this wave's three fragments (pieces) of a K16 step, 16 bytes per lane each, straight from L2 into registers ----
        const unsigned char* wgrp = reinterpret_cast<const unsigned char*>(wfrag) +
                                    ((size_t)g * d.ksteps * d.mblocks + (size_t)mt * MB + wmq) * 3 * 1024;     // (wave-uniform)
        const size_t wstep = (size_t)d.mblocks * 3 * 1024;
        // B operand register sets.  With two, the next step's operands are requested BEFORE this step's matrix instructions: measured
        // (round 5) on the 2-block / three-workgroups-per-CU build, the only one with room for it: 1.36 ms against 1.29 -- not used
        constexpr int BSETS = 1;
        bf16x8 af[2][3], bfr[BSETS][NB][3];
#pragma unroll
        for (int q = 0; q < 6; ++q) af[q / 3][q % 3] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        auto load_a = [&](int s, auto o_tag) {
            constexpr int o = decltype(o_tag)::value;
            const unsigned char* src = wgrp + (size_t)s * wstep;
#pragma unroll
            for (int p = 0; p < 3; ++p) cp_gload16(af[o][p], src + p * 1024, (unsigned)lane * 16u);
        };
        const unsigned bbase = lds0 + (unsigned)(kh_ * PLANE + n31 * 16);
        int cg = 0, tap = 0, kh = 0, kw = 0;                 // position of the current step
        auto fetch_b = [&](auto set_tag, int cg_, int kh__, int kw__) {   // the B operands of the step at that position -> bfr[set]
            constexpr int st = decltype(set_tag)::value;
            const unsigned bt = bbase + (unsigned)((cg_ & 1) * PBUF + (kh__ * PC + kw__) * 16);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) bfr[st][j][p] = *(lds_cbf8*)(bt + (unsigned)(j * PC * 16 + p * 2 * PLANE));
        };
        f32x16 acc[NB], tq[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; tq[j][e] = 0.f; }
        const int S = d.ksteps;
        __builtin_amdgcn_s_barrier();                        // (every wave has left the previous tile's patches)
        issue_patch(0, 0);
        load_a(0, std::integral_constant<int, 0>());
        load_a(1, std::integral_constant<int, 1>());
        cp_wait<0>();
        if (tid == 0) s_next[parity] = nx + (int)gridDim.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // patch 0 is complete for every wave
        fetch_b(std::integral_constant<int, 0>(), 0, 0, 0);
        CP_LAP(7);
        int fold = 0;
        // One K16 step.  The waves of a workgroup are NOT in lockstep: weights come per wave from L2 (two steps ahead, two register
        // sets), the B operands from the shared patch (a step ahead), and the only barrier is at a channel group's first step --
        // while one wave of a SIMD waits for memory or LDS, the other one's matrix instructions keep the pipe busy.
        auto step = [&](auto o_tag, int s_) {
            constexpr int o = decltype(o_tag)::value;
            const bool group_start = tap == 0 && s_ > 0;
            if (group_start) {
                // own pieces of this group's patch (issued a whole group ago) and the weights of this step have landed; behind the
                // barrier every wave's have -- and every wave has finished reading the other buffer
                if (s_ + 1 < S) cp_wait_for<3>(af[o][0], af[o][1], af[o][2]); else cp_wait_for<0>(af[o][0], af[o][1], af[o][2]);
#ifndef CP_X_NOBAR
                __builtin_amdgcn_s_barrier();
#endif
#ifndef CP_X_NOFETCH
                fetch_b(std::integral_constant<int, BSETS == 2 ? o : 0>(), cg, 0, 0);
#endif
            } else {
                // the weights of this step (younger: the next step's three loads -- and the patch, when it was issued a step ago)
#if !defined(CP_X_NOA) && !defined(CP_X_NOPATCH)
                if (s_ + 1 >= S) cp_wait_for<0>(af[o][0], af[o][1], af[o][2]);
                else if (tap == 1 && cg + 1 < d.Cg16) cp_wait_for<3 + NPW>(af[o][0], af[o][1], af[o][2]);
                else cp_wait_for<3>(af[o][0], af[o][1], af[o][2]);
#endif
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            CP_LAP(4);
            // the next step's position
            int n_kw = kw + 1, n_kh = kh, n_tap = tap + 1, n_cg = cg;
            if (n_kw == d.KW) { n_kw = 0; ++n_kh; }
            if (n_tap == d.ntaps) { n_tap = 0; n_kh = 0; ++n_cg; }
#ifndef CP_X_NOFETCH
            if (BSETS == 2 && n_tap != 0 && s_ + 1 < S) fetch_b(std::integral_constant<int, BSETS == 2 ? (o ^ 1) : 0>(), n_cg, n_kh, n_kw);
#endif
            __builtin_amdgcn_sched_barrier(0);
            constexpr int bs = BSETS == 2 ? o : 0;
#ifndef CP_X_NOMFMA
            // six products per block into the block sums, the five small ones first
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][2], bfr[bs][j][0], tq[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][1], bfr[bs][j][1], tq[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][0], bfr[bs][j][2], tq[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][1], bfr[bs][j][0], tq[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][0], bfr[bs][j][1], tq[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NB; ++j) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][0], bfr[bs][j][0], tq[j], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            CP_LAP(0);
            // behind the matrix instructions (their operands are read at issue): the next group's patch at a group's first step
            // (every wave is past the barrier: nobody reads that buffer any more), the weights two steps ahead into the set just
            // used, the next step's B operands (unless that step starts a group: then after its barrier)
#ifndef CP_X_NOPATCH
            if (tap == 0 && cg + 1 < d.Cg16) issue_patch(cg + 1, (cg + 1) & 1);
#endif
#ifndef CP_X_NOA
            if (s_ + 2 < S) load_a(s_ + 2, o_tag);
#endif
            kw = n_kw; kh = n_kh; tap = n_tap; cg = n_cg;
#ifndef CP_X_NOFETCH
            if (BSETS == 1 && tap != 0 && s_ + 1 < S) fetch_b(std::integral_constant<int, 0>(), cg, kh, kw);
#endif
            CP_LAP(3);
#ifdef CP_X_NOFOLD
            if (s_ + 1 == S)
#else
            if (++fold == FP)
#endif
            {                                                // the block sums join the accumulators: ONE rounding per FP steps
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[j] += tq[j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) tq[j][e] = 0.f;
                }
                fold = 0;
            }
            CP_LAP(2);
        };
        int s2 = 0;
        for (; s2 + 1 < S; s2 += 2) {
            step(std::integral_constant<int, 0>(), s2);
            step(std::integral_constant<int, 1>(), s2 + 1);
        }
        if (s2 < S) step(std::integral_constant<int, 0>(), s2);
        // ---- epilogue: bias + ReLU -> f32 NCHW planes; accumulator register 4 q + e = row 8 q + 4 (lane / 32) + e of the block ----
        const int ow = x0 + n31;
        const int m_blk = (mt * MB + wmq) * 32;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int oh = y0 + j;
            if (oh >= d.OH || ow >= d.OW) continue;
            float* ocol = out + ((size_t)b * d.groups + g) * d.OC * d.OHp * d.OWp + (size_t)(oh + d.opad) * d.OWp + ow + d.opad;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m0 = __builtin_amdgcn_readfirstlane(m_blk + 8 * q);
                if (m0 >= d.OC) continue;
                const f32x4v bl = *reinterpret_cast<const f32x4v*>(bias + g * d.OC + m0 + 4 * kh_);   // this lane's 4 rows
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[j][4 * q + e] + bl[e];
                    if (d.relu) v = v > 0.f ? v : 0.f;
#ifdef CP_X_NOSTORE
                    if (v == 12345.678f)
#endif
                    ocol[(size_t)(m0 + 4 * kh_ + e) * d.OHp * d.OWp] = v;
                }
            }
        }
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
        CP_LAP(7);
    }
#ifdef CP_TIME
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 8; ++i) cp_dbg[(((KH == 5 ? 0 : 1) * 256 + blockIdx.x) * 8 + wave) * 8 + i] = tacc[i];
#endif
}

}  // namespace
#endif
