// cnn_conv_pieces.hpp -- conv2..conv5 of cnn/deploy.prototxt (:56-174) as DIRECT convolutions on the matrix cores with the f32
// operands cut into 16-bit pieces whose products are exact.  Included by vpk_cnn.hip (after cnn_split_gemm.hpp: bf16x8, split3, dma16).
//
// Arithmetic.  Two ways to cut an f32 number (template parameter NP):
//   NP = 3  three bf16 pieces, EXACT (8 + 8 + 8 significand bits); of the nine partial products the six with i + j <= 2 carry
//           everything above 2^-24 of the product -- six v_mfma_f32_32x32x16_bf16 per K16 step and 32 x 32 block;
//   NP = 2  two fp16 pieces of the number times a power of two (split2h below): 22 of 24 significand bits, THREE
//           v_mfma_f32_32x32x16_f16 per step -- the default (vpk_cnn_set_algorithm(4)), because the matrix cores are POWER-limited with
//           operands that change between instructions (1.7 PFLOP/s sustained, scripts/ubench/mfma_f16_pairs.hip, against the 2.5
//           dense peak): the six-product kernel ran at 83 % of that ceiling and could only get faster by executing less.
// Every product of two pieces is exact in f32; what is rounded is the accumulation, and that is organised in BLOCK SUMS: the
// products of FP consecutive K16 steps (a kernel row of conv2: 5 taps x 16 channels; all nine taps of a 3 x 3 layer) accumulate
// from zero -- per step the small products first -- and the block sum joins the running accumulator with one f32 addition.
// Error against the float64 net (scripts/cnn_accuracy.py, B = 3, relative to each blob's largest value): conv2 0.21e-6 (NP = 3) /
// 0.25e-6 (NP = 2); conv3..5 0.4-0.5e-6 (NP = 2); the f32-input direct kernels 1.0-2.0e-6, Winograd on the f32 cores 0.5-0.7e-6.
//
// Data movement:
//   * activations as piece planes: [image][channel group of 16][piece x k half = 2 NP][y][x] -> 16 bytes (8 values = the B operand of
//     one lane for one pixel), with the convolution's zero border.  Written by to_planes_kernel, by this kernel's epilogue for the
//     next convolution (conv3 -> conv4 -> conv5) and by cnn_norm_pool_planes.hpp (pool2 -> conv3).  A tile = 128 output channels
//     x 4 rows x 32 columns; per channel group its RAW input patch ((4 + KH - 1) rows x 40 columns x 2 NP planes) comes once by
//     LDS-DMA, a channel group ahead, spread over the group's blocks, and every tap reads it at a shifted address -- no im2col.
//     LDS layout of a plane: [column group of 8][row (8)][column] 16-byte words, groups 1152 bytes apart: a DMA instruction writes
//     8 rows x 8 columns with ONE lane-address register for the whole kernel, and the two or three groups a 16-lane quarter of a
//     ds_read_b128 touches fall on different banks (at 1024 bytes: 2-way conflicts, measured);
//   * weights: A fragments [group][K16 step][32-row block][piece][lane][8] (L2-resident) go per wave straight from L2 into
//     registers, D steps ahead -- the waves of a workgroup share nothing but the patch, the only barrier is at a channel group's
//     first step;
//   * four waves per workgroup (one 32-row block of the output channels each, four 32 x 32 accumulator blocks), two workgroups per
//     CU.
// Instruction issue (round 5, SQ counters in profiles/r05_pmc_conv2_issue.txt): the first version spent ~130 scalar / branch / vector
// instructions per step on bookkeeping beside 24 matrix instructions and kept the pipe 62 % busy whatever the memory schedule.
// Now a block of FP steps is straight-line code (tap offsets are immediates, waits constants, branches per block), a step goes
// row by row with the next step's operands requested behind each row, and the accumulators are touched only by inline asm
// (so that the register allocator cannot move them): ~25 other instructions per 12 (24) matrix instructions, 0 spills.
// Where it stands (B = 102, kernels alone): conv2 0.60-0.67 ms (was 1.20 on triples with the first loop, 1.46 Winograd), conv3 / 4 / 5
// 0.40 / 0.34 / 0.21 ms (Winograd: 0.73 / 0.55 / 0.37): 1.2-1.4 PFLOP/s executed = 72-85 % of the sustained ceiling; the rest is memory latency
// (timing ablations in DESIGN.md section 8: the patch DMA's completion is tied to the weight waits by the in-order vmcnt).
#ifndef VPK_CNN_CONV_PIECES_HPP_
#define VPK_CNN_CONV_PIECES_HPP_

namespace {

constexpr int CP_THREADS = 256;            // four waves; TWO workgroups per CU (one's prologue / epilogue under the other's products)
constexpr int CP_TC = 32;                       // columns of a tile (its rows: the kernel's NB)

struct PieceDims {
    int B;                                      // images
    int Cg16, CGtot, Hp, Wp;                    // input: channel groups (of 16) per conv group / in the tensor; padded plane
    int OC, OH, OW, groups;                     // OC = output channels per conv group
    int KW, ntaps, ksteps;                      // kernel width; KH * KW; K16 steps per conv group (Cg16 * ntaps)
    int mblocks;                                // 32-row blocks per conv group in the packed weights
    int mtiles, rtiles, ctiles;                 // tiles per (image, conv group): output-channel tiles, row tiles, column tiles
    int relu;
    int OHp, OWp, opad;                         // f32 NCHW output planes
    long long in_image;                         // bytes per image of the input tensor (planes of 16-byte words)
    int o_cgtot, o_Hp, o_Wp, o_pad;             // the NEXT layer's piece planes (written instead of the f32 planes when the kernel gets them)
    float o_ascale;                             //   and its activation scale
    unsigned* range_word;                       //   and where an activation beyond fp16's range is reported (split2h_guard), as bit
    unsigned range_bit;                         //   range_bit (the consuming layer's)
    float oscale;                               // 1 / (weight scale x activation scale) of the fp16-pair operands (a power of two); 1 for bf16 pieces
};

// fp16 PAIRS (round 5): an f32 operand x is h0 + h1 with h0 = fp16(x), h1 = fp16(x - h0): 22 significand bits of its 24 (the
// remainder is at most 2^-23 |x|), and of the four partial products h_i h_j (each EXACT in f32: 11 x 11 bits) the three with
// i + j <= 1 leave out h1 h1' -- at most 2^-22, typically 2^-24 of the product (|h1| <= 2^-11 |x|) -- HALF the matrix instructions of the
// bf16 triples, for a product that is within 8 x 2^-24 of exact in the worst case and 2^-24 in the median
// (tests/test_split_precision_math.py) where an f32 FMA chain rounds every partial sum to 2^-24 of ITS magnitude.  fp16's exponent range is what needs care, and powers of two take care of it (exactly):
//   * weights: x 2^k per layer, the largest in [2^13, 2^14): every weight down to 2^-17 of the largest keeps a normal second piece;
//   * activations: x a power of two per CONSUMING layer, fixed at load by a calibration forward (vpk_cnn.hip: calibrate): the layer's
//     input blob for a synthetic raster, computed by the f32 direct kernels, is brought to a maximum in [64, 128).  The pair misses
//     the scaled value x by at most max(2^-23 |x|, 2^-25) (tests/test_split_precision_math.py): full precision from 2^-8 to 2^+9 of
//     the calibration maximum; below, the second piece is a denormal -- the matrix cores multiply fp16 denormals, measured with
//     scripts/ubench/mfma_f16_pairs.hip -- and the ABSOLUTE error stays below 2^-25 / scale; 2^9 above it fp16 overflows (a net whose activations for real rasters are 500 x those of the
//     calibration raster; CP_DEFAULT_ASCALE is the uncalibrated fallback);
//   * the epilogue multiplies by the exact reciprocal of both.
// Whether a layer may use this is decided by MEASUREMENT against the float64 net (tests/test_gpu_cnn.py: no further from it than the
// f32 direct kernels, at every tap, also for nets whose blobs are 128 x larger / smaller than the synthetic net's).
// (CP_DEFAULT_ASCALE, split2h, split2h_guard, range_report: cnn_pairs.hpp -- conv1's pooling stage writes pairs too)

// f32 NCHW planes (with their zero border) -> piece planes: [image][channel group of 16][piece x k half (2 NP)][y][x] 16-byte words
// (8 values = the B operand of one lane for one pixel).  NP = 3: bf16 triples; NP = 2: fp16 pairs of ascale x.  One workgroup
// per (image, channel group, row): 16 channels x Wp values in, 2 NP x Wp words out.
template <int NP>
__global__ __launch_bounds__(256) void to_planes_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C, int Hp,
                                                        int Wp, float ascale, unsigned* __restrict__ range_word, unsigned range_bit) {
    const int y = blockIdx.x, cg = blockIdx.y, b = blockIdx.z;
    const float* src = in + (((size_t)b * C + cg * 16) * Hp + y) * Wp;
    u32x4* dst = reinterpret_cast<u32x4*>(out) + (((size_t)b * (C >> 4) + cg) * (2 * NP) * Hp + y) * Wp;
    bool bad = false;
    for (int idx = threadIdx.x; idx < 2 * Wp; idx += 256) {      // (k half, x)
        const int h = idx / Wp, x = idx - h * Wp;
        unsigned short p[3][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = src[(size_t)(8 * h + e) * Hp * Wp + x];
            if (NP == 3) split3(v, p[0][e], p[1][e], p[2][e]);
            else split2h_guard(v * ascale, p[0][e], p[1][e], bad);
        }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            u32x4 w4;
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = (unsigned)p[q][2 * e] | ((unsigned)p[q][2 * e + 1] << 16);
            dst[(size_t)(2 * q + h) * Hp * Wp + x] = w4;
        }
    }
    if (NP == 2) range_report(bad, range_word, range_bit);
}

template <int N>
__device__ __forceinline__ void cp_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);           // (nothing that reads a loaded register moves above the wait)
}

// A 16-byte global load the COMPILER DOES NOT KNOW TO BE A LOAD (round 5).  hipcc keeps a scoreboard of the vector-memory operations it
// has emitted and puts its own s_waitcnt in front of the first use of a loaded register; with LDS-DMA issued from inline asm in between
// (which it cannot count) it falls back to `s_waitcnt vmcnt(0)` -- in front of every K16 step's first matrix instruction in the first
// version, which turned "weights two steps ahead" into "everything in flight must land now".  Issued from asm, completion is counted
// by hand (cp_wait: a bare s_waitcnt and a scheduling barrier; a wait that names the registers as "+v" operands makes hipcc COPY them
// into the asm's operand registers BEFORE the wait -- stale data).  The destination is a "+v" operand -- the load overwrites the
// register the variable already lives in -- so that the compiler has no fresh value to copy into place either.
template <int OFF>
__device__ __forceinline__ void cp_gload16(bf16x8& dst, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "+v"(dst) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}

// KH: kernel size (5: conv2; 3 is kept for measurements on the 3 x 3 layers).  A tile = 128 output channels x (NB = 4 rows x 32
// columns).  Four waves, one per 32-row block of the output channels, each all 4 x 32 pixels (four 32 x 32 blocks).
//
// THE LOOP IS WRITTEN FOR THE INSTRUCTION ISSUE, not only for the memory system (round 5, the counters of profiles/r05_pmc_conv2.txt):
// the first version spent 2.5 scalar instructions, 0.45 branches and 1.2 other vector instructions per matrix instruction on
// per-step bookkeeping (tap position, conditional waits, 64-bit addresses) -- ~130 instructions per step and wave beside its 24
// matrix instructions, as long to issue as those take to execute -- and the two waves of a SIMD do not hide that for each other:
// the arbiter alternates between them while both have matrix work, they finish a step together and then do their bookkeeping
// together (pipe 62 % busy, whatever the memory schedule).  Here a BLOCK of FP steps (a kernel row of conv2; all nine taps of
// a 3 x 3 layer) is straight-line code: tap offsets are immediates of the LDS reads, the weight pieces immediates of one
// scalar pointer, waits are constants, and the only branches are per block.
// RH = 1: the four waves own four 32-channel blocks x the tile's 4 rows (128 channels x 4 rows x 32 columns).  RH = 2: two blocks x two
// row halves (64 channels x 8 rows x 32 columns) -- conv4's: its 192 channels per group are three such tiles (as 128-channel tiles they
// were two, a quarter of the second one padding: 0.36 -> 0.32 ms); conv3 and conv5 measured 6-9 % SLOWER in this shape and keep RH = 1.
template <int KH, int NB, int NP, int RH>
__global__ __launch_bounds__(CP_THREADS, 2) void conv_pieces_kernel(PieceDims d, const unsigned short* __restrict__ act,
                                                                     const unsigned short* __restrict__ wfrag,
                                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                                     unsigned short* __restrict__ out_planes,
                                                                     int* __restrict__ tile_counter, int total_tiles) {
    constexpr int MB = 4 / RH;                               // 32-row blocks per tile
    constexpr int TR = RH * NB;                              // rows of a tile
    constexpr int PR = TR + KH - 1, PC = CP_TC + KH - 1;     // patch rows / columns
    constexpr int KPP = (PC + 7) / 8;                        // column groups of 8 per plane
    constexpr int PARTS = PR > 8 ? 2 : 1;                    // DMA instructions per column group: 8 rows x 8 columns each (the second: PR - 8 rows)
    constexpr int GST = (PR > 8 ? PR * 128 : 1024) + 128;    // bytes between column groups: the rows x 128 bytes, + 128 so that the two or three
                                                             // groups a 16-lane quarter of a ds_read_b128 touches fall on different banks
    static_assert(GST % 256 == 128, "odd multiple of 128 bytes");
    constexpr int PLANE = KPP * GST;                         // bytes per plane in LDS: [column group of 8][row (8)][column] 16-byte words
    constexpr int NPL = 2 * NP;                              // planes per channel group: piece x k half
    constexpr int PBUF = NPL * PLANE;                        // bytes per patch buffer
    constexpr int NDMA = NPL * KPP * PARTS;                  // patch DMA instructions per channel group
    constexpr int NPW = (NDMA + 3) / 4;                      //   ... and wave
    constexpr int NPR = NP == 3 ? 6 : 3;                     // products per K16 step and 32 x 32 block
    // weight register sets = how many steps ahead the weights are requested.  fp16 pairs (8 registers a set): one per step of a
    // kernel row (5) / three for the nine taps -- the set is a function of the step's place in the block, every block is the same
    // code, and a request has 5 (3) steps to come back from L2 (with two sets the pairs' steps, half as long as the triples',
    // waited for their weights 29 % of the time: SQ_WAIT_ANY).  bf16 triples (12 registers a set): two sets alternating per step,
    // two blocks make a period when FP is odd.  Round 6 measured SIX sets for the 3 x 3 layers (a request then has six steps, and a patch
    // DMA instruction issued behind a block's first step seven, before the in-order vmcnt makes a weight wait a wait for it -- DESIGN.md
    // section 8's lead): conv3 0.413 -> 0.413 ms, conv5 0.217 -> 0.220, conv4 (RH = 2: ten patch rows, two DMA instructions per column
    // group) 0.348 -> 0.335, same box.  Kept where it pays: six sets for RH = 2 (two blocks then make a period: 9 steps, sets 0..5, 0..2 |
    // 3..5, 0..5), three elsewhere.  So the unhidden latency of these kernels is not the DMA's place in the vmcnt queue.
    constexpr int NSETS = NP == 3 ? 2 : (KH == 5 ? 5 : (RH == 2 ? 6 : 3));
    constexpr int D = NSETS;                                 // steps between a weight request and its use
    constexpr int KHB = KH == 5 ? 5 : 1;                     // blocks per channel group
    constexpr int FP = KH == 5 ? 5 : 9;                      // K16 steps per block = per block sum (a kernel row of conv2, all taps of a 3 x 3 layer)
    constexpr int NPWB = (NPW + KHB - 1) / KHB;              // patch DMA instructions per wave and block
    static_assert(PR <= 16 && NB == 4 && (RH == 1 || RH == 2) && KH * KH == KHB * FP && (NP == 2 || NP == 3), "8 row slots per DMA instruction; two halves");
    __shared__ __attribute__((aligned(16))) unsigned char cp_lds[2 * PBUF];
    __shared__ int s_next[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wmq = RH == 1 ? wave : (wave & 1);             // this wave: block wmq of the tile's output channels,
    const int rh = RH == 1 ? 0 : (wave >> 1);                //            rows rh * NB .. + NB - 1 of the tile
    const int n31 = lane & 31, kh_ = lane >> 5;
    const unsigned lds0 = lds_addr(cp_lds);
    typedef __attribute__((address_space(3))) const bf16x8 lds_cbf8;
    // a lane's B operand for pixel column n31 + kw: word ((column / 8) * 64 + row * 8 + column % 8) of plane (piece, k half = lane / 32)
    unsigned vb[KH];
#pragma unroll
    for (int kw = 0; kw < KH; ++kw) {
        const unsigned c = (unsigned)(n31 + kw);
        vb[kw] = lds0 + (unsigned)(kh_ * PLANE + rh * NB * 128) + (c >> 3) * (unsigned)GST + (c & 7u) * 16u;
    }
    const unsigned voff_a = (unsigned)lane * 16u;
    int parity = 0;
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
        const int y0 = rt * TR, x0 = ct * CP_TC;
        // ---- patch loader: a DMA instruction brings 8 rows x 8 columns of one plane (lane = row * 8 + column); the lane part of
        //      its address is the same for all of them (rows past the plane repeat the last one; columns past the row read on into
        //      the next row or plane -- finite data for output pixels that are never stored) ----
        const unsigned char* in_g = reinterpret_cast<const unsigned char*>(act) + (size_t)b * d.in_image +
                                    (size_t)g * d.Cg16 * NPL * d.Hp * d.Wp * 16;
        const size_t plane_bytes = (size_t)d.Hp * d.Wp * 16;
        const int yr = y0 + (lane >> 3) < d.Hp ? y0 + (lane >> 3) : d.Hp - 1;
        const unsigned voff_p = (unsigned)(yr * d.Wp + x0 + (lane & 7)) * 16u;
        const int yr2 = y0 + 8 + (lane >> 3) < d.Hp ? y0 + 8 + (lane >> 3) : d.Hp - 1;       // (PARTS = 2: rows 8 .. PR - 1, lanes 0 .. 8 (PR - 8) - 1)
        const unsigned voff_p2 = (unsigned)(yr2 * d.Wp + x0 + (lane & 7)) * 16u;
        auto issue_dma = [&](int cg, int k) __attribute__((always_inline)) {   // this wave's k-th instruction of channel group cg
            int q = wave * NPW + k;
            q = q < NDMA ? q : NDMA - 1;                     // (surplus instructions repeat the last one)
            const int part = q / (NPL * KPP), q1 = q - part * (NPL * KPP);
            const int pl = q1 / KPP, kk = q1 - pl * KPP;     // wave-uniform
            const unsigned char* src = in_g + ((size_t)cg * NPL + pl) * plane_bytes + (size_t)kk * 128;
            const unsigned dst = lds0 + (unsigned)((cg & 1) * PBUF + pl * PLANE + kk * GST);
            if (PARTS == 1 || part == 0) dma16(voff_p, src, __builtin_amdgcn_readfirstlane(dst));
            else if (lane < 8 * (PR - 8)) dma16(voff_p2, src, __builtin_amdgcn_readfirstlane(dst + 1024u));   // (masked lanes write nothing)
        };
        // ---- weights: this wave's three fragments (pieces) of a K16 step, 16 bytes per lane each, straight from L2 into registers ----
        const unsigned char* wgrp = reinterpret_cast<const unsigned char*>(wfrag) +
                                    ((size_t)g * d.ksteps * d.mblocks + (size_t)mt * MB + wmq) * NP * 1024;    // (wave-uniform)
        const unsigned wstep = (unsigned)d.mblocks * NP * 1024;
        bf16x8 af[NSETS][NP], bfr[NB][NP];
#pragma unroll
        for (int q = 0; q < NSETS * NP; ++q) af[q / NP][q % NP] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        auto load_a = [&](const unsigned char* src, auto o_tag) __attribute__((always_inline)) {
            constexpr int o = decltype(o_tag)::value;
            cp_gload16<0>(af[o][0], src, voff_a);
            cp_gload16<1024>(af[o][1], src, voff_a);
            if constexpr (NP == 3) cp_gload16<2048>(af[o][2], src, voff_a);
        };
        // B operands of tile row j for tap k of a block whose scalar LDS offset is sb (buffer, kernel row)
        auto fetch_b = [&](auto row_tag, auto k_tag, unsigned va) __attribute__((always_inline)) {   // va = tap_address(k, sb)
            constexpr int j = decltype(row_tag)::value, k = decltype(k_tag)::value;
            constexpr int dkh = KH == 5 ? 0 : k / KH;
#pragma unroll
            for (int p = 0; p < NP; ++p) bfr[j][p] = *(lds_cbf8*)(va + (unsigned)((j + dkh) * 128 + p * 2 * PLANE));
        };
        auto tap_address = [&](auto k_tag, unsigned sb) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            return vb[KH == 5 ? k : k % KH] + sb;
        };
        auto fetch_all = [&](auto k_tag, unsigned sb) __attribute__((always_inline)) {
            const unsigned va = tap_address(k_tag, sb);
            fetch_b(std::integral_constant<int, 0>(), k_tag, va); fetch_b(std::integral_constant<int, 1>(), k_tag, va);
            fetch_b(std::integral_constant<int, 2>(), k_tag, va); fetch_b(std::integral_constant<int, 3>(), k_tag, va);
        };
        // The accumulators are register PAIRS that only inline asm adds to (cp_fold): left to the compiler, the sum of accumulator and
        // block sum lands in the block sum's registers, the accumulators wander between two homes over the unrolled blocks and
        // ~30 of them are spilled (and every scratch reload is a `s_waitcnt vmcnt(0)` on the weights in flight).
        f32x2v acc[NB][8];
        f32x16 tq[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = f32x2v{0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 16; ++e) tq[j][e] = 0.f;
        }
        const int S = d.ksteps;
        __builtin_amdgcn_s_barrier();                        // (every wave has left the previous tile's patches)
#pragma unroll
        for (int k = 0; k < NPW; ++k) issue_dma(0, k);
        asm volatile("s_nop 4" ::: "memory");                // (the weight pointer may come from v_readfirstlane: VALU-writes-SGPR -> VMEM)
        load_a(wgrp, std::integral_constant<int, 0>());
        load_a(wgrp + wstep, std::integral_constant<int, 1>());
        if constexpr (NSETS > 2) load_a(wgrp + 2 * wstep, std::integral_constant<int, 2>());
        if constexpr (NSETS > 3) { load_a(wgrp + 3 * wstep, std::integral_constant<int, 3>()); load_a(wgrp + 4 * wstep, std::integral_constant<int, 4>()); }
        if constexpr (NSETS > 5) load_a(wgrp + 5 * wstep, std::integral_constant<int, 5>());
        const unsigned char* anext = wgrp + (size_t)D * wstep;   // the next weights to request (step s + D behind step s; none behind the last D)
        cp_wait<0>();
        if (tid == 0) s_next[parity] = nx + (int)gridDim.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                        // patch 0 is complete for every wave
        fetch_all(std::integral_constant<int, 0>(), 0u);
        // The matrix instructions of one tile row (the products of a K16 step, the small ones first) into its block sum.  FIRST: the step
        // opens a block -- the first product starts from zero (C = 0: no register clearing).  (A row's instructions form a dependent
        // chain; the matrix pipe forwards the accumulator, scripts/ubench/mfma_bf16_chain.hip.)
        auto products = [&](auto o_tag, auto row_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr int o = decltype(o_tag)::value, j = decltype(row_tag)::value;
            constexpr bool first = decltype(first_tag)::value;
            constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0}, PB[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0};
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < NPR; ++p) {
                if constexpr (NP == 3)
                    tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[o][PA[p]], bfr[j][PB[p]], first && p == 0 ? zero : tq[j], 0, 0, 0);
                else
                    tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[o][PA[p]]), __builtin_bit_cast(f16x8, bfr[j][PB[p]]),
                                                                   first && p == 0 ? zero : tq[j], 0, 0, 0);
            }
        };
        // a row's finished block sum joins its accumulators: ONE rounding per FP steps (asm: see the accumulators' comment; the
        // matrix-instruction result -> VALU read hazard is not counted by the compiler for asm: s_nop; where the fold of row j - 1
        // follows the products of row j the result is ~100 cycles old anyway)
        auto fold_row = [&](auto row_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(row_tag)::value;
            asm volatile("s_nop 7" ::: "memory");
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const f32x2v t2 = {tq[j][2 * e], tq[j][2 * e + 1]};
                f32x2v& a2 = acc[j][e];                      // (a plain use: an asm operand alone does not make the lambda capture `acc`)
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2) : "v"(t2));
            }
        };
        int kb = 0, cg = 0;                                  // block of the channel group (kernel row of conv2); channel group
        // One K16 step (tap k of the block), ROW BY ROW: [a row's matrix instructions][the same row's B operands of the NEXT step] --
        // the row's operand registers are free once its instructions have issued (operands are read at issue), the requests go out
        // in the shadow of the row's last matrix instruction and have three rows' instructions to land: no second register set, and
        // almost nothing of a step's bookkeeping is issued while the matrix pipe has nothing of this wave's to do.  The waves of a
        // workgroup are NOT in lockstep: weights come per wave from L2 (two steps ahead, two register sets), the B operands from
        // the shared patch, and the only barrier is at a channel group's first step.  In a block's last step a row's fold follows the
        // NEXT row's matrix instructions (VALU beside the matrix pipe).
        // Vector-memory queue (in order): at the top of a step [weights s][weights s + 1]; behind step 0 of a block with patch work
        // [weights s + 2][NPWB patch instructions] join it -- steps 1 and 2 leave those in flight, step 3 waits for them.
        // TAIL: the tile's last block -- nothing is requested behind its last two steps (a load in flight into registers the
        // compiler considers dead would land in whatever the epilogue keeps there), and the last step's weights are the queue's last.
        auto step = [&](auto o_tag, auto k_tag, bool gs, bool dma_block, bool dma_prev, bool more_blocks, bool tail, unsigned sb) __attribute__((always_inline)) {
            constexpr int k = decltype(k_tag)::value;
            constexpr bool last = k + 1 == FP;
            // this step's weights have landed; younger in the queue: the next D - 1 steps' (fewer at the tile's end), and the patch
            // instructions issued behind step 0 of this block (of the previous one at step 0 when D = FP) for D steps
            constexpr int AHEAD = NP * (D - 1), AHEAD_TAIL = NP * (FP - 1 - k < D - 1 ? FP - 1 - k : D - 1);
            // (not at a channel group's first step: the patch it is about to read includes the previous block's instructions)
            const bool dma_out = (k >= 1 && k <= D) ? dma_block : (k == 0 && D == FP ? dma_prev && !gs : false);
            if (tail) cp_wait<AHEAD_TAIL>();
            else if (dma_out) cp_wait<AHEAD + NPWB>();
            else cp_wait<AHEAD>();
            if (k == 0 && gs) {
                // own pieces of this group's patch (issued at least a block ago) have landed; behind the barrier every wave's have --
                // and every wave has finished reading the other buffer
                __builtin_amdgcn_s_barrier();
                fetch_all(k_tag, sb);
            }
            const unsigned va_next = k + 1 < FP ? tap_address(std::integral_constant<int, (k + 1 < FP ? k + 1 : 0)>(), sb)
                                                : tap_address(std::integral_constant<int, 0>(), sb + 128u);
#define CP_ROW(J)                                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            products(o_tag, std::integral_constant<int, J>(), std::integral_constant<bool, k == 0>());                     \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            if constexpr (k + 1 < FP) fetch_b(std::integral_constant<int, J>(), std::integral_constant<int, k + 1>(), va_next); \
            else if (more_blocks) fetch_b(std::integral_constant<int, J>(), std::integral_constant<int, 0>(), va_next);    \
            if constexpr (last && J > 0) fold_row(std::integral_constant<int, (J > 0 ? J - 1 : 0)>());                     \
            __builtin_amdgcn_sched_barrier(0);
            CP_ROW(0) CP_ROW(1) CP_ROW(2) CP_ROW(3)
#undef CP_ROW
            if (k < FP - D || !tail) { load_a(anext, o_tag); anext += wstep; }   // into the set just used
            if (k == 0 && dma_block) {
#pragma unroll
                for (int i = 0; i < NPWB; ++i) issue_dma(cg + 1, kb * NPWB + i < NPW ? kb * NPWB + i : NPW - 1);
            }
            if constexpr (last) { asm volatile("s_nop 15" ::: "memory"); fold_row(std::integral_constant<int, NB - 1>()); }
            __builtin_amdgcn_sched_barrier(0);
        };
        // FP steps make a block, written out.  fp16 pairs: every block is the same code (set = step % NSETS, NSETS divides FP);
        // bf16 triples: the set alternates per step, so two blocks make a period when FP is odd.
        bool dma_prev = false;
        auto block = [&](auto a_tag) __attribute__((always_inline)) {
            constexpr int a = decltype(a_tag)::value;
            const bool gs = kb == 0 && cg > 0;
            const bool dma_block = cg + 1 < d.Cg16 && kb * NPWB < NPW;
            const bool more = kb + 1 < KHB;                  // (the next block continues this channel group: its first operands can be requested)
            const unsigned sb = (unsigned)((cg & 1) * PBUF + kb * 128);
            const bool tail = !more && cg + 1 == d.Cg16;
            static_assert(FP == 5 || FP == 9, "the steps of a block are written out for 5 and 9");
            static_assert(NP == 3 || (2 * FP) % NSETS == 0, "two blocks must be a whole number of set rotations");
#define CP_STEP(n) step(std::integral_constant<int, (NP == 3 ? (a + n) & 1 : (a + n) % NSETS)>(), std::integral_constant<int, n>(), gs, dma_block, dma_prev, more, tail, sb);
            CP_STEP(0) CP_STEP(1) CP_STEP(2) CP_STEP(3) CP_STEP(4)
            if constexpr (FP == 9) { CP_STEP(5) CP_STEP(6) CP_STEP(7) CP_STEP(8) }
#undef CP_STEP
            dma_prev = dma_block;
            if (++kb == KHB) { kb = 0; ++cg; }
        };
        const int nblocks = S / FP;
        if constexpr (NP == 3) {
            int bi = 0;
            for (; bi + 2 <= nblocks; bi += 2) {
                block(std::integral_constant<int, 0>());
                block(std::integral_constant<int, FP & 1>());
            }
            if (bi < nblocks) block(std::integral_constant<int, 0>());
        } else if constexpr (FP % NSETS == 0) {
            for (int bi = 0; bi < nblocks; ++bi) block(std::integral_constant<int, 0>());
        } else {                                             // (six sets, nine steps: two blocks make a period)
            int bi = 0;
            for (; bi + 2 <= nblocks; bi += 2) {
                block(std::integral_constant<int, 0>());
                block(std::integral_constant<int, FP % NSETS>());
            }
            if (bi < nblocks) block(std::integral_constant<int, 0>());
        }
        // ---- epilogue: bias + ReLU -> f32 NCHW planes; accumulator register 4 q + e = row 8 q + 4 (lane / 32) + e of the block.
        //      The tile's coordinates are derived AGAIN from the (laundered) tile index and lane offset: kept from the top of the
        //      tile they would be a dozen registers live through the loop, which has none to spare ----
        {
            int te = tile;
            unsigned le = voff_a;
            asm volatile("" : "+s"(te), "+v"(le));
            const int e_mt = te % d.mtiles; te /= d.mtiles;
            const int e_ct = te % d.ctiles; te /= d.ctiles;
            const int e_rt = te % d.rtiles; te /= d.rtiles;
            const int e_b = te % d.B, e_g = te / d.B;
            const int e_n31 = (int)(le >> 4) & 31, e_kh = (int)(le >> 9);
            const int ow = e_ct * CP_TC + e_n31;
            const int m_blk = (e_mt * MB + wmq) * 32;
            // one scalar base per stored plane, one 32-bit lane offset per row of the tile (64 stores per lane: address arithmetic per
            // store -- 64-bit multiplies in the first version -- was a tenth of the kernel's VALU time)
            const unsigned plane = (unsigned)(d.OHp * d.OWp);
            const unsigned char* obase = reinterpret_cast<const unsigned char*>(out + ((size_t)e_b * d.groups + e_g) * d.OC * plane);
            const int oh0 = e_rt * TR + rh * NB;             // this wave's first row of the blob
            const unsigned vo0 = ((unsigned)(4 * e_kh) * plane + (unsigned)((oh0 + d.opad) * d.OWp + ow + d.opad)) * 4u;
            f32x4v bl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                    // this lane's 4 rows of each 8-row slice (the packed weights' padding rows: clamped)
                const int m0 = m_blk + 8 * q < d.OC ? m_blk + 8 * q : d.OC - 8;
                bl[q] = *reinterpret_cast<const f32x4v*>(bias + e_g * d.OC + m0 + 4 * e_kh);
            }
            if (NP == 2 && out_planes) {
                // the NEXT layer's input: fp16 pairs of o_ascale x (ReLU(result)) in its piece planes (interior only: the border is the
                // arena's zeros).  A lane holds 4 consecutive channels of an 8-channel word, lane + 32 the other 4: 8-byte stores that
                // the two halves of the wave complete to whole words.
                unsigned char* pbase = reinterpret_cast<unsigned char*>(out_planes) + (size_t)e_b * d.o_cgtot * 4 * d.o_Hp * d.o_Wp * 16;
                bool bad = false;
                const unsigned wplane = (unsigned)(d.o_Hp * d.o_Wp) * 16u;
                const unsigned vo_p = (unsigned)((oh0 + d.o_pad) * d.o_Wp + ow + d.o_pad) * 16u + 8u * (unsigned)e_kh;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    if (ow >= d.OW || oh0 + j >= d.OH) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m0 = __builtin_amdgcn_readfirstlane(m_blk + 8 * q);
                        if (m0 >= d.OC) continue;
                        const int c0 = e_g * d.OC + m0;                                  // first channel of the 8-channel word (wave-uniform)
                        unsigned short h0[4], h1[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[j][2 * q + e / 2][e % 2] * d.oscale + bl[q][e];
                            if (d.relu) v = v > 0.f ? v : 0.f;
                            split2h_guard(v * d.o_ascale, h0[e], h1[e], bad);
                        }
                        const unsigned w0 = ((unsigned)(c0 >> 4) * 4u + (unsigned)((c0 >> 3) & 1)) * wplane + vo_p + (unsigned)(j * d.o_Wp) * 16u;
                        *reinterpret_cast<u32x2*>(pbase + w0) = u32x2{(unsigned)h0[0] | ((unsigned)h0[1] << 16), (unsigned)h0[2] | ((unsigned)h0[3] << 16)};
                        *reinterpret_cast<u32x2*>(pbase + (w0 + 2u * wplane)) = u32x2{(unsigned)h1[0] | ((unsigned)h1[1] << 16), (unsigned)h1[2] | ((unsigned)h1[3] << 16)};
                    }
                }
                range_report(bad, d.range_word, d.range_bit);
            } else
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (ow >= d.OW || oh0 + j >= d.OH) continue;       // (one exec-mask region per row of the tile)
                const unsigned vo = vo0 + (unsigned)(j * d.OWp) * 4u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m0 = __builtin_amdgcn_readfirstlane(m_blk + 8 * q);
                    if (m0 >= d.OC) continue;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[j][2 * q + e / 2][e % 2] * d.oscale + bl[q][e];
                        if (d.relu) v = v > 0.f ? v : 0.f;
                        *reinterpret_cast<float*>(const_cast<unsigned char*>(obase) + (vo + (unsigned)(m0 + e) * plane * 4u)) = v;
                    }
                }
            }
        }
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
}

}  // namespace
#endif
