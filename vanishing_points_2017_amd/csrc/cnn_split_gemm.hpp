// cnn_split_gemm.hpp -- f32-accurate implicit-GEMM convolution on the bf16 matrix cores (included by vpk_cnn.hip).
//
// gfx950 has no TF32-like mode: f32-input MFMA runs at 1/16 of the bf16 rate.  An f32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits: x1 = bf16(x), x2 = bf16(x - x1), x3 = x - x1 - x2, rounding to nearest), a
// product of two bf16 numbers is exact in f32, and the bf16 MFMA accumulates in f32.  So
//     a * b = sum_{i,j} a_i b_j      (9 exact partial products)
// and the six products with i + j <= 4 carry everything but a_2 b_3 + a_3 b_2 + a_3 b_3 <= 2^-24 |a b| -- the size of the
// single rounding an f32 FMA makes anyway (tests/test_split_precision_math.py).  Six bf16 MFMAs per f32 MFMA-equivalent
// = 6/16 of the matrix-pipe time of the native f32 path.  (cuBLAS ships the same idea as "BF16x9" FP32 emulation; "x6" drops the terms that are
// below the accumulator's own rounding.)  tests/test_gpu_cnn.py compares both paths with an fp64 oracle: their errors
// are of the same size.
//
// Data layout (chosen for the matrix cores, not inherited from the f32 path):
//   activations  [image][y][x][channel / 16][piece 0..2][16 channels] bf16, with the convolution's zero border
//                (split_nhwc_kernel writes them from the f32 NCHW planes); K steps = (channel group, kh, kw), so a K16 step = 16
//                consecutive channels of one tap = 96 contiguous bytes per pixel (all three pieces): the gather touches
//                1.5 cache lines per pixel instead of 3 half-used ones;
//   weights      pre-split and pre-permuted on the host into MFMA fragment order:
//                [group][k16 step][32-row block][piece][lane][8 bf16]  (1 KB = one A operand of v_mfma_f32_32x32x16_bf16);
//   LDS stage    a list of 1 KB fragments, A blocks first then B blocks, each exactly as the MFMA wants it in registers:
//                an A fragment is written by ONE global_load_lds_dwordx4 (lane l brings the 16 bytes lane l will later read
//                back with one conflict-free ds_read_b128); a block of 32 columns arrives as [column][piece][k half] (3 KB,
//                three DMA instructions of 64 consecutive 16-byte chunks) and is read back with a 96-byte lane stride.
// Tile = (WAVES_M * TM * 32) x 256 outputs: WAVES_M x WAVES_N waves (8 with three stages in flight and one workgroup per
// CU, or 4 with two stages and two workgroups per CU), each wave TM x (8 / WAVES_N) blocks of 32 x 32.  K16 steps run over
// the taps first, then over the channel groups.
#ifndef VPK_CNN_SPLIT_GEMM_HPP_
#define VPK_CNN_SPLIT_GEMM_HPP_

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct SplitDims {
    int B, Cg, Ctot, Hp, Wp;        // input: channels per group / in total; padded plane
    int OC, OH, OW, groups;         // OC = output channels per group
    int KW, ntaps, csteps, ksteps;  // kernel width; KH * KW; K16 steps per tap (Cg / 16); K16 steps in total (ntaps * csteps)
    int mblocks;                    // 32-row blocks per group in the packed weights
    int N;                          // B * OH * OW
    int relu;
    int OHp, OWp, opad;             // f32 NCHW output planes
};

constexpr int SG_BN = 256;          // columns per tile: 8 blocks of 32

// f32 -> three bf16 pieces, each the round-to-nearest-even bf16 of what is left: x - p0 and (x - p0) - p1 are exact in
// f32 and the last remainder has at most 8 significant bits, so p0 + p1 + p2 == x exactly.  Rounding (not truncating)
// keeps |p1| <= 2^-8 |x| and |p2| <= 2^-16 |x|, which bounds the dropped products at 2^-24 |a b|.
__device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    const unsigned b = __float_as_uint(x);
    return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    const unsigned b0 = bf16_rne_bits(x);
    const float r1 = x - __uint_as_float(b0);
    const unsigned b1 = bf16_rne_bits(r1);
    const float r2 = r1 - __uint_as_float(b1);
    p0 = (unsigned short)(b0 >> 16);
    p1 = (unsigned short)(b1 >> 16);
    p2 = (unsigned short)(__float_as_uint(r2) >> 16);
}

// f32 NCHW planes (with their zero border) -> three bf16 NHWC pieces.  One workgroup per (image, row): the row's C x Wp
// values are transposed through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void split_nhwc_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C,
                                                         int Hp, int Wp) {
    extern __shared__ float sn_tile[];                   // [C][Wp + 1]
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ld = Wp + 1;
    for (int idx = tid; idx < C * Wp; idx += 256) {
        const int c = idx / Wp, x = idx - c * Wp;
        sn_tile[c * ld + x] = in[(((size_t)b * C + c) * Hp + y) * Wp + x];
    }
    __syncthreads();
    const int C2 = C >> 1;
    unsigned* o = reinterpret_cast<unsigned*>(out + ((size_t)b * Hp + y) * Wp * C * 3);
    for (int idx = tid; idx < Wp * C2; idx += 256) {
        const int x = idx / C2, c = (idx - x * C2) * 2;
        unsigned short a0, a1, a2, b0, b1, b2;
        split3(sn_tile[c * ld + x], a0, a1, a2);
        split3(sn_tile[(c + 1) * ld + x], b0, b1, b2);
        const int base = (((x * (C >> 4) + (c >> 4)) * 3) * 16 + (c & 15)) >> 1;     // in 4-byte words
        o[base] = (unsigned)a0 | ((unsigned)b0 << 16);
        o[base + 8] = (unsigned)a1 | ((unsigned)b1 << 16);
        o[base + 16] = (unsigned)a2 | ((unsigned)b2 << 16);
    }
}

// -DSG_TIME: s_memtime around the phases of a stage, read back by scripts/split_gemm_phase_times.py (development only)
#ifdef SG_TIME
__device__ long long sg_dbg[256 * 8 * 8];
#define SG_T(i) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; }
#else
#define SG_T(i)
#endif
// SPLIT_OUT: the result goes straight into the NEXT convolution's input format (three bf16 pieces, channels innermost)
// instead of f32 NCHW planes.
template <int WAVES_M, int WAVES_N, int TM, int NST, int WPE, bool SPLIT_OUT = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WPE) void conv_gemm_split_kernel(SplitDims d, const unsigned short* __restrict__ act,
                                                                        const unsigned short* __restrict__ wfrag,
                                                                        const float* __restrict__ bias, void* __restrict__ out_,
                                                                        int* __restrict__ tile_counter, int total_tiles) {
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int TN = SG_BN / 32 / WAVES_N;
    constexpr int MB = WAVES_M * TM;                     // 32-row blocks per tile
    constexpr int NB = SG_BN / 32;                       // 32-column blocks per tile
    constexpr int NA = MB * 3, NBF = NB * 3;             // fragments per stage
    constexpr int STAGE_BYTES = (NA + NBF) * 1024;
    static_assert(NST * STAGE_BYTES <= 160 * 1024 - 64, "the stages must fit the CU's LDS");
    constexpr int AHEAD = NST - 1;
    __shared__ __attribute__((aligned(16))) unsigned char sg_lds[NST * STAGE_BYTES];
    __shared__ int s_next[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const unsigned lds0 = lds_addr(sg_lds);
    const int mtiles = d.mblocks / MB;
    const int ntiles = (d.N + SG_BN - 1) / SG_BN;
    const int ohw = d.OH * d.OW;
    // A fragments of a stage are dealt round-robin to the waves (fragment f = block f / 3, piece f % 3, to wave f % NWAVES);
    // every wave brings NB / NWAVES column blocks
    constexpr int CB = NB / NWAVES;                      // column blocks per wave
    const int na_mine = (NA - wave + NWAVES - 1) / NWAVES;   // wave-uniform
    int parity = 0;
#ifdef SG_TIME
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int tile = blockIdx.x; tile < total_tiles;) {
        int nx = 0;
        if (tid == 0)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(nx) : "v"(0), "v"(1), "s"(tile_counter) : "memory");
        int bid = tile;
        const int mt = bid % mtiles; bid /= mtiles;
        const int nt = bid % ntiles;
        const int g = bid / ntiles;
        SG_T(6)
        // ---- B gather: this wave brings column block `wave` (32 columns x 96 bytes per stage) as 192 chunks of 16 bytes;
        //      chunk index = column * 6 + piece * 2 + k half, lane l of instruction q brings chunk 64 q + l ----
        // Offsets are 32-bit and relative to the tile's FIRST image (a tile of SG_BN columns spans a few images at most):
        // relative to the arena they would pass 2^32 bytes from image ~1765 on (conv2: 65 x 65 x 96 x 6 B per image), and
        // a forward call may carry up to 4096 images.  The 64-bit part goes into the scalar base.
        const int b_first = (nt * SG_BN) / ohw;
        unsigned boff[3 * CB];
#pragma unroll
        for (int q = 0; q < 3 * CB; ++q) {
            const int chunk = q * 64 + lane;
            const int col = chunk / 6, part = chunk - col * 6;   // col: 0 .. 32 CB - 1 (this wave's blocks are adjacent)
            int n = nt * SG_BN + wave * CB * 32 + col;
            n = n < d.N ? n : d.N - 1;                   // tail columns re-read the last valid one
            const int b = n / ohw, r = n - b * ohw;
            const int oh = r / d.OW, ow = r - oh * d.OW;
            boff[q] = (unsigned)((((b - b_first) * d.Hp + oh) * d.Wp + ow) * d.Ctot) * 6u + (unsigned)part * 16u;
        }
        const unsigned char* bgrp = reinterpret_cast<const unsigned char*>(act) + (size_t)g * d.Cg * 6 +
                                    (size_t)b_first * d.Hp * d.Wp * d.Ctot * 6;
        // ---- A fragments: contiguous 1 KB pieces of the packed weights ----
        const unsigned char* wgrp = reinterpret_cast<const unsigned char*>(wfrag) +
                                    ((size_t)g * d.ksteps * d.mblocks + (size_t)mt * MB) * 3 * 1024;
        unsigned aoff[(NA + NWAVES - 1) / NWAVES];        // fragment f = wave + NWAVES q of the stage, this lane's 16 bytes
#pragma unroll
        for (int q = 0; q < (NA + NWAVES - 1) / NWAVES; ++q) aoff[q] = (unsigned)((wave + NWAVES * q) * 1024 + lane * 16);
        // Stages are issued in order, so the (channel group, kh, kw) of the next one is carried along instead of being
        // recomputed: two integer divisions by run-time divisors per stage cost more than the stage's DMA instructions.
        // K16 steps run over the taps first, then over the channel groups (consecutive stages re-read the same 96 bytes
        // per pixel, shifted by one pixel / one row).
        int i_kw = 0, i_kh = 0, i_c0 = 0, i_buf = 0;
        const unsigned char* i_w = wgrp;
        const size_t wstep = (size_t)d.mblocks * 3 * 1024;
        auto issue = [&]() {
            const unsigned stage = lds0 + (unsigned)(i_buf * STAGE_BYTES);
#pragma unroll
            for (int q = 0; q < (NA + NWAVES - 1) / NWAVES; ++q) {
                const int f = wave + NWAVES * q;          // wave-uniform
                if (f < NA) dma16(aoff[q], i_w, __builtin_amdgcn_readfirstlane(stage + (unsigned)f * 1024u));
            }
            const unsigned char* bst = bgrp + ((size_t)(i_kh * d.Wp + i_kw) * d.Ctot + i_c0) * 6;
#pragma unroll
            for (int q = 0; q < 3 * CB; ++q)
                dma16(boff[q], bst, __builtin_amdgcn_readfirstlane(stage + (unsigned)((NA + wave * 3 * CB + q) * 1024)));
            i_w += wstep;
            i_buf = i_buf + 1 == NST ? 0 : i_buf + 1;
            if (++i_kw == d.KW) {
                i_kw = 0;
                if (++i_kh == d.KW) { i_kh = 0; i_c0 += 16; }
            }
        };
        auto wait_stage = [&](bool keep_one_in_flight) {   // until only the newest stage's DMA (if any) is outstanding
            if (!keep_one_in_flight) wait_vmcnt<0>();
            else if (na_mine == 1) wait_vmcnt<3 * CB + 1>();
            else if (na_mine == 2) wait_vmcnt<3 * CB + 2>();
            else if (na_mine == 3) wait_vmcnt<3 * CB + 3>();
            else if (na_mine == 4) wait_vmcnt<3 * CB + 4>();
            else wait_vmcnt<3 * CB + 5>();
        };
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int nk = d.ksteps;
        issue();
        if (AHEAD > 1 && nk > 1) issue();
        wait_stage(AHEAD > 1 && nk > 1);
        asm volatile("" : "+v"(nx));                     // the atomic's result has landed (it is older than stage 0)
        if (tid == 0) s_next[parity] = nx + (int)gridDim.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        SG_T(0)
        int buf = NST - 1;
        for (int t = 0; t < nk; ++t) {
            buf = buf + 1 == NST ? 0 : buf + 1;
            if (t + AHEAD < nk) issue();
            SG_T(1)
            const unsigned char* stage = sg_lds + buf * STAGE_BYTES;
            bf16x8 af[TM][3];
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(stage + ((wm * TM + i) * 3 + p) * 1024 + lane * 16);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bf16x8 bfr[3];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    bfr[p] = *reinterpret_cast<const bf16x8*>(stage + (NA + (wn * TN + j) * 3) * 1024 +
                                                              ((lane & 31) * 6 + p * 2 + (lane >> 5)) * 16);
                // the six partial products, smallest first
#ifdef SG_SMALL_SEPARATE
                // (experiment) the five small products of a step are summed on their own -- from zero, so their roundings are
                // 2^-8 of the accumulator's -- and join the accumulator with ONE addition: two roundings of the running sum per
                // step instead of six
                f32x16 tsm[TM];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) tsm[i][e] = 0.f;
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) tsm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[0], tsm[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) tsm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[1], tsm[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) tsm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[2], tsm[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) tsm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[0], tsm[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) tsm[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[1], tsm[i], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[0], acc[i][j], 0, 0, 0);
                    acc[i][j] += tsm[i];
                }
                if (false)
#endif
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[0], acc[i][j], 0, 0, 0);
                }
            }
            SG_T(2)
            wait_stage(AHEAD > 1 && t + 2 < nk);         // stage t + 1 has landed (own pieces) ...
            SG_T(3)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // ... for every wave; stage t's buffer is free again
            SG_T(4)
        }
        // ---- epilogue: bias + ReLU, f32 NCHW planes (accumulator register 4 q + e = row 8 q + 4 (lane / 32) + e) ----
        const int khalf = lane >> 5;
        const int oplane = d.OHp * d.OWp;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nn = nt * SG_BN + (wn * TN + j) * 32 + (lane & 31);
            if (nn >= d.N) continue;
            const int bb = nn / ohw, rr = nn - bb * ohw;
            const int yy = rr / d.OW, xx = rr - yy * d.OW;
            float* ocol = reinterpret_cast<float*>(out_) + ((size_t)bb * d.groups + g) * d.OC * oplane +
                          (size_t)(yy + d.opad) * d.OWp + xx + d.opad;
            unsigned short* opix = reinterpret_cast<unsigned short*>(out_) +
                                   ((size_t)(bb * d.OHp + yy + d.opad) * d.OWp + xx + d.opad) * (size_t)(d.groups * d.OC) * 3;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m0 = __builtin_amdgcn_readfirstlane((mt * MB + wm * TM + i) * 32 + 8 * q);
                    if (m0 >= d.OC) continue;
                    const f32x4v bl = *reinterpret_cast<const f32x4v*>(bias + g * d.OC + m0 + 4 * khalf);   // this lane's 4 rows
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[i][j][4 * q + e] + bl[e];
                        if (d.relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    }
                    if (SPLIT_OUT) {
                        const int oc = g * d.OC + m0 + 4 * khalf;       // 4 consecutive channels of one 16-channel group
                        unsigned short pc[3][4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) split3(v[e], pc[0][e], pc[1][e], pc[2][e]);
                        unsigned short* o = opix + (oc >> 4) * 48 + (oc & 15);
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            uint2 w2;
                            w2.x = (unsigned)pc[p][0] | ((unsigned)pc[p][1] << 16);
                            w2.y = (unsigned)pc[p][2] | ((unsigned)pc[p][3] << 16);
                            *reinterpret_cast<uint2*>(o + p * 16) = w2;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ocol[(m0 + 4 * khalf + e) * oplane] = v[e];
                    }
                }
        }
        SG_T(5)
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
#ifdef SG_TIME
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 8; ++i) sg_dbg[(blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
#endif
}
#ifdef SG_TIME
extern "C" int vpk_dbg_sg(long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sg_dbg), sizeof(long long) * 256 * 8 * 8); }
#endif

#endif
