// cnn_split_gemm.hpp -- f32-accurate implicit-GEMM convolution on the bf16 matrix cores (included by vpk_cnn.hip).
//
// gfx950 has no TF32-like mode: f32-input MFMA runs at 1/16 of the bf16 rate.  An f32 number is EXACTLY the sum of three
// bf16 numbers (8 + 8 + 8 significand bits, obtained by truncation: x1 = hi16(x), x2 = hi16(x - x1), x3 = x - x1 - x2), a
// product of two bf16 numbers is exact in f32, and the bf16 MFMA accumulates in f32.  So
//     a * b = sum_{i,j} a_i b_j      (9 exact partial products)
// and the six products with i + j <= 4 carry everything above 2^-24 |a b| -- the size of the single rounding an f32 FMA
// makes anyway; the three dropped ones are below 2^-31 |a b|.  Six bf16 MFMAs per f32 MFMA-equivalent = 6/16 of the time
// of the native f32 matrix path.  (cuBLAS ships the same idea as "BF16x9" FP32 emulation; "x6" drops the terms that are
// below the accumulator's own rounding.)  tests/test_gpu_cnn.py compares both paths with an fp64 oracle: their errors
// are of the same size.
//
// Data layout (chosen for the matrix cores, not inherited from the f32 path):
//   activations  [piece 0..2][image][y][x][channel] bf16, planes carry the convolution's zero border (split_nhwc_kernel
//                writes them from the f32 NCHW planes); K order = (kh, kw, channel), so a K16 step = 16 consecutive
//                channels of one tap = 32 contiguous bytes per pixel and piece;
//   weights      pre-split and pre-permuted on the host into MFMA fragment order:
//                [group][k16 step][32-row block][piece][lane][8 bf16]  (1 KB = one A operand of v_mfma_f32_32x32x16_bf16);
//   LDS stage    a list of 1 KB fragments, A blocks first then B blocks, each exactly as the MFMA wants it in registers:
//                every fragment is written by ONE global_load_lds_dwordx4 (lane l brings the 16 bytes lane l will later
//                read back with one conflict-free ds_read_b128).
// Tile = (WAVES_M * TM * 32) x 256 outputs, 8 waves (WAVES_M x 4 in N... see below), three stages in flight.
#ifndef VPK_CNN_SPLIT_GEMM_HPP_
#define VPK_CNN_SPLIT_GEMM_HPP_

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct SplitDims {
    int B, Cg, Ctot, Hp, Wp;        // input: channels per group / in total; padded plane
    int OC, OH, OW, groups;         // OC = output channels per group
    int KW, csteps, ksteps;         // kernel width; K16 steps per tap (Cg / 16); K16 steps in total (KH * KW * csteps)
    int mblocks;                    // 32-row blocks per group in the packed weights
    int N;                          // B * OH * OW
    int relu;
    int OHp, OWp, opad;             // f32 NCHW output planes
    long long plane_bytes;          // distance between two pieces of the activations, bytes
};

constexpr int SG_THREADS = 512;
constexpr int SG_BN = 256;          // columns per tile: 8 blocks of 32

// f32 -> three bf16 pieces (truncation; exact: the three add up to x)
__device__ __forceinline__ void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    const unsigned b0 = __float_as_uint(x) & 0xffff0000u;
    const float r1 = x - __uint_as_float(b0);
    const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(b1);
    p0 = (unsigned short)(b0 >> 16);
    p1 = (unsigned short)(b1 >> 16);
    p2 = (unsigned short)(__float_as_uint(r2) >> 16);
}

// f32 NCHW planes (with their zero border) -> three bf16 NHWC pieces.  One workgroup per (image, row): the row's C x Wp
// values are transposed through LDS so that both sides are coalesced.
__global__ __launch_bounds__(256) void split_nhwc_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, int C,
                                                         int Hp, int Wp, long long plane_elems) {
    extern __shared__ float sn_tile[];                   // [C][Wp + 1]
    const int y = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int ld = Wp + 1;
    for (int idx = tid; idx < C * Wp; idx += 256) {
        const int c = idx / Wp, x = idx - c * Wp;
        sn_tile[c * ld + x] = in[(((size_t)b * C + c) * Hp + y) * Wp + x];
    }
    __syncthreads();
    const int C2 = C >> 1;
    unsigned* o0 = reinterpret_cast<unsigned*>(out + ((size_t)b * Hp + y) * Wp * C);
    unsigned* o1 = reinterpret_cast<unsigned*>(out + plane_elems + ((size_t)b * Hp + y) * Wp * C);
    unsigned* o2 = reinterpret_cast<unsigned*>(out + 2 * plane_elems + ((size_t)b * Hp + y) * Wp * C);
    for (int idx = tid; idx < Wp * C2; idx += 256) {
        const int x = idx / C2, c = (idx - x * C2) * 2;
        unsigned short a0, a1, a2, b0, b1, b2;
        split3(sn_tile[c * ld + x], a0, a1, a2);
        split3(sn_tile[(c + 1) * ld + x], b0, b1, b2);
        o0[idx] = (unsigned)a0 | ((unsigned)b0 << 16);
        o1[idx] = (unsigned)a1 | ((unsigned)b1 << 16);
        o2[idx] = (unsigned)a2 | ((unsigned)b2 << 16);
    }
}

template <int WAVES_M, int TM>
__global__ __launch_bounds__(SG_THREADS, 2) void conv_gemm_split_kernel(SplitDims d, const unsigned short* __restrict__ act,
                                                                        const unsigned short* __restrict__ wfrag,
                                                                        const float* __restrict__ bias, float* __restrict__ out,
                                                                        int* __restrict__ tile_counter, int total_tiles) {
    constexpr int WAVES_N = 8 / WAVES_M;
    constexpr int TN = SG_BN / 32 / WAVES_N;
    constexpr int MB = WAVES_M * TM;                     // 32-row blocks per tile
    constexpr int NB = SG_BN / 32;                       // 32-column blocks per tile
    constexpr int NA = MB * 3, NBF = NB * 3;             // fragments per stage
    constexpr int STAGE_BYTES = (NA + NBF) * 1024;
    constexpr int NST = 3;
    static_assert(NST * STAGE_BYTES <= 160 * 1024 - 64, "three stages must fit the CU's LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char sg_lds[];
    __shared__ int s_next[2];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const unsigned lds0 = lds_addr(sg_lds);
    const int mtiles = d.mblocks / MB;
    const int ntiles = (d.N + SG_BN - 1) / SG_BN;
    const int ohw = d.OH * d.OW;
    // A fragments of a stage are dealt round-robin to the waves: fragment f (block f / 3, piece f % 3) to wave f % 8
    const int na_mine = (NA - wave + 7) / 8;             // wave-uniform
    int parity = 0;
    for (int tile = blockIdx.x; tile < total_tiles;) {
        int nx = 0;
        if (tid == 0)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(nx) : "v"(0), "v"(1), "s"(tile_counter) : "memory");
        int bid = tile;
        const int mt = bid % mtiles; bid /= mtiles;
        const int nt = bid % ntiles;
        const int g = bid / ntiles;
        // ---- B gather: this wave brings column block `wave` (32 columns x 16 channels x 3 pieces per stage) ----
        int n = nt * SG_BN + wave * 32 + (lane & 31);
        n = n < d.N ? n : d.N - 1;                       // tail columns re-read the last valid one
        const int b = n / ohw, r = n - b * ohw;
        const int oh = r / d.OW, ow = r - oh * d.OW;
        const unsigned boff = (unsigned)(((b * d.Hp + oh) * d.Wp + ow) * d.Ctot) * 2u + (unsigned)(lane >> 5) * 16u;
        const unsigned char* bgrp = reinterpret_cast<const unsigned char*>(act) + (size_t)g * d.Cg * 2;
        // ---- A fragments: contiguous 1 KB pieces of the packed weights ----
        const unsigned char* wgrp = reinterpret_cast<const unsigned char*>(wfrag) +
                                    ((size_t)g * d.ksteps * d.mblocks + (size_t)mt * MB) * 3 * 1024;
        const unsigned aoff = (unsigned)lane * 16u;
        auto issue = [&](int s, int buf) {
            const unsigned stage = lds0 + (unsigned)(buf * STAGE_BYTES);
            const unsigned char* wst = wgrp + (size_t)s * d.mblocks * 3 * 1024;
#pragma unroll
            for (int q = 0; q < (NA + 7) / 8; ++q) {
                const int f = wave + 8 * q;               // wave-uniform
                if (f < NA) dma16(aoff, wst + (size_t)f * 1024, __builtin_amdgcn_readfirstlane(stage + (unsigned)f * 1024u));
            }
            const int tap = s / d.csteps, c0 = (s - tap * d.csteps) * 16;
            const int kh = tap / d.KW, kw = tap - kh * d.KW;
            const unsigned char* bst = bgrp + ((size_t)(kh * d.Wp + kw) * d.Ctot + c0) * 2;
#pragma unroll
            for (int p = 0; p < 3; ++p)
                dma16(boff, bst + (size_t)p * d.plane_bytes,
                      __builtin_amdgcn_readfirstlane(stage + (unsigned)((NA + wave * 3 + p) * 1024)));
        };
        auto wait_stage = [&](bool keep_one_in_flight) {   // until only the newest stage's DMA (if any) is outstanding
            if (!keep_one_in_flight) wait_vmcnt<0>();
            else if (na_mine == 1) wait_vmcnt<4>();
            else if (na_mine == 2) wait_vmcnt<5>();
            else wait_vmcnt<6>();
        };
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int nk = d.ksteps;
        issue(0, 0);
        if (nk > 1) issue(1, 1);
        wait_stage(nk > 1);
        asm volatile("" : "+v"(nx));                     // the atomic's result has landed (it is older than stage 0)
        if (tid == 0) s_next[parity] = nx + (int)gridDim.x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            const int buf = t % NST;
            if (t + 2 < nk) issue(t + 2, (t + 2) % NST);
            const unsigned char* stage = sg_lds + buf * STAGE_BYTES;
            bf16x8 af[TM][3], bfr[TN][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i][p] = *reinterpret_cast<const bf16x8*>(stage + ((wm * TM + i) * 3 + p) * 1024 + lane * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j][p] = *reinterpret_cast<const bf16x8*>(stage + (NA + (wn * TN + j) * 3 + p) * 1024 + lane * 16);
            }
            // the six partial products, smallest first
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bfr[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bfr[j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bfr[j][0], acc[i][j], 0, 0, 0);
                }
            wait_stage(t + 2 < nk);                      // stage t + 1 has landed (own pieces) ...
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                // ... for every wave; stage t's buffer is free again
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
            float* ocol = out + ((size_t)bb * d.groups + g) * d.OC * oplane + (size_t)(yy + d.opad) * d.OWp + xx + d.opad;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m0 = __builtin_amdgcn_readfirstlane((mt * MB + wm * TM + i) * 32 + 8 * q);
                    if (m0 >= d.OC) continue;
                    const float* bp = bias + g * d.OC + m0;          // wave-uniform: scalar load of 8 floats
                    float bl[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) bl[e] = bp[e];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc[i][j][4 * q + e] + (khalf ? bl[4 + e] : bl[e]);
                        if (d.relu) v = v > 0.f ? v : 0.f;
                        ocol[(m0 + 4 * khalf + e) * oplane] = v;
                    }
                }
        }
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
}

#endif
