// cnn_conv1_pieces.hpp -- conv1 + relu1 + norm1 + pool1 (cnn/deploy.prototxt:9-55) on the bf16 matrix cores with EXACT
// operands.  Included by vpk_cnn.hip (after cnn_split_gemm.hpp: bf16x8, lds_barrier, C1_* / C1D_LD constants).
//
// conv1's input is an 8-bit raster (evaluation.py:34-38: float(image) - mean, no scaling).  An integer 0..255 IS a bf16
// number (8 significant bits), every f32 weight is exactly the sum of three bf16 pieces (cnn_split_gemm.hpp), a product of
// two bf16 numbers is exact in f32 and the bf16 MFMA accumulates in f32 -- so
//     conv1(x - mean)[oc][p] + bias[oc] = sum_k (w1 + w2 + w3)[oc][k] x[p, k]  +  ( bias[oc] - sum_k w[oc][k] mean[p, k] )
// with THREE bf16 matrix products per f32 product, none of them rounded, at 3/16 of the f32-input MFMA's pipe time; the
// bracket is a constant of the model (a 123 x 123 x 96 map made in float64 at load, `cmap`) added to the accumulators.  Not
// an approximation of the f32 path: the only roundings left are the accumulator's (one per 32 products instead of one per
// product) and the constant's -- tests/test_gpu_cnn.py measures it against the float64 net beside the f32 direct kernel.
//
// Shape of the kernel (one 768-thread workgroup per CU, persistent over a queue of work items):
//   * tile = the 7 x 17 patch of conv outputs of conv1_direct_kernel (pools to 3 x 8; neighbours share a row / column);
//   * K = (kernel row, tap) padded to 12 rows x 16 taps: a K step of v_mfma_f32_16x16x32_bf16 = two kernel rows x 16 taps (the
//     taps 11..15 and row 11 meet zero weights), six steps; lane (column c, k group q) reads its eight consecutive pixels
//     of row 2 s + q / 2 -- 16 bytes of the RAW patch, kept in LDS as bf16 (36 x 80 pixels = 5.6 KB) -- no im2col;
//   * wave w owns 16 output channels (w % 6) x 64 columns (w / 6) and keeps its 18 weight fragments (6 steps x 3 pieces) in
//     72 registers for the kernel's lifetime: the matrix instructions' A operands never touch LDS;
//   * work item = (patch position, group of images): the constant's 4 x 4 values per lane are fetched once per item;
//   * epilogue as in conv1_direct_kernel: ReLU -> LDS patch [channel][column] -> LRN across channels in place -> 3 x 3 / 2
//     max pool (windows clipped like Caffe's: positions outside the blob hold 0, every real value is >= 0) -> pool1's planes
//     with conv2's border.  bf16 MFMAs do run beside VALU work (the f32-input ones do not).
#ifndef VPK_CNN_CONV1_PIECES_HPP_
#define VPK_CNN_CONV1_PIECES_HPP_

namespace {

constexpr int C1B_THREADS = 768;
constexpr int C1T_LD = 108;                            // row stride (floats) of the [column][channel + 4] output patch
constexpr int C1B_PROWS = 36, C1B_PCOLS = 80;          // raw patch in LDS (pixels): rows 4 * 6 + 12, columns 4 * 16 + 16
constexpr int C1B_LDW = C1B_PCOLS;                     // row stride of the raw patch in LDS (bf16 elements; 88 / 96 / 104 / 192 measured: no faster)
constexpr int C1B_XS = C1B_PROWS * C1B_LDW;            // bf16 per patch buffer
constexpr int C1B_STEPS = 6;                           // K steps of 32
constexpr int C1B_PATCHES = C1_TR * C1_TC;             // 168 patch positions per image

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const u32x2 lds_cu32x2;

// [m tile 6][K step 6][piece 3][lane 64][8] bf16: the A operand of v_mfma_f32_16x16x32_bf16 for output channels
// 16 mt + lane % 16, k = 8 (lane / 16) + e -> kernel row 2 s + (lane / 32), tap 8 ((lane / 16) % 2) + e (zero beyond 10)
// np = 2: scaled fp16 pairs instead (cnn_conv_pieces.hpp; the raster's integers are exact fp16 numbers too): x scale, two pieces
inline void conv1_pieces_weights(const float* w, std::vector<unsigned short>& out, int np = 3, float scale = 1.f) {
    out.assign((size_t)6 * C1B_STEPS * np * 64 * 8, 0);
    auto rne = [](float x) { unsigned b; memcpy(&b, &x, 4); return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u; };
    for (int mt = 0; mt < 6; ++mt)
        for (int s = 0; s < C1B_STEPS; ++s)
            for (int ln = 0; ln < 64; ++ln)
                for (int e = 0; e < 8; ++e) {
                    const int q = ln >> 4, kh = 2 * s + (q >> 1), kw = 8 * (q & 1) + e, oc = 16 * mt + (ln & 15);
                    if (kh > 10 || kw > 10) continue;
                    const float x = w[(size_t)oc * 121 + kh * 11 + kw];
                    if (np == 2) {
                        const float xs = x * scale;
                        const _Float16 h0 = (_Float16)xs;
                        const _Float16 h1 = (_Float16)(xs - (float)h0);
                        const size_t at2 = ((((size_t)mt * C1B_STEPS + s) * 2) * 64 + ln) * 8 + e;
                        memcpy(&out[at2], &h0, 2);
                        memcpy(&out[at2 + 512], &h1, 2);
                        continue;
                    }
                    const unsigned b0 = rne(x);                    // the three pieces of split3() (cnn_split_gemm.hpp)
                    float f0; memcpy(&f0, &b0, 4);
                    const float r1 = x - f0;
                    const unsigned b1 = rne(r1);
                    float f1; memcpy(&f1, &b1, 4);
                    const float r2 = r1 - f1;
                    unsigned b2; memcpy(&b2, &r2, 4);
                    const size_t at = ((((size_t)mt * C1B_STEPS + s) * 3) * 64 + ln) * 8 + e;
                    out[at] = (unsigned short)(b0 >> 16);
                    out[at + 512] = (unsigned short)(b1 >> 16);
                    out[at + 1024] = (unsigned short)(b2 >> 16);
                }
}

// cmap[oh][ow][oc] = bias[oc] - sum_k w[oc][k] mean[4 oh + kh][4 ow + kw], in float64, rounded once
inline void conv1_pieces_cmap(const float* w, const float* bias, const float* mean, std::vector<float>& out) {
    out.assign((size_t)C1_OUT * C1_OUT * 96, 0.f);
    std::vector<double> wt((size_t)121 * 96);
    for (int oc = 0; oc < 96; ++oc)
        for (int k = 0; k < 121; ++k) wt[(size_t)k * 96 + oc] = (double)w[(size_t)oc * 121 + k];
    for (int oh = 0; oh < C1_OUT; ++oh)
        for (int ow = 0; ow < C1_OUT; ++ow) {
            double acc[96];
            for (int oc = 0; oc < 96; ++oc) acc[oc] = 0.0;
            for (int kh = 0; kh < 11; ++kh)
                for (int kw = 0; kw < 11; ++kw) {
                    const double m = (double)mean[(size_t)(4 * oh + kh) * 500 + 4 * ow + kw];
                    const double* wr = &wt[(size_t)(kh * 11 + kw) * 96];
                    for (int oc = 0; oc < 96; ++oc) acc[oc] += wr[oc] * m;
                }
            float* o = &out[((size_t)oh * C1_OUT + ow) * 96];
            for (int oc = 0; oc < 96; ++oc) o[oc] = (float)((double)bias[oc] - acc[oc]);
        }
}

template <int NP>
__global__ __launch_bounds__(C1B_THREADS, 3) void conv1_pieces_kernel(const unsigned char* __restrict__ sphere,
                                                                      const unsigned short* __restrict__ wfrag,
                                                                      const float* __restrict__ cmap, float* __restrict__ out,
                                                                      int OHp, int OWp, int opad, int batch, int group, float oscale,
                                                                      int* __restrict__ item_counter, int total_items,
                                                                      unsigned short* __restrict__ out_planes, float p_ascale,
                                                                      unsigned* __restrict__ range_word) {
    __shared__ __attribute__((aligned(16))) unsigned short Xs[2][C1B_XS];
    // The output patch as [column][channel] (round 6; it was [channel][column]): a lane's four accumulator values are four consecutive
    // channels of one column, the LRN's window runs along the channels and a pooling thread takes four channels of a pixel -- every
    // phase of the epilogue moves 16 bytes per LDS instruction where it moved 4 (wave-level LDS instructions per tile: ~870 -> ~250;
    // counters of the old layout: LDS 0.72 busy, 42 % bank conflicts, matrix pipes 0.30 -- profiles/r05_pmc_conv1_issue.txt).  Channel c
    // sits at index c + 4 (16-byte aligned groups of four), indices 2, 3 and 100, 101 are the LRN's zero halo; the row stride of 108
    // floats = 27 x 16 bytes keeps the 16 lanes of a quarter wave (consecutive columns) on different banks.
    __shared__ __attribute__((aligned(16))) float Ct[128][C1T_LD];
    __shared__ int s_next[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = wave % 6, chalf = wave / 6;
    const int q = lane >> 4, c16 = lane & 15;
    if (tid < 128) {
        *reinterpret_cast<f32x4*>(&Ct[tid][0]) = f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&Ct[tid][100]) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- this wave's weight fragments: 18 x 16 bytes per lane, for the lifetime of the workgroup ----
    bf16x8 A[C1B_STEPS][NP];
#pragma unroll
    for (int s = 0; s < C1B_STEPS; ++s)
#pragma unroll
        for (int p = 0; p < NP; ++p)
            A[s][p] = *reinterpret_cast<const bf16x8*>(wfrag + ((((size_t)mtile * C1B_STEPS + s) * NP + p) * 64 + lane) * 8);
    // ---- B operands: column of the patch this lane feeds in N tile j, and where its eight pixels of K step 0 start ----
    lds_cu32x2* bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int c = 64 * chalf + 16 * j + c16;
        c = c < C1_PR * C1_PC ? c : C1_PR * C1_PC - 1;                  // columns 119..127 repeat the last position, unused
        const int crow = c / C1_PC, ccol = c - crow * C1_PC;
        bp[j] = (lds_cu32x2*)&Xs[0][(4 * crow + (q >> 1)) * C1B_LDW + 4 * ccol + 8 * (q & 1)];
    }
    // ---- patch loader: thread t < 720 brings the four pixels (row t / 20, columns 4 (t % 20) ..) as one 4-byte word ----
    const bool p_on = tid < C1B_PROWS * (C1B_PCOLS / 4);
    const int p_row = p_on ? tid / (C1B_PCOLS / 4) : 0, p_q = p_on ? tid - p_row * (C1B_PCOLS / 4) : 0;
    auto patch_offset = [&](int pr, int pc) {                           // byte offset inside an image; overhang is clamped
        const int y = 4 * (C1_PR - 1) * pr + p_row, x = 4 * (C1_PC - 1) * pc + 4 * p_q;   //  (it only meets zero weights and
        return (y < 500 ? y : 499) * 500 + (x < 496 ? x : 496);         //   conv outputs outside the blob, which are zeroed)
    };
    auto patch_store = [&](unsigned v, int buf) {                        // uint8 -> bf16: exact, the high half of the f32 (fp16: exact too)
        if (p_on) {
            u32x2 w2;
            if (NP == 3) {
                w2[0] = (__float_as_uint((float)(v & 255u)) >> 16) | (__float_as_uint((float)((v >> 8) & 255u)) & 0xffff0000u);
                w2[1] = (__float_as_uint((float)((v >> 16) & 255u)) >> 16) | (__float_as_uint((float)(v >> 24)) & 0xffff0000u);
            } else {
                auto h = [](unsigned x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(float)x); };
                w2[0] = h(v & 255u) | (h((v >> 8) & 255u) << 16);
                w2[1] = h((v >> 16) & 255u) | (h(v >> 24) << 16);
            }
            *reinterpret_cast<u32x2*>(&Xs[buf][p_row * C1B_LDW + 4 * p_q]) = w2;
        }
    };
    // ---- epilogue roles ----
    const bool pool_on = tid < 96 * C1_QR * 2;                           // pooling: (channel, pooled row, half of its 8 outputs)
    // pooling straight into conv2's piece planes (round 6; out_planes != nullptr): thread t < 24 x 24 owns one pooled PIXEL of the patch
    // and FOUR channels = half of a 16-byte word -- [image][channel group of 16][piece x k half][y][x], cnn_conv_pieces.hpp -- as scaled
    // fp16 pairs (split2h_guard: clamped, flagged): 8-byte stores that two threads complete to a word (as the conv epilogues do); the f32
    // pool1 blob and to_planes_kernel's pass over it are not needed then
    // (the roles' indices are derived per tile: kept across the K loop they cost registers the loop does not have)
    const bool pl_on = tid < 24 * C1_QR * C1_QC;
    bool pl_bad = false;

    int item = blockIdx.x, parity = 0, buf = 0;
    if (item >= total_items) return;
    int pr = (item % C1B_PATCHES) / C1_TC, pc = item % C1_TC;
    int b = (item / C1B_PATCHES) * group;
    int b1 = b + group < batch ? b + group : batch;
    if (tid == 0) s_next[0] = atomicAdd(item_counter, 1) + (int)gridDim.x;
    int poff = patch_offset(pr, pc);
    patch_store(*reinterpret_cast<const unsigned*>(sphere + (size_t)b * 250000 + poff), 0);
    bool fresh = true;                                                   // first tile of a work item: fetch its constants
    f32x4 cin[4];
    float cap[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { cin[j] = f32x4{0.f, 0.f, 0.f, 0.f}; cap[j] = 0.f; }
    lds_barrier();
    for (;;) {
        if (fresh) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = 64 * chalf + 16 * j + c16;
                const int cc = c < C1_PR * C1_PC ? c : C1_PR * C1_PC - 1;
                const int crow = cc / C1_PC, ccol = cc - crow * C1_PC;
                const int oh = (C1_PR - 1) * pr + crow, ow = (C1_PC - 1) * pc + ccol;
                const bool inside = c < C1_PR * C1_PC && oh < C1_OUT && ow < C1_OUT;
                const int ohc = oh < C1_OUT ? oh : C1_OUT - 1, owc = ow < C1_OUT ? ow : C1_OUT - 1;
                cin[j] = *reinterpret_cast<const f32x4*>(cmap + ((size_t)ohc * C1_OUT + owc) * 96 + 16 * mtile + 4 * q);
                cap[j] = inside ? 3.402823466e38f : 0.f;
            }
            fresh = false;
        }
        // ---- K loop: 6 steps x 3 pieces x 4 N tiles, smallest pieces first; the operands of step s + 1 are requested
        //      before the matrix instructions of step s ----
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2 braw[2][4][2];
        auto operands = [&](int s) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lds_cu32x2* src = bp[j] + buf * (C1B_XS / 4) + s * (2 * C1B_LDW / 4);
                braw[s & 1][j][0] = src[0];
                braw[s & 1][j][1] = src[1];
            }
        };
        operands(0);
#pragma unroll
        for (int s = 0; s < C1B_STEPS; ++s) {
            if (s + 1 < C1B_STEPS) operands(s + 1);
            bf16x8 bf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 t4 = {braw[s & 1][j][0][0], braw[s & 1][j][0][1], braw[s & 1][j][1][0], braw[s & 1][j][1][1]};
                bf[j] = __builtin_bit_cast(bf16x8, t4);
            }
#pragma unroll
            for (int p = NP - 1; p >= 0; --p)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (NP == 3) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s][p], bf[j], acc[j], 0, 0, 0);
                    else acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A[s][p]), __builtin_bit_cast(f16x8, bf[j]), acc[j], 0, 0, 0);
                }
        }
        lds_barrier();                                                   // (the previous tile's pooling has read Cs)
        // ---- what comes next: the same item's next image, or the next item's first (its index was stored before at
        //      least one barrier ago) -- its raw patch is requested now and stored behind the LRN ----
        int n_item = item, n_pr = pr, n_pc = pc, n_b = b + 1, n_b1 = b1;
        bool n_fresh = false;
        if (n_b >= b1) {
            n_item = __builtin_amdgcn_readfirstlane(s_next[parity]);
            n_pr = (n_item % C1B_PATCHES) / C1_TC; n_pc = n_item % C1_TC;
            n_b = (n_item / C1B_PATCHES) * group;
            n_b1 = n_b + group < batch ? n_b + group : batch;
            n_fresh = true;
        }
        const bool n_on = n_item < total_items;
        unsigned pre = 0;
        if (n_on) {
            if (n_fresh) poff = patch_offset(n_pr, n_pc);
            pre = *reinterpret_cast<const unsigned*>(sphere + (size_t)n_b * 250000 + poff);
        }
        // ---- + constant, ReLU -> Ct[column][channel + 4]; positions outside the conv blob become 0 ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 64 * chalf + 16 * j + c16;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r)                                  // accumulator register r holds row 4 (lane / 16) + r
                v[r] = __builtin_amdgcn_fmed3f(NP == 3 ? acc[j][r] + cin[j][r] : __builtin_fmaf(acc[j][r], oscale, cin[j][r]), 0.f, cap[j]);
            *reinterpret_cast<f32x4*>(&Ct[col][16 * mtile + 4 * q + 4]) = v;
        }
        lds_barrier();
        // ---- LRN across channels (deploy.prototxt:34-44): out = v (1 + alpha / 5 sum of the 5 squares)^-0.75; this thread:
        //      column lp, channels 16 lcg .. 16 lcg + 15 (raw[k] = channel 16 lcg + k - 2; rows 0, 1, 98, 99 are the zero halo)
        int tl_ = tid;
        asm volatile("" : "+v"(tl_));                                    // (roles derived per tile: see the pooling roles)
        const int lp = tl_ & 127, lcg = tl_ >> 7;                        // LRN: column, group of 16 channels
        float raw[20];
        {
            const float* rp = &Ct[lp][16 * lcg + 2];                     // channels 16 lcg - 2 ..: 8 bytes, 4 x 16 bytes, 8 bytes
            const f32x2v a = *reinterpret_cast<const f32x2v*>(rp), z = *reinterpret_cast<const f32x2v*>(rp + 18);
            raw[0] = a[0]; raw[1] = a[1]; raw[18] = z[0]; raw[19] = z[1];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(rp + 2 + 4 * g4);
                raw[2 + 4 * g4] = m[0]; raw[3 + 4 * g4] = m[1]; raw[4 + 4 * g4] = m[2]; raw[5 + 4 * g4] = m[3];
            }
        }
        lds_barrier();                                                   // every raw value has been read
        {
#pragma clang fp contract(off)      // (the association and roundings of conv1_direct_kernel's lrn_two)
            float sq[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) sq[k] = raw[k] * raw[k];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                             // four channels = one 16-byte store
                f32x4 res;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {                         // channels 16 lcg + i, + i + 1: windows raw[i .. i + 4], raw[i + 1 .. i + 5]
                    const int i = 4 * g4 + 2 * h2;
                    const float psa = sq[i] + sq[i + 1], psb = sq[i + 2] + sq[i + 3];
                    const float c = psb + sq[i + 4];
                    const float w0 = c + psa, w1 = (c + sq[i + 1]) + sq[i + 5];
                    const float s0 = __builtin_fmaf(w0, 1e-4f / 5.f, 1.f), s1 = __builtin_fmaf(w1, 1e-4f / 5.f, 1.f);
                    const float r0 = __builtin_amdgcn_rsqf(s0), r1 = __builtin_amdgcn_rsqf(s1);
                    res[2 * h2] = raw[i + 2] * (r0 * __builtin_amdgcn_sqrtf(r0));
                    res[2 * h2 + 1] = raw[i + 3] * (r1 * __builtin_amdgcn_sqrtf(r1));
                }
                *reinterpret_cast<f32x4*>(&Ct[lp][16 * lcg + 4 + 4 * g4]) = res;
            }
        }
        if (n_on) patch_store(pre, buf ^ 1);                             // (that buffer was last read in the previous tile's K loop)
        if (n_fresh && tid == 0 && n_on) s_next[parity ^ 1] = atomicAdd(item_counter, 1) + (int)gridDim.x;
        lds_barrier();
        // ---- 3 x 3 / stride 2 max pool: (channel, pooled row, half) = 4 outputs from 3 x 9 values ----
        if (out_planes) {
            if (pl_on) {
                int tl = tid;
                asm volatile("" : "+v"(tl));                         // (not hoisted out of the tile loop)
                const int pl_g = tl / (C1_QR * C1_QC), pl_pix = tl % (C1_QR * C1_QC), pl_y = pl_pix / C1_QC, pl_x = pl_pix % C1_QC;
                const float* pl_src = &Ct[2 * pl_y * C1_PC + 2 * pl_x][4 * pl_g + 4];
                const int ph = C1_QR * pr + pl_y, pw = C1_QC * pc + pl_x;
                unsigned short h0[4], h1[4];
                f32x4 m4 = {0.f, 0.f, 0.f, 0.f};                     // (every value is >= 0: ReLU, then a positive factor)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(pl_src + (dy * C1_PC + dx) * C1T_LD);
#pragma unroll
                        for (int k = 0; k < 4; ++k) m4[k] = __builtin_fmaxf(m4[k], t4[k]);
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) split2h_guard(m4[k] * p_ascale, h0[k], h1[k], pl_bad);
                if (ph < C1_POOL && pw < C1_POOL) {
                    const u32x2 a = {(unsigned)h0[0] | ((unsigned)h0[1] << 16), (unsigned)h0[2] | ((unsigned)h0[3] << 16)};
                    const u32x2 c = {(unsigned)h1[0] | ((unsigned)h1[1] << 16), (unsigned)h1[2] | ((unsigned)h1[3] << 16)};
                    const size_t wpl = (size_t)OHp * OWp;
                    const int cg8 = pl_g >> 1;                       // the word's eight channels; this thread: its half pl_g & 1
                    u32x2* dst = reinterpret_cast<u32x2*>(reinterpret_cast<u32x4*>(out_planes) + (((size_t)b * 6 + (cg8 >> 1)) * 4 + (cg8 & 1)) * wpl +
                                                          (size_t)(ph + opad) * OWp + pw + opad) + (pl_g & 1);
                    dst[0] = a;                                      // piece 0, this k half
                    dst[4 * wpl] = c;                                // piece 1 (2 planes x 2 halves of a word further)
                }
            }
        } else
        if (pool_on) {
            int tl = tid;
            asm volatile("" : "+v"(tl));                                 // (derived per tile, like the plane-writing roles)
            const int pk = tl / (C1_QR * 2), prem = tl % (C1_QR * 2), ppy = prem >> 1, phalf = prem & 1;
            const float* pool_src = &Ct[2 * ppy * C1_PC + 8 * phalf][pk + 4];
            float cm[9];
#pragma unroll
            for (int x = 0; x < 9; ++x)
                cm[x] = __builtin_fmaxf(__builtin_fmaxf(pool_src[x * C1T_LD], pool_src[(C1_PC + x) * C1T_LD]), pool_src[(2 * C1_PC + x) * C1T_LD]);
            const int ph = C1_QR * pr + ppy, pw0 = C1_QC * pc + 4 * phalf;
            if (ph < C1_POOL) {
                float* o = out + ((size_t)b * 96 + pk) * OHp * OWp + (size_t)(ph + opad) * OWp + pw0 + opad;
#pragma unroll
                for (int px = 0; px < 4; ++px)
                    if (pw0 + px < C1_POOL) o[px] = __builtin_fmaxf(__builtin_fmaxf(cm[2 * px], cm[2 * px + 1]), cm[2 * px + 2]);
            }
        }
        if (!n_on) break;
        if (n_fresh) parity ^= 1;
        item = n_item; pr = n_pr; pc = n_pc; b = n_b; b1 = n_b1; fresh = n_fresh;
        buf ^= 1;
    }
    if (out_planes) range_report(pl_bad, range_word, 1u << 1);         // (conv2 is the consuming layer)
}


// (Round 6, built, measured, removed: the same kernel in workgroups of SIX waves -- a wave per 16-channel tile, both column halves one after
//  the other, two items per thread in the epilogue -- so that TWO workgroups share a CU and one's K loop runs under the other's epilogue.
//  Same bits, and 0.455 ms against 0.365: the second workgroup never becomes resident.  The waves of a workgroup are placed on the SIMDs
//  in turn starting from the first, so two 6-wave workgroups would put 2 + 2 waves on SIMDs 0 and 1, and 4 x 168 VGPRs exceed the
//  512 of a SIMD (SQ_WAVE_CYCLES per kernel-ms: 0.55 of the 12-wave kernel's).  The serial phases can only be overlapped INSIDE one
//  workgroup -- and that was built and measured too: six PRODUCER waves (a wave per 16-channel tile, both column halves, the K loops of
//  tile t) beside six CONSUMER waves (LRN, pooling, stores and the next raw patch for tile t - 1) on a double-buffered output patch,
//  three barriers per tile, work items (patch position, image) two tiles ahead.  Same bits, no deadlock, and 0.55 ms: six waves spread
//  2 + 2 + 1 + 1 over the SIMDs do not keep the matrix pipes fed the way twelve do (a wave's K loop is a chain of LDS operand reads and
//  dependent matrix instructions; three waves per SIMD hide that, one or two do not), and the two roles in one kernel spill (51 dwords).
//  So this kernel's 0.355 ms stand: its phases are serial, but each of them runs at the occupancy it needs.)

}  // namespace
#endif
