// cnn_dense_pieces.hpp -- fc6 of cnn/deploy.prototxt (:192-210: InnerProduct 57 600 -> 4 096, weights (out, in)) on the bf16 matrix
// cores with exact operands.  Included by vpk_cnn.hip (after cnn_split_gemm.hpp: bf16x8, split3, dma16, lds_addr).
//
// Why: at B = 102 the layer is 48 GFLOP over a 0.94 GB weight matrix.  On the f32-input matrix instructions (157 TF) the products
// take 0.31 ms at peak -- longer than streaming the weights from HBM (0.17 ms at 5.5 TB/s) -- so the layer was matrix-pipe-bound
// at 0.40 ms and paid twice beside the EM.  With every f32 operand split into three bf16 pieces (exact; six bf16 products per f32
// product carry everything above 2^-24 of it, cnn_split_gemm.hpp) the products need 0.13 ms of the bf16 pipes and the layer is
// what it should be: a weight stream.
//   * weights stay f32 in HBM in Caffe's own layout (no 1.4 GB pre-split copy): a lane reads 64 contiguous bytes of its row
//     (two lanes = one 128-byte line) two chunks ahead and splits them in registers (~160 integer / f32 operations per 48 matrix
//     instructions: free);
//   * activations (pool5, 102 x 57 600 f32) are split once per forward into B-fragment order (dense_split_kernel, 35 MB) and come
//     through a three-stage LDS ring by DMA, shared by the workgroup's eight waves;
//   * K order inside a chunk of 32: matrix step A takes k = 16 h + 0..7, step B k = 16 h + 8..15 (h = k half of the lane) -- any
//     assignment of k to the instruction's 16 slots is valid as long as both operands use it, and this one makes a lane's 16
//     values contiguous in memory;
//   * block sums as in cnn_conv_pieces.hpp: the products of 8 chunks accumulate from zero and join the accumulator with one
//     addition; split-K partials are reduced by splitk_reduce_kernel in a fixed order (deterministic).
#ifndef VPK_CNN_DENSE_PIECES_HPP_
#define VPK_CNN_DENSE_PIECES_HPP_

namespace {

constexpr int DP_THREADS = 512;
constexpr int DP_BM = 256, DP_BN = 128;         // rows (outputs) x columns (images) of a tile
constexpr int DP_CHUNK = 32;                    // k per chunk = two K16 steps
template <int NP> constexpr int DP_STAGE = NP * 2 * 4 * 1024;   // bytes of one chunk's B fragments: [piece][step][column block][lane][8]
constexpr int DP_NST = 3;
constexpr int DP_FOLD = 8;                      // chunks per block sum

struct DenseDims {
    int N, K, OC;                               // images, inputs, outputs
    int chunks, kparts, cpp;                    // K / 32; split-K parts; chunks per part
    int mtiles, ntiles;
    float wscale, oscale;                       // fp16 pairs (NP = 2): weights x wscale before the split; partials x oscale (powers of two)
};

// X f32 [N][K] -> B fragments [column tile][chunk][piece][step][column block][lane][8] (columns >= N: zeros): NP = 3 bf16 triples,
// NP = 2 fp16 pairs of ascale x (cnn_conv_pieces.hpp)
template <int NP>
__global__ __launch_bounds__(256) void dense_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, int N, int K,
                                                          int chunks, float ascale, unsigned* __restrict__ range_word, unsigned range_bit) {
    const int c = blockIdx.x, nt = blockIdx.y;
    const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = nt * DP_BN + 32 * j + (lane & 31), h = lane >> 5;
    float v[16];
    if (n < N) {
        const f32x4v* src = reinterpret_cast<const f32x4v*>(x + (size_t)n * K + (size_t)c * DP_CHUNK + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4v t = src[q];
            v[4 * q] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = 0.f;
    }
    unsigned short p[3][16];
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (NP == 3) split3(v[e], p[0][e], p[1][e], p[2][e]);
        else split2h_guard(v[e] * ascale, p[0][e], p[1][e], bad);
    }
    if (NP == 2) range_report(bad, range_word, range_bit);
    u32x4* dst = reinterpret_cast<u32x4*>(out + ((size_t)nt * chunks + c) * (DP_STAGE<NP> / 2));
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w4;
#pragma unroll
            for (int e = 0; e < 4; ++e) w4[e] = (unsigned)p[q][8 * s + 2 * e] | ((unsigned)p[q][8 * s + 2 * e + 1] << 16);
            dst[(((q * 2 + s) * 4 + j) * 64) + lane] = w4;
        }
}

// Caffe's (out, in) weight matrix -> tile order [row tile of 256][chunk of 32 k][row][k] (rows past OC: zeros), once at load: the
// kernel then streams each tile's K range as one contiguous run (in Caffe's layout a wave's 32 rows are 32 separate 128-byte
// pieces 230 KB apart: 3.0 TB/s; in tile order see DESIGN.md)
__global__ void dense_tile_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int OC, int K, int chunks, long long total4) {
    const long long o4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o4 >= total4) return;
    const int kk4 = (int)(o4 % (DP_CHUNK / 4));
    const int r = (int)((o4 / (DP_CHUNK / 4)) % DP_BM);
    const long long tc = o4 / ((DP_CHUNK / 4) * DP_BM);
    const int c = (int)(tc % chunks);
    const int mt = (int)(tc / chunks);
    const int row = mt * DP_BM + r;
    f32x4v v = {0.f, 0.f, 0.f, 0.f};
    if (row < OC) v = *reinterpret_cast<const f32x4v*>(w + (size_t)row * K + (size_t)c * DP_CHUNK + 4 * kk4);
    reinterpret_cast<f32x4v*>(out)[o4] = v;
}

// fp16 pairs, PRE-SPLIT (round 6): a scaled pair is 2 + 2 bytes -- exactly an f32 weight's four -- so the weights can be stored as the
// kernel's A fragments at no cost in traffic: [row tile][chunk][wave][K16 step (2)][piece (2)][lane][8 halves], made once at load
// (rows past OC: zeros).  The split in registers (~110 VALU instructions per chunk and wave, all eight waves doing it together
// between two barriers, the matrix pipes idle meanwhile: pipes 0.30 busy, 3.8 TB/s) is gone from the loop.
__global__ void dense_pair_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int OC, int K, int chunks,
                                          float wscale, long long total16) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte word = 8 halves
    if (o >= total16) return;
    const int lane = (int)(o & 63);
    const int q = (int)((o >> 6) & 1), s_ = (int)((o >> 7) & 1), wv = (int)((o >> 8) & 7);
    const long long tc = o >> 11;
    const int c = (int)(tc % chunks), mt = (int)(tc / chunks);
    const int row = mt * DP_BM + 32 * wv + (lane & 31), h = lane >> 5;
    unsigned short p[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = row < OC ? w[(size_t)row * K + (size_t)c * DP_CHUNK + 16 * h + 8 * s_ + e] * wscale : 0.f;
        unsigned short h0, h1;
        split2h(x, h0, h1);
        p[e] = q == 0 ? h0 : h1;
    }
    u32x4 w4;
#pragma unroll
    for (int e = 0; e < 4; ++e) w4[e] = (unsigned)p[2 * e] | ((unsigned)p[2 * e + 1] << 16);
    reinterpret_cast<u32x4*>(out)[o] = w4;
}

// the loop on pre-split pairs: per chunk and wave four 16-byte loads (2 steps x 2 pieces) three chunks deep (three register sets:
// the set of chunk c is still the matrix instructions' operand when chunk c + 2 is requested), the B fragments as before
__global__ __launch_bounds__(DP_THREADS, 2) void dense_pairs_kernel(DenseDims d, const unsigned short* __restrict__ wpair,
                                                                    const unsigned short* __restrict__ xfrag, float* __restrict__ part,
                                                                    int* __restrict__ item_counter, int total_items) {
    constexpr int NP = 2;
    __shared__ __attribute__((aligned(16))) unsigned char dp_lds[DP_NST * DP_STAGE<NP>];
    __shared__ int s_next[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = lds_addr(dp_lds);
    typedef __attribute__((address_space(3))) const bf16x8 lds_cbf8;
    int parity = 0;
    for (int item = blockIdx.x; item < total_items;) {
        int nx = 0;
        if (tid == 0)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(nx) : "v"(0), "v"(1), "s"(item_counter) : "memory");
        int t = item;
        const int mt = t % d.mtiles; t /= d.mtiles;
        const int nt = t % d.ntiles;
        const int ks = t / d.ntiles;
        const int c0 = ks * d.cpp, c1 = c0 + d.cpp < d.chunks ? c0 + d.cpp : d.chunks;
        // this wave's fragments of chunk c: 4 KB at ((mt * chunks + c) * 8 + wave) * 4 KB; (step s, piece q) at ((2 s + q) * 64 + lane) * 16
        const bf16x8* wbase = reinterpret_cast<const bf16x8*>(wpair) + ((size_t)mt * d.chunks * 8 + wave) * 256 + lane;
        const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(xfrag) + (size_t)nt * d.chunks * DP_STAGE<NP>;
        bf16x8 af[3][2][NP];
        auto load_a = [&](int c, auto o_tag) {
            constexpr int o = decltype(o_tag)::value;
            const bf16x8* src = wbase + (size_t)c * (8 * 256);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < NP; ++q) af[o][s][q] = src[(2 * s + q) * 64];
        };
        auto issue_b = [&](int c) {                           // 8 NP fragments of 1 KB: wave w brings NP w .. NP w + NP - 1
            const unsigned char* src = xsrc + (size_t)c * DP_STAGE<NP>;
            const unsigned dst = lds0 + (unsigned)((c % DP_NST) * DP_STAGE<NP>);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int f = NP * wave + k;
                dma16((unsigned)(f * 1024 + lane * 16), src, __builtin_amdgcn_readfirstlane(dst + (unsigned)f * 1024u));
            }
        };
        f32x16 acc[4], tq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; tq[j][e] = 0.f; }
        __builtin_amdgcn_s_barrier();                        // (every wave has left the previous item's stages)
        load_a(c0, std::integral_constant<int, 0>());
        issue_b(c0);
        if (c0 + 1 < c1) { load_a(c0 + 1, std::integral_constant<int, 1>()); issue_b(c0 + 1); }
        int fold = 0;
        auto chunk = [&](auto o_tag, auto o2_tag, int c) {   // o: this chunk's register set, o2: the set of chunk c + 2
            constexpr int o = decltype(o_tag)::value;
            // chunk c's weights and B fragments (own pieces) have landed; younger: chunk c + 1's 4 + NP requests
            if (c + 1 < c1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (c == c0 && tid == 0) { asm volatile("" : "+v"(nx)); s_next[parity] = nx + (int)gridDim.x; }
            __builtin_amdgcn_s_barrier();                    // ... for every wave; and every wave is done with chunk c - 1's stage
            if (c + 2 < c1) { load_a(c + 2, o2_tag); issue_b(c + 2); }   // (stage (c + 2) % 3 and set (c + 2) % 3 held chunk c - 1)
            const unsigned stage = lds0 + (unsigned)((c % DP_NST) * DP_STAGE<NP> + lane * 16);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 bfr[4][NP];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < NP; ++q) bfr[j][q] = *(lds_cbf8*)(stage + (unsigned)((((q * 2 + s) * 4) + j) * 1024));
                constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};   // (small products first)
#pragma unroll
                for (int pr = 0; pr < 3; ++pr)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[o][s][PA[pr]]),
                                                                       __builtin_bit_cast(f16x8, bfr[j][PB[pr]]), tq[j], 0, 0, 0);
            }
            if (++fold == DP_FOLD || c + 1 == c1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] += tq[j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) tq[j][e] = 0.f;
                }
                fold = 0;
            }
        };
        int c = c0;
        for (; c + 2 < c1; c += 3) {
            chunk(std::integral_constant<int, 0>(), std::integral_constant<int, 2>(), c);
            chunk(std::integral_constant<int, 1>(), std::integral_constant<int, 0>(), c + 1);
            chunk(std::integral_constant<int, 2>(), std::integral_constant<int, 1>(), c + 2);
        }
        if (c < c1) chunk(std::integral_constant<int, 0>(), std::integral_constant<int, 2>(), c);
        if (c + 1 < c1) chunk(std::integral_constant<int, 1>(), std::integral_constant<int, 0>(), c + 1);
        // ---- partials [k part][image][output]: accumulator register 4 q + e = row 8 q + 4 h + e of this wave's 32-row block ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nt * DP_BN + 32 * j + r31;
            if (n >= d.N) continue;
            float* prow = part + ((size_t)ks * d.N + n) * d.OC + mt * DP_BM + 32 * wave + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mt * DP_BM + 32 * wave + 8 * q >= d.OC) continue;
                f32x4v v4 = {acc[j][4 * q] * d.oscale, acc[j][4 * q + 1] * d.oscale, acc[j][4 * q + 2] * d.oscale, acc[j][4 * q + 3] * d.oscale};
                *reinterpret_cast<f32x4v*>(prow + 8 * q) = v4;
            }
        }
        item = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
}

template <int NP>
__global__ __launch_bounds__(DP_THREADS, 2) void dense_pieces_kernel(DenseDims d, const float* __restrict__ w,
                                                                     const unsigned short* __restrict__ xfrag, float* __restrict__ part,
                                                                     int* __restrict__ item_counter, int total_items) {
    __shared__ __attribute__((aligned(16))) unsigned char dp_lds[DP_NST * DP_STAGE<NP>];
    __shared__ int s_next[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r31 = lane & 31, h = lane >> 5;
    const unsigned lds0 = lds_addr(dp_lds);
    typedef __attribute__((address_space(3))) const bf16x8 lds_cbf8;
    int parity = 0;
    for (int item = blockIdx.x; item < total_items;) {
        int nx = 0;
        if (tid == 0)
            asm volatile("s_nop 4\n\tglobal_atomic_add %0, %1, %2, %3 sc0" : "=v"(nx) : "v"(0), "v"(1), "s"(item_counter) : "memory");
        int t = item;
        const int mt = t % d.mtiles; t /= d.mtiles;
        const int nt = t % d.ntiles;
        const int ks = t / d.ntiles;
        const int c0 = ks * d.cpp, c1 = c0 + d.cpp < d.chunks ? c0 + d.cpp : d.chunks;
        const int row = mt * DP_BM + 32 * wave + r31;
        const bool row_ok = row < d.OC;
        // (weights in TILE order, dense_tile_weights_kernel: a chunk of a tile is 32 KB contiguous, a wave's 32 rows 4 KB of it)
        const f32x4v* wrow = reinterpret_cast<const f32x4v*>(w + ((size_t)mt * d.chunks * DP_BM + 32 * wave + r31) * DP_CHUNK + 16 * h);
        const unsigned char* xsrc = reinterpret_cast<const unsigned char*>(xfrag) + (size_t)nt * d.chunks * DP_STAGE<NP>;
        f32x4v araw[2][4];                                   // the 16 weights of this lane for chunks c and c + 1 (two register sets)
        auto load_a = [&](int c, auto o_tag) {
            constexpr int o = decltype(o_tag)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) araw[o][q] = wrow[(size_t)c * (DP_BM * DP_CHUNK / 4) + q];
        };
        auto issue_b = [&](int c) {                           // 8 NP fragments of 1 KB: wave w brings NP w .. NP w + NP - 1
            const unsigned char* src = xsrc + (size_t)c * DP_STAGE<NP>;
            const unsigned dst = lds0 + (unsigned)((c % DP_NST) * DP_STAGE<NP>);
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int f = NP * wave + k;
                dma16((unsigned)(f * 1024 + lane * 16), src, __builtin_amdgcn_readfirstlane(dst + (unsigned)f * 1024u));
            }
        };
        f32x16 acc[4], tq[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; tq[j][e] = 0.f; }
        __builtin_amdgcn_s_barrier();                        // (every wave has left the previous item's stages)
        load_a(c0, std::integral_constant<int, 0>());
        issue_b(c0);
        if (c0 + 1 < c1) { load_a(c0 + 1, std::integral_constant<int, 1>()); issue_b(c0 + 1); }
        int fold = 0;
        // (Round 5, measured: making the NEXT chunk's pieces between the groups of matrix instructions of the current one -- two
        // operand sets, scheduling barriers, B fragments fetched per group -- was slower, 0.42 ms against 0.33: every group then waits
        // for its own LDS reads.)
        auto chunk = [&](auto o_tag, int c) {
            constexpr int o = decltype(o_tag)::value;
            // chunk c's weights and B fragments (own pieces) have landed; younger: chunk c + 1's 4 + NP requests
            if (c + 1 < c1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 + NP) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (c == c0 && tid == 0) { asm volatile("" : "+v"(nx)); s_next[parity] = nx + (int)gridDim.x; }
            __builtin_amdgcn_s_barrier();                    // ... for every wave; and every wave is done with chunk c - 1's stage
            // split this lane's 16 weights: step A = values 0..7, step B = values 8..15
            bf16x8 af[2][NP];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned short p[3][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float wv = row_ok ? araw[o][2 * s + (e >> 2)][e & 3] : 0.f;
                    if (NP == 3) split3(wv, p[0][e], p[1][e], p[2][e]);
                    else split2h(wv * d.wscale, p[0][e], p[1][e]);
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    u32x4 w4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w4[e] = (unsigned)p[q][2 * e] | ((unsigned)p[q][2 * e + 1] << 16);
                    af[s][q] = __builtin_bit_cast(bf16x8, w4);
                }
            }
            if (c + 2 < c1) { load_a(c + 2, o_tag); issue_b(c + 2); }   // (stage (c + 2) % 3 held chunk c - 1)
            const unsigned stage = lds0 + (unsigned)((c % DP_NST) * DP_STAGE<NP> + lane * 16);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 bfr[4][NP];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < NP; ++q) bfr[j][q] = *(lds_cbf8*)(stage + (unsigned)((((q * 2 + s) * 4) + j) * 1024));
                constexpr int PA[6] = {NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0, 0}, PB[6] = {0, 1, NP == 3 ? 2 : 0, 0, 1, 0};   // (small products first)
#pragma unroll
                for (int pr = 0; pr < (NP == 3 ? 6 : 3); ++pr)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (NP == 3) tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][PA[pr]], bfr[j][PB[pr]], tq[j], 0, 0, 0);
                        else tq[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[s][PA[pr]]),
                                                                            __builtin_bit_cast(f16x8, bfr[j][PB[pr]]), tq[j], 0, 0, 0);
                    }
            }
            if (++fold == DP_FOLD || c + 1 == c1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] += tq[j];
#pragma unroll
                    for (int e = 0; e < 16; ++e) tq[j][e] = 0.f;
                }
                fold = 0;
            }
        };
        int c = c0;
        for (; c + 1 < c1; c += 2) {
            chunk(std::integral_constant<int, 0>(), c);
            chunk(std::integral_constant<int, 1>(), c + 1);
        }
        if (c < c1) chunk(std::integral_constant<int, 0>(), c);
        // ---- partials [k part][image][output]: accumulator register 4 q + e = row 8 q + 4 h + e of this wave's 32-row block ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = nt * DP_BN + 32 * j + r31;
            if (n >= d.N) continue;
            float* prow = part + ((size_t)ks * d.N + n) * d.OC + mt * DP_BM + 32 * wave + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mt * DP_BM + 32 * wave + 8 * q >= d.OC) continue;
                f32x4v v4 = {acc[j][4 * q] * d.oscale, acc[j][4 * q + 1] * d.oscale, acc[j][4 * q + 2] * d.oscale, acc[j][4 * q + 3] * d.oscale};
                *reinterpret_cast<f32x4v*>(prow + 8 * q) = v4;
            }
        }
        item = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
}

}  // namespace
#endif
