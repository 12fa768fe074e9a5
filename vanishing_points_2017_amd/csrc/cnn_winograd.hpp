// cnn_winograd.hpp -- conv3 / conv4 / conv5 of cnn/deploy.prototxt (:104-174: 3 x 3, stride 1, pad 1, 30 x 30 planes) by
// Winograd's minimal filtering F(2 x 2, 3 x 3) on the f32 matrix cores.  Included by vpk_cnn.hip (uses its DMA helpers).
//
// A 2 x 2 block of outputs of one channel is  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  with d the 4 x 4 input patch, g the 3 x 3
// filter and (.) the element-wise product summed over the input channels: 16 multiplications per input channel instead
// of 36 -- the three layers hold 53 % of the net's arithmetic, and the f32-input MFMA is the slowest matrix instruction
// of the part (1/16 of the bf16 rate), so the transforms' additions are cheap against the products they save.  Per
// position p = (i, j) of the transformed 4 x 4 tile the sum over input channels is a plain GEMM
//     M_p[oc][tile] = sum_ic U_p[oc][ic] V_p[ic][tile],      U = G g G^T (made once at load), V = B^T d B,
// sixteen of them.  All f32, accumulation in f32 like the direct kernel; the rounding differs from a direct sum (the
// products are of transformed operands): measured against the float64 evaluation of the same net in
// tests/test_gpu_cnn.py, like the split-bf16 path.
//
// One 512-thread workgroup per CU works on (64 output channels) x (64 tiles = 256 output pixels) x (all 16 positions):
//   * wave w owns the positions 2 w and 2 w + 1: two 64 x 64 accumulator tiles = 2 x 2 x 2 MFMA tiles of 32 x 32 (128 VGPRs);
//   * the K loop runs over chunks of 8 input channels.  U of the chunk (16 x 8 x 64 floats = 32 KB, stored in that order at
//     load) comes by LDS-DMA one chunk ahead; V of the chunk is MADE in the workgroup: thread (channel = wave, tile =
//     lane) fetches its 4 x 4 patch one chunk ahead (eight 8-byte loads straight from the bordered planes: the patch origin
//     of tile (ty, tx) is pixel (2 ty, 2 tx) of the padded plane), transforms it after the chunk's MFMAs and writes its 16
//     values to Vs[p][channel][tile] -- every workgroup of the same tiles redoes this for its 64 output channels, 32 additions
//     per 16 x 64 products;
//   * epilogue: the sixteen M_p of an (oc, tile) sit in eight different waves, so the accumulators go through LDS a quarter at
//     a time ([p][16 rows][64 tiles] = 64 KB over the idle U buffers), a thread gathers the 16 values of its (row, tile),
//     applies A^T . A, bias and ReLU and stores the 2 x 2 outputs into the consumer's bordered planes.
#ifndef VPK_CNN_WINOGRAD_HPP_
#define VPK_CNN_WINOGRAD_HPP_

namespace {

constexpr int WG_THREADS = 512;
constexpr int WG_TB = 64;            // tiles (of 2 x 2 outputs) per workgroup tile
constexpr int WG_OCB = 64;           // output channels per workgroup tile
constexpr int WG_KC = 8;             // input channels per LDS stage
constexpr int WG_TILES_PER_IMAGE = 15 * 15;

struct WinoDims {
    int IC, OC, groups;              // channels per group
    int ctot_in, ctot_out;           // channels of the input / output blob
    int tiles;                       // B * 225
    int ocblocks;                    // OC / 64 per group
    int chunks;                      // IC / 8
    int OHp, OWp, opad;              // output planes: (oh, ow) at (oh + opad, ow + opad) of an OHp x OWp plane
    int relu;
};

// U[g][oc block][chunk][p][kc][64] from Caffe's OIHW weights (host, at load): U_p = (G g G^T)_p in float64, rounded once
inline void winograd_weights(const float* w, int G, int OC, int IC, std::vector<float>& out) {
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int ocb = OC / WG_OCB, chunks = IC / WG_KC;
    out.assign((size_t)G * OC * IC * 16, 0.f);
    for (int g = 0; g < G; ++g)
        for (int oc = 0; oc < OC; ++oc)
            for (int ic = 0; ic < IC; ++ic) {
                const float* k = w + ((size_t)(g * OC + oc) * IC + ic) * 9;
                double t[4][3];
                for (int i = 0; i < 4; ++i)
                    for (int c = 0; c < 3; ++c) t[i][c] = Gm[i][0] * k[c] + Gm[i][1] * k[3 + c] + Gm[i][2] * k[6 + c];
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        const double u = t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2];
                        const size_t at = (((((size_t)g * ocb + oc / WG_OCB) * chunks + ic / WG_KC) * 16 + (i * 4 + j)) * WG_KC +
                                           ic % WG_KC) * WG_OCB + oc % WG_OCB;
                        out[at] = (float)u;
                    }
            }
}

#ifdef W5_TIME
// development: shader-clock breakdown of the kernel's phases per wave (scripts/w5_phase_times.py; a build with -DW5_TIME)
__device__ long long w5_dbg[256 * 12 * 8];
__device__ long long w3_dbg[256 * 8 * 8];       // the 3 x 3 kernel's (its last launch: conv5)
#define W5_LAP(slot) do { const long long now_ = __builtin_readcyclecounter(); tacc[slot] += now_ - tlast; tlast = now_; } while (0)
#else
#define W5_LAP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(WG_THREADS, 1) void conv3x3_winograd_kernel(WinoDims d, const float* __restrict__ in,
                                                                         const float* __restrict__ U,
                                                                         const float* __restrict__ bias,
                                                                         float* __restrict__ out, int* __restrict__ tile_counter,
                                                                         int total_tiles) {
    __shared__ __attribute__((aligned(16))) float Us[2][16][WG_KC][WG_OCB];     // 64 KB
    __shared__ __attribute__((aligned(16))) float Vs[2][16][WG_KC][WG_TB];      // 64 KB
    float (*Ms)[16][WG_TB] = reinterpret_cast<float (*)[16][WG_TB]>(&Us[0][0][0][0]);   // epilogue: [p][row][tile] over Us
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const unsigned us_base = lds_addr(&Us[0][0][0][0]);
    const int nblocks = (d.tiles + WG_TB - 1) / WG_TB;
    // persistent workgroups over a dynamic tile queue (the first gridDim.x tiles are static): beside another stream's kernel
    // that holds CUs (the EM), a static deal leaves the workgroups of the busy XCDs behind
    // (two slots used in turn, like the other persistent kernels: with one, thread 0 of a wave that has gone round could
    //  overwrite the index before a slower wave has read it -- no barrier separates that read from the next write)
    __shared__ int s_next[2];
    int parity = 0;
#ifdef W5_TIME
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#endif
    for (int tile = blockIdx.x; tile < total_tiles;) {
        if (tid == 0) s_next[parity] = atomicAdd(tile_counter, 1) + (int)gridDim.x;   // read after the K loop's barriers
        int t = tile;
        const int ob = t % d.ocblocks; t /= d.ocblocks;     // output-channel blocks of the same tiles next to each other (L2)
        const int tb = t % nblocks;
        const int g = t / nblocks;
        // ---- this thread's tile: the one it transforms (channel `wave` of every chunk) and stores results for ----
        const int n_raw = tb * WG_TB + lane;
        const bool n_ok = n_raw < d.tiles;
        const int n = n_ok ? n_raw : d.tiles - 1;
        const int b = n / WG_TILES_PER_IMAGE, rr0 = n - b * WG_TILES_PER_IMAGE;
        const int ty = rr0 / 15, tx = rr0 - ty * 15;
        const int b_first = (tb * WG_TB) / WG_TILES_PER_IMAGE;                   // scalar: the block's first image
        const float* in_base = in + ((size_t)b_first * d.ctot_in + (size_t)g * d.IC) * 1024;
        const unsigned poff = (unsigned)((((b - b_first) * d.ctot_in + wave) * 1024 + 2 * ty * 32 + 2 * tx) * 4);
        const float* u_base = U + ((size_t)(g * d.ocblocks + ob) * d.chunks) * (16 * WG_KC * WG_OCB);

        float2 raw[4][2];                               // the 4 x 4 patch of the NEXT chunk's channel
        auto fetch = [&](int c) {
            const char* p = (const char*)(in_base + (size_t)c * WG_KC * 1024) + poff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                raw[r][0] = *(const float2*)(p + r * 128);
                raw[r][1] = *(const float2*)(p + r * 128 + 8);
            }
        };
        auto transform = [&](int buf) {                 // V = B^T d B -> Vs[buf][p][wave][lane]
            float dd[4][4], tt[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { dd[r][0] = raw[r][0].x; dd[r][1] = raw[r][0].y; dd[r][2] = raw[r][1].x; dd[r][3] = raw[r][1].y; }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                tt[0][c] = dd[0][c] - dd[2][c];
                tt[1][c] = dd[1][c] + dd[2][c];
                tt[2][c] = dd[2][c] - dd[1][c];
                tt[3][c] = dd[1][c] - dd[3][c];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                Vs[buf][4 * i + 0][wave][lane] = tt[i][0] - tt[i][2];
                Vs[buf][4 * i + 1][wave][lane] = tt[i][1] + tt[i][2];
                Vs[buf][4 * i + 2][wave][lane] = tt[i][2] - tt[i][1];
                Vs[buf][4 * i + 3][wave][lane] = tt[i][1] - tt[i][3];
            }
        };
        auto issue_u = [&](int c, int buf) {            // 32 KB = 32 pieces of 1 KB; wave w moves pieces w, w + 8, w + 16, w + 24
            const float* src = u_base + (size_t)c * (16 * WG_KC * WG_OCB);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int piece = wave + 8 * q;
                dma16((unsigned)lane * 16u, src + piece * 256,
                      __builtin_amdgcn_readfirstlane(us_base + (unsigned)((buf * 16 * WG_KC * WG_OCB + piece * 256) * 4)));
            }
        };

        f32x16 acc[2][2][2];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[pp][i][j][r] = 0.f;

        W5_LAP(7);
        issue_u(0, 0);
        fetch(0);
        transform(0);
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        W5_LAP(3);
        for (int c = 0; c < d.chunks; ++c) {
            const int buf = c & 1;
            const bool more = c + 1 < d.chunks;
            if (more) { issue_u(c + 1, buf ^ 1); fetch(c + 1); }
            // operands of step k2 + 2 are requested before the MFMAs of step k2 (left alone the compiler puts every step's LDS
            // reads right in front of their use: one exposed LDS round trip per step)
            float af[2][2][2], bf[2][2][2];
            auto operands = [&](int k2) {
                const int o = (k2 >> 1) & 1;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int p = 2 * wave + pp;
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[o][pp][i] = Us[buf][p][k2 + khalf][i * 32 + l31];
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[o][pp][j] = Vs[buf][p][k2 + khalf][j * 32 + l31];
                }
            };
            operands(0);
#pragma unroll
            for (int k2 = 0; k2 < WG_KC; k2 += 2) {
                if (k2 + 2 < WG_KC) operands(k2 + 2);
                __builtin_amdgcn_sched_barrier(0);
                const int o = (k2 >> 1) & 1;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[pp][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[o][pp][i], bf[o][pp][j], acc[pp][i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            W5_LAP(1);
            if (more) transform(buf ^ 1);
            W5_LAP(0);
            wait_vmcnt<0>();                            // the next chunk's U has landed (own pieces) ...
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // ... for every wave; this chunk's buffers are free
            W5_LAP(2);
        }

        // ---- epilogue: Y = A^T M A per (output channel, tile), a quarter of the rows at a time through LDS ----
        const int oplane = d.OHp * d.OWp;
        float* obase = out + ((size_t)b * d.ctot_out + (size_t)g * d.OC + (size_t)ob * WG_OCB) * oplane +
                       (size_t)(2 * ty + d.opad) * d.OWp + 2 * tx + d.opad;
        const float* bbase = bias + g * d.OC + ob * WG_OCB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            Ms[2 * wave + pp][i * 8 + 4 * khalf + e][j * 32 + l31] = acc[pp][i][j][4 * r + e];
            W5_LAP(4);
            __syncthreads();
            W5_LAP(5);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int row = wave + 8 * u;           // id = tid + 512 u: row = id / 64, tile = lane
                float m[16];
#pragma unroll
                for (int p = 0; p < 16; ++p) m[p] = Ms[p][row][lane];
                float s[4][2];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    s[a][0] = m[4 * a] + m[4 * a + 1] + m[4 * a + 2];
                    s[a][1] = m[4 * a + 1] - m[4 * a + 2] - m[4 * a + 3];
                }
                const int ocl = (row >> 3) * 32 + 8 * r + (row & 7);
                const float bv = bbase[ocl];
                float y[2][2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    y[0][cc] = s[0][cc] + s[1][cc] + s[2][cc] + bv;
                    y[1][cc] = s[1][cc] - s[2][cc] - s[3][cc] + bv;
                }
                if (n_ok) {
                    float* o = obase + (size_t)ocl * oplane;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            float v = y[dy][dx];
                            if (d.relu) v = v > 0.f ? v : 0.f;
                            o[dy * d.OWp + dx] = v;
                        }
                }
            }
            W5_LAP(6);
            __syncthreads();
            W5_LAP(5);
        }
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
#ifdef W5_TIME
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 8; ++i) w3_dbg[(blockIdx.x * 8 + wave) * 8 + i] = tacc[i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv2 (deploy.prototxt:56-75: 5 x 5, stride 1, pad 2, two groups of 48 -> 128 channels, 61 x 61 planes) by F(2 x 2, 5 x 5):
// 36 instead of 100 products per 2 x 2 outputs and input channel.  Interpolation points 0, 1, -1, 2, -1/2, infinity; the rows
// of B^T are scaled to small integers (the inverse factors go into G, applied in float64 at load), so the input transform is
// exact multiplications and rounded additions:
//     B^T = [ 2  3 -4 -3  2  0 ]    G = diag(1/2, 1/6, 1/6, 1/30, 16/15, 1/2) . [ 1    0    0    0     0   ]    A^T = [ 1  1  1  1   1   0 ]
//           [ 0  2  5  1 -2  0 ]                                                [ 1    1    1    1     1   ]          [ 0  1 -1  2 -1/2  1 ]
//           [ 0  2  1 -5  2  0 ]                                                [ 1   -1    1   -1     1   ]
//           [ 0 -1 -2  1  2  0 ]                                                [ 1    2    4    8    16   ]
//           [ 0 -2  1  2 -1  0 ]                                                [ 1  -1/2  1/4 -1/8  1/16  ]
//           [ 0  2  3 -4 -3  2 ]                                                [ 0    0    0    0     1   ]
// (f32 error against float64, measured in NumPy on this layer's data before the kernel was written: 7e-7 of the blob's
// scale, the direct f32 sum over K = 1200: 8e-7.)  Same structure as the 3 x 3 kernel above with other sizes: 768 threads
// (twelve waves: three of the 36 positions each), 64 output channels x 32 tiles per workgroup tile, chunks of 4 input
// channels (U of a chunk = 36 KB by LDS-DMA).  The 6 x 6 input transform is done by ALL threads -- VALU work and f32 MFMAs
// of one SIMD do not overlap, so a transform left to a few waves would make their SIMDs the stragglers -- in two passes:
// columns (B^T d, straight from the loaded patch column), then rows ((.) B).  The six threads of a (tile, channel) pair
// are six consecutive lanes of ONE wave (ten pairs per wave: 4 channels x 30 tiles = 120 pairs on twelve waves -- which
// is why a workgroup tile has 30 tiles, two of the MFMA tile's 32 columns stay empty), so the hand-over between the
// passes is a wave-private LDS exchange without a workgroup barrier and the K loop has ONE barrier per chunk (the first
// version -- 32 tiles, passes on different waves, two barriers per chunk -- kept the matrix pipes busy 38 % of the time,
// its waves waiting 43 % of their cycles: profiles/r04_pmc_mfma.json).  The last
// tile row / column (outputs 60, "61") reads input rows / columns up to 65 of a 65-pixel plane: the index is clamped to
// 64, which IS the zero border the missing pixel would hold.
constexpr int W5_THREADS = 768;
constexpr int W5_TB = 30;            // tiles per workgroup tile (of the 32 columns of an MFMA tile: see the transform's lane map)
constexpr int W5_NB = 32;            // columns of the B operand
constexpr int W5_OCB = 64;
constexpr int W5_KC = 4;
constexpr int W5_P = 36;
constexpr int W5_TPI = 31 * 31;      // tiles per image (61 x 61 outputs)
constexpr int W5_PLANE = 65 * 65;

// U[g][oc block][chunk][p][kc][64] for conv2 (host, at load; float64, rounded once)
inline void winograd5_weights(const float* w, int G, int OC, int IC, std::vector<float>& out) {
    static const double sc[6] = {0.5, 1.0 / 6, 1.0 / 6, 1.0 / 30, 16.0 / 15, 0.5};
    static const double E[6][5] = {{1, 0, 0, 0, 0}, {1, 1, 1, 1, 1}, {1, -1, 1, -1, 1}, {1, 2, 4, 8, 16},
                                   {1, -0.5, 0.25, -0.125, 0.0625}, {0, 0, 0, 0, 1}};
    const int ocb = OC / W5_OCB, chunks = IC / W5_KC;
    out.assign((size_t)G * OC * IC * W5_P, 0.f);
    for (int g = 0; g < G; ++g)
        for (int oc = 0; oc < OC; ++oc)
            for (int ic = 0; ic < IC; ++ic) {
                const float* k = w + ((size_t)(g * OC + oc) * IC + ic) * 25;
                double t[6][5];
                for (int i = 0; i < 6; ++i)
                    for (int c = 0; c < 5; ++c) {
                        double a = 0;
                        for (int r = 0; r < 5; ++r) a += sc[i] * E[i][r] * k[r * 5 + c];
                        t[i][c] = a;
                    }
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) {
                        double u = 0;
                        for (int c = 0; c < 5; ++c) u += t[i][c] * sc[j] * E[j][c];
                        const size_t at = (((((size_t)g * ocb + oc / W5_OCB) * chunks + ic / W5_KC) * W5_P + (i * 6 + j)) * W5_KC +
                                           ic % W5_KC) * W5_OCB + oc % W5_OCB;
                        out[at] = (float)u;
                    }
            }
}

// B^T x for one vector of six, with the common differences shared (18 VALU operations instead of 26: the transform's
// VALU time is a quarter of the kernel's)
__device__ __forceinline__ void w5_bt(const float (&x)[6], float (&o)[6]) {
    const float a = x[3] - x[1], b = x[4] - x[2];
    o[0] = fmaf(2.f, x[0] + x[4], fmaf(-3.f, a, -4.f * x[2]));       //  2 x0 + 3 x1 - 4 x2 - 3 x3 + 2 x4
    o[1] = fmaf(2.f, x[1] - x[4], fmaf(5.f, x[2], x[3]));            //  2 x1 + 5 x2 +   x3 - 2 x4
    o[2] = fmaf(2.f, x[1] + x[4], fmaf(-5.f, x[3], x[2]));           //  2 x1 +   x2 - 5 x3 + 2 x4
    o[3] = fmaf(2.f, b, a);                                          //   -x1 - 2 x2 +   x3 + 2 x4
    o[4] = fmaf(2.f, a, -b);                                         // -2 x1 +   x2 + 2 x3 -   x4
    o[5] = fmaf(2.f, x[1] + x[5], fmaf(-3.f, b, -4.f * x[3]));       //  2 x1 + 3 x2 - 4 x3 - 3 x4 + 2 x5
}

struct Wino5Dims {
    int IC, OC, groups, ctot_in, ctot_out;
    int tiles;                       // B * 961
    int ocblocks, chunks;
    int relu;
};

__global__ __launch_bounds__(W5_THREADS, 1) void conv5x5_winograd_kernel(Wino5Dims d, const float* __restrict__ in,
                                                                         const float* __restrict__ U,
                                                                         const float* __restrict__ bias,
                                                                         float* __restrict__ out, int* __restrict__ tile_counter,
                                                                         int total_tiles) {
    __shared__ __attribute__((aligned(16))) float Us[2][W5_P][W5_KC][W5_OCB];    // 72 KB
    __shared__ __attribute__((aligned(16))) float Vs[2][W5_P][W5_KC][W5_NB];     // 36 KB
    __shared__ __attribute__((aligned(16))) float Sw[12][10][6][7];              // 20 KB: per wave, B^T d of its ten pairs [pair][i][column] (row stride 7)
    float (*Ms)[16][W5_NB] = reinterpret_cast<float (*)[16][W5_NB]>(&Us[0][0][0][0]);   // epilogue: [p][row][tile] = 72 KB over Us
    __shared__ int s_next[2];                          // two slots used in turn (see the 3 x 3 kernel)
    int parity = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    // transform role: lane = 6 q + tc: pair q (0..9) of this wave, column (pass 1) / row (pass 2) tc; lanes 60..63 idle
    const bool t_on = lane < 60;
    const int tq = t_on ? lane / 6 : 9, tc = t_on ? lane - 6 * (lane / 6) : 5;
    const int tpair = 10 * wave + tq;                   // 0..119 = channel * 30 + tile
    const int tch = tpair / W5_TB, ttile = tpair - tch * W5_TB;
    const unsigned us_base = lds_addr(&Us[0][0][0][0]);
    const int nblocks = (d.tiles + W5_TB - 1) / W5_TB;
    for (int q = tid; q < 2 * W5_P * W5_KC * W5_NB; q += W5_THREADS) (&Vs[0][0][0][0])[q] = 0.f;   // (columns 30, 31 stay zero)
    __syncthreads();
#ifdef W5_TIME
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tlast = __builtin_readcyclecounter();
#endif
    for (int tile = blockIdx.x; tile < total_tiles;) {
        if (tid == 0) s_next[parity] = atomicAdd(tile_counter, 1) + (int)gridDim.x;     // read after the K loop's barriers
        int t = tile;
        const int ob = t % d.ocblocks; t /= d.ocblocks;
        const int tb = t % nblocks;
        const int g = t / nblocks;
        const int b_first = (tb * W5_TB) / W5_TPI;
        const float* in_base = in + ((size_t)b_first * d.ctot_in + (size_t)g * d.IC) * W5_PLANE;
        unsigned doff[6];                               // byte offsets of this lane's patch column tc, rows 0..5 (clamped into the plane)
        {
            int n = tb * W5_TB + ttile;
            n = n < d.tiles ? n : d.tiles - 1;
            const int b = n / W5_TPI, rr0 = n - b * W5_TPI;
            const int ty = rr0 / 31, tx = rr0 - ty * 31;
            const int col = 2 * tx + tc < 64 ? 2 * tx + tc : 64;
            const unsigned base = (unsigned)(((b - b_first) * d.ctot_in + tch) * W5_PLANE + col);
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                const int row = 2 * ty + r < 64 ? 2 * ty + r : 64;
                doff[r] = (base + (unsigned)(row * 65)) * 4u;
            }
        }
        const float* u_base = U + ((size_t)(g * d.ocblocks + ob) * d.chunks) * (W5_P * W5_KC * W5_OCB);
        float dcol[6];                                  // the patch column of the next chunk
        auto fetch = [&](int c) {
            const char* p = (const char*)(in_base + (size_t)c * W5_KC * W5_PLANE);
#pragma unroll
            for (int r = 0; r < 6; ++r) dcol[r] = *(const float*)(p + doff[r]);
        };
        auto transform = [&](int buf) {                 // V = B^T d B of this wave's ten pairs -> Vs[buf][p][channel][tile]
            float o[6], x[6];
            w5_bt(dcol, o);                             // column tc of B^T d
#pragma unroll
            for (int i = 0; i < 6; ++i) Sw[wave][tq][i][tc] = o[i];
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();            // (the LDS serves a wave's accesses in issue order: no wait needed)
#pragma unroll
            for (int c = 0; c < 6; ++c) x[c] = Sw[wave][tq][tc][c];     // row tc of B^T d
            w5_bt(x, o);
            if (t_on) {
#pragma unroll
                for (int j = 0; j < 6; ++j) Vs[buf][6 * tc + j][tch][ttile] = o[j];
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_wave_barrier();            // (the next call's writes to Sw come after these reads)
        };
        auto issue_u = [&](int c, int buf) {            // 36 KB = 36 pieces of 1 KB: wave w moves pieces w, w + 12, w + 24
            const float* src = u_base + (size_t)c * (W5_P * W5_KC * W5_OCB);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int piece = wave + 12 * q;
                dma16((unsigned)lane * 16u, src + piece * 256,
                      __builtin_amdgcn_readfirstlane(us_base + (unsigned)((buf * W5_P * W5_KC * W5_OCB + piece * 256) * 4)));
            }
        };
        f32x16 acc[3][2];
#pragma unroll
        for (int pp = 0; pp < 3; ++pp)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pp][i][r] = 0.f;

        W5_LAP(7);
        issue_u(0, 0);
        fetch(0);
        transform(0);
        if (d.chunks > 1) fetch(1);
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        W5_LAP(3);
        for (int c = 0; c < d.chunks; ++c) {
            const int buf = c & 1;
            const bool more = c + 1 < d.chunks;
            if (more) issue_u(c + 1, buf ^ 1);
            float af[2][3][2], bf[2][3];
            auto operands = [&](int k2) {
                const int o = (k2 >> 1) & 1;
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) {
                    const int p = 3 * wave + pp;
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[o][pp][i] = Us[buf][p][k2 + khalf][i * 32 + l31];
                    bf[o][pp] = Vs[buf][p][k2 + khalf][l31];
                }
            };
            // (Tried: the three waves of a SIMD transforming at different points of the chunk -- before, between, after the
            // MFMAs.  No gain: a wave's VALU instructions queue behind the other waves' 64-cycle matrix instructions either
            // way -- the hardware serves the older waves' MFMAs first, so the waves drift apart by themselves -- and the
            // transform costs 26 % of a wave's time against 36 % for the MFMA phase, scripts/w5_phase_times.py.)
            auto next_v = [&]() {                       // (dcol holds chunk c + 1; Vs[buf ^ 1] was last read in chunk c - 1)
                W5_LAP(1);
                if (more) transform(buf ^ 1);
                if (c + 2 < d.chunks) fetch(c + 2);
                W5_LAP(0);
            };
            operands(0);
#pragma unroll
            for (int k2 = 0; k2 < W5_KC; k2 += 2) {
                if (k2 + 2 < W5_KC) operands(k2 + 2);
                __builtin_amdgcn_sched_barrier(0);
                const int o = (k2 >> 1) & 1;
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        acc[pp][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[o][pp][i], bf[o][pp], acc[pp][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            next_v();
            W5_LAP(1);
            wait_vmcnt<0>();                            // the next chunk's U has landed (own pieces) ...
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // ... for every wave; this chunk's buffers are free
            W5_LAP(2);
        }

        // ---- epilogue: Y = A^T M A per (output channel, tile), a quarter of the rows at a time through LDS ----
        const int row = tid >> 5;                       // threads 0..511: (row 0..15, tile l31)
        const int n_raw = tb * W5_TB + l31;
        const bool n_ok = l31 < W5_TB && n_raw < d.tiles;
        const int n = n_ok ? n_raw : d.tiles - 1;
        const int b = n / W5_TPI, rr0 = n - b * W5_TPI;
        const int ty = rr0 / 31, tx = rr0 - ty * 31;
        const bool last_y = 2 * ty + 1 >= 61, last_x = 2 * tx + 1 >= 61;
        float* obase = out + ((size_t)b * d.ctot_out + (size_t)g * d.OC + (size_t)ob * W5_OCB) * 3721 + (size_t)(2 * ty) * 61 + 2 * tx;
        const float* bbase = bias + g * d.OC + ob * W5_OCB;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Ms[3 * wave + pp][i * 8 + 4 * khalf + e][l31] = acc[pp][i][4 * r + e];
            W5_LAP(4);
            __syncthreads();
            W5_LAP(5);
            if (tid < 512) {
                float s[6][2];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    float m[6];
#pragma unroll
                    for (int q = 0; q < 6; ++q) m[q] = Ms[6 * a + q][row][l31];
                    s[a][0] = m[0] + m[1] + m[2] + m[3] + m[4];
                    s[a][1] = m[1] - m[2] + 2.f * m[3] - 0.5f * m[4] + m[5];
                }
                const int ocl = (row >> 3) * 32 + 8 * r + (row & 7);
                const float bv = bbase[ocl];
                float y[2][2];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    y[0][cc] = s[0][cc] + s[1][cc] + s[2][cc] + s[3][cc] + s[4][cc] + bv;
                    y[1][cc] = s[1][cc] - s[2][cc] + 2.f * s[3][cc] - 0.5f * s[4][cc] + s[5][cc] + bv;
                }
                if (n_ok) {
                    float* o = obase + (size_t)ocl * 3721;
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 2; ++dx) {
                            float v = y[dy][dx];
                            if (d.relu) v = v > 0.f ? v : 0.f;
                            if (!(dy && last_y) && !(dx && last_x)) o[dy * 61 + dx] = v;
                        }
                }
            }
            W5_LAP(6);
            __syncthreads();
            W5_LAP(5);
        }
        tile = __builtin_amdgcn_readfirstlane(s_next[parity]);
        parity ^= 1;
    }
#ifdef W5_TIME
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 8; ++i) w5_dbg[(blockIdx.x * 12 + wave) * 8 + i] = tacc[i];
#endif
}


}  // namespace
#endif
