// cnn_norm_pool_planes.hpp -- norm2 + pool2 (cnn/deploy.prototxt:82-103: LRN across 5 channels, alpha 1e-4, beta 0.75; MAX pool 3 x 3 /
// stride 2) as a stream over the channels, writing conv3's input directly in the format conv_pieces_kernel reads: scaled fp16 pairs
// in piece planes (cnn_conv_pieces.hpp) -- or the f32 planes when a caller taps pool2.  Included by vpk_cnn.hip after
// cnn_conv_pieces.hpp (split2h).
//
// Same walk as lrn5_pool3s2_stream_kernel (vpk_cnn.hip): a workgroup owns TPH pooled rows x the whole width of one image -- in an
// unpadded NCHW plane one contiguous run of (2 TPH + 1) W floats per channel -- and a range of channels; a thread keeps the 5-deep
// raw window of its (up to four) pixels in registers, every raw value is read once, fully coalesced; the normalised planes of a
// batch of channels go to LDS (double buffered, one barrier per batch).  What differs (round 5):
//   * a batch is EIGHT channels = one 16-byte word of a piece plane, and the pooling phase is one thread per pooled PIXEL, eight
//     channels each: its row / column / addresses are computed once per workgroup, not per output and batch (the old kernel spent
//     ~55 VALU instructions per input element, most of them index arithmetic of the pooling phase);
//   * LDS rows are W + 1 floats so that a window row is one aligned 8-byte read + one 4-byte read (6 reads per window, not 9);
//   * no window is clipped at this shape (61 -> 30: 2 * 29 + 2 = 60; checked by the host), so the pooling loop has no bounds tests;
//   * LRN expressions and their order are lrn5_pool3s2_stream_kernel's: the same f32 bits go into the maximum.
#ifndef VPK_CNN_NORM_POOL_PLANES_HPP_
#define VPK_CNN_NORM_POOL_PLANES_HPP_

namespace {

template <int TPH>
__global__ __launch_bounds__(256) void lrn5_pool3s2_planes_kernel(const float* __restrict__ in, float* __restrict__ out_f32,
                                                                  unsigned short* __restrict__ out_planes, int C, int H, int W, int PH,
                                                                  int PW, float alpha, int PHp, int PWp, int opad, int cgroups, float ascale,
                                                                  unsigned* __restrict__ range_word, unsigned range_bit) {
    // PMAX: floats per normalised plane in LDS.  832 (>= TR x (W + 1) = 13 x 62 at this shape; it was 1024 = the thread slots) makes the
    // double-buffered batch 53 KB, so that THREE workgroups share a CU instead of two (round 6)
    // (four per CU with tiles of four pooled rows, 37 KB: no faster, 0.126 against 0.124 ms)
    constexpr int CB = 8, TR = 2 * TPH + 1, SLOTS = 4, PMAX = 832;
    __shared__ __attribute__((aligned(16))) float plane[2][CB][PMAX];
    const int tiles_h = (PH + TPH - 1) / TPH;
    const int th = blockIdx.x % tiles_h, cgi = (blockIdx.x / tiles_h) % cgroups, b = blockIdx.x / (tiles_h * cgroups);
    const int cper = C / cgroups, c_lo = cgi * cper, c_hi = c_lo + cper;      // this workgroup's channels [c_lo, c_hi): multiples of 8
    const int ph0 = th * TPH, h0 = 2 * ph0;
    const int HW = H * W, npix = TR * W, LW = W + 1;     // npix <= PMAX, TR * LW <= PMAX (checked by the host)
    const float* x = in + (size_t)b * C * HW + (size_t)h0 * W;
    bool ok[SLOTS];
    int ld_off[SLOTS], st_off[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int off = threadIdx.x + 256 * i;
        const int r = off / W, q = off - r * W;
        ok[i] = off < npix && h0 + r < H;                // (rows past the blob: zeros)
        ld_off[i] = ok[i] ? off : 0;
        st_off[i] = off < npix ? r * LW + q : -1;
    }
    float v0[SLOTS], v1[SLOTS], v2[SLOTS], v3[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {                    // raw values of the channels c_lo - 2 .. c_lo + 1 (zeros outside the blob)
        v0[i] = (ok[i] && c_lo >= 2) ? x[(size_t)(c_lo - 2) * HW + ld_off[i]] : 0.f;
        v1[i] = (ok[i] && c_lo >= 1) ? x[(size_t)(c_lo - 1) * HW + ld_off[i]] : 0.f;
        v2[i] = ok[i] ? x[(size_t)c_lo * HW + ld_off[i]] : 0.f;
        v3[i] = (ok[i] && c_lo + 1 < C) ? x[(size_t)(c_lo + 1) * HW + ld_off[i]] : 0.f;
    }
    // pooling role: thread t < TPH * PW owns pooled pixel (t / PW, t % PW) of the tile
    const bool p_on = (int)threadIdx.x < TPH * PW && ph0 + (int)threadIdx.x / PW < PH;
    const int oy = p_on ? (int)threadIdx.x / PW : 0, ox = p_on ? (int)threadIdx.x % PW : 0;
    const float* win = &plane[0][0][2 * oy * LW + 2 * ox];                     // (even offset: LW * 2 oy + 2 ox)
    const size_t out_pix = (size_t)(ph0 + oy + opad) * PWp + ox + opad;
    const float an = alpha / 5.f;
    int buf = 0;
    bool bad = false;                                    // a scaled value beyond fp16's range (split2h_guard)
    float nx[CB][SLOTS], nn[CB][SLOTS];                  // raw values of this batch's / the next batch's channels (+2)
    auto fetch = [&](int cb, float (&dst)[CB][SLOTS]) {
        const int cmax = C - 1;
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const int c4 = cb + k + 2;
            const float* row = x + (size_t)(c4 < cmax ? c4 : cmax) * HW;       // (wave-uniform; clamped: masked when used)
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) dst[k][i] = row[ld_off[i]];
        }
    };
    fetch(c_lo, nx);
    for (int cb = c_lo; cb < c_hi; cb += CB) {
        if (cb + CB < c_hi) fetch(cb + CB, nn);          // the next batch's loads are in flight under this batch's work
#pragma unroll
        for (int k = 0; k < CB; ++k)
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) {
                const float v4 = (ok[i] && cb + k + 2 < C) ? nx[k][i] : 0.f;
                const float sc = 1.f + an * (v0[i] * v0[i] + v1[i] * v1[i] + v2[i] * v2[i] + v3[i] * v3[i] + v4 * v4);
                const float r = __builtin_amdgcn_rsqf(sc);                     // sc^-0.75 = rsq(sc) * sqrt(rsq(sc)) (1 ulp, sc >= 1)
                if (st_off[i] >= 0) plane[buf][k][st_off[i]] = v2[i] * (r * __builtin_amdgcn_sqrtf(r));
                v0[i] = v1[i]; v1[i] = v2[i]; v2[i] = v3[i]; v3[i] = v4;
            }
#pragma unroll
        for (int k = 0; k < CB; ++k)
#pragma unroll
            for (int i = 0; i < SLOTS; ++i) nx[k][i] = nn[k][i];
        __syncthreads();
        if (p_on) {
            float m[CB];
            const float* w0 = win + buf * (CB * PMAX);
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                float mm = 0.f;                          // (every value is >= 0: ReLU, then a positive factor)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const f32x2v a = *reinterpret_cast<const f32x2v*>(w0 + k * PMAX + dy * LW);
                    const float c = w0[k * PMAX + dy * LW + 2];
                    mm = __builtin_fmaxf(mm, __builtin_fmaxf(__builtin_fmaxf(a[0], a[1]), c));
                }
                m[k] = mm;
            }
            if (out_planes) {                            // conv3's input: word (channel group of 16, piece x k half, y, x)
                unsigned short h0_[CB], h1_[CB];
#pragma unroll
                for (int k = 0; k < CB; ++k) split2h_guard(m[k] * ascale, h0_[k], h1_[k], bad);
                u32x4 a, c;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a[e] = (unsigned)h0_[2 * e] | ((unsigned)h0_[2 * e + 1] << 16);
                    c[e] = (unsigned)h1_[2 * e] | ((unsigned)h1_[2 * e + 1] << 16);
                }
                const size_t wpl = (size_t)PHp * PWp;
                u32x4* dst = reinterpret_cast<u32x4*>(out_planes) + (((size_t)b * (C >> 4) + (cb >> 4)) * 4 + ((cb >> 3) & 1)) * wpl + out_pix;
                dst[0] = a;
                dst[2 * wpl] = c;
            } else {
#pragma unroll
                for (int k = 0; k < CB; ++k) out_f32[((size_t)b * C + cb + k) * PHp * PWp + out_pix] = m[k];
            }
        }
        buf ^= 1;
    }
    if (out_planes) range_report(bad, range_word, range_bit);
}

}  // namespace
#endif
