// cnn_pairs.hpp -- the scaled fp16 PAIR of an f32 number and its guarded form, shared by every kernel that writes pairs
// (cnn_conv1_pieces.hpp's pooling stage, cnn_conv_pieces.hpp, cnn_norm_pool_planes.hpp, cnn_dense_pieces.hpp).  The arithmetic is
// described in cnn_conv_pieces.hpp; included by vpk_cnn.hip first.
#ifndef VPK_CNN_PAIRS_HPP_
#define VPK_CNN_PAIRS_HPP_

namespace {

constexpr float CP_DEFAULT_ASCALE = 0.125f;

__device__ __forceinline__ void split2h(float x, unsigned short& h0, unsigned short& h1) {
    const _Float16 a = (_Float16)x;
    const _Float16 b = (_Float16)(x - (float)a);
    h0 = __builtin_bit_cast(unsigned short, a);
    h1 = __builtin_bit_cast(unsigned short, b);
}

// The GUARDED split every activation writer uses (round 6).  fp16's largest finite number is 65 504: a scaled activation beyond it would
// become h0 = inf, h1 = x - inf = -inf and the next layer's products NaN -- silently.  Instead the value is clamped to +-65 504 (NaN too:
// fmax / fmin return the other operand) and `bad` remembers it; the kernel ORs the consuming layer's bit into the handle's range word
// (range_report), which vpk_cnn_range_flags reads: a net whose activations leave the calibrated range is an ERROR the caller sees
// (VPK_ERR_RANGE), never a response map of NaNs.  Cost: a compare, a scalar OR and a v_med3 per stored value, in the epilogues only.
constexpr float CP_H_MAX = 65504.f;
__device__ __forceinline__ void split2h_guard(float x, unsigned short& h0, unsigned short& h1, bool& bad) {
    bad |= !(__builtin_fabsf(x) < CP_H_MAX);
    split2h(__builtin_fminf(__builtin_fmaxf(x, -CP_H_MAX), CP_H_MAX), h0, h1);
}
__device__ __forceinline__ void range_report(bool bad, unsigned* __restrict__ range_word, unsigned bit) {
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(range_word, bit);
}

}  // namespace
#endif
