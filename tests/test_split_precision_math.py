"""The arithmetic behind vpk_cnn_set_precision(1) (csrc/cnn_split_gemm.hpp), checked in NumPy on the CPU: an f32 number
is exactly the sum of three bf16 numbers (each the round-to-nearest bf16 of what is left), products of bf16 numbers are
exact in f32, and the six partial products with i + j <= 4 reproduce the f32 product to within the rounding an f32 FMA
makes anyway (2^-24 of the product; with truncated pieces it would be 2^-21).  The HIP
kernels themselves are compared with a float64 evaluation of the net in tests/test_gpu_cnn.py."""
import numpy as np


def _trunc_bf16(x):
    """The bf16 number below |x| (keeps the top 16 bits of the f32 word), as f32."""
    return (np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)


def _rne_bf16(x):
    """Round-to-nearest-even bf16 of x, as f32 (the bit trick of csrc/cnn_split_gemm.hpp: bf16_rne_bits)."""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((b + 0x7fff + ((b >> 16) & 1)) & 0xffff0000).astype(np.uint32)).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    p0 = _rne_bf16(x)
    r1 = x - p0                      # exact in f32: |r1| <= half a bf16 ulp of x, a multiple of x's f32 ulp
    p1 = _rne_bf16(r1)
    r2 = r1 - p1                     # exact, and <= 8 significant bits: already a bf16 number
    return p0, p1, r2


def _samples(rng, n):
    mant = rng.uniform(1.0, 2.0, n).astype(np.float32)
    expo = rng.integers(-20, 20, n)
    sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
    return (sign * mant * np.exp2(expo)).astype(np.float32)


def test_three_truncated_bf16_pieces_add_up_exactly():
    rng = np.random.default_rng(1)
    x = np.concatenate([_samples(rng, 200000), np.float32([0.0, 1.0, -1.0, 1.7e38, 1e-30, np.float32(1) / 3])])
    p0, p1, p2 = split3(x)
    for p in (p0, p1, p2):
        assert np.array_equal(_trunc_bf16(p), p)                       # each piece IS a bf16 number
    assert np.array_equal((p0.astype(np.float64) + p1.astype(np.float64)) + p2.astype(np.float64), x.astype(np.float64))


def test_six_partial_products_carry_the_f32_product():
    rng = np.random.default_rng(2)
    a, b = _samples(rng, 100000), _samples(rng, 100000)
    pa, pb = split3(a), split3(b)
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = np.zeros_like(exact)
    for i in range(3):
        for j in range(3):
            prod = pa[i].astype(np.float64) * pb[j].astype(np.float64)
            assert np.array_equal(prod.astype(np.float32).astype(np.float64), prod)   # bf16 x bf16 is exact in f32
            if i + j <= 2:                                             # pieces 0-based: the six with i + j <= 2
                six += prod
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() <= 2.0 ** -24                                    # the three dropped terms
    f32_rounding = np.abs((a * b).astype(np.float64) - exact) / np.abs(exact)
    assert np.median(rel) <= np.median(f32_rounding)                  # typically smaller than f32's own product rounding


def test_dot_products_match_f32_accumulation():
    """A K = 2400 dot product (conv2's depth) summed in f32: the split form's distance to the float64 result is of the
    size of the plain f32 form's."""
    rng = np.random.default_rng(3)
    err_plain, err_split = [], []
    for _ in range(200):
        a = rng.standard_normal(2400).astype(np.float32) * 0.05
        b = np.maximum(rng.standard_normal(2400), 0).astype(np.float32) * 3
        exact = float(a.astype(np.float64) @ b.astype(np.float64))
        plain = np.float32(0)
        for k in range(0, 2400, 16):                                   # f32 accumulation in chunks, like the matrix pipe
            plain = np.float32(plain + np.float32((a[k:k + 16].astype(np.float64) * b[k:k + 16]).sum()))
        pa, pb = split3(a), split3(b)
        acc = np.float32(0)
        for k in range(0, 2400, 16):
            for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
                acc = np.float32(acc + np.float32((pa[i][k:k + 16].astype(np.float64) * pb[j][k:k + 16]).sum()))
        err_plain.append(abs(float(plain) - exact))
        err_split.append(abs(float(acc) - exact))
    assert np.mean(err_split) <= 3.0 * np.mean(err_plain) + 1e-9
