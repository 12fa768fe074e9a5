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


# ---- scaled fp16 pairs (vpk_cnn_set_algorithm(4), csrc/cnn_conv_pieces.hpp: split2h) --------------------------------------------

def split2h(x):
    """h0 = fp16(x), h1 = fp16(x - h0), as f64 (NumPy rounds float16 conversions to nearest even, like v_cvt_f16_f32)."""
    x = np.asarray(x, dtype=np.float32)
    h0 = x.astype(np.float16)
    h1 = (x - h0.astype(np.float32)).astype(np.float16)       # x - h0 is exact in f32
    return h0.astype(np.float64), h1.astype(np.float64)


def test_an_fp16_pair_holds_22_bits_of_an_f32_number_in_the_normal_window():
    """The pair misses x by at most max(2^-23 |x|, 2^-25): for |x| in [2^-2, 65504) -- the second piece's half step, 2^-25, is below
    2^-23 |x| there -- that is 2^-23 |x|, one bit short of an f32 number's own half-ulp."""
    rng = np.random.default_rng(3)
    x = (rng.choice([-1.0, 1.0], 200000) * rng.uniform(1.0, 2.0, 200000) * np.exp2(rng.integers(-2, 15, 200000))).astype(np.float32)
    h0, h1 = split2h(x)
    rel = np.abs(h0 + h1 - x.astype(np.float64)) / np.abs(x.astype(np.float64))
    assert rel.max() <= 2.0 ** -23
    # below the window the second piece is denormal: the ABSOLUTE error stays below half a denormal step, 2^-25
    small = (rng.uniform(0.0, 0.25, 100000)).astype(np.float32)
    h0, h1 = split2h(small)
    assert np.abs(h0 + h1 - small.astype(np.float64)).max() <= 2.0 ** -25


def test_three_partial_products_of_fp16_pairs_carry_the_f32_product():
    """h_i h_j is exact in f32 (11 x 11 bits); the three products with i + j <= 1 differ from a b by the dropped h1 h1' (|h1| <= 2^-11
    |x|: <= 2^-22 |a b|, typically 2^-24) and the two representation residues (<= 2^-23 each): within 8 x 2^-24 of the product in the
    worst case and 2^-24 in the median -- the size of ONE f32 rounding, where an FMA chain makes one per term."""
    rng = np.random.default_rng(4)
    def sample(n):
        return (rng.choice([-1.0, 1.0], n) * rng.uniform(1.0, 2.0, n) * np.exp2(rng.integers(0, 12, n))).astype(np.float32)
    a, b = sample(100000), sample(100000) / np.float32(4)        # (both inside the window [2^-2, 65504))
    (a0, a1), (b0, b1) = split2h(a), split2h(b)
    for p in (a0 * b0, a0 * b1, a1 * b0):
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)          # exact in f32
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(a0 * b0 + a0 * b1 + a1 * b0 - exact) / np.abs(exact)
    assert rel.max() <= 8 * 2.0 ** -24 and np.median(rel) <= 2.0 ** -24
    # a dot product of 1200 terms (conv2's K) in this arithmetic against a sequential f32 FMA chain, both against float64
    K, n = 1200, 2000
    x = rng.uniform(0.0, 3.0, (n, K)).astype(np.float32)
    w = (rng.standard_normal(K) * 0.03).astype(np.float32) * np.float32(2.0 ** 16)   # scaled like the library's weights
    (x0, x1), (w0, w1) = split2h(x), split2h(w)
    pairs = (x0 @ w0 + x0 @ w1 + x1 @ w0)
    ref = x.astype(np.float64) @ w.astype(np.float64)
    chain = np.zeros(n, np.float32)
    for k in range(K):
        chain = (chain.astype(np.float64) + x[:, k].astype(np.float64) * np.float64(w[k])).astype(np.float32)
    scale = np.abs(ref).max()
    assert np.abs(pairs - ref).max() / scale < np.abs(chain.astype(np.float64) - ref).max() / scale


def test_executed_flop_bookkeeping_of_the_default_configuration():
    """cnn.Net.executed_flop is bench.py's roofline numerator: under the defaults the piece layers execute three (conv1: three bf16)
    products per f32 product plus tile padding -- never fewer flops than 3 x the layer's algorithmic count, never more than 3.6 x."""
    import importlib.util
    import os
    import sys
    import types
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = types.ModuleType("vanishing_points_2017_amd")          # (the package's __init__ needs the GPU library: load cnn.py alone)
    pkg.__path__ = [os.path.join(here, "vanishing_points_2017_amd")]
    saved = sys.modules.get("vanishing_points_2017_amd")
    try:
        sys.modules["vanishing_points_2017_amd"] = pkg
        spec = importlib.util.spec_from_file_location("vanishing_points_2017_amd.cnn", os.path.join(here, "vanishing_points_2017_amd", "cnn.py"))
        try:
            cnn = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(cnn)
        except Exception:
            import pytest
            pytest.skip("cnn.py needs the runtime to import")
    finally:
        if saved is not None:
            sys.modules["vanishing_points_2017_amd"] = saved
        else:
            sys.modules.pop("vanishing_points_2017_amd", None)
    for layer in ("conv2", "conv3", "conv4", "conv5", "fc6", "fc7"):
        ex, pipe = cnn.Net.executed_flop(layer, fusion=3, precision=0, algorithm=4, batch=102)
        assert pipe == "f16"
        assert 3.0 <= ex / cnn.Net.LAYER_FLOP[layer] <= 3.8, (layer, ex / cnn.Net.LAYER_FLOP[layer])
    ex, pipe = cnn.Net.executed_flop("conv2", fusion=3, precision=0, algorithm=2, batch=102)
    assert pipe == "bf16" and 6.0 <= ex / cnn.Net.LAYER_FLOP["conv2"] <= 7.0
    assert cnn.Net.executed_flop("fc8", algorithm=4, batch=102)[1] == "f32"
