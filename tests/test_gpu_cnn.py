"""HIP AlexNet-500 forward vs the torch-CPU fp32 restatement (oracle/cnn_torch.py), seeded
synthetic weights.  Tolerance (fp32 parity mode): per-layer max abs error <= 2e-4 * (1 + max|ref|),
final 20x20 sigmoid map <= 2e-5 abs.  Different fp32 summation orders (MFMA k-order vs MKL-DNN)
are the only source of difference."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def net_and_ref():
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, synth
    w = cnn.synthetic_weights(3)
    mean = cnn.synthetic_mean(3)
    from vanishing_points_2017_amd import sphere_mapping
    sphere = sphere_mapping.raster_batch([synth.make_scene(7000 + i, 150 + 40 * i, 3)["l"] for i in range(3)])
    ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True)
    net = cnn.Net(w, mean)
    return net, sphere, ref, taps


@pytest.mark.parametrize("tap", list(range(11)))
def test_layer_outputs(net_and_ref, tap):
    from oracle import cnn_torch
    net, sphere, ref, taps = net_and_ref
    out, got = net.forward(sphere, tap=tap)
    want = taps[cnn_torch.TAPS[tap]].reshape(got.shape)
    scale = 1.0 + np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= 2e-4 * scale, (cnn_torch.TAPS[tap], err, scale)
    assert np.abs(out - ref).max() <= 2e-5


def test_batch_invariance_and_single_image_call(net_and_ref):
    from vanishing_points_2017_amd import cnn
    net, sphere, ref, _ = net_and_ref
    full = net.forward(sphere)
    assert np.abs(full - ref).max() <= 2e-5
    one = cnn.caffe_forward(net, sphere[1])
    assert one.shape == (20, 20) and one.dtype == np.float32
    assert np.abs(one - full[1]).max() <= 1e-6        # batch composition does not change a result
    big = net.forward(np.concatenate([sphere] * 5)[:13])
    assert np.array_equal(big[:3], full)              # deterministic: same bits for the same image


@pytest.mark.parametrize("batch", [13, 102])
def test_bench_batch_against_the_oracle(batch):
    """B = 102 is bench.py's batch (other tile tails than B = 3, the persistent tile queue goes round more than
    once, split-K partials of the dense layers span 102 rows); B = 13 an odd size.  The rasters are bench.py's own."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    from vanishing_points_2017_amd import sphere_mapping
    sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=batch)])
    ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True)
    net = cnn.Net(w, mean)
    try:
        for mode in (0, 1):                                # native f32 matrix path; split-bf16 convolutions (other tiles:
            net.set_precision(mode)                        #  256-column tiles, 4- and 8-wave workgroups, chained epilogues)
            for tap in (1, 2, 7, 8):                       # pool1, conv2, pool5, fc6
                out, got = net.forward(sphere, tap=tap)
                want = taps[cnn_torch.TAPS[tap]].reshape(got.shape)
                assert np.abs(got - want).max() <= 2e-4 * (1.0 + np.abs(want).max()), (mode, cnn_torch.TAPS[tap])
                assert np.abs(out - ref).max() <= 2e-5, mode
    finally:
        net.set_precision(0)
    assert np.abs(ref - 0.5).max() > 1e-3                  # the response maps are not a constant


def test_fused_first_stage_equals_separate_kernels(net_and_ref):
    """conv1 + norm1 + pool1 as one kernel (2-D patches, LRN and pooling out of LDS) against the separate conv1 and
    LRN/pool kernels: the same K order and the same LRN expression, so pool1 agrees to rounding -- including
    the clipped windows at the right / bottom border (61 = 7 x 8 + 5 columns, 20 x 3 + 1 rows of patches)."""
    net, sphere, ref, taps = net_and_ref
    net.set_fusion(0)
    out_s, pool_s = net.forward(sphere, tap=1)
    for mode in (1, 2, 3):             # 1: direct f32 kernel, 2: implicit-GEMM kernel with the fused epilogue, 3 (default): direct
        net.set_fusion(mode)           #    kernel on the bf16 matrix cores with exact operands (cnn_conv1_pieces.hpp)
        out_f, pool_f = net.forward(sphere, tap=1)
        # (separate compilations and, for mode 1, another summation order over the 121 taps)
        assert np.abs(pool_f - pool_s).max() <= 2e-5 * (1 + np.abs(pool_s).max()), mode
        # (mode 3 rounds differently -- exact products, one accumulator rounding per 32 of them -- and every later layer amplifies it)
        assert np.abs(out_f - out_s).max() <= (2e-6 if mode != 3 else 6e-6), mode
    net.set_fusion(3)


def test_conv1_on_exact_bf16_pieces_against_the_float64_net():
    """vpk_cnn_set_fusion(3), the default first stage: conv1 as THREE bf16 matrix products per f32 product -- the uint8 raster
    is exact in one bf16 piece, each weight in three, `- mean` is a constant map added to the accumulators
    (cnn_conv1_pieces.hpp).  Measured against the SAME net evaluated in float64, beside the f32-input direct kernel
    (vpk_cnn_set_fusion(1)): at pool1 (what the stage writes) and at the output its error must not exceed the f32
    kernel's (the rule for a layer on bf16 pieces: factor 1, plus 2^-24 of the blob's scale for ties), at B = 3 and at an
    odd batch of 11 that ends inside a work item's group of images."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    net = cnn.Net(w, mean)
    report = {}
    try:
        for batch in (3, 11):
            sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=batch, start=20)])
            ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
            want = taps[cnn_torch.TAPS[1]]
            scale = float(np.abs(want).max())
            err = {}
            for mode in (1, 3):
                net.set_fusion(mode)
                out, got = net.forward(sphere, tap=1)
                err[mode] = (float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(out - ref).max()))
            report[batch] = (err, scale)
            assert err[3][0] <= 2e-5 * scale, (batch, err, scale)
            assert err[3][0] <= err[1][0] + 6e-8 * scale, (batch, err, scale)
            assert err[3][1] <= 2e-5 and err[1][1] <= 2e-5
    finally:
        net.set_fusion(3)
    print({k: ([round(e[0] / v[1], 9) for e in v[0].values()]) for k, v in report.items()})


def test_split_bf16_convolutions_are_as_accurate_as_the_f32_matrix_path():
    """vpk_cnn_set_precision(1): conv2..5 as six bf16 matrix products per f32 product (cnn_split_gemm.hpp).  Both
    precisions are measured against the SAME net evaluated in float64: the split path's error must be of the size of
    the native f32 DIRECT path's (f32 accumulation noise; vpk_cnn_set_algorithm(0) -- the default Winograd kernels sum
    fewer products and are more accurate than either), at every tap and at the output."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    from vanishing_points_2017_amd import sphere_mapping
    sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=3)])
    ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
    net = cnn.Net(w, mean)
    net.set_algorithm(0)
    try:
        for tap in (2, 4, 5, 6, 8):                            # conv2, conv3, conv4, conv5, fc6
            want = taps[cnn_torch.TAPS[tap]]
            err = []
            for mode in (0, 1):
                net.set_precision(mode)
                out, got = net.forward(sphere, tap=tap)
                err.append((np.abs(got.reshape(want.shape) - want).max(), np.abs(out - ref).max()))
            scale = np.abs(want).max()
            assert err[0][0] <= 2e-5 * scale and err[1][0] <= 2e-5 * scale, (cnn_torch.TAPS[tap], err, scale)
            assert err[1][0] <= 3.0 * err[0][0] + 1e-7 * scale, (cnn_torch.TAPS[tap], err)
            assert err[1][1] <= 2e-5 and err[0][1] <= 2e-5
    finally:
        net.set_precision(0)
        net.set_algorithm(4)


def test_split_path_beyond_4_gib_of_activations():
    """vpk_cnn_forward takes up to 4096 images per call; in the split-precision path conv2's input is 65 x 65 x 96 x 6 B =
    2.4 MB per image, so image 1765 starts 4 GiB into the arena (conv4/5: ~1820, conv3: ~2730).  The gather offsets are
    32-bit and relative to a tile's first image (cnn_split_gemm.hpp), never to the arena: with 1800 images -- six
    distinct rasters repeated -- every image must give the bits its twin among the first six gives, and those must be
    the native path's result to f32 rounding.  (Relative to the arena, images 1765.. silently read other images'
    activations.)"""
    from vanishing_points_2017_amd import cnn, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    from vanishing_points_2017_amd import sphere_mapping
    six = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=6)])
    net = cnn.Net(w, mean)
    native = net.forward(six)
    sphere = np.concatenate([six] * 300)
    try:
        net.set_precision(1)
        out = net.forward(sphere)
    finally:
        net.set_precision(0)
    assert out.shape == (1800, 20, 20)
    assert np.abs(out[:6] - native).max() <= 2e-5
    twins = out.reshape(300, 6, 20, 20)
    assert np.array_equal(twins, np.broadcast_to(twins[:1], twins.shape))
    # the default path (fp16 pairs): conv2's f32 blob is 3.8 MB per image -- image 1128 starts 4 GiB into it --, the piece planes
    # 1.6 MB; per-image bases are 64-bit, offsets inside an image 32-bit (cnn_conv_pieces.hpp, cnn_norm_pool_planes.hpp)
    out = net.forward(sphere)
    twins = out.reshape(300, 6, 20, 20)
    assert np.array_equal(twins, np.broadcast_to(native[None], twins.shape))


def test_misaligned_rasters_are_refused_and_the_fused_input_path_equals_the_pre_pass(net_and_ref):
    """conv1's loader reads four adjacent raster bytes as one word: a raster pointer that is not 4-byte aligned is an
    argument error (never a silent wrong read).  And the two ways of forming float(raster) - mean -- in conv1's loader
    (default) and as the fp32 pre-pass of the unfused path -- give the same response map to rounding."""
    import torch
    from vanishing_points_2017_amd._lib import VpkError
    from vanishing_points_2017_amd import cnn
    _, sphere, ref, _ = net_and_ref
    net = cnn.Net(cnn.synthetic_weights(3), cnn.synthetic_mean(3))   # (other tests have loaded other weights on the handle)
    rt = net.rt
    with rt.on_stream():                                  # (the net's stream: the copies below are ordered before its kernels)
        flat = torch.zeros(sphere.size + 8, dtype=torch.uint8, device=rt.tdev)
        flat[1:1 + sphere.size] = torch.from_numpy(sphere.reshape(-1)).to(rt.tdev)
        with pytest.raises(VpkError):
            net.forward_device(flat[1:1 + sphere.size].view(sphere.shape))
        flat[4:4 + sphere.size] = torch.from_numpy(sphere.reshape(-1)).to(rt.tdev)
        out = net.forward_device(flat[4:4 + sphere.size].view(sphere.shape))
    rt.synchronize()
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() <= 2e-5
    net.set_fusion(0)
    unf = net.forward(sphere)
    net.set_fusion(3)
    assert np.abs(got - unf).max() <= 2e-5


def test_default_path_is_no_further_from_the_float64_net_than_the_f32_direct_kernels():
    """The rule for moving a layer onto pieces (or onto another algorithm): against the SAME net evaluated in float64 its error
    must not exceed, at any tap, that of the f32-input DIRECT kernels (vpk_cnn_set_fusion(1), vpk_cnn_set_algorithm(0): one f32
    FMA chain per output) -- factor 1, plus 2^-24 of the blob's scale for ties.  Checked for the defaults -- conv1 on exact bf16
    pieces (cnn_conv1_pieces.hpp), conv2..5 and fc6 on scaled fp16 PAIRS with block sums (algorithm 4: cnn_conv_pieces.hpp,
    cnn_dense_pieces.hpp), fc7 / fc8 f32 -- and for round 5's first default (algorithm 2: conv2 and fc6 on exact bf16 triples,
    conv3..5 Winograd).  B = 3 and an odd batch of 7."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    net = cnn.Net(w, mean)
    report = {}
    try:
        for batch in (3, 7):
            sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=batch, start=10)])
            ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
            for tap in (1, 2, 3, 4, 5, 6, 8):                      # pool1, conv2, pool2, conv3, conv4, conv5, fc6
                want = taps[cnn_torch.TAPS[tap]]
                scale = float(np.abs(want).max())
                err = {}
                for name, (fusion, algorithm) in (("direct_f32", (1, 0)), ("default", (3, 4)), ("triples", (3, 2))):
                    net.set_fusion(fusion)
                    net.set_algorithm(algorithm)
                    out, got = net.forward(sphere, tap=tap)
                    err[name] = (float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(out - ref).max()))
                report[(batch, cnn_torch.TAPS[tap])] = (err, scale)
                for name in ("default", "triples"):
                    assert err[name][0] <= err["direct_f32"][0] + 6e-8 * scale, (name, batch, cnn_torch.TAPS[tap], err, scale)
                    assert err[name][1] <= err["direct_f32"][1] + 6e-8, (name, batch, cnn_torch.TAPS[tap], err)
                    assert err[name][1] <= 2e-5
    finally:
        net.set_fusion(3)
        net.set_algorithm(4)
    print({k: {n: round(e[0] / v[1], 9) for n, e in v[0].items()} for k, v in report.items()})


def test_forward_is_bit_reproducible_at_the_bench_batch():
    """Twelve forwards of bench.py's batch must give the same bits (and every tap twice): the persistent tile queues hand tiles to
    workgroups in a different order every launch, LDS patches are filled by DMA a channel group ahead and weights arrive by hand-counted
    waits -- a wait that lets a matrix instruction read a patch one DMA instruction early shows up as a rare wrong tile (round 5: one
    forward in six, 1e-3 off at conv2), not as a consistent error, and only repetition finds it."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=102)])
    ref = cnn_torch.forward(w, mean, sphere)
    net = cnn.Net(w, mean)
    try:
        for algorithm in (4, 2):
            net.set_algorithm(algorithm)
            first = net.forward(sphere)
            assert np.abs(first - ref).max() <= 2e-5, algorithm
            for _ in range(11):
                assert np.array_equal(net.forward(sphere), first), algorithm
            for tap in (2, 4, 5, 6, 8):                        # conv2, conv3, conv4, conv5, fc6
                a = net.forward(sphere, tap=tap)[1]
                for _ in range(2):
                    assert np.array_equal(net.forward(sphere, tap=tap)[1], a), (algorithm, tap)
        # a second load of the same model calibrates to the same activation scales: the same bits again
        net.set_algorithm(4)
        first = net.forward(sphere[:7])
        assert np.array_equal(cnn.Net(w, mean).forward(sphere[:7]), first)
    finally:
        net.set_algorithm(4)


def test_fp16_pairs_over_the_operand_range():
    """The fp16 pairs of vpk_cnn_set_algorithm(4) live in fp16's exponent range: weights are brought there by a power of two per layer,
    activations by a power of two per consuming layer that vpk_cnn_load CALIBRATES on the loaded weights (include/vpk.h:
    vpk_cnn_calibrate).  Nets whose blobs are 128 x larger / smaller than the synthetic net's -- conv2's weights and bias x 2^7, conv3's
    weights x 2^-7, and the other way round -- must therefore come out like the synthetic net: no further from the float64 net than the
    f32 direct kernels, at every tap behind conv2.  (What calibration canNOT absorb -- an INPUT whose activations are far from the
    calibration rasters' -- is test_default_path_on_dense_rasters_and_the_all_255_image and test_range_guard_* below.)"""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    base = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=3, start=20)])
    report = {}
    for f2 in (128.0, 1.0 / 128.0):
        w = {k: (np.array(v[0], copy=True), np.array(v[1], copy=True)) for k, v in base.items()}
        w["conv2"] = (w["conv2"][0] * np.float32(f2), w["conv2"][1] * np.float32(f2))
        w["conv3"] = (w["conv3"][0] * np.float32(1.0 / f2), w["conv3"][1])
        ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
        net = cnn.Net(w, mean)
        for tap in (2, 3, 4, 5, 6, 8):                             # conv2, pool2, conv3, conv4, conv5, fc6
            want = taps[cnn_torch.TAPS[tap]]
            scale = float(np.abs(want).max())
            err = {}
            for name, (fusion, algorithm) in (("direct_f32", (1, 0)), ("pairs", (3, 4))):
                net.set_fusion(fusion)
                net.set_algorithm(algorithm)
                out, got = net.forward(sphere, tap=tap)
                assert np.isfinite(got).all() and np.isfinite(out).all()
                err[name] = (float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(out - ref).max()))
            report[(f2, cnn_torch.TAPS[tap])] = {n: round(e[0] / scale, 9) for n, e in err.items()}
            assert err["pairs"][0] <= err["direct_f32"][0] + 6e-8 * scale, (f2, cnn_torch.TAPS[tap], err, scale)
            assert err["pairs"][1] <= err["direct_f32"][1] + 6e-8, (f2, cnn_torch.TAPS[tap], err)
    print(report)


def test_winograd_convolutions_against_the_float64_net():
    """vpk_cnn_set_algorithm(1): conv2 by Winograd F(2 x 2, 5 x 5) and conv3..5 by F(2 x 2, 3 x 3) on the f32 matrix
    cores (cnn_winograd.hpp).  Direct and Winograd paths are measured against the SAME net evaluated in float64 (B = 3 and
    an odd batch of 7 whose tiles end inside a workgroup's block): at conv2 / pool2 / conv3 / conv4 / conv5 / fc6 and at
    the output the Winograd path's error must stay within the bar of every other CNN test (2e-5 of the blob's scale at
    the taps, 2e-5 at the output) and within a small factor of the direct path's own rounding noise."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    net = cnn.Net(w, mean)
    report = {}
    try:
        for batch in (3, 7):
            sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=batch, start=10)])
            ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
            for tap in (2, 3, 4, 5, 6, 8):                         # conv2, pool2, conv3, conv4, conv5, fc6
                want = taps[cnn_torch.TAPS[tap]]
                err = []
                for mode in (0, 1):
                    net.set_algorithm(mode)
                    out, got = net.forward(sphere, tap=tap)
                    err.append((float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(out - ref).max())))
                scale = float(np.abs(want).max())
                report[(batch, cnn_torch.TAPS[tap])] = (err, scale)
                assert err[1][0] <= 2e-5 * scale, (batch, cnn_torch.TAPS[tap], err, scale)
                assert err[1][0] <= 6.0 * err[0][0] + 1e-7 * scale, (batch, cnn_torch.TAPS[tap], err)
                assert err[1][1] <= 2e-5 and err[0][1] <= 2e-5
    finally:
        net.set_algorithm(4)                                       # the library's default
    print({k: ([round(e[0] / v[1], 9) for e in v[0]]) for k, v in report.items()})


def _dense_rasters():
    """Rasters of the densities the YUD-shape tests never reach: an ECD-shape scene with > 1000 lines (configs[2]), an HLW-shape scene
    (configs[3]) with > 800, a stress scene (configs[4], N = 1000), and the largest input the uint8 boundary admits."""
    from vanishing_points_2017_amd import sphere_mapping, synth
    picks = []
    for cid, lo in ((3, 1000), (4, 800), (5, 1000)):
        for sc in synth.config_scenes(cid, count=40):
            if sc["l"].shape[0] >= lo:
                picks.append(sc["l"])
                break
    assert len(picks) == 3
    sphere = sphere_mapping.raster_batch(picks)
    return np.concatenate([sphere, np.full((1, 500, 500), 255, np.uint8)]), [p.shape[0] for p in picks]


def test_default_path_on_dense_rasters_and_the_all_255_image():
    """The factor-1 rule of test_default_path_is_no_further_... where the fp16 pairs are EXPOSED: the activation
    scales are fixed at load, so what matters is an input whose blobs are far larger than a sparse raster's.  Rasters of 1000-line
    scenes (configs[2], [3], [4]: evaluation.py:12-14 with those line counts) and the all-255 image, every tap and the output,
    against the float64 net: the default (fusion 3, algorithm 4) no further from it than the f32 direct kernels, nothing clamped."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    sphere, counts = _dense_rasters()
    assert min(counts) >= 800 and sphere[:3].mean() > 40 and sphere[:3].max() > 200, (counts, sphere.mean(), sphere.max())
    net = cnn.Net(w, mean)
    ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
    report = {}
    try:
        for tap in (1, 2, 3, 4, 5, 6, 7, 8, 9):                    # pool1 .. fc7
            want = taps[cnn_torch.TAPS[tap]]
            scale = float(np.abs(want).max())
            err = {}
            for name, (fusion, algorithm) in (("direct_f32", (1, 0)), ("default", (3, 4))):
                net.set_fusion(fusion)
                net.set_algorithm(algorithm)
                out, got = net.forward(sphere, tap=tap)            # (raises VpkRangeError if anything was clamped)
                assert np.isfinite(got).all() and np.isfinite(out).all()
                err[name] = (float(np.abs(got.reshape(want.shape) - want).max()), float(np.abs(out - ref).max()))
            report[cnn_torch.TAPS[tap]] = {n: round(e[0] / scale, 9) for n, e in err.items()}
            assert err["default"][0] <= err["direct_f32"][0] + 6e-8 * scale, (cnn_torch.TAPS[tap], err, scale)
            assert err["default"][1] <= err["direct_f32"][1] + 6e-8, (cnn_torch.TAPS[tap], err)
            assert err["default"][1] <= 2e-5
        assert net.range_flags() == 0
    finally:
        net.set_fusion(3)
        net.set_algorithm(4)
    print(report)


def test_calibration_covers_the_dense_rasters_with_headroom():
    """The scales vpk_cnn_load picks come from the built-in calibration set (sparse noise, 1000 blended strokes, all-255): every
    blob of a dense raster and of the all-255 image, multiplied by its layer's scale, must stay below 2^7 -- at least 2^9 of headroom
    to fp16's 65 504 -- and the largest must not be tiny either (> 2^-3: the pairs keep 22 bits down to 2^-13 of the calibration
    maximum).  Recalibrating on caller rasters and restoring the saved scales gives back the same bits."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import cnn
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    sphere, _ = _dense_rasters()
    net = cnn.Net(w, mean)
    scales = net.activation_scales()
    assert scales.shape == (6,) and all(np.frexp(float(x))[0] == 0.5 for x in scales)
    _, taps = cnn_torch.forward(w, mean, sphere, want_taps=True)
    for sc, tap in zip(scales, (1, 3, 4, 5, 7, 8)):                # inputs of conv2, conv3, conv4, conv5, fc6, fc7
        m = float(np.abs(taps[cnn_torch.TAPS[tap]]).max()) * float(sc)
        assert 0.125 < m < 128.0, (cnn_torch.TAPS[tap], m, sc)
    first = net.forward(sphere)
    net.calibrate(sphere[:2])                                       # the caller's rasters alone decide
    mine = net.activation_scales()
    assert all(np.frexp(float(x))[0] == 0.5 for x in mine)
    again = net.forward(sphere)
    assert np.abs(again - first).max() <= 2e-6
    net.calibrate(None)                                             # the built-in set again
    assert np.array_equal(net.activation_scales(), scales)
    assert np.array_equal(net.forward(sphere), first)


@pytest.mark.parametrize("layer", [0, 1, 2, 3, 4, 5])
def test_range_guard_reports_a_clamped_activation_and_never_lets_a_nan_out(layer):
    """split2h_guard (cnn_conv_pieces.hpp): a scaled activation that reaches fp16's 65 504 is clamped and the consuming layer's bit set;
    vpk_cnn_range_flags turns that into VPK_ERR_RANGE.  Provoked by setting ONE layer's scale far too high (the other five stay
    calibrated): the forward must report exactly that layer, its response maps must be finite (round 5: inf - inf = NaN from there on),
    the flag must clear on reading, and restoring the calibrated scales must restore the bits."""
    from vanishing_points_2017_amd import cnn, sphere_mapping, synth
    from vanishing_points_2017_amd._lib import VpkError, VpkRangeError
    w = cnn.synthetic_weights(0)
    mean = cnn.synthetic_mean(0)
    sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=5, start=30)])
    net = cnn.Net(w, mean)
    rt = net.rt
    good = net.activation_scales()
    first = net.forward(sphere)
    assert net.range_flags() == 0
    bad = good.copy()
    bad[layer] = good[layer] * np.float32(2.0 ** 16)               # (a YUD raster's blobs are up to 8 x below the calibration maximum of 32..64)
    try:
        net.set_activation_scales(bad)
        with pytest.raises(VpkRangeError) as ei:
            net.forward(sphere)
        assert ei.value.flags == 1 << (layer + 1), (ei.value.flags, layer)
        assert cnn.Net.RANGE_LAYERS[layer] in str(ei.value)
        d = rt.torch.from_numpy(sphere).to(rt.tdev)
        out = net.forward_device(d)
        rt.synchronize()
        assert np.isfinite(out.cpu().numpy()).all()
        assert net.range_flags() == 1 << (layer + 1)
        assert net.range_flags() == 0                               # (read and cleared)
        for tap in (2, 4, 6, 8):                                    # the f32 taps stay finite as well
            d_out, d_tap = net.forward_device(d, tap=tap)
            rt.synchronize()
            assert np.isfinite(d_tap.cpu().numpy()).all() and np.isfinite(d_out.cpu().numpy()).all()
        net.range_flags()
        with pytest.raises(VpkError):
            net.set_activation_scales([1.0, 2.0, 3.0, 4.0, 8.0, 16.0])      # 3 is not a power of two
    finally:
        net.set_activation_scales(good)
    assert np.array_equal(net.forward(sphere), first)
    assert net.range_flags() == 0
