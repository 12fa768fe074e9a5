"""GPU sphere rasteriser (csrc/vpk_raster.hip: the reference's matplotlib / Agg pipeline restated stage by stage)
against the REFERENCE's own rasters stored in tests/golden/*.npz (sphere_mapping.sphere_line_plot run in the build
container by oracle/make_golden.py) and against the CPU restatement oracle/agg_raster.py.

Bar: the raster is integer work -- bit-exact.  Every default golden must be reproduced pixel for pixel; downstream,
find_initial_vps on the GPU raster must give the golden's initial VPs (1e-13) and the EM started from the lines alone
(raster made here) the reference's assignments bit for bit."""
import numpy as np
import pytest

from conftest import golden_cases
from golden_util import abserr, check_em_result, em_kwargs, load

pytestmark = pytest.mark.gpu

# (the two hard1row goldens carry rasters captured from round 2's GPU rasteriser, not matplotlib's)
REF_RASTER = [c for c in golden_cases() if not c.startswith("hard1row")]


@pytest.mark.parametrize("name", REF_RASTER)
def test_raster_equals_the_references_raster(name):
    from vanishing_points_2017_amd import sphere_mapping
    g = load(name)
    l = g["l"].copy()
    got = sphere_mapping.sphere_line_plot(l, 500, alpha=0.1)
    assert got.shape == (500, 500) and got.dtype == np.uint8
    assert np.array_equal(l, g["l"])                       # f = 1 leaves the caller's lines unchanged
    diff = np.abs(got.astype(int) - g["sphere_image"].astype(int))
    assert diff.max() == 0, "%d pixels differ from the reference's raster (max %d)" % ((diff > 0).sum(), diff.max())


def test_single_line_peaks_at_25_and_the_frame_is_black():
    from vanishing_points_2017_amd import sphere_mapping
    img = sphere_mapping.sphere_line_plot(np.array([[0.3, 1.0, 0.2]]), 500, alpha=0.1)
    assert img.max() == 25                                 # alpha8 = 26 -> 65280 * 26 / 65306, truncated
    assert (img > 0).sum() > 500
    assert img[:, 0].max() == 0 and img[0, :].max() == 0   # the axes' left / top spines cover column 0 / row 0


def test_batch_other_sizes_alpha_and_odd_lines_against_the_cpu_restatement():
    """A ragged batch in one launch (queue of images per workgroup), another canvas size and alpha, and lines a data
    set never has: vertical (b = 0: the curve jumps between +-pi/2), b tiny, all-zero (every sample NaN: nothing is
    drawn), huge c.  Against oracle/agg_raster.py, which is pinned against matplotlib itself."""
    from oracle import agg_raster
    from vanishing_points_2017_amd import sphere_mapping, synth
    a = synth.make_scene(77, 90, 3, raster=None)["l"]
    b = synth.make_scene(78, 7, 3, raster=None)["l"]
    odd = np.array([[1, 0.0, 0.3], [1, 1e-9, 0.3], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], [0.3, -1e-6, -2.0], [1, 1, 1e6],
                    [5, 0.01, 0.01], [0, 0, 1.0]])
    got = sphere_mapping.raster_batch([a, b, odd, a[:1]], size=500, alpha=0.1)
    for lines, r in zip([a, b, odd, a[:1]], got):
        assert np.array_equal(r, agg_raster.raster(lines))
    small = sphere_mapping.raster_batch([a[:30]], size=250, alpha=0.1)[0]
    assert np.array_equal(small, agg_raster.raster(a[:30], size=250))
    strong = sphere_mapping.raster_batch([a[:30]], size=500, alpha=0.5)[0]
    assert np.array_equal(strong, agg_raster.raster(a[:30], alpha=0.5))


@pytest.mark.parametrize("name", [c for c in REF_RASTER if "init_vp" not in load(c)])
def test_initial_vps_and_em_from_the_lines_alone(name):
    """What the raster feeds (sphere_mapping.py:36-72 -> find_initial_vps, vp_localisation.py:111-165 -> EM): with the
    raster made on the GPU from the lines, the initial VPs are the golden's and the whole EM run is the reference's."""
    from vanishing_points_2017_amd import em as gem, kernels, sphere_mapping
    g = load(name)
    ras = sphere_mapping.sphere_line_plot(g["l"].copy(), 500, alpha=0.1)
    if "i_v0" in g:
        v0, _ = kernels.init_vps(g["cnn_response"], ras)
        assert abserr(v0, g["i_v0"]) <= 1e-13
    kw = {k: v for k, v in em_kwargs(g).items() if k != "init_vp"}
    res = gem.em_batch([{"l": g["l"].copy(), "lp": g["lp"], "cnn_response": g["cnn_response"], "sphere_image": ras}], **kw)[0]
    check_em_result(res, g)


def test_wave_parallel_simplifier_equals_the_sequential_machine(monkeypatch):
    """simplify_kernel (one wave per line, 64 samples at a time, runs folded with reductions) against the vertex-by-vertex
    machine of raster_device.hpp run for every line (VPK_RASTER_SEQUENTIAL=1): same pixels on 1 500 random lines -- steep,
    flat, through the poles, tiny and huge coefficients -- and on lines with non-finite samples (always sequential)."""
    from vanishing_points_2017_amd import sphere_mapping
    rng = np.random.default_rng(2024)
    sets = []
    for k in range(6):
        l = rng.normal(size=(250, 3))
        l[:, 1] *= 10.0 ** rng.uniform(-6, 1, size=250)            # b small: steep curves, jumps across the canvas
        l[:, 2] *= 10.0 ** rng.uniform(-3, 3, size=250)
        if k == 5:
            l[::7, 1] = 0.0                                        # vertical lines: atan(+-inf), NaN where the numerator is 0 too
            l[3::50] = 0.0
        sets.append(l)
    fast = sphere_mapping.raster_batch(sets, size=500, alpha=0.1)
    monkeypatch.setenv("VPK_RASTER_SEQUENTIAL", "1")
    slow = sphere_mapping.raster_batch(sets, size=500, alpha=0.1)
    monkeypatch.delenv("VPK_RASTER_SEQUENTIAL")
    assert np.array_equal(fast, slow), "%d pixels differ" % (fast != slow).sum()
    assert fast.max() > 100                                        # (lines do pile up: the comparison is not of empty canvases)


def test_empty_images_in_a_batch_and_extreme_canvas_sizes():
    """Images without lines between others (the frame alone), a 64-px and a 1000-px canvas (row tables, LDS rows and the
    blend's column segments are sized from the canvas), against the CPU restatement."""
    from oracle import agg_raster
    from vanishing_points_2017_amd import sphere_mapping, synth
    a = synth.make_scene(77, 40, 3, raster=None)["l"]
    sets = [a[:0], a, a[:0], a[:3], a[:0]]
    got = sphere_mapping.raster_batch(sets, size=500, alpha=0.1)
    for lines, r in zip(sets, got):
        assert np.array_equal(r, agg_raster.raster(lines))
    assert np.array_equal(sphere_mapping.raster_batch([a[:5]], size=64, alpha=0.1)[0], agg_raster.raster(a[:5], size=64))
    assert np.array_equal(sphere_mapping.raster_batch([a[:5]], size=1000, alpha=0.2)[0],
                          agg_raster.raster(a[:5], size=1000, alpha=0.2))


def test_lines_far_from_the_principal_point_need_no_more_room_than_others():
    """|c| >> |a|, |b|: the curve is an arch whose rows are crossed twice, hundreds of pixels apart.  With one range of cells
    per row such a polygon asked for 100 000 pool entries (mostly the gap), a hundred of them overran the call's coverage pool
    and -- with the shared bump counter -- took lines of OTHER images with them.  Rows now hold two ranges split at the apex
    (gap dropped where nothing is drawn in it).  Against the CPU restatement, in a batch and alone; and the wrapper raises
    rather than return an incomplete raster (checked by making sure no flag is up)."""
    from oracle import agg_raster
    from vanishing_points_2017_amd import sphere_mapping, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    rng = np.random.default_rng(99)
    wide = rng.normal(size=(60, 3))
    wide[:, 2] *= 50
    steep = rng.normal(size=(60, 3))
    steep[:, 1] *= 0.01
    ordinary = synth.make_scene(503, 60, 3, raster=None)["l"]
    sets = [ordinary, wide, steep, wide[:1]]
    got = sphere_mapping.raster_batch(sets, size=500, alpha=0.1)
    for lines, r in zip(sets, got):
        assert np.array_equal(r, agg_raster.raster(lines))
    assert not sphere_mapping.raster_flags(get_runtime(0), len(sets)).any()
    one = np.array([[-0.26247709, -1.59752339, -74.31417043]])       # (alone: eight canvases of pool, several LDS bands)
    assert np.array_equal(sphere_mapping.raster_batch([one], size=500, alpha=0.1)[0], agg_raster.raster(one))


def test_alternative_parametrisation_equals_the_references_raster():
    """sphere_line_plot(..., alternative=True) (sphere_mapping.py:58-59: beta = atan(-c / (a cos + b sin)), a curve with a
    pole -- the polyline jumps across the whole canvas where the denominator changes sign): the reference's own rasters
    (tests/golden/rasteralt.npz, oracle/make_raster_alt_golden.py), pixel for pixel; the default curve right after it on
    the same handle is unaffected by the switch."""
    import os
    from conftest import GOLDEN
    from vanishing_points_2017_amd import sphere_mapping
    g = np.load(os.path.join(GOLDEN, "rasteralt.npz"))
    for k in "abc":
        l = g["l_" + k].copy()
        got = sphere_mapping.sphere_line_plot(l, 500, alpha=0.1, alternative=True)
        diff = got.astype(int) - g["raster_" + k].astype(int)
        assert not diff.any(), "set %s: %d pixels differ" % (k, (diff != 0).sum())
    plain = load("tiny_n12")
    assert np.array_equal(sphere_mapping.sphere_line_plot(plain["l"].copy(), 500, alpha=0.1), plain["sphere_image"])


def test_a_batch_of_only_empty_images_and_a_bare_call():
    """No line in the whole call (the simplifier's launch has nothing to do, l may be NULL): every canvas is the frame
    alone, as the reference's figure is when its loop over the lines runs zero times (sphere_mapping.py:54)."""
    from oracle import agg_raster
    from vanishing_points_2017_amd import sphere_mapping
    none = np.zeros((0, 3))
    want = agg_raster.raster(none)
    got = sphere_mapping.raster_batch([none, none, none], size=500, alpha=0.1)
    assert got.shape == (3, 500, 500)
    for r in got:
        assert np.array_equal(r, want)
    assert np.array_equal(sphere_mapping.sphere_line_plot(none.copy(), 500, alpha=0.1), want)


def test_workspace_growth_between_calls_with_the_same_offsets():
    """The offsets / image order / sample table cached in the handle's workspace must not survive a re-allocation of that
    workspace: the same batch structure at 64 px and then at 1000 px (the workspace grows; the new block may sit at the
    old address) and back; and the flags call refuses another batch than the raster call's."""
    from oracle import agg_raster
    from vanishing_points_2017_amd import _lib, sphere_mapping, synth
    from vanishing_points_2017_amd.runtime import Runtime
    rt = Runtime(0)                                                  # a fresh handle: its workspace starts empty
    a = synth.make_scene(77, 40, 3)["l"]
    sets = [a[:5], a[5:9]]
    for size in (64, 1000, 64, 250):
        got = sphere_mapping.raster_batch(sets, size=size, alpha=0.1, runtime=rt)
        for lines, r in zip(sets, got):
            assert np.array_equal(r, agg_raster.raster(lines, size=size)), size
    with pytest.raises(_lib.VpkError):
        sphere_mapping.raster_flags(rt, 3)
    assert not sphere_mapping.raster_flags(rt, 2).any()
