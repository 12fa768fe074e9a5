"""The EM paths that only unusual images take, on the GPU (round-2 review: "never executed by any test").

* vpk_em_set_lds_panel shrinks the LDS budget the workgroup plans with, so small images -- the reference's own goldens
  -- go down the paths big images take by necessity: the chunked smoother (smooth_blocks: operand rows staged through
  LDS a few at a time, one summation chain per column), smooth_full in several passes, the split's distance matrix and
  direction vectors in HBM (cluster2 instead of cluster2_lds).  Bar: the goldens' (assignments exact, VPs 1e-4).
* images with 2 100 and 3 000 lines (past every LDS panel) against the CPU oracle;
* a 512-image stress launch (every CU, slots recycled) against one-image launches, bit for bit;
* all 2 018 images of the HLW-shape config in ONE launch, the stored ones against the reference's results.
"""
import numpy as np
import pytest

from conftest import golden_cases
from golden_util import abserr, check_em_result, em_kwargs, gpu_rasters, load, relerr

pytestmark = pytest.mark.gpu

CASES = golden_cases()


def _scene(g):
    return {"l": g["l"].copy(), "lp": g["lp"], "cnn_response": g["cnn_response"],
            "sphere_image": g["sphere_image"], "init_vp": g.get("init_vp")}


@pytest.mark.parametrize("budget", [64, 96, 700, 2048, 6144])
def test_small_lds_budgets_send_the_goldens_down_the_big_image_paths(budget):
    """64 / 96 doubles: no operand panel at all (smooth_blocks with 2-12 rows per chunk), split sets of more than 21 / 32
    lines stage their direction vectors in HBM and cluster there; 700: panels of a few dozen lines; 2048 / 6144: round 1's
    budgets (smooth_full, partly in several passes, LDS clustering for the small sets)."""
    from vanishing_points_2017_amd import em as gem
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0)
    rt.handle.em_set_lds_panel(budget)
    try:
        for name in CASES:
            g = load(name)
            kw = {k: v for k, v in em_kwargs(g).items() if k != "init_vp"}
            res = gem.em_batch([_scene(g)], **kw)[0]
            check_em_result(res, g)
    finally:
        rt.handle.em_set_lds_panel(0)


@pytest.mark.parametrize("n,m", [(2100, 8), (2100, 13), (3000, 8), (3000, 20)])
def test_weight_matrix_beyond_the_lds_panel(n, m):
    """vpk_weight_matrix where no whole operand panel fits: smooth_full in passes of 8 VPs (2100 x 13), smooth_blocks
    (3000 lines) -- against the NumPy expression of vp_localisation.py:515-524."""
    from vanishing_points_2017_amd import kernels
    rng = np.random.RandomState(n + m)
    lsim = rng.rand(n, n) ** 8
    lsim = lsim + lsim.T
    np.fill_diagonal(lsim, 0.0)
    pvl = rng.rand(m, n)
    pvl /= pvl.sum(0)
    lw = rng.rand(n) * 0.4
    got = kernels.weight_matrix(pvl, lw, lsim)
    w_ = pvl * lw
    want = (w_ + lw * (w_ @ lsim)) / (1 + lw * lsim.sum(0))
    assert relerr(got, want) <= 1e-12


@pytest.mark.parametrize("seed,n", [(8100, 2100), (8200, 3000)])
def test_images_past_every_lds_panel_against_the_oracle(seed, n):
    """2 100 lines: smooth_full from the HBM copy of the operands, 8 VPs per pass; 3 000 lines: smooth_blocks.  Whole EM
    runs against the CPU oracle, the bar of every other parity test."""
    from oracle import em_numpy
    from vanishing_points_2017_amd import em as gem, synth
    sc = gpu_rasters([synth.make_scene(seed, n, 3)])[0]
    ref = em_numpy.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                            sphere_image=sc["sphere_image"])
    res = gem.em_batch([sc])[0]
    assert res["status"] == 0 and res["iterations"] == ref["iterations"]
    assert res["vp"].shape == ref["vp"].shape
    assert np.array_equal(res["vp_assoc"], ref["vp_assoc"])
    assert abserr(res["vp"], ref["vp"]) <= 1e-4
    assert np.array_equal(res["counts"], ref["counts"])


def test_stress_launch_of_512_images_equals_one_image_launches():
    """BASELINE configs[4] as bench.py --workload stress runs it: 512 images (16 distinct scenes) of 1000 lines, 8
    supplied VPs, 50 forced iterations in ONE launch -- two images per CU's worth of queue, every slot reused.  Every
    copy must equal its scene's one-image launch in every output bit."""
    from vanishing_points_2017_amd import em as gem, synth
    kw = dict(num_iter=50, do_split=False, do_merge=False, final_convergence=-1)
    base = []
    for i in range(16):
        s = synth.make_scene(5000 + i, 1000, 8)
        s["init_vp"] = synth.stress_init_vps(5000 + i)
        base.append(s)
    base = gpu_rasters(base)
    single = [gem.em_batch([dict(s, l=s["l"].copy())], **kw)[0] for s in base]
    res = gem.em_batch([dict(base[i % 16], l=base[i % 16]["l"].copy()) for i in range(512)], **kw)
    for i, r in enumerate(res):
        a = single[i % 16]
        assert r["status"] == a["status"] == 0 and r["iterations"] == a["iterations"] == 49
        for key in ("vp", "sigma", "counts", "counts_weighted", "vp_assoc"):
            assert np.array_equal(r[key], a[key]), (i, key)


def test_all_2018_hlw_shape_images_in_one_launch():
    """BASELINE configs[3]: the whole HLW-shape set (2 018 images, 100..1000 lines) through ONE vpk_em_batch launch: 256
    persistent workgroups, every slot reused about eight times, largest images first.  The 64 images the reference's
    results are stored for (tests/golden/full_c4.npz) must meet the parity bar inside that launch exactly as they do in
    their own small batch; every image must come back with a regular status."""
    import os
    from vanishing_points_2017_amd import em as gem, parity, synth
    if not os.path.isfile(parity.golden_path(4)):
        pytest.skip("no stored reference results for config 4")
    ref = parity.ReferenceResults(4)
    scenes = list(synth.config_scenes(4))
    assert len(scenes) == 2018
    stored = {int(i) for i in ref.index}
    for k in stored:
        assert parity.input_sha(scenes[k]) == ref.get(k)["input_sha"]
    res = gem.em_batch(scenes)                            # from the lines alone: the rasters are made on the way (evaluation.py:175)
    for k in stored:                                      # ... and are the reference's
        assert parity.raster_sha(scenes[k]["sphere_image"]) == ref.get(k)["raster_sha"], k
    assert len(res) == 2018 and all(r["status"] in (0, 1, 2) for r in res)
    assert sum(r["status"] == 0 for r in res) >= 2000
    cert = parity.instability_certificates()             # (images on which the reference's own answer moves under a one-ulp input change)
    bad = [(k, parity.compare_one(res[k], ref.get(k))) for k in sorted(stored)
           if not parity.passes(parity.compare_one(res[k], ref.get(k))) and not cert.get((4, k), {}).get("unstable")]
    assert not bad, bad[:4]
