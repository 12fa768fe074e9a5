"""Parity tests proper: the HIP EM path, called through the C-ABI (libvpk.so), against
(a) golden vectors captured from the reference and (b) the CPU oracle on fresh seeded scenes.

Bar (BASELINE.json north_star): line->VP assignments bit-exact, VP directions within 1e-4.
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


@pytest.mark.parametrize("name", [c for c in CASES if "i_v0" in load(c)])
def test_kernels_against_reference_intermediates(name):
    from vanishing_points_2017_amd import kernels
    g = load(name)
    lsim, lscore, langle = kernels.pairwise(g["lp"])
    if "i_lsim" in g:
        assert abserr(lsim, g["i_lsim"]) <= 1e-12
    else:
        assert abserr(lsim[::17, :], g["i_lsim_rows"]) <= 1e-12
    assert np.array_equal(lsim, lsim.T)
    assert abserr(lsim.sum(axis=1), g["i_lsim_rowsum"]) <= 1e-10
    assert abserr(lscore, g["i_lscore"]) <= 1e-12
    assert abserr(langle, g["i_langles"]) <= 1e-13
    v0, w = kernels.init_vps(g["cnn_response"], g["sphere_image"])
    assert abserr(v0, g["i_v0"]) <= 1e-13
    assert np.array_equal(w, g["i_pdf_weights"])            # float32 prior weights are bit-exact
    m0 = g["i_v0"].shape[0]
    s = np.ones(m0) * (np.pi / (1.282 * 20)) * 1e-6
    pv, lvsq, pvl, pl, _ = kernels.estep(g["lp"], g["cnn_response"], g["i_v0"], s)
    assert relerr(pv, g["i_p_v0"]) <= 1e-11
    assert abserr(lvsq, g["i_lvsq0"]) <= 1e-13
    # exponent lvsq/(2s), s ~ 1.2e-7: ulp-level differences in lvsq / exp are amplified ~1e9-fold
    assert relerr(pl, g["i_p_l0"]) <= 1e-6
    assert abserr(pvl, g["i_p_vl0"]) <= 1e-6
    from oracle import em_numpy as em
    w0 = kernels.weight_matrix(g["i_p_vl0"], g["i_lweight"], em.calc_lsim(g["lp"], sigma=1))
    assert relerr(w0, g["i_w0"]) <= 1e-10
    lnorm = g["l"] / np.sqrt((g["l"] ** 2).sum(1))[:, None]
    vp, valid = kernels.mstep(lnorm, g["i_w0"])
    assert valid.all()
    # The kernel takes the bottom eigenvector of the 3x3 weighted scatter where the reference takes
    # the third right singular vector (LAPACK).  They agree wherever the weighted line set has
    # numerical rank >= 2; rank-deficient rows (one supporting line: VPs the reference prunes at
    # vp_localisation.py:250) have a null vector decided by LAPACK's rounding noise.
    for m in range(vp.shape[0]):
        r = g["i_w0"][m] / g["i_w0"][m].max()
        sv = np.linalg.svd(r[:, None] * lnorm, compute_uv=False)
        if sv[1] >= 1e-3:
            assert abserr(vp[m], g["i_mstep0"][m]) <= 1e-9
    assert sum(g["i_counts0"] >= 3) >= 1
    # calc_vp_line_counts (E14) on its own, on the reference's decision metric: counts and assignments exact
    counts, counts_w, assoc = kernels.line_counts(g["lp"], g["i_v0"], s, g["i_w0"], g["i_lweight"])
    assert np.array_equal(assoc, g["i_assoc0"])
    assert np.array_equal(counts, g["i_counts0"])


def test_cluster2_matches_sklearn():
    import warnings
    import sklearn.cluster as cluster
    from oracle import em_numpy as em
    from vanishing_points_2017_amd import kernels
    rs = np.random.RandomState(5)
    for n in (9, 17, 40, 83, 300):
        ang = rs.uniform(0, np.pi, n)
        lp = np.stack([np.cos(ang), np.sin(ang), np.zeros(n), np.zeros(n)], 1) * rs.uniform(0.1, 1, (n, 1))
        rows = np.repeat(np.arange(n), n).reshape(n, n)
        ld = 1 - em.pair_cosangle(lp, 2, rows, rows.T)
        np.fill_diagonal(ld, 0)
        model = cluster.AgglomerativeClustering(linkage="average", connectivity=ld, n_clusters=2,
                                                metric="precomputed")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model.fit_predict(ld)
        labels, flags = kernels.cluster2(ld)
        assert flags == 0          # random angles: no two candidate merges are exactly tied
        assert np.array_equal(labels, model.labels_)


def test_cluster2_flags_exact_tie():
    """Four directions twice over: the first merge candidates are exactly tied and sklearn's choice depends on
    Python heap order -> VPK_EM_FLAG_SPLIT_TIE (both the one-wave LDS version and the workgroup version)."""
    from oracle import em_numpy as em
    from vanishing_points_2017_amd import kernels
    for reps in (2, 40):           # 8 lines (LDS path), 160 lines (workgroup path)
        ang = np.tile(np.array([0.0, 0.5, 0.25, 0.75]) * np.pi, reps)
        n = ang.shape[0]
        lp = np.stack([np.cos(ang), np.sin(ang), np.zeros(n), np.zeros(n)], 1)
        rows = np.repeat(np.arange(n), n).reshape(n, n)
        ld = 1 - em.pair_cosangle(lp, 2, rows, rows.T)
        np.fill_diagonal(ld, 0)
        _, flags = kernels.cluster2(ld)
        assert flags & 1


@pytest.mark.parametrize("name", CASES)
def test_full_run_matches_reference_golden(name):
    from vanishing_points_2017_amd import em as gem
    g = load(name)
    kw = {k: v for k, v in em_kwargs(g).items() if k != "init_vp"}
    res = gem.em_batch([_scene(g)], **kw)[0]
    assert abserr(res["l"], g["l_normalised"]) <= 1e-15
    check_em_result(res, g)


def test_drop_in_call_surface():
    """vp_localisation.expectation_maximisation: reference name, defaults, in-place l, result keys."""
    from vanishing_points_2017_amd import vp_localisation
    g = load("yud_n120")
    l = g["l"].copy()
    res = vp_localisation.expectation_maximisation(l, g["lp"].copy(), g["cnn_response"],
                                                   sphere_image=g["sphere_image"])
    assert abserr(l, g["l_normalised"]) <= 1e-15          # caller's array normalised in place
    for key in ("vp_assoc", "vp", "counts", "counts_weighted", "count_id", "decision_metric",
                "iterations", "distribution", "sigma"):
        assert key in res
    check_em_result(res, g)
    assert res["decision_metric"].shape == (res["vp"].shape[0], g["lp"].shape[0])
    assert relerr(res["decision_metric"].max(axis=0), g["o_decision_metric_colmax"]) <= 1e-6
    with pytest.raises(AssertionError):
        vp_localisation.expectation_maximisation(l, g["lp"], g["cnn_response"], sphere_image=g["sphere_image"],
                                                 distance_measure="dotprod")
    with pytest.raises(ValueError):                       # np.vstack([]) at vp_localisation.py:165
        vp_localisation.expectation_maximisation(l, g["lp"], g["cnn_response"],
                                                 sphere_image=np.zeros((500, 500), np.uint8))


def test_ragged_batch_equals_single_runs():
    """All default-parameter goldens in ONE launch (ragged N, dynamic queue) == one by one."""
    from vanishing_points_2017_amd import em as gem
    names = [c for c in CASES if not any(k.startswith("kw_") for k in load(c))]
    gs = [load(n) for n in names]
    res = gem.em_batch([_scene(g) for g in gs] * 3)       # 3 copies: more images than one wave of slots
    for k, r in enumerate(res):
        check_em_result(r, gs[k % len(gs)])
    # determinism: the copies agree bit for bit
    for k in range(len(gs)):
        for rep in (1, 2):
            a, b = res[k], res[k + rep * len(gs)]
            assert np.array_equal(a["vp"], b["vp"]) and np.array_equal(a["vp_assoc"], b["vp_assoc"])


def test_workgroup_cap_queues_images():
    """vpk_em_set_workgroups: fewer persistent workgroups than images -> the images queue inside the launch;
    results are bit-identical to the uncapped launch."""
    from vanishing_points_2017_amd import em as gem
    from vanishing_points_2017_amd.runtime import get_runtime
    names = [c for c in CASES if not any(k.startswith("kw_") for k in load(c))]
    gs = [load(n) for n in names]
    ref = gem.em_batch([_scene(g) for g in gs])
    rt = get_runtime(0)
    rt.handle.em_set_workgroups(3)
    try:
        res = gem.em_batch([_scene(g) for g in gs])
    finally:
        rt.handle.em_set_workgroups(0)
    for a, b, g in zip(ref, res, gs):
        check_em_result(b, g)
        assert np.array_equal(a["vp"], b["vp"]) and np.array_equal(a["vp_assoc"], b["vp_assoc"])
        assert a["iterations"] == b["iterations"]


def test_against_oracle_on_fresh_scenes():
    """YUD-shape scenes not in the golden set: GPU vs CPU oracle, same seeded inputs."""
    from oracle import em_numpy as em
    from vanishing_points_2017_amd import em as gem, synth
    scenes = gpu_rasters(synth.config_scenes(2, count=12, start=30))
    res = gem.em_batch(scenes)
    for sc, r in zip(scenes, res):
        ref = em.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                          sphere_image=sc["sphere_image"])
        assert r["status"] == 0
        assert r["iterations"] == ref["iterations"]
        assert np.array_equal(r["vp_assoc"], ref["vp_assoc"])
        assert abserr(r["vp"], ref["vp"]) <= 1e-4
        assert np.array_equal(r["counts"], ref["counts"])


def test_edge_cases_in_one_ragged_batch():
    """Degenerate images next to ordinary ones in one launch: no lines, two lines, a blank CNN response (the
    reference raises ValueError at vp_localisation.py:165), and 40 supplied VPs (more than one smoother pass
    and more hypotheses than the 32 accumulators).  Checked against the CPU oracle."""
    from oracle import em_numpy as em
    from vanishing_points_2017_amd import em as gem, synth
    base = gpu_rasters(synth.config_scenes(2, count=3, start=50))
    empty = dict(base[0], l=np.zeros((0, 3)), lp=np.zeros((0, 4)))
    two = dict(base[1], l=base[1]["l"][:2].copy(), lp=base[1]["lp"][:2].copy())
    blank = dict(base[2], cnn_response=np.zeros((20, 20), dtype=np.float32))
    res = gem.em_batch([base[0], empty, two, blank, base[1]])
    for k in (0, 4):
        sc = base[0] if k == 0 else base[1]
        ref = em.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                          sphere_image=sc["sphere_image"])
        assert res[k]["status"] == 0 and np.array_equal(res[k]["vp_assoc"], ref["vp_assoc"])
        assert abserr(res[k]["vp"], ref["vp"]) <= 1e-4
    assert res[1]["status"] != 0 and res[1]["vp"] is None                 # nothing to estimate from
    ref2 = em.expectation_maximisation(two["l"].copy(), two["lp"].copy(), two["cnn_response"].copy(),
                                       sphere_image=two["sphere_image"])
    if ref2["vp"] is None:
        assert res[2]["vp"] is None
    else:
        assert np.array_equal(res[2]["vp_assoc"], ref2["vp_assoc"]) and abserr(res[2]["vp"], ref2["vp"]) <= 1e-4
    assert res[3]["status"] == 2                                          # VPK_EM_NO_INITIAL_VP
    with pytest.raises(ValueError):
        em.expectation_maximisation(blank["l"].copy(), blank["lp"].copy(), blank["cnn_response"].copy(),
                                    sphere_image=blank["sphere_image"])
    # 40 hypotheses supplied by the caller
    rs = np.random.RandomState(3)
    v = rs.normal(size=(40, 3))
    v[:, 2] = np.abs(v[:, 2]) + 0.2
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    many = dict(base[2], init_vp=v)
    kw = dict(do_split=False, num_iter=12, final_convergence=-1)
    got = gem.em_batch([many], **kw)[0]
    ref = em.expectation_maximisation(many["l"].copy(), many["lp"].copy(), many["cnn_response"].copy(),
                                      sphere_image=many["sphere_image"], init_vp=v, **kw)
    assert got["status"] == 0 and got["iterations"] == ref["iterations"]
    assert np.array_equal(got["vp_assoc"], ref["vp_assoc"])
    assert abserr(got["vp"], ref["vp"]) <= 1e-4


@pytest.mark.parametrize("name", ["tiny_n12", "clean3_n60", "yud_n120", "yud_n250", "nosplit_n150"])
def test_distribution_matches_the_reference(name):
    """EM_result['distribution'] (vp_localisation.py:441): the PDF tuple of the reference's last calc_probabilities call
    (tests/golden/dist_<case>.npz, oracle/make_dist_golden.py) against vpk_em_set_distribution_out -- same shapes
    (probability_functions.py:120), values to the accuracy the E-step unit tests hold after a whole run."""
    import os
    from conftest import GOLDEN
    from vanishing_points_2017_amd import em as gem, vp_localisation
    from vanishing_points_2017_amd.probability_functions import PDF
    g = load(name)
    want = dict(np.load(os.path.join(GOLDEN, "dist_" + name + ".npz")))
    kw = {k: v for k, v in em_kwargs(g).items() if k != "init_vp"}
    res = gem.em_batch([_scene(g)], want_distribution=True, **kw)[0]
    check_em_result(res, g)
    p = res["distribution"]
    assert isinstance(p, PDF)
    for field, key in (("v", "p_v"), ("lv", "p_lv"), ("vl", "p_vl"), ("l", "p_l"), ("lvsq", "lvsq"), ("angles", "angles")):
        assert getattr(p, field).shape == want[key].shape, field
    # observed: angles 4e-13, p_v 4e-13 (relative), lvsq 2e-13, p_lv 2e-11 of its maximum, p_l 7e-11 (relative), p_vl 2e-12
    assert abserr(p.angles, want["angles"]) <= 1e-10
    assert relerr(p.v, want["p_v"]) <= 1e-10
    assert abserr(p.lvsq, want["lvsq"]) <= 1e-10
    scale = np.abs(want["p_lv"]).max()
    assert abserr(p.lv, want["p_lv"]) <= 1e-9 * scale
    assert relerr(p.l, want["p_l"]) <= 1e-8
    assert abserr(p.vl, want["p_vl"]) <= 1e-9                        # posteriors in [0, 1]
    assert np.allclose(p.vl.sum(axis=0)[p.l > 1e-12], 1.0, atol=1e-9)   # calc_pvl: columns sum to one
    # without the request the key stays None (and the drop-in surface returns the tuple when asked)
    assert gem.em_batch([_scene(g)], **kw)[0]["distribution"] is None
    r2 = vp_localisation.expectation_maximisation(g["l"].copy(), g["lp"].copy(), g["cnn_response"],
                                                  sphere_image=g["sphere_image"], return_distribution=True, **kw)
    assert isinstance(r2["distribution"], PDF) and np.array_equal(r2["distribution"].vl, p.vl)


def test_distribution_request_is_consumed_by_one_call_and_validated():
    """vpk_em_set_distribution_out applies to the next vpk_em_batch call only; incomplete buffer sets and time-sliced
    launches are refused (include/vpk.h)."""
    import ctypes
    import torch
    from vanishing_points_2017_amd import _lib, em as gem
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0)
    g = load("clean3_n60")
    d = gem.upload_batch(rt, [_scene(g)])
    p = gem._params({})
    l0 = d["l"].clone()
    out1 = gem.em_batch_device(rt, d["offsets"], d["l"], d["lp"], d["cnn"], d["sphere"], None, p, want_distribution=True)
    rt.synchronize()
    keep = {k: v.clone() for k, v in out1["dist"].items()}
    assert float(keep["p_vl"].abs().sum()) > 0
    for v in out1["dist"].values():
        v.fill_(-7.0)
    d["l"].copy_(l0)
    gem.em_batch_device(rt, d["offsets"], d["l"], d["lp"], d["cnn"], d["sphere"], None, p)      # no request: nothing written
    rt.synchronize()
    assert all(bool((v == -7.0).all()) for v in out1["dist"].values())
    bad = _lib.EmDistOut(rt.ptr(keep["p_v"]), None, rt.ptr(keep["p_l"]), rt.ptr(keep["p_lv"]), rt.ptr(keep["p_vl"]),
                         rt.ptr(keep["lvsq"]))
    assert rt.lib.vpk_em_set_distribution_out(rt.h, ctypes.byref(bad)) != 0
    rt.handle.em_set_time_slice(5.0, 64)
    try:
        ok = _lib.EmDistOut(*[rt.ptr(keep[k]) for k in ("p_v", "angles", "p_l", "p_lv", "p_vl", "lvsq")])
        assert rt.lib.vpk_em_set_distribution_out(rt.h, ctypes.byref(ok)) != 0
    finally:
        rt.handle.em_set_time_slice(0.0, 0)
    assert rt.lib.vpk_em_set_distribution_out(rt.h, None) == 0


def _with_smoother(mode, fn):
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0)
    rt.handle.em_set_smoother(mode)
    try:
        return fn()
    finally:
        rt.handle.em_set_smoother(0)


def test_row_sliced_smoother_gives_the_bits_of_the_round2_kernels():
    """vpk_em_set_smoother: the row-sliced weight_matrix kernel (default) and the round-1/2 kernels sum every
    (column, VP) in the same order -- eight row slices, ascending rows, slices in order -- so whole EM runs agree in
    EVERY output bit: VPs, variances, the decision metric, iteration counts, assignments.  YUD-shape images (102, the
    bench's batch: every N from 100 to 400, up to 34 hypotheses), small and odd line counts (slices that are short or
    empty), and ECD-shape images that do not fit the row-sliced panel and fall back."""
    from vanishing_points_2017_amd import em as gem, synth
    scenes = gpu_rasters(synth.config_scenes(2, count=102))
    for n in (3, 7, 8, 9, 15, 17, 33, 57, 63, 64, 65, 66, 127, 129):
        sc = synth.make_scene(900 + n, max(n, 40), 3)
        scenes.append(dict(sc, l=sc["l"][:n].copy(), lp=sc["lp"][:n].copy()))
    scenes += list(synth.config_scenes(3, count=6))          # ECD-shape: 300..1200 lines, partly in passes
    scenes += [synth.make_scene(8300 + n, n, 5) for n in (1300, 1600)]
    scenes = gpu_rasters(scenes)
    new = _with_smoother(0, lambda: gem.em_batch([dict(s, l=s["l"].copy()) for s in scenes], want_metric=True))
    old = _with_smoother(1, lambda: gem.em_batch([dict(s, l=s["l"].copy()) for s in scenes], want_metric=True))
    rows = _with_smoother(2, lambda: gem.em_batch([dict(s, l=s["l"].copy()) for s in scenes], want_metric=True))
    ok = 0
    for a, b, r3 in zip(new, old, rows):                  # row-sliced (default) | round-1/2 kernels | sparse (round 4, optional)
        assert a["status"] == b["status"] == r3["status"]
        if a["status"] != 0:
            continue
        ok += 1
        assert a["iterations"] == b["iterations"] == r3["iterations"]
        for key in ("vp", "sigma", "counts", "counts_weighted", "vp_assoc", "decision_metric"):
            assert np.array_equal(a[key], b[key]), key
            assert np.array_equal(a[key], r3[key]), key
    assert ok >= 112


@pytest.mark.parametrize("n", [511, 512, 513, 700, 1000, 1023, 1200])
def test_tiled_pair_pass_gives_the_bits_of_the_row_by_row_pass(n):
    """calc_lsim + line_rating_knn for images of 512 lines and more (round 6): pass 1 walks tiles of 16 rows x 64 columns so that a
    mirrored 128-byte line is written whole by one wave (em_device.hpp: pairwise_tiles).  Same pair function, same positions:
    lsim, the kNN score and the line angles must equal the row-by-row pass (vpk_em_set_smoother(1)) bit for bit; 511 lines
    take the row-by-row pass under both settings."""
    from vanishing_points_2017_amd import kernels, synth
    sc = synth.make_scene(4400 + n, n, 5)
    new = _with_smoother(0, lambda: kernels.pairwise(sc["lp"]))
    old = _with_smoother(1, lambda: kernels.pairwise(sc["lp"]))
    for a, b in zip(new, old):
        assert np.array_equal(a, b)
    assert np.array_equal(new[0], new[0].T)


@pytest.mark.parametrize("n,m", [(100, 25), (245, 22), (380, 24), (400, 32), (64, 3), (65, 9), (9, 2), (500, 17),
                                 (900, 8), (1000, 8), (1100, 8), (700, 30), (1100, 20), (1500, 12), (1700, 9), (1800, 8), (600, 40)])
def test_weight_matrix_row_sliced_vs_round2_vs_numpy(n, m):
    """vpk_weight_matrix under both smoother settings on random inputs: identical bits, and both within rounding of
    the NumPy expression of vp_localisation.py:515-524.  (900..1024 lines: the round-2 kernel's single-chain mode,
    which the row-sliced kernel leaves alone; 700 x 30 ... 1700 x 9: the whole panel does not fit the LDS and the
    row-sliced kernel runs in passes of 8-24 VPs, like the round-2 kernel did; 1800 lines: past it.)"""
    from vanishing_points_2017_amd import kernels
    rng = np.random.RandomState(n * 100 + m)
    lsim = rng.rand(n, n) ** 6
    lsim = lsim + lsim.T
    np.fill_diagonal(lsim, 0.0)
    pvl = rng.rand(m, n)
    pvl /= pvl.sum(0)
    lw = rng.rand(n) * 0.4
    new = _with_smoother(0, lambda: kernels.weight_matrix(pvl, lw, lsim))
    old = _with_smoother(1, lambda: kernels.weight_matrix(pvl, lw, lsim))
    assert np.array_equal(new, old)
    assert np.array_equal(new, _with_smoother(2, lambda: kernels.weight_matrix(pvl, lw, lsim)))
    w_ = pvl * lw
    want = (w_ + lw * (w_ @ lsim)) / (1 + lw * lsim.sum(0))
    assert relerr(new, want) <= 1e-12
