"""Logic check of the UNMODIFIED EM device source (csrc/em_device.hpp) on CPU.

tests/hostsim compiles the device code with g++ against a single-lane stand-in for
wave_prims.hpp and runs it here against the reference-generated goldens and sklearn.  This is
test infrastructure (it lets the CPU suite exercise the kernel's control flow); the parity
tests proper are the -m gpu ones, which call the HIP build through the C-ABI."""
import numpy as np
import pytest

from conftest import golden_cases
from golden_util import abserr, check_em_result, em_kwargs, load, relerr
from hostsim import simlib
from oracle import em_numpy as em

CASES = golden_cases()


@pytest.mark.parametrize("name", CASES)
def test_full_run_matches_reference_golden(name):
    g = load(name)
    kw = em_kwargs(g)
    res = simlib.em_single(g["l"].copy(), g["lp"], g["cnn_response"], g["sphere_image"], **kw)
    if res["status"] != 0:
        res["vp"] = None
    assert abserr(res["l"], g["l_normalised"]) <= 1e-15
    check_em_result(res, g, vp_tol=1e-9)


@pytest.mark.parametrize("name", [c for c in CASES if "i_v0" in load(c)])
def test_pieces(name):
    g = load(name)
    lsim, lscore, langle = simlib.pairwise(g["lp"])
    if "i_lsim" in g:
        assert abserr(lsim, g["i_lsim"]) <= 1e-13
    assert abserr(lsim.sum(axis=1), g["i_lsim_rowsum"]) <= 1e-11
    assert np.array_equal(lsim, lsim.T)
    assert abserr(lscore, g["i_lscore"]) <= 1e-13
    assert abserr(langle, g["i_langles"]) <= 1e-14
    v0, w = simlib.init_vps(g["cnn_response"], g["sphere_image"])
    assert abserr(v0, g["i_v0"]) <= 1e-14
    assert np.array_equal(w, g["i_pdf_weights"])          # float32 prior weights bit-exact
    m0 = v0.shape[0]
    s = np.ones(m0) * (np.pi / (1.282 * 20)) * 1e-6
    pv, lvsq, pvl, _ = simlib.estep(g["lp"], g["cnn_response"], g["i_v0"], s)
    assert relerr(pv, g["i_p_v0"]) <= 1e-12
    assert abserr(lvsq.T, g["i_lvsq0"]) <= 1e-14
    assert relerr(pvl, g["i_p_vl0"]) <= 1e-7
    w0 = simlib.weight_matrix(g["i_p_vl0"], g["i_lweight"], em.calc_lsim(g["lp"], sigma=1))
    assert relerr(w0, g["i_w0"]) <= 1e-10


def test_cluster2_matches_sklearn():
    import warnings
    import sklearn.cluster as cluster
    rs = np.random.RandomState(5)
    for n in (9, 17, 40, 83):
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
        labels, flags = simlib.cluster2(ld)
        if flags == 0:          # no exact tie: the merge order is unambiguous
            assert np.array_equal(labels, model.labels_)


def test_degenerate_no_initial_vp():
    g = load("tiny_n12")
    res = simlib.em_single(g["l"].copy(), g["lp"], g["cnn_response"], np.zeros((500, 500), np.uint8))
    assert res["status"] == 2      # the reference raises ValueError from np.vstack([]) (vp_localisation.py:165)
    with pytest.raises(ValueError):
        em.expectation_maximisation(g["l"].copy(), g["lp"].copy(), g["cnn_response"].copy(),
                                    sphere_image=np.zeros((500, 500), np.uint8))
