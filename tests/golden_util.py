import os

import numpy as np

from conftest import GOLDEN


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def em_kwargs(g):
    kw = {}
    for k in g:
        if k.startswith("kw_"):
            v = g[k]
            kw[k[3:]] = v.item() if v.ndim == 0 else v
    if "init_vp" in g:
        kw["init_vp"] = g["init_vp"]
    return kw


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.nanmax(np.abs(a - b) / np.maximum(1e-300, np.maximum(np.abs(a), np.abs(b)))))


def abserr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.nanmax(np.abs(a - b))) if a.size else 0.0


def check_em_result(res, g, vp_tol=1e-4):
    """The parity bar of BASELINE.json: line->VP assignments bit-exact, VP directions <= 1e-4."""
    if int(g["o_status"]) != 0:
        assert res["vp"] is None
        return
    assert res["vp"] is not None
    assert res["iterations"] == int(g["o_iterations"])
    assert res["vp"].shape == g["o_vp"].shape
    assert np.array_equal(res["vp_assoc"], g["o_vp_assoc"])
    assert abserr(res["vp"], g["o_vp"]) <= vp_tol
    assert np.array_equal(res["counts"], g["o_counts"])
    assert relerr(res["counts_weighted"], g["o_counts_weighted"]) <= 1e-9
    assert relerr(res["sigma"], g["o_sigma"]) <= 1e-4
