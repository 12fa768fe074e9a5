"""calc_horizon / auc ports against reference-generated goldens (host code, no GPU)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_cases
from golden_util import load
from vanishing_points_2017_amd import auc, calc_horizon


@pytest.mark.parametrize("name", golden_cases())
def test_horizon_matches_reference(name):
    g = load(name)
    if int(g["o_status"]) != 0:
        pytest.skip("no VPs")
    res = {"vp": g["o_vp"], "counts": g["o_counts"]}
    hp1, hp2, zvp, hvp1, hvp2, combo = calc_horizon.calculate_horizon_and_ortho_vp(res, maxbest=20,
                                                                                   theta_vmin=np.pi / 10)
    assert np.array_equal(hp1, g["h_hP1"]) and np.array_equal(hp2, g["h_hP2"])
    assert np.array_equal(np.asarray(zvp, dtype=np.float64), g["h_zVP"])
    assert np.array_equal(hvp1, g["h_hVP1"]) and np.array_equal(hvp2, g["h_hVP2"])
    assert np.array_equal(np.asarray(combo), g["h_best_combo"])


def test_horizon_fallbacks():
    for m in (0, 1, 2):
        vps = np.array([[0.1, 0.2, 0.97], [0.9, 0.1, 0.4]])[:m].reshape(m, 3)
        out = calc_horizon.calculate_horizon_and_ortho_vp({"vp": vps, "counts": np.ones(m)}, maxbest=20)
        assert len(out) == 6 and np.all(np.isfinite(out[0][:2]) | (m == 2))


def test_auc_matches_reference():
    g = dict(np.load(os.path.join(GOLDEN, "auc.npz")))
    for k in range(5):
        a, pts = auc.calc_auc(g["err%d" % k], cutoff=0.25)
        assert a == float(g["auc%d" % k])
        assert np.array_equal(pts, g["pts%d" % k])
