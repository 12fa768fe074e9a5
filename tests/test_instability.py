"""Why configs[1] scene 86 is exempt from the bit-exact bar (tests/test_gpu_full_configs.py: UNSTABLE).

The EM of that image never converges (99 iterations).  With the per-VP variance clamped to [1e-200, 1e-6]
(vp_localisation.py:306-307) one iteration amplifies a perturbation of a VP, and over 99 iterations a ONE-ulp
change of ONE input coordinate moves the final VPs by 1e-5 .. 1e-2 and flips line->VP assignments -- in the
CPU oracle here, and in the reference itself (oracle/ref_instability.py, numbers in DESIGN.md section 4).
An implementation whose exp/acos/asin differ from glibc's in the last bit (ocml on the GPU) is such a
perturbation.  The control scene converges in a few iterations and does not move at all."""
import numpy as np

from oracle import em_numpy as em
from vanishing_points_2017_amd import synth


def _spread(idx, trials):
    sc = next(synth.config_scenes(2, count=1, start=idx))

    def run(lp):
        return em.expectation_maximisation(sc["l"].copy(), lp.copy(), sc["cnn_response"].copy(),
                                           sphere_image=sc["sphere_image"])
    base = run(sc["lp"])
    rs = np.random.RandomState(0)
    moved, flipped = [], []
    for t in range(trials):
        lp = sc["lp"].copy()
        i, j = rs.randint(lp.shape[0]), rs.randint(4)
        lp[i, j] = np.nextafter(lp[i, j], 10.0 if t % 2 else -10.0)
        r = run(lp)
        assert r["vp"].shape == base["vp"].shape
        moved.append(np.abs(r["vp"] - base["vp"]).max())
        flipped.append(int((r["vp_assoc"] != base["vp_assoc"]).sum()))
    return base, moved, flipped


def test_scene_86_is_unstable_under_one_ulp_input_changes():
    base, moved, flipped = _spread(86, 6)
    assert base["iterations"] == 99
    assert max(moved) > 1e-4          # beyond the VP-direction bar of BASELINE.json
    assert max(flipped) >= 1          # and assignments flip


def test_a_converging_scene_is_stable():
    base, moved, flipped = _spread(0, 3)
    assert base["iterations"] < 20
    assert max(moved) < 1e-9 and max(flipped) == 0
