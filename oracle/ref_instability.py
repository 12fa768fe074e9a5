"""Is the REFERENCE itself stable on a given scene?  (build container only; test infrastructure)

Runs the reference's own expectation_maximisation (vp_localisation.py:168-450, loaded by ref_shim)
on a seeded scene and on copies whose segment end points differ by ONE ulp in one coordinate of one
line, and reports how many line->VP assignments and how far the VPs move.  A scene where a 1-ulp
input change flips assignments has no well-defined "bit-exact" answer: any implementation whose
transcendentals differ from NumPy/glibc in the last bit (ocml on the GPU) lands on another member
of the same family.  Used for configs[1] scene 86.

Usage: python oracle/ref_instability.py <config> <index> [trials]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402
from vanishing_points_2017_amd import synth  # noqa: E402


def perturbations(lp, trials, seed=0, trials_all=0):
    """`trials` copies with ONE coordinate of ONE line moved by one ulp, then `trials_all` copies with EVERY coordinate moved
    by -1, 0 or +1 ulp (seeded): still an input within one ulp of the original everywhere, but one that reaches every
    intermediate of the EM the way another exp / acos implementation does (yields line = coordinate = -1 for those)."""
    rs = np.random.RandomState(seed)
    for t in range(trials):
        q = lp.copy()
        i, j = rs.randint(lp.shape[0]), rs.randint(4)
        q[i, j] = np.nextafter(q[i, j], 10.0 if t % 2 else -10.0)
        yield i, j, q
    for t in range(trials_all):
        step = rs.randint(-1, 2, size=lp.shape)
        q = np.where(step > 0, np.nextafter(lp, 10.0), np.where(step < 0, np.nextafter(lp, -10.0), lp))
        yield -1, -1, q


def main(argv):
    warnings.filterwarnings("ignore")
    import joblib
    cfg, idx = int(argv[0]), int(argv[1])
    trials = int(argv[2]) if len(argv) > 2 else 4
    vpl = load_reference()["vp_localisation"]
    sc = next(synth.config_scenes(cfg, count=1, start=idx))

    def run(lp):
        with joblib.parallel_backend("multiprocessing"):
            return vpl.expectation_maximisation(sc["l"].copy(), lp.copy(), sc["cnn_response"].copy(),
                                                sphere_image=sc["sphere_image"])
    base = run(sc["lp"])
    print("reference, config %d scene %d: %d iterations, %d VPs" % (cfg, idx, base["iterations"], base["vp"].shape[0]))
    for i, j, q in perturbations(sc["lp"], trials):
        r = run(q)
        same = r["vp"].shape == base["vp"].shape
        print("  line %d coordinate %d moved by 1 ulp: %d iterations, %d VPs, %d assignments differ, max VP change %s"
              % (i, j, r["iterations"], r["vp"].shape[0], int((r["vp_assoc"] != base["vp_assoc"]).sum()),
                 ("%.3g" % np.abs(r["vp"] - base["vp"]).max()) if same else "n/a (VP count differs)"), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
