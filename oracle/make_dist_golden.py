"""EM_result['distribution'] of the REFERENCE (vp_localisation.py:441: the probability_functions.PDF of its last
calc_probabilities call) for a few golden cases -> tests/golden/dist_<case>.npz.

TEST INFRASTRUCTURE, build container only (needs /root/reference; see ref_shim.py).  The inputs are the ones stored
in tests/golden/<case>.npz (same lines, CNN response, raster, keywords); the run's vp / vp_assoc are checked against
that golden before anything is written.  Only data is written.

Usage:  python oracle/make_dist_golden.py [case ...]
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_shim import load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
DEFAULT_CASES = ["tiny_n12", "clean3_n60", "yud_n120", "yud_n250", "nosplit_n150"]


def main(argv):
    import joblib
    warnings.filterwarnings("ignore")
    mods = load_reference()
    vpl = mods["vp_localisation"]
    for name in (argv or DEFAULT_CASES):
        g = dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True))
        kwargs = {k[3:]: g[k].item() for k in g if k.startswith("kw_")}
        if "init_vp" in g:
            kwargs["init_vp"] = g["init_vp"]
        with joblib.parallel_backend("multiprocessing"):
            res = vpl.expectation_maximisation(g["l"].copy(), g["lp"].copy(), g["cnn_response"].copy(),
                                               sphere_image=g["sphere_image"], **kwargs)
        assert np.array_equal(res["vp"], g["o_vp"]) and np.array_equal(res["vp_assoc"], g["o_vp_assoc"]), name
        p = res["distribution"]
        np.savez_compressed(os.path.join(GOLDEN, "dist_" + name + ".npz"), p_v=p.v, p_lv=p.lv, p_vl=p.vl, p_l=p.l,
                            lvsq=p.lvsq, angles=p.angles)
        print("%-14s M=%d N=%d  p_v %s lv %s vl %s l %s lvsq %s angles %s" % (
            name, res["vp"].shape[0], g["lp"].shape[0], p.v.shape, p.lv.shape, p.vl.shape, p.l.shape, p.lvsq.shape,
            p.angles.shape))


if __name__ == "__main__":
    main(sys.argv[1:])
