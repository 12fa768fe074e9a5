"""Generate tests/golden/*.npz by running the REFERENCE's own hot-path modules.

TEST INFRASTRUCTURE, build container only (needs /root/reference; see ref_shim.py).
The reference ships no tests or golden vectors, so parity is
pinned by these captured vectors: inputs + intermediates + outputs of
``vp_localisation.expectation_maximisation``, ``calc_horizon`` and ``auc`` on seeded
synthetic scenes (``vanishing_points_2017_amd.synth``).  Only data is written --
no reference source text travels.

Usage:  python oracle/make_golden.py [case ...]
"""
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_shim import load_reference  # noqa: E402
from vanishing_points_2017_amd import synth  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
EVENTS = {"split": 0, "merge": 0, "abort": 0, "final_merge": 0}      # filled by the wrappers of make_full_goldens.instrument
EVENTS_THRESH = [1e-3]

# name: dict(seed, n, vps, outlier_frac, em kwargs, stress?)
CASES = {
    "clean3_n60": dict(seed=1001, n=60, vps=3, outlier_frac=0.0),
    "yud_n120": dict(seed=2001, n=120, vps=3),
    "yud_n200": dict(seed=2002, n=200, vps=3),
    "yud_n250": dict(seed=2003, n=250, vps=4),
    "yud_n330": dict(seed=2007, n=330, vps=3),
    "ecd_n300_v8": dict(seed=3001, n=300, vps=8),
    "ecd_n400_v5": dict(seed=3002, n=400, vps=5),
    "nosplit_n150": dict(seed=2011, n=150, vps=3, em=dict(do_split=False, do_merge=False)),
    "noweights_n100": dict(seed=2012, n=100, vps=3, em=dict(use_weights=False)),
    "stress_n1000": dict(seed=5000, n=1000, vps=8, stress=True,
                         em=dict(num_iter=50, do_split=False, do_merge=False, final_convergence=-1)),
    "stress_n300": dict(seed=5001, n=300, vps=8, stress=True,
                        em=dict(num_iter=50, do_split=False, do_merge=False, final_convergence=-1)),
    "tiny_n12": dict(seed=2020, n=12, vps=2, outlier_frac=0.0),
    # inputs captured from the GPU pipeline (GPU-rasterised sphere image) where a VP wins exactly one
    # line in the hard-assignment M-step, so LAPACK's 1 x 3 null vector decides (:353-392)
    "hard1row_n387": dict(inputs="hard1row_n387"),
    "hard1row_n289": dict(inputs="hard1row_n289"),
    # control flow the default parameters rarely reach (searched with the CPU restatement's event trace):
    # a wider merge threshold makes merge_vps (:633-684) fire at i = 10 next to a split, merge several pairs in
    # the finalisation and ABORT on a pair whose pooled variance exceeds 0.01 (:666-670, s[k] written first)
    "periodicmerge_n220": dict(seed=2208, n=220, vps=6, em=dict(merge_thresh=0.12)),
    "mergeabort_n200": dict(seed=2102, n=200, vps=6, em=dict(merge_thresh=0.07)),
}
FULL_INTERMEDIATES_MAX_N = 260


def reference_raster(sm, l, size=500):
    import matplotlib
    matplotlib.rcParams["lines.linewidth"] = 1.0      # matplotlib 1.5.1 default (requirements.txt:7)
    return sm.sphere_line_plot(l.copy(), size, alpha=0.1, f=1.0)


def run_case(name, spec, mods):
    import joblib
    vpl = mods["vp_localisation"]
    prob = mods["probability_functions"]
    ch = mods["calc_horizon"]
    sm = mods["sphere_mapping"]
    if "inputs" in spec:     # stored inputs (tests/golden_inputs/*.npz): data only
        src = dict(np.load(os.path.join(ROOT, "tests", "golden_inputs", spec["inputs"] + ".npz")))
        sc = {"l": src["l"], "lp": src["lp"], "cnn_response": src["cnn_response"],
              "true_vps": np.zeros((0, 3)), "true_horizon": np.zeros(3)}
        sphere = src["sphere_image"]
    else:
        sc = synth.make_scene(spec["seed"], spec["n"], spec["vps"],
                              outlier_frac=spec.get("outlier_frac", 0.25), raster=None)
        sphere = reference_raster(sm, sc["l"])
    l0, lp = sc["l"], sc["lp"]
    cnn = sc["cnn_response"]
    kwargs = dict(spec.get("em", {}))
    init_vp = synth.stress_init_vps(spec["seed"]) if spec.get("stress") else None
    spec = dict(spec, n=lp.shape[0])
    out = {"l": l0, "lp": lp, "cnn_response": cnn, "sphere_image": sphere,
           "true_vps": sc["true_vps"], "true_horizon": sc["true_horizon"]}
    if init_vp is not None:
        out["init_vp"] = init_vp
        kwargs["init_vp"] = init_vp
    for k, v in spec.get("em", {}).items():
        out["kw_" + k] = np.array(v)
    ev = {"split": 0, "merge": 0, "abort": 0, "final_merge": 0}
    if not getattr(vpl, "_vpk_instrumented", False):
        from make_full_goldens import instrument
        instrument(mods, EVENTS, EVENTS_THRESH)
        vpl._vpk_instrumented = True
    EVENTS.update(ev)
    EVENTS_THRESH[0] = kwargs.get("merge_thresh", 1e-3)

    with joblib.parallel_backend("multiprocessing"):
        # --- intermediates, by calling the reference's own functions ---------------------------
        n = lp.shape[0]
        use_w = kwargs.get("use_weights", True)
        if n <= FULL_INTERMEDIATES_MAX_N and use_w:
            l = l0.copy()
            lsim = vpl.calc_lsim(lp, sigma=1)
            for i in range(n):
                l[i, :] /= np.linalg.norm(l[i, :])
            v0 = vpl.find_initial_vps(sphere, cnn, 25)
            pdfpar = prob.pdf_params(cnn)
            langles = vpl.lines_angles(lp)
            llen = np.array([np.linalg.norm(lp[i, 0:2] - lp[i, 2:4]) for i in range(n)])
            lscore = vpl.line_rating_knn(lp, k2=4)
            lweight = llen * np.clip(lscore, 0.2, 1)
            m0 = v0.shape[0]
            s = np.ones(m0) * pdfpar.sigma * 1e-6
            v = np.zeros((2, m0, 3))
            v[0] = v0
            p = prob.calc_probabilities(0, pdfpar, v, l, lp, s, llen, "angle")
            w = vpl.weight_matrix(p.vl, lweight, lsim, bias=1)
            counts, counts_w, assoc = vpl.calc_vp_line_counts(v[0], l, lp, s, w, lweight, "angle",
                                                              thresh=1.96 ** 2)
            newvp = np.array([vpl.calc_new_vanishing_point(l, w[m, :]) for m in range(m0)])
            out.update(i_lscore=lscore, i_lweight=lweight, i_v0=v0, i_pdf_weights=pdfpar.weights,
                       i_pdf_means=pdfpar.means, i_langles=langles, i_p_v0=p.v, i_lvsq0=p.lvsq,
                       i_p_vl0=p.vl, i_p_l0=p.l, i_w0=w, i_counts0=counts, i_assoc0=assoc,
                       i_mstep0=newvp, i_lsim_rowsum=lsim.sum(axis=1))
            if n <= 130:
                out["i_lsim"] = lsim
            else:
                out["i_lsim_rows"] = lsim[::17, :]

        # --- the full run -------------------------------------------------------------------
        t0 = time.time()
        l = l0.copy()
        res = vpl.expectation_maximisation(l, lp.copy(), cnn.copy(), sphere_image=sphere, **kwargs)
        dt = time.time() - t0
    out["ref_seconds"] = np.array(dt)
    out["o_events"] = np.array([EVENTS[k] for k in ("split", "merge", "abort", "final_merge")])
    out["l_normalised"] = l
    if res["vp"] is None:
        out["o_status"] = np.array(1)
    else:
        out["o_status"] = np.array(0)
        out.update(o_vp=res["vp"], o_vp_assoc=res["vp_assoc"], o_counts=res["counts"],
                   o_counts_weighted=res["counts_weighted"], o_sigma=res["sigma"],
                   o_iterations=np.array(res["iterations"]),
                   o_decision_metric_colmax=res["decision_metric"].max(axis=0))
        hp1, hp2, zvp, hvp1, hvp2, combo = ch.calculate_horizon_and_ortho_vp(
            res, maxbest=20, theta_vmin=np.pi / 10)
        out.update(h_hP1=hp1, h_hP2=hp2, h_zVP=np.asarray(zvp, dtype=np.float64), h_hVP1=hvp1,
                   h_hVP2=hvp2, h_best_combo=np.asarray(combo))
    print("%-16s N=%4d  %.1fs  iters=%s  M=%s" % (
        name, lp.shape[0], dt, res["iterations"], None if res["vp"] is None else res["vp"].shape[0]))
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)


def auc_golden(mods):
    """Golden vectors for auc.calc_auc (auc.py:5-37) on seeded error arrays."""
    auc = mods["auc"]
    rs = np.random.RandomState(77)
    out = {}
    for k, (n, scale) in enumerate([(77, 0.05), (78, 0.3), (2018, 0.12), (5, 0.01), (40, 1.0)]):
        err = np.abs(rs.standard_cauchy(n)) * scale
        a, pts = auc.calc_auc(err.copy(), cutoff=0.25)
        out["err%d" % k] = err
        out["auc%d" % k] = np.array(a)
        out["pts%d" % k] = pts
    np.savez_compressed(os.path.join(GOLDEN, "auc.npz"), **out)
    print("auc golden written")


def main(argv):
    warnings.filterwarnings("ignore")
    os.makedirs(GOLDEN, exist_ok=True)
    mods = load_reference()
    names = argv or (list(CASES) + ["auc"])
    for name in names:
        if name == "auc":
            auc_golden(mods)
        else:
            run_case(name, CASES[name], mods)


if __name__ == "__main__":
    main(sys.argv[1:])
