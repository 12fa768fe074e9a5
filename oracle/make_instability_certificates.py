"""Instability certificates: is the REFERENCE's own answer on an image defined to the last bit of its inputs?

TEST INFRASTRUCTURE, build container only (needs /root/reference; see ref_shim.py).  Input: the images on which the HIP
path missed the parity bar against the reference's stored results (the list scripts/full_parity.py writes on the GPU box,
gpurun_out/parity_failures.json, or "config:image" arguments).  For each of them the reference itself -- its
sphere_line_plot raster (sphere_mapping.py:36-72) and expectation_maximisation (vp_localisation.py:168-450), loaded by
ref_shim -- is run on the image and on `trials` copies whose segment end points differ by ONE ulp in one coordinate of one
line (seeded choice) -- and, with --all T2, on T2 further copies in which EVERY coordinate moved by -1, 0 or +1 ulp (seeded):
the same bound on the input change, but one that reaches every intermediate of the run, as another implementation's exp /
acos does; a degenerate hypothesis (a VP that two lines pin down, variance collapsed to ~1e-30, whose next E-step depends on
whether 1 - |cos| rounds to exactly 0) only shows under such a change unless the single coordinate happens to belong to one of
its two lines --, and tests/golden/instability.npz records per image

    config, index, trials, trials_all, iterations (unperturbed run)
    max_vp_move        largest change of a VP component among the perturbed runs with the same VP count
    max_assoc_flips    most line->VP assignments that changed
    iterations_stable  every perturbed run took the unperturbed run's iteration count
    num_vp_stable      ... and returned as many VPs
    unstable           max_vp_move > 1e-4, or an assignment flipped, or iterations / VP count changed: the parity bar
                       (assignments bit-exact, VPs 1e-4) is not something the reference itself meets on this image

A one-ulp input change stands for what any other implementation of exp / acos / asin does to the EM's intermediates
(tests/test_gpu_math.py measures the device's: up to 1-2 ulp, a few per cent of the arguments).  The test
(tests/test_gpu_full_configs.py) exempts an image from the full bar only if its certificate says `unstable`.

Usage:  python oracle/make_instability_certificates.py [--trials T] [--all T2] [--from gpurun_out/parity_failures.json] [cfg:idx ...]
"""
import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from ref_shim import load_reference  # noqa: E402
from vanishing_points_2017_amd import synth  # noqa: E402
from make_golden import reference_raster  # noqa: E402
from ref_instability import perturbations  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "instability.npz")
FIELDS = ("config", "index", "trials", "trials_all", "iterations", "max_vp_move", "max_assoc_flips", "iterations_stable", "num_vp_stable",
          "unstable")


def certify(mods, cfg, idx, trials, trials_all=0):
    import joblib
    vpl = mods["vp_localisation"]
    sc = next(synth.config_scenes(cfg, count=1, start=idx))
    sphere = reference_raster(mods["sphere_mapping"], sc["l"])

    def run(lp):
        with joblib.parallel_backend("multiprocessing"):
            return vpl.expectation_maximisation(sc["l"].copy(), lp.copy(), sc["cnn_response"].copy(), sphere_image=sphere)
    base = run(sc["lp"])
    move, flips, it_ok, nv_ok = 0.0, 0, True, True
    for i, j, q in perturbations(sc["lp"], trials, seed=idx, trials_all=trials_all):
        r = run(q)
        it_ok &= r["iterations"] == base["iterations"]
        same = r["vp"].shape == base["vp"].shape
        nv_ok &= same
        if same:
            move = max(move, float(np.abs(r["vp"] - base["vp"]).max()))
        flips = max(flips, int((r["vp_assoc"] != base["vp_assoc"]).sum()))
        print("  c%d #%d line %d coord %d: %d iterations, %d VPs, %d flips, VP move %s" % (
            cfg, idx, i, j, r["iterations"], r["vp"].shape[0], int((r["vp_assoc"] != base["vp_assoc"]).sum()),
            "%.3g" % np.abs(r["vp"] - base["vp"]).max() if same else "n/a"), flush=True)
    unstable = move > 1e-4 or flips > 0 or not it_ok or not nv_ok
    return {"config": cfg, "index": idx, "trials": trials, "trials_all": trials_all, "iterations": int(base["iterations"]), "max_vp_move": move,
            "max_assoc_flips": flips, "iterations_stable": it_ok, "num_vp_stable": nv_ok, "unstable": unstable}


def main(argv):
    warnings.filterwarnings("ignore")
    trials, trials_all, todo = 6, 0, []
    while argv:
        a = argv.pop(0)
        if a == "--trials":
            trials = int(argv.pop(0))
        elif a == "--all":
            trials_all = int(argv.pop(0))
        elif a == "--from":
            for cfg, lst in json.load(open(argv.pop(0))).items():
                todo += [(int(cfg), int(i)) for i in lst]
        else:
            cfg, idx = a.split(":")
            todo.append((int(cfg), int(idx)))
    have = {}
    if os.path.isfile(OUT):
        g = np.load(OUT)
        for k in range(len(g["config"])):
            have[(int(g["config"][k]), int(g["index"][k]))] = {f: (g[f][k].item() if f in g.files else 0) for f in FIELDS}
    mods = load_reference()
    for cfg, idx in todo:
        c = certify(mods, cfg, idx, trials, trials_all)
        have[(cfg, idx)] = c
        print("config %d image %d: %s" % (cfg, idx, c), flush=True)
        keys = sorted(have)
        np.savez_compressed(OUT, **{f: np.array([have[k][f] for k in keys]) for f in FIELDS})


if __name__ == "__main__":
    main(sys.argv[1:])
