"""Run the REFERENCE's own raster + EM over whole BASELINE.json configs and store the results as fixtures.

TEST INFRASTRUCTURE, build container only (needs /root/reference; see ref_shim.py).  For every
seeded scene of a config (``vanishing_points_2017_amd.synth.config_scenes`` -- the generator behind
bench.py and the parity tests; it yields LINES and a response map, no raster) this does what the
reference does with a datum (evaluation.py:175, :332-350): ``sphere_image =
sphere_mapping.sphere_line_plot(lines, 500, alpha=0.1)`` (sphere_mapping.py:36-72, under the installed
matplotlib with the pinned version's default line width, see make_golden.reference_raster), then
``vp_localisation.expectation_maximisation`` (vp_localisation.py:168-450) on that raster and
``calc_horizon.calculate_horizon_and_ortho_vp`` (calc_horizon.py:19-225), and writes ONE compact
file per config:

    tests/golden/full_c<config>.npz
        index      image indices inside the config (seed = 1000 * config + index)
        n_lines    lines per image;   status  0 = VPs, 1 = all-None result, 2 = ValueError (:165)
        iterations, num_vp, ref_seconds
        assoc      concatenated vp_assoc (int16; offsets = cumsum(n_lines))
        vp / sigma / counts / counts_w   concatenated per-VP rows (offsets = cumsum(num_vp))
        hP1, hP2, combo                  horizon end points and best_combo (calc_horizon.py)
        ev_split / ev_merge / ev_abort / ev_final_merge    control-flow events seen in the reference
        input_sha  first 8 bytes of sha1(l | lp | cnn_response): the tests refuse to compare when the
                   regenerated inputs differ from the ones the reference saw
        raster_sha first 8 bytes of sha1(sphere_image) of the REFERENCE's raster: vpk_sphere_raster has to
                   reproduce it from the lines (the raster itself, 250 KB per image, is not stored)
        raster_sum sum of the raster's pixels (a second, human-readable check)

Only data is written -- no reference source text travels.  The event counters come from wrapping the
reference's merge_vps / split_best_vp in memory (nothing is written into the reference tree).

Usage:  python oracle/make_full_goldens.py <config> [count] [stride]
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
from vanishing_points_2017_amd.parity import input_sha, raster_sha  # noqa: E402
from make_golden import reference_raster  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def instrument(mods, ev, merge_thresh=1e-3):
    """Count control-flow events inside the reference (in-memory wrappers).  merge_thresh: the keyword the run
    uses (the final merge is called with ten times that, vp_localisation.py:190)."""
    vpl = mods["vp_localisation"]
    prob = mods["probability_functions"]
    orig_merge, orig_split, orig_prob = vpl.merge_vps, vpl.split_best_vp, prob.calc_probabilities
    state = {"in_merge": False, "esteps": 0}

    def calc_probabilities(*a, **k):
        if state["in_merge"]:
            state["esteps"] += 1
        return orig_prob(*a, **k)

    def merge_vps(i, v, s, l, thresh, *a, **k):
        mb = v.shape[1]
        state["in_merge"], state["esteps"] = True, 0
        try:
            out = orig_merge(i, v, s, l, thresh, *a, **k)
        finally:
            state["in_merge"] = False
        merged = mb - out["v"].shape[1]
        mt = merge_thresh[0] if isinstance(merge_thresh, list) else merge_thresh   # a list: read at call time
        final = thresh > 5 * mt
        ev["final_merge" if final else "merge"] += merged
        ev["abort"] += state["esteps"] - merged        # an E-step inside merge_vps that removed no VP (:668-670)
        return out

    def split_best_vp(i, v, s, *a, **k):
        mb = v.shape[1]
        out = orig_split(i, v, s, *a, **k)
        ev["split"] += out["v"].shape[1] - mb
        return out

    vpl.merge_vps, vpl.split_best_vp, prob.calc_probabilities = merge_vps, split_best_vp, calc_probabilities


def run_config(cfg, indices, out_path, mods):
    import joblib
    vpl = mods["vp_localisation"]
    ch = mods["calc_horizon"]
    ev = {"split": 0, "merge": 0, "abort": 0, "final_merge": 0}
    instrument(mods, ev)
    rec = {k: [] for k in ("index", "n_lines", "status", "iterations", "num_vp", "ref_seconds", "assoc", "vp", "sigma",
                           "counts", "counts_w", "hP1", "hP2", "combo", "ev_split", "ev_merge", "ev_abort",
                           "ev_final_merge", "input_sha", "raster_sha", "raster_sum")}
    with joblib.parallel_backend("multiprocessing"):
        for idx in indices:
            sc = next(synth.config_scenes(cfg, count=1, start=idx))
            sc["sphere_image"] = reference_raster(mods["sphere_mapping"], sc["l"])
            for k in ev:
                ev[k] = 0
            n = sc["lp"].shape[0]
            t0 = time.time()
            status = 0
            try:
                res = vpl.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                                   sphere_image=sc["sphere_image"])
                if res["vp"] is None:
                    status = 1
            except ValueError:
                res, status = None, 2
            dt = time.time() - t0
            rec["index"].append(idx); rec["n_lines"].append(n); rec["status"].append(status)
            rec["ref_seconds"].append(dt); rec["input_sha"].append(input_sha(sc))
            rec["raster_sha"].append(raster_sha(sc["sphere_image"])); rec["raster_sum"].append(int(sc["sphere_image"].sum()))
            for k in ev:
                rec["ev_" + k].append(ev[k])
            if status == 0:
                m = res["vp"].shape[0]
                rec["iterations"].append(res["iterations"]); rec["num_vp"].append(m)
                rec["assoc"].append(res["vp_assoc"].astype(np.int16))
                rec["vp"].append(res["vp"]); rec["sigma"].append(res["sigma"])
                rec["counts"].append(res["counts"]); rec["counts_w"].append(res["counts_weighted"])
                hp1, hp2, _, _, _, combo = ch.calculate_horizon_and_ortho_vp(res, maxbest=20, theta_vmin=np.pi / 10)
                rec["hP1"].append(hp1); rec["hP2"].append(hp2); rec["combo"].append(np.asarray(combo, dtype=np.int32))
            else:
                rec["iterations"].append(0); rec["num_vp"].append(0)
                rec["assoc"].append(np.full(n, -1, np.int16))
                rec["hP1"].append(np.zeros(3)); rec["hP2"].append(np.zeros(3)); rec["combo"].append(np.full(3, -1, np.int32))
            print("c%d #%d N=%d %.1fs status=%d iters=%s M=%s events=%s" % (
                cfg, idx, n, dt, status, rec["iterations"][-1], rec["num_vp"][-1], dict(ev)), flush=True)
            save(rec, out_path)


def save(rec, path):
    def cat(k, shape, dtype):
        return np.concatenate(rec[k]) if rec[k] else np.zeros(shape, dtype)
    out = {k: np.asarray(rec[k]) for k in ("index", "n_lines", "status", "iterations", "num_vp", "ref_seconds",
                                           "ev_split", "ev_merge", "ev_abort", "ev_final_merge", "input_sha", "raster_sha",
                                           "raster_sum")}
    out["assoc"] = cat("assoc", (0,), np.int16)
    out["vp"] = cat("vp", (0, 3), np.float64)
    for k in ("sigma", "counts", "counts_w"):
        out[k] = cat(k, (0,), np.float64)
    for k in ("hP1", "hP2", "combo"):
        out[k] = np.stack(rec[k])
    tmp = path + ".tmp.npz"
    np.savez_compressed(tmp, **out)
    os.replace(tmp, path)


def main(argv):
    warnings.filterwarnings("ignore")
    cfg = int(argv[0])
    total = synth.CONFIGS[cfg][1]
    count = int(argv[1]) if len(argv) > 1 else total
    stride = int(argv[2]) if len(argv) > 2 else 1
    indices = list(range(0, total, stride))[:count]
    os.makedirs(GOLDEN, exist_ok=True)
    mods = load_reference()
    run_config(cfg, indices, os.path.join(GOLDEN, "full_c%d.npz" % cfg), mods)


if __name__ == "__main__":
    main(sys.argv[1:])
