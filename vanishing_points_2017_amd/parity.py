"""Comparison of EM results with stored results of the REFERENCE (tests/golden/full_c<config>.npz).

The files are written in the build container by make_full_goldens.py (test infrastructure), which runs the
reference's own expectation_maximisation (vp_localisation.py:168-450) and
calculate_horizon_and_ortho_vp (calc_horizon.py:19-225) over every seeded scene of a BASELINE.json
config.  This module only reads that data and applies the parity bar of BASELINE.json's north_star:
line->VP assignments bit-exact, iteration count equal, VP directions within 1e-4.  Used by the -m gpu
tests and by bench.py's "parity" object; it computes nothing of the hot path itself."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VP_TOL = 1e-4


def golden_path(config_id):
    return os.path.join(ROOT, "tests", "golden", "full_c%d.npz" % config_id)


def input_sha(scene):
    """First 8 bytes of sha1(l | lp | cnn_response) -- ties a stored result to the inputs the reference was given.  The
    raster is not an input: the reference makes it from the lines (evaluation.py:175), see raster_sha."""
    h = hashlib.sha1()
    for k in ("l", "lp", "cnn_response"):
        h.update(np.ascontiguousarray(scene[k]).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def raster_sha(sphere_image):
    """First 8 bytes of sha1 of a 500 x 500 uint8 raster: the stored value is that of the reference's own
    sphere_line_plot (sphere_mapping.py:36-72) on the scene's lines."""
    h = hashlib.sha1(np.ascontiguousarray(sphere_image, dtype=np.uint8).tobytes())
    return np.frombuffer(h.digest()[:8], dtype=np.uint64)[0]


def instability_certificates(path=None):
    """{(config, image): certificate} from tests/golden/instability.npz (written in the build container by the test
    infrastructure from runs of the REFERENCE itself): for an image the HIP path once failed the bar on, how far the
    reference's own result moves when one input coordinate moves by one ulp.  'unstable' = the reference's own VPs moved
    by more than the 1e-4 bar, or assignments flipped, or its iteration / VP count changed."""
    path = path or os.path.join(ROOT, "tests", "golden", "instability.npz")
    if not os.path.isfile(path):
        return {}
    g = np.load(path, allow_pickle=False)
    out = {}
    for k in range(len(g["config"])):
        out[(int(g["config"][k]), int(g["index"][k]))] = {
            "trials": int(g["trials"][k]), "max_vp_move": float(g["max_vp_move"][k]),
            "max_assoc_flips": int(g["max_assoc_flips"][k]), "iterations_stable": bool(g["iterations_stable"][k]),
            "num_vp_stable": bool(g["num_vp_stable"][k]), "unstable": bool(g["unstable"][k])}
    return out


class ReferenceResults(object):
    """Per-image views into one full_c<config>.npz."""

    def __init__(self, config_id, path=None):
        g = np.load(path or golden_path(config_id), allow_pickle=False)
        self.config_id = config_id
        self.g = {k: g[k] for k in g.files}
        self.index = self.g["index"]
        self._lo = np.concatenate([[0], np.cumsum(self.g["n_lines"])])
        self._vo = np.concatenate([[0], np.cumsum(self.g["num_vp"])])
        self._pos = {int(i): k for k, i in enumerate(self.index)}

    def __len__(self):
        return len(self.index)

    def has(self, image_index):
        return int(image_index) in self._pos

    def get(self, image_index):
        k = self._pos[int(image_index)]
        g = self.g
        lo, hi, vo, vh = self._lo[k], self._lo[k + 1], self._vo[k], self._vo[k + 1]
        return {"status": int(g["status"][k]), "iterations": int(g["iterations"][k]),
                "vp_assoc": g["assoc"][lo:hi].astype(np.int64), "vp": g["vp"][vo:vh], "sigma": g["sigma"][vo:vh],
                "counts": g["counts"][vo:vh], "counts_weighted": g["counts_w"][vo:vh],
                "hP1": g["hP1"][k], "hP2": g["hP2"][k], "combo": g["combo"][k],
                "input_sha": g["input_sha"][k], "raster_sha": g["raster_sha"][k] if "raster_sha" in g else None,
                "ref_seconds": float(g["ref_seconds"][k]),
                "events": {e: int(g["ev_" + e][k]) for e in ("split", "merge", "abort", "final_merge")}}


def compare_one(res, ref):
    """One image: dict of booleans / deltas.  ``res``: a reference-style result dict with 'status'
    (em.em_batch); ``ref``: ReferenceResults.get()."""
    out = {"status": res["status"] == ref["status"], "iterations": False, "assoc": False, "counts": False,
           "num_vp": False, "vp_err": np.inf, "assoc_diff": -1}
    if ref["status"] != 0 or res["status"] != 0:
        ok = out["status"]
        out.update(iterations=ok, assoc=ok, counts=ok, num_vp=ok, vp_err=0.0 if ok else np.inf, assoc_diff=0 if ok else -1)
        return out
    out["iterations"] = res["iterations"] == ref["iterations"]
    out["num_vp"] = res["vp"].shape == ref["vp"].shape
    if res["vp_assoc"].shape == ref["vp_assoc"].shape:
        out["assoc_diff"] = int((res["vp_assoc"] != ref["vp_assoc"]).sum())
        out["assoc"] = out["assoc_diff"] == 0
    if out["num_vp"]:
        out["vp_err"] = float(np.abs(res["vp"] - ref["vp"]).max()) if ref["vp"].size else 0.0
        out["counts"] = bool(np.array_equal(res["counts"], ref["counts"]))
    return out


def passes(c):
    return bool(c["status"] and c["iterations"] and c["assoc"] and c["num_vp"] and c["counts"] and c["vp_err"] <= VP_TOL)


def summarise(comparisons):
    """{image_index: compare_one(...)} -> the counts bench.py reports."""
    n = len(comparisons)
    return {"images": n,
            "assoc_exact": sum(1 for c in comparisons.values() if c["assoc"]),
            "iterations_equal": sum(1 for c in comparisons.values() if c["iterations"]),
            "vp_within_1e-4": sum(1 for c in comparisons.values() if c["num_vp"] and c["vp_err"] <= VP_TOL),
            "all_criteria": sum(1 for c in comparisons.values() if passes(c)),
            "failing_images": sorted(int(i) for i, c in comparisons.items() if not passes(c))}
