"""Golden raster of the reference's `alternative` curve (sphere_mapping.py:58-59).

TEST INFRASTRUCTURE, build container only (needs /root/reference; see ref_shim.py).  Runs the reference's own
``sphere_line_plot(lines, 500, alpha=0.1, alternative=True)`` under the installed matplotlib (line width of the pinned
version, see make_golden.reference_raster) on seeded line sets and stores lines + rasters in
tests/golden/rasteralt.npz.  Only data is written.

Usage:  python oracle/make_raster_alt_golden.py
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
from vanishing_points_2017_amd import synth  # noqa: E402


def main():
    warnings.filterwarnings("ignore")
    import matplotlib
    matplotlib.rcParams["lines.linewidth"] = 1.0      # matplotlib 1.5.1 default (requirements.txt:7)
    sm = load_reference(["sphere_mapping"])["sphere_mapping"]
    out = {}
    sets = {"a": synth.make_scene(77, 40, 3)["l"], "b": synth.make_scene(2001, 120, 3)["l"],
            # lines the curve's pole treats differently: through the principal point, axis-parallel, far away, degenerate
            "c": np.array([[1, 0.0, 0.3], [0.0, 1.0, 0.2], [0.3, -1e-6, -2.0], [1, 1, 1e6], [5, 0.01, 0.01], [0, 0, 1.0],
                           [0.7, 0.7, 0.0], [1e-9, 1.0, 0.5]])}
    for k, l in sets.items():
        out["l_" + k] = np.ascontiguousarray(l, dtype=np.float64)
        out["raster_" + k] = sm.sphere_line_plot(out["l_" + k].copy(), 500, alpha=0.1, f=1.0, alternative=True)
        print(k, out["l_" + k].shape[0], "lines, mean grey %.3f" % out["raster_" + k].mean())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rasteralt.npz"), **out)


if __name__ == "__main__":
    main()
