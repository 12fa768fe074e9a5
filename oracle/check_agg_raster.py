"""Pins oracle/agg_raster.py against matplotlib itself (build container only: needs the reference tree and matplotlib).

    python oracle/check_agg_raster.py [n_lines]

Renders seeded lines one at a time and as whole sets with the reference's own sphere_line_plot (sphere_mapping.py:36-72,
loaded through oracle/ref_shim.py) and with the restatement, and reports the pixels that differ."""
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
from oracle import agg_raster, ref_shim  # noqa: E402


def mpl_raster(sm, l, size=500):
    import matplotlib
    matplotlib.rcParams["lines.linewidth"] = 1.0      # matplotlib 1.5.1's default (requirements.txt:7)
    return sm.sphere_line_plot(np.asarray(l, dtype=np.float64).copy(), size, alpha=0.1, f=1.0)


def main():
    import matplotlib
    matplotlib.use("Agg")
    sm = ref_shim.load_reference(["sphere_mapping"])["sphere_mapping"]
    from vanishing_points_2017_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sc = synth.make_scene(4242, max(n, 40), 3, raster=None)
    bad_lines = 0
    for i in range(n):
        want = mpl_raster(sm, sc["l"][i:i + 1])
        got = agg_raster.raster(sc["l"][i:i + 1])
        d = np.abs(want.astype(int) - got.astype(int))
        if d.max() > 0:
            bad_lines += 1
            ys, xs = np.nonzero(d)
            print("line %d: %d pixels differ (max %d), e.g. (y=%d, x=%d): mpl %d, restatement %d; nonzero %d vs %d" % (
                i, len(ys), d.max(), ys[0], xs[0], want[ys[0], xs[0]], got[ys[0], xs[0]], (want > 0).sum(), (got > 0).sum()))
    print("single lines: %d of %d differ" % (bad_lines, n))
    t = time.time()
    want = mpl_raster(sm, sc["l"][:40])
    t1 = time.time()
    got = agg_raster.raster(sc["l"][:40])
    d = np.abs(want.astype(int) - got.astype(int))
    print("40 lines at once: %d pixels differ (max %d); matplotlib %.1f s, restatement %.1f s" % ((d > 0).sum(), d.max(), t1 - t, time.time() - t1))


if __name__ == "__main__":
    main()
