"""Pins oracle/agg_raster.py against matplotlib itself (build container only: needs the reference tree and matplotlib).

    python oracle/check_agg_raster.py [n_lines]

Renders seeded lines -- ordinary scene lines, near-vertical / flat / far-off / steep curves, degenerate ones (b = 0, all
zero) -- one at a time and as whole sets (also at size 250 and alpha 0.5) with the reference's own sphere_line_plot
(sphere_mapping.py:36-72, loaded through oracle/ref_shim.py) and with the restatement, and reports the pixels that
differ.  Last run (matplotlib 3.10.8): 0 of 200 single lines, 0 pixels of the three sets."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rs = np.random.RandomState(5)
    lines = list(synth.make_scene(4242, 40, 3, raster=None)["l"])
    for k in range(n):                       # ordinary, near-vertical, flat, far-off and steep curves
        scale = [(1, 1, 1), (1, 1e-3, 1), (1e-3, 1, 1e-3), (1, 1, 30), (10, 0.1, 0.1)][k % 5]
        lines.append(rs.randn(3) * np.array(scale))
    lines += [np.array(o, dtype=float) for o in ([1, 0.0, 0.3], [1, 1e-9, 0.3], [0.0, 1.0, 0.0], [0.3, -1e-6, -2.0], [1, 1, 1e6],
                                                 [5, 0.01, 0.01], [0, 0, 1.0], [0.0, 0.0, 0.0], [1, -1, 0], [0, 1, 1])]
    bad = 0
    for i, l in enumerate(lines):
        want = mpl_raster(sm, l[None])
        got = agg_raster.raster(l[None])
        d = np.abs(want.astype(int) - got.astype(int))
        if d.max() > 0:
            bad += 1
            print("line %d %s: %d pixels differ (max %d)" % (i, l, (d > 0).sum(), d.max()))
    print("single lines: %d of %d differ" % (bad, len(lines)))
    L = np.array(lines)
    import matplotlib as mpl
    for size, alpha in ((500, 0.1), (250, 0.1), (500, 0.5)):
        mpl.rcParams["lines.linewidth"] = 1.0
        t = time.time()
        want = sm.sphere_line_plot(L.copy(), size, alpha=alpha, f=1.0)
        t1 = time.time()
        got = agg_raster.raster(L, size=size, alpha=alpha)
        d = np.abs(want.astype(int) - got.astype(int))
        print("%d lines at once, size %d, alpha %.1f: %d pixels differ (max %d); matplotlib %.1f s, restatement %.1f s" % (
            len(L), size, alpha, (d > 0).sum(), d.max(), t1 - t, time.time() - t1))


if __name__ == "__main__":
    main()
