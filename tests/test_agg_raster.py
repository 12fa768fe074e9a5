"""oracle/agg_raster.py -- the CPU restatement of the reference's matplotlib / Agg rasteriser -- against the rasters the
reference itself produced (tests/golden/*.npz, sphere_mapping.sphere_line_plot under matplotlib in the build container;
oracle/check_agg_raster.py repeats the comparison against a live matplotlib there).  Integer work: bit-exact."""
import numpy as np
import pytest

from golden_util import load
from oracle import agg_raster


@pytest.mark.parametrize("name", ["tiny_n12", "clean3_n60", "noweights_n100"])
def test_restatement_reproduces_the_references_raster(name):
    g = load(name)
    got = agg_raster.raster(g["l"])
    assert np.array_equal(got, g["sphere_image"])


def test_blend_and_single_line():
    # white with alpha8 = 26 over black at full coverage: 65280 * 26 / 65306 -> 25; the reference's "one line peaks at 25"
    assert agg_raster.blend_white(0, 255) == 25
    assert agg_raster.blend_white(0, 34) == 2 and agg_raster.blend_white(0, 35) == 3      # a = round(26 cover / 255)
    img = agg_raster.raster(np.array([[0.3, 1.0, 0.2]]))
    assert img.max() == 25 and img[:, 0].max() == 0 and img[:, 1].max() < 25              # spines: column 0 black, 1 dimmed


def test_simplifier_keeps_the_curve_within_a_ninth_of_a_pixel():
    x, y = agg_raster.curve_pixels(np.array([0.4, 1.0, -0.7]))
    pts = np.array(agg_raster.simplify(x, y))
    assert 10 < len(pts) < 200 and pts[0, 0] == 0.0 and pts[-1, 0] == 500.0
    # every original sample lies within the threshold of the simplified polyline
    seg = np.searchsorted(pts[:, 0], x, side="right") - 1
    seg = np.clip(seg, 0, len(pts) - 2)
    p0, p1 = pts[seg], pts[seg + 1]
    d = p1 - p0
    t = np.clip(((x - p0[:, 0]) * d[:, 0] + (y - p0[:, 1]) * d[:, 1]) / np.maximum((d * d).sum(1), 1e-30), 0, 1)
    dist = np.hypot(x - (p0[:, 0] + t * d[:, 0]), y - (p0[:, 1] + t * d[:, 1]))
    assert dist.max() <= 1.0 / 9.0 + 1e-9


def test_alternative_curve_against_the_references_rasters():
    """sphere_line_plot(..., alternative=True) (sphere_mapping.py:58-59): the reference's own rasters of three line sets
    (tests/golden/rasteralt.npz, made by oracle/make_raster_alt_golden.py), pixel for pixel."""
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "rasteralt.npz"))
    for k in "ac":
        assert np.array_equal(agg_raster.raster(g["l_" + k], alternative=True), g["raster_" + k]), k
