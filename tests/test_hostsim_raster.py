"""The rasteriser's PRODUCT arithmetic (csrc/raster_device.hpp: simplifier, stroker, cell walker with its clip box and its
closed-form restart of AGG's row DDA, calculate_alpha, blender) compiled for the host by tests/hostsim/sim_raster.cpp and run
serially -- the CPU suite's check of the code the GPU kernels are made of (the kernels' own orchestration: LDS pools, row
bands, atomics, the blend order across workgroups, is what tests/test_gpu_raster.py covers).  Bit-exact against the
reference's own rasters (tests/golden) and the restatement oracle/agg_raster.py."""
import os
import sys

import numpy as np
import pytest

from golden_util import load
from oracle import agg_raster

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))
import simlib  # noqa: E402


@pytest.mark.parametrize("name", ["tiny_n12", "clean3_n60", "noweights_n100", "yud_n330", "stress_n300"])
def test_product_arithmetic_reproduces_the_references_raster(name):
    g = load(name)
    got, flags = simlib.sim_raster(g["l"], 500, 0.1)
    assert flags == 0
    diff = got.astype(int) - g["sphere_image"].astype(int)
    assert not diff.any(), "%d pixels differ" % (diff != 0).sum()


def test_odd_lines_sizes_and_alpha_against_the_restatement():
    odd = np.array([[1, 0.0, 0.3], [1, 1e-9, 0.3], [0.0, 1.0, 0.0], [0.0, 0.0, 0.0], [0.3, -1e-6, -2.0], [1, 1, 1e6],
                    [5, 0.01, 0.01], [0, 0, 1.0]])
    got, flags = simlib.sim_raster(odd, 500, 0.1)
    assert flags == 0 and np.array_equal(got, agg_raster.raster(odd))
    rng = np.random.default_rng(5)
    l = rng.normal(size=(25, 3))
    got, _ = simlib.sim_raster(l, 250, 0.5)
    assert np.array_equal(got, agg_raster.raster(l, size=250, alpha=0.5))
    got, _ = simlib.sim_raster(l[:0], 250, 0.1)                      # no lines: the frame alone
    assert np.array_equal(got, agg_raster.raster(l[:0], size=250))


def test_an_edge_walked_in_shares_gives_the_cells_of_the_edge_walked_whole():
    """coverage_kernel gives the rows of a long edge to up to 32 threads; each restarts AGG's incremental x DDA in closed
    form at its first row.  Same accumulators as the single walk, for edges inside, across and outside the clip box."""
    rng = np.random.default_rng(11)
    size = 64
    n = 0
    for _ in range(400):
        x1, y1, x2, y2 = rng.uniform(-20, size + 20, 4)
        if rng.random() < 0.2:
            x2 = x1 + rng.uniform(-0.3, 0.3)                          # nearly vertical
        if rng.random() < 0.2:
            y2 = y1 + rng.uniform(-0.3, 0.3)                          # within one row
        for k in (2, 5, 32):
            assert simlib.sim_edge_shares(x1, y1, x2, y2, size, k) == 0, (x1, y1, x2, y2, k)
            n += 1
    assert n == 1200


def test_product_arithmetic_on_the_alternative_curve():
    """The `alternative` parametrisation (sphere_mapping.py:58-59) through the product's simplifier / stroker / cells: the
    reference's own rasters (tests/golden/rasteralt.npz), pixel for pixel."""
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "rasteralt.npz"))
    for k in "abc":
        got, flags = simlib.sim_raster(g["l_" + k], 500, 0.1, alternative=True)
        assert flags == 0 and np.array_equal(got, g["raster_" + k]), k
