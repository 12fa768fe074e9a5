"""Front end (evaluation.py:121-251 of the reference).

Pinned against the reference: the arithmetic AROUND the detector -- `detect_lsd_lines` (pixel -> normalised
coordinates) and the homogeneous lines of `create_data_dict_single` -- by running the reference's own functions on
known detector output (oracle/make_frontend_golden.py -> tests/golden/frontend.npz).
Not pinnable: the detector (the reference's is an absent submodule).  For it the tests check the contract the LSD
paper states: straight edges are found with sub-pixel accuracy, white noise gives (almost) no detection."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

G = dict(np.load(os.path.join(GOLDEN, "frontend.npz")))


@pytest.mark.parametrize("case", [0, 1, 2])
def test_normalisation_equals_the_reference_functions(case):
    from vanishing_points_2017_amd import frontend
    h, w = G["shape%d" % case]
    raw = G["raw%d" % case]
    res = frontend.detect_lsd_lines(np.full((h, w), 0.5), detector=lambda image: raw.copy())
    assert np.array_equal(res["segments"], G["segments%d" % case])        # same operations, same order: same bits
    assert np.array_equal(res["nfa"], G["nfa%d" % case])
    rgb = G["rgb%d" % case]
    single = frontend.create_data_dict_single(rgb, 500, detector=lambda image: G["raw_small%d" % case].copy(),
                                              sphere_fn=lambda lines, size, alpha: None)
    assert np.array_equal(single["lines"]["line_segments"], G["single_segments%d" % case])
    assert np.array_equal(single["lines"]["lines"], G["single_lines%d" % case])
    assert tuple(single["lines"]["image_shape"]) == tuple(G["single_shape%d" % case])


def test_rgb2gray_weights():
    from vanishing_points_2017_amd import frontend
    rgb = np.zeros((2, 2, 3), np.uint8)
    rgb[0, 0] = (255, 0, 0); rgb[0, 1] = (0, 255, 0); rgb[1, 0] = (0, 0, 255); rgb[1, 1] = (255, 255, 255)
    g = frontend.rgb2gray(rgb)
    assert np.allclose(g, [[0.2125, 0.7154], [0.0721, 1.0]], atol=1e-15)


def test_resize_to_fit_keeps_the_aspect_ratio():
    from vanishing_points_2017_amd import frontend
    img = np.zeros((1333, 2000, 3), np.uint8)
    assert frontend.resize_to_fit(img, 640).shape == (427, 640, 3)         # example.py: target_size = 640
    assert frontend.resize_to_fit(np.zeros((2000, 1500, 3), np.uint8), 800).shape == (800, 600, 3)


def _render(segments, h, w, width=2.5):
    """Dark anti-aliased strokes on a light background (distance to the segment, clamped)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 220.0)
    for x1, y1, x2, y2 in segments:
        dx, dy = x2 - x1, y2 - y1
        t = np.clip(((xx - x1) * dx + (yy - y1) * dy) / (dx * dx + dy * dy), 0, 1)
        d = np.hypot(xx - (x1 + t * dx), yy - (y1 + t * dy))
        img = np.minimum(img, 220.0 - 180.0 * np.clip(width / 2 + 0.5 - d, 0, 1))
    return img


def _seg_dist(a, b):
    """max distance of the end points of b to the infinite line through a, and the angle between them (deg)."""
    ax, ay = a[2] - a[0], a[3] - a[1]
    n = np.hypot(ax, ay)
    d = max(abs((b[0] - a[0]) * ay - (b[1] - a[1]) * ax), abs((b[2] - a[0]) * ay - (b[3] - a[1]) * ax)) / n
    ang = np.degrees(np.arccos(min(1.0, abs(ax * (b[2] - b[0]) + ay * (b[3] - b[1])) / (n * np.hypot(b[2] - b[0], b[3] - b[1])))))
    return d, ang


def test_detector_finds_rendered_segments_with_subpixel_accuracy():
    from vanishing_points_2017_amd import lsd
    rs = np.random.RandomState(4)
    true = [(40, 50, 300, 70), (60, 200, 280, 120), (150, 20, 170, 230), (20, 230, 120, 140), (200, 30, 310, 220)]
    img = _render(true, 256, 336) + rs.normal(0, 1.5, (256, 336))
    det = lsd.detect_line_segments(img)
    assert det.shape[1] == 7 and det.shape[0] >= 2 * len(true) - 2             # each stroke has two edges
    assert (det[:, 6] > 0).all() and np.allclose(det[:, 5], 0.125)            # -log10(NFA) > 0; p = 22.5 / 180
    for t in true:
        best = [(_seg_dist(t, d), d) for d in det]
        close = [b for b in best if b[0][0] <= 2.2 and b[0][1] <= 1.5]          # an edge lies ~1.25 px off the stroke's axis
        assert close, ("no detection along", t)
        length = sum(np.hypot(d[2] - d[0], d[3] - d[1]) for _, d in close)
        assert length >= 1.2 * np.hypot(t[2] - t[0], t[3] - t[1])               # both edges, most of their length
    # every detection lies along one of the strokes: nothing is invented
    for d in det:
        assert min(_seg_dist(t, d)[0] for t in true) <= 3.0


def test_detector_is_quiet_on_white_noise():
    from vanishing_points_2017_amd import lsd
    rs = np.random.RandomState(0)
    total = 0
    for k in range(3):
        total += lsd.detect_line_segments(rs.uniform(0, 255, (200, 200))).shape[0]
    assert total <= 1          # the a-contrario bound: about one false alarm per image at most (eps = 1)


def test_front_end_feeds_the_reference_schema():
    """image -> grey -> segments -> homogeneous lines, y up, long side = [-1, 1] (no raster: that needs the GPU)."""
    from vanishing_points_2017_amd import frontend
    img = _render([(40, 50, 300, 70), (150, 20, 170, 230)], 256, 336)
    rgb = np.repeat(img[:, :, None], 3, 2).astype(np.uint8)
    d = frontend.create_data_dict_single(rgb, 500, sphere_fn=lambda lines, size, alpha: "raster")
    seg, lines = d["lines"]["line_segments"], d["lines"]["lines"]
    assert d["sphere_image"] == "raster" and seg.shape[1] == 4 and lines.shape == (seg.shape[0], 3)
    assert np.abs(seg[:, [0, 2]]).max() <= 1.0 and np.abs(seg[:, [1, 3]]).max() <= 256.0 / 336.0 + 1e-9
    p1 = np.c_[seg[:, 0:2], np.ones(len(seg))]
    assert np.abs((lines * p1).sum(1)).max() <= 1e-12                       # every line passes through its first end point
