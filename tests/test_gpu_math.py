"""The premise of the parity bar, measured: how the device's double-precision exp / acos / asin / atan / sqrt / sin / cos /
log -- the ocml functions the EM and raster kernels call, compiled as they are in csrc/vpk_em.hip (vpk_math_probe) -- compare
with NumPy's on the arguments the path feeds them (probability_functions.py:99-176: exp(-lvsq / 2 sigma^2), acos / asin of
cosines, the prior's exp; vp_localisation.py:700-776: acos / cos of line angles; sphere_mapping.py:61: atan, sin, cos).

The reference's results can agree with any other implementation's to the last bit only where these functions do.  The test
prints, per function, the largest difference in ulp and the fraction of arguments whose results differ at all (the numbers
DESIGN.md section 4 quotes come from this test's output, gpurun_out/ulp_report.json), and asserts what the EM relies on:
sqrt is correctly rounded (bit-equal), everything else within 2 ulp."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 20


def _args():
    rs = np.random.RandomState(20241002)
    u = lambda lo, hi: rs.uniform(lo, hi, N)
    return {
        # exp(-lvsq / (2 s)): s in [1e-12, 1e-6] after the first iterations, lvsq in [0, 1] -> everything from 0 down to underflow
        "exp": np.concatenate([-10.0 ** u(-8, 2.88)[: N // 2], u(-40, 0)[: N // 2]]),
        # |cos| of angles between directions, clipped to [0, 1] (and the signed range for completeness)
        "acos": np.concatenate([u(0, 1)[: N // 2], 1 - 10.0 ** u(-16, 0)[: N // 4], u(-1, 1)[: N // 4]]),
        "asin": np.concatenate([u(-1, 1)[: N // 2], np.sin(u(-np.pi / 2, np.pi / 2))[: N // 2]]),
        # (-a sin - c cos) / b of the sphere curve: any magnitude, both signs
        "atan": np.concatenate([u(-4, 4)[: N // 2], (10.0 ** u(-6, 6) * np.sign(u(-1, 1)))[: N // 2]]),
        "sqrt": 10.0 ** u(-30, 4),
        # 9 x (angle difference) clipped to [-pi/2, pi/2]; alpha in [-pi/2, pi/2]; VP angles
        "sin": u(-np.pi, np.pi),
        "cos": u(-np.pi, np.pi),
        # log of weighted sums (the variance update exp(log a - log b), vp_localisation.py:301-304)
        "log": 10.0 ** u(-300, 3),
    }


def _ulp_diff(a, b):
    """|a - b| in units of the spacing of b (0 where both are equal, NaN or infinite alike)."""
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    with np.errstate(invalid="ignore", over="ignore"):
        d = np.abs(a - b) / np.spacing(np.abs(b))
    return np.where(same, 0.0, d)


def test_device_elementary_functions_against_numpy():
    from vanishing_points_2017_amd import kernels
    report = {}
    for name, x in _args().items():
        got = kernels.math_probe(name, x)
        with np.errstate(all="ignore"):
            want = getattr(np, {"acos": "arccos", "asin": "arcsin", "atan": "arctan"}.get(name, name))(x)
        d = _ulp_diff(got, want)
        report[name] = {"arguments": int(x.shape[0]), "max_ulp": float(np.nanmax(d)), "mismatch_rate": float((d > 0).mean()),
                        "range": [float(x.min()), float(x.max())]}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ulp_report.json"), "w") as fh:
        json.dump(report, fh, indent=1)
    print(json.dumps(report))
    assert report["sqrt"]["max_ulp"] == 0.0                     # IEEE: correctly rounded on both sides
    for name, r in report.items():
        assert r["max_ulp"] <= 2.0, (name, r)                   # last-bit differences only -- but they exist
