"""Horizon-error AUC -- port of the reference's auc.calc_auc (auc.py:5-37): sorted errors ->
cumulative fraction curve, value at the cutoff interpolated as at :20-28, trapezoid area / cutoff."""
import numpy as np

_trapezoid = getattr(np, "trapezoid", None) or np.trapz


def calc_auc(error_array, cutoff=0.25):
    err = np.sort(np.asarray(error_array).squeeze())
    n = err.shape[0]
    pts = np.zeros((n, 2))
    pts[:, 0] = err
    pts[:, 1] = (np.arange(n) + 1) * 1.0 / n
    mid = 1.
    for i in range(1, n):                         # the last crossing wins, as in the reference loop
        if err[i - 1] < cutoff < err[i]:
            mid = (err[i - 1] * pts[i - 1, 1] + err[i] * pts[i, 1]) / (err[i] + err[i - 1])
    tail = np.array([cutoff, 1]) if pts[-1, 0] < cutoff else np.array([cutoff, mid])
    pts = np.vstack([pts, tail])
    pts = pts[np.argsort(pts[:, 0]), :]
    sel = pts[:, 0] <= cutoff
    auc = _trapezoid(pts[sel, 1], pts[sel, 0])    # sklearn.metrics.auc == trapezoid rule (auc.py:33)
    return auc / cutoff, pts
