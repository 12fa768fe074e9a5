"""CPU oracle: NumPy float64 restatement of the reference's EM vanishing-point refinement.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker -- the product path (``vanishing_points_2017_amd``) never does and fails
loudly when its HIP library is missing.

Parity pin: every function below is checked against golden vectors produced by
running the reference's own modules (shimmed in memory, ``oracle/ref_shim.py``)
in the build container -- see ``oracle/make_golden.py`` and
``tests/test_oracle_golden.py``.  The reference ships no tests or known-answer
vectors of its own.

This is a *vectorised* restatement: same arithmetic per element (so intermediates
agree with the reference to ~1e-15), array-at-a-time instead of the reference's
Python double loops, and no joblib process pools.  It is therefore much faster
than the reference and must be labelled "port" wherever it is timed.

Third-party arithmetic on the path is called exactly where the reference calls it:
``numpy.linalg.svd`` (vp_localisation.py:466,595), ``numpy.argsort``
(vp_localisation.py:47,57,123,546; probability_functions.py:84) and
``sklearn.cluster.AgglomerativeClustering`` (vp_localisation.py:574-576).

All citations are file:line under /root/reference.
"""
from collections import namedtuple

import numpy as np

PI = np.pi


# ----------------------------------------------------------------------------------------------
# fused dot products: the reference's scalar code calls np.dot / np.linalg.norm on 2- and 3-vectors,
# which NumPy's BLAS evaluates as fma(x_{n-1}, y_{n-1}, ... fma(x1, y1, x0*y0)).  NumPy has no fma
# ufunc, so a 10-line C helper (oracle/_fma.c, built on first use with gcc) provides it.
# ----------------------------------------------------------------------------------------------
_FMA = None


def _fma_lib():
    global _FMA
    if _FMA is None:
        import ctypes
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "_build", "libvpkfma.so")
        src = os.path.join(here, "_fma.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            os.makedirs(os.path.dirname(so), exist_ok=True)
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-builtin", src, "-o", so, "-lm"])
        _FMA = ctypes.CDLL(so)
        _FMA.vpk_vfma.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t]
        _FMA.vpk_vpow2.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_size_t]
    return _FMA


def pow2_scalar(a):
    """x ** 2 as a NumPy float64 scalar computes it: libm pow(x, 2.0)."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty(a.shape)
    _fma_lib().vpk_vpow2(a.ctypes.data, out.ctypes.data, out.size)
    return out


def fma(a, b, c):
    """Element-wise fused multiply-add round(a*b + c) with broadcasting."""
    a, b, c = np.broadcast_arrays(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64),
                                  np.asarray(c, dtype=np.float64))
    a, b, c = np.ascontiguousarray(a), np.ascontiguousarray(b), np.ascontiguousarray(c)
    out = np.empty(a.shape)
    _fma_lib().vpk_vfma(a.ctypes.data, b.ctypes.data, c.ctypes.data, out.ctypes.data, out.size)
    return out


def dot2(ax, ay, bx, by):
    """np.dot of 2-vectors as the reference's BLAS rounds it."""
    return fma(ay, by, np.asarray(ax) * np.asarray(bx))


def dot3(ax, ay, az, bx, by, bz):
    return fma(az, bz, fma(ay, by, np.asarray(ax) * np.asarray(bx)))


def norm2(x, y):
    return np.sqrt(dot2(x, y, x, y))

PDFParams = namedtuple("PDFParams", "means weights sigma")      # probability_functions.py:4
PDF = namedtuple("PDF", "v lv vl l lvsq angles")                # probability_functions.py:5


# ----------------------------------------------------------------------------------------------
# coordinate_conversion.py
# ----------------------------------------------------------------------------------------------
def index_to_angle(index, shape):
    """coordinate_conversion.py:4-20."""
    a, b = index[0], index[1]
    m, n = shape[0], shape[1]
    return np.array([(a - 0.5 * m + 0.5) * PI / m, (b - 0.5 * n + 0.5) * PI / n])


def angle_to_index(angle, shape):
    """coordinate_conversion.py:23-35."""
    m, n = shape[0], shape[1]
    return np.array([(angle[0] / PI + 0.5 - 0.5 / m) * m, (angle[1] / PI + 0.5 - 0.5 / n) * n])


def angle_to_point(angle):
    """coordinate_conversion.py:38-50 (sign(0) = 0 zeroes the point, as in the reference)."""
    alpha, beta = angle[0], angle[1]
    p = np.array([np.sin(alpha) * np.cos(beta), np.sin(beta), np.cos(alpha) * np.cos(beta)])
    return p * np.sign(p[2])


def point_to_angle(point):
    """coordinate_conversion.py:53-61."""
    beta = np.arcsin(point[1])
    inner = np.maximum(np.minimum(point[0] / np.cos(beta), 1), -1)
    return np.array([np.arcsin(inner), beta])


# ----------------------------------------------------------------------------------------------
# initial VPs from the CNN grid + sphere raster
# ----------------------------------------------------------------------------------------------
def find_maxima(r):
    """vp_localisation.py:13-31.  Strict 4-neighbour maxima; a neighbour whose index is <= 0
    (not < 0) or out of range counts as 0 -- row/column 0 never act as neighbours (:24-25)."""
    r = np.asarray(r)
    zero = np.zeros_like(r)
    vu = zero.copy(); vu[:, :-1] = r[:, 1:]          # a+1 < A
    vd = zero.copy(); vd[:, 2:] = r[:, 1:-1]         # a-1 > 0
    vl = zero.copy(); vl[2:, :] = r[1:-1, :]         # b-1 > 0
    vr = zero.copy(); vr[:-1, :] = r[1:, :]          # b+1 < B
    return ((r > vu) & (r > vd) & (r > vl) & (r > vr)).astype(np.float64)


def find_initial_vps(sphere_image, cnn_response, num_max):
    """vp_localisation.py:111-165.  Raises ValueError (np.vstack of []) when no cell survives."""
    sphere = sphere_image[::-1, :]
    ra_n, rb_n = cnn_response.shape
    sa, sb = sphere_image.shape
    maxima = find_maxima(cnn_response).flatten()
    flat = cnn_response.flatten()
    best = np.argsort(flat[maxima == 1])[::-1]
    maxima[np.where(maxima == 1)[0][best[num_max:]]] = 0
    maxima = maxima.reshape(cnn_response.shape)
    vps = []
    for ra in range(ra_n):
        for rb in range(rb_n):
            if maxima[ra, rb] != 1:
                continue
            blk = sphere[ra * sa // ra_n:(ra + 1) * sa // ra_n, rb * sb // rb_n:(rb + 1) * sb // rb_n]
            mx = blk.max()
            if mx == 0:                                   # :137-142 (nothing > 0 survives)
                continue
            rows, cols = np.nonzero(blk >= mx)            # row-major order, as np.where on the flat slice
            avg = np.zeros(2)
            for r_, c_ in zip(rows, cols):                # sequential accumulation (:148-151)
                avg += (r_, c_)
            avg /= len(rows)
            idx = np.array([avg[1] + rb * sb // rb_n, avg[0] + ra * sa // ra_n])   # (col, row) :155-158
            vps.append(angle_to_point(index_to_angle(idx, sphere_image.shape)))
    return np.vstack(vps)


def pdf_params(cnn_response, confidence=1.282):
    """probability_functions.py:62-96.  The weight normalisation runs in float32 because
    ``cnn_response`` is float32 (flatten() copies, the in-place divisions keep the dtype)."""
    a_n, b_n = cnn_response.shape
    sigma = PI / (confidence * a_n)
    alphas = np.tile(np.linspace(-(a_n - 1.0) / a_n * PI / 2, (a_n - 1.0) / a_n * PI / 2, a_n), (b_n, 1))
    betas = np.tile(np.linspace(-(b_n - 1.0) / b_n * PI / 2, (b_n - 1.0) / b_n * PI / 2, b_n), (a_n, 1)).T
    weights = cnn_response.flatten()
    order = np.argsort(weights)[::-1]
    weights[order[100:]] = 0
    weights /= np.sum(weights)
    weights /= (2 * PI * sigma * sigma)
    means = np.zeros((a_n * b_n, 2))
    means[:, 0] = alphas.flatten()
    means[:, 1] = betas.flatten()
    return PDFParams(means=means, weights=weights, sigma=sigma)


# ----------------------------------------------------------------------------------------------
# E-step
# ----------------------------------------------------------------------------------------------
def calc_angles(v):
    """probability_functions.py:252-259.  v: (M,3) -> (M,2) = (alpha, beta)."""
    ang = np.zeros((v.shape[0], 2))
    ang[:, 1] = np.arcsin(v[:, 1])
    inner = v[:, 0] / np.cos(ang[:, 1])
    inner = np.maximum(np.minimum(inner, 1), -1)
    ang[:, 0] = np.arcsin(inner)
    return ang


def calc_pdf(pdfpar, x, y):
    """probability_functions.py:8-40.  Mixture prior with four wrap images; the fifth term
    duplicates the fourth (:25-26).  Components are accumulated in index order like the
    reference; the five exponentials are summed left to right (np.sum of 5 values)."""
    means, weights, sigma = pdfpar
    resp = np.zeros(x.shape[0])
    k = -0.5 / (sigma * sigma)
    for n in np.nonzero(weights > 0)[0]:
        ma, mb = means[n, 0], means[n, 1]
        d1 = (x - ma) * (x - ma) + (y - mb) * (y - mb)
        d2 = (x - ma + PI) * (x - ma + PI) + (y + mb) * (y + mb)
        d3 = (x - ma - PI) * (x - ma - PI) + (y + mb) * (y + mb)
        d4 = (x + ma) * (x + ma) + (y - mb - PI) * (y - mb - PI)
        p = np.exp(d1 * k)
        p = p + np.exp(d2 * k)
        p = p + np.exp(d3 * k)
        e4 = np.exp(d4 * k)
        p = p + e4
        p = p + e4
        resp += p * weights[n]
    return resp


def calc_lvsq_angle(v, lp):
    """probability_functions.py:157-176.  v: (M,3) VPs; returns (N,M) of (1-|cos|)^2 between
    (segment midpoint - projected VP) and the segment direction."""
    vx = v[:, 0] / v[:, 2]
    vy = v[:, 1] / v[:, 2]
    lmx = 0.5 * (lp[:, 0] + lp[:, 2])
    lmy = 0.5 * (lp[:, 1] + lp[:, 3])
    v2x = lp[:, 0] - lp[:, 2]
    v2y = lp[:, 1] - lp[:, 3]
    v1x = lmx[:, None] - vx[None, :]
    v1y = lmy[:, None] - vy[None, :]
    dot = dot2(v1x, v1y, v2x[:, None], v2y[:, None])
    n1 = norm2(v1x, v1y)
    n2 = norm2(v2x, v2y)
    c = 1 - np.abs(dot / (n1 * n2[:, None]))
    return pow2_scalar(c)                                    # (...)**2 on a float64 scalar (:174)


def calc_probabilities(pdfpar, v, lp, s):
    """probability_functions.py:99-147 ("angle" branch).  ``v`` is the (M,3) slice the
    reference indexes as v[i].  Floors ``s`` at 1e-200 IN PLACE (:139)."""
    m_n = v.shape[0]
    angles = calc_angles(v)
    p_v = calc_pdf(pdfpar, angles[:, 0], angles[:, 1])
    lvsq = calc_lvsq_angle(v, lp)
    np.maximum(s, 1e-200, out=s)                              # :139  (s[m] > 1e-200 else 1e-200)
    p_lv = np.exp(-(lvsq / (2 * s)[None, :])) * (1.0 / np.sqrt(2 * PI * s))[None, :]
    if m_n:
        p_l = np.dot(p_lv, p_v)
    else:
        p_l = np.zeros(lp.shape[0])
    p_l = np.maximum(p_l, 1e-12)
    p_vl = (p_lv.T * p_v[:, None]) / p_l[None, :]
    return PDF(v=p_v, lv=p_lv, vl=p_vl, l=p_l, lvsq=lvsq, angles=angles)


# ----------------------------------------------------------------------------------------------
# segment geometry (vp_localisation.py:700-776), all-pairs form
# ----------------------------------------------------------------------------------------------
def _seg_point_dist(ax, ay, bx, by, px, py):
    """vp_localisation.py:743-758: distance from point p to segment a-b (clamped projection).
    The reference squares the *norm* of (b-a) (:747), reproduced as sqrt-then-square."""
    dx, dy = bx - ax, by - ay
    nrm = norm2(dx, dy)
    with np.errstate(divide="ignore", invalid="ignore"):
        param = dot2(px - ax, py - ay, dx, dy) / np.square(nrm)
    cx = np.where(param < 0, ax, np.where(param > 1, bx, ax + param * dx))
    cy = np.where(param < 0, ay, np.where(param > 1, by, ay + param * dy))
    ex, ey = cx - px, cy - py
    return norm2(ex, ey)


def pair_distance_closest(lp):
    """vp_localisation.py:727-740 for every ordered pair (i,j): min over the four
    end-point-to-segment distances.  Returns (N,N); the diagonal is not special-cased here."""
    x1, y1, x2, y2 = (lp[:, k][:, None] for k in range(4))        # line i  (rows)
    u1, w1, u2, w2 = (lp[:, k][None, :] for k in range(4))        # line j  (cols)
    d1 = _seg_point_dist(x1, y1, x2, y2, u1, w1)
    d2 = _seg_point_dist(x1, y1, x2, y2, u2, w2)
    d4 = _seg_point_dist(u1, w1, u2, w2, x1, y1)
    d5 = _seg_point_dist(u1, w1, u2, w2, x2, y2)
    return np.minimum(np.minimum(d1, d2), np.minimum(d4, d5))


def pair_cosangle(lp, f, rows=None, cols=None):
    """vp_localisation.py:715-724 for pairs: cos(clip(f * acos(|cos angle|), -pi/2, pi/2))."""
    vx = lp[:, 0] - lp[:, 2]
    vy = lp[:, 1] - lp[:, 3]
    nrm = norm2(vx, vy)
    if rows is None:
        ax, ay, an = vx[:, None], vy[:, None], nrm[:, None]
        bx, by, bn = vx[None, :], vy[None, :], nrm[None, :]
    else:
        ax, ay, an = vx[rows], vy[rows], nrm[rows]
        bx, by, bn = vx[cols], vy[cols], nrm[cols]
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.abs(dot2(ax, ay, bx, by) / (an * bn))
    dphi = np.abs(np.arccos(np.clip(c, -1, 1)))
    return np.cos(np.clip(f * dphi, -PI / 2, PI / 2))


def line_lengths(lp):
    """vp_localisation.py:761-762."""
    dx = lp[:, 0] - lp[:, 2]
    dy = lp[:, 1] - lp[:, 3]
    return norm2(dx, dy)


def pair_proximity(lp, dist, sigma):
    """vp_localisation.py:708-712 on a precomputed closest-distance matrix."""
    ln = line_lengths(lp)
    sg = sigma * np.minimum(ln[:, None], ln[None, :])
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.exp(-(dist * dist) / (2 * sg * sg))


def calc_lsim(lp, sigma=0.1, dist=None):
    """vp_localisation.py:87-108.  Symmetric N x N similarity, zero diagonal.  The reference
    evaluates the lower triangle as lines_similarity(lp[i], lp[j]) (j < i) and mirrors it;
    the all-pairs form here is symmetric by construction (every term is symmetric in fp)."""
    if dist is None:
        dist = pair_distance_closest(lp)
    sim = pair_cosangle(lp, 9) * pair_proximity(lp, dist, sigma)
    low = np.tril(sim, -1)
    return low + low.T


def line_rating_knn(lp, k1=10, k2=3, sigma=1, dist=None):
    """vp_localisation.py:34-84.  kNN line score: among the k1 nearest segments (closest
    distance, diagonal = 4) take the k2 best aligned (cos with f = 9) and average
    proximity * cos."""
    n = lp.shape[0]
    k1 = min(k1, n)
    k2 = min(k2, n)
    if dist is None:
        dist = pair_distance_closest(lp)
    ldist = dist.copy()
    np.fill_diagonal(ldist, 4)                                       # :82
    nn = np.argsort(ldist, axis=1)[:, 0:k1]                          # :47-48
    rows = np.repeat(np.arange(n), k1).reshape(n, k1)
    cosphi = pair_cosangle(lp, 9, rows, nn)                          # :55
    best = np.argsort(cosphi, axis=1)[:, ::-1][:, 0:k2]              # :57-59
    lj = np.take_along_axis(nn, best, 1)
    cb = np.take_along_axis(cosphi, best, 1)
    ln = line_lengths(lp)
    sg = sigma * np.minimum(ln[:, None], ln[lj])
    d = dist[rows[:, :k2], lj]
    prox = np.exp(-(d * d) / (2 * sg * sg))
    terms = prox * cb
    score = np.zeros(n)
    for k in range(k2):                                              # np.sum of k2 values, in order
        score = score + terms[:, k]
    return score / k2


def lines_angles(lp):
    """vp_localisation.py:765-776: undirected segment angle folded into [0, pi/2]."""
    vx = lp[:, 0] - lp[:, 2]
    vy = lp[:, 1] - lp[:, 3]
    nrm = norm2(vx, vy)
    phi = np.abs(np.arccos(np.clip(vx / nrm, -1, 1)))
    return np.where(phi > PI / 2, PI - phi, phi)


# ----------------------------------------------------------------------------------------------
# smoothing + M-step
# ----------------------------------------------------------------------------------------------
class Smoother(object):
    """vp_localisation.py:515-524 with the per-column denominator hoisted."""

    def __init__(self, lsim, lweight, bias=1):
        self.lsim = lsim
        self.lweight = lweight
        self.bias = bias
        self.den = 1 + bias * lweight * np.sum(lsim, axis=1)        # lsim symmetric: col sum == row sum

    def __call__(self, p_vl):
        w_ = p_vl * self.lweight[None, :]
        return (w_ + self.bias * self.lweight[None, :] * np.dot(w_, self.lsim)) / self.den[None, :]


def calc_new_vanishing_point(l, w):
    """vp_localisation.py:453-479: smallest right singular vector of diag(w/max w) * l."""
    if np.size(w) == 0:
        return None
    wm = np.max(w)
    if wm == 0:
        return None
    try:
        mat = (w / wm)[:, None] * l
        _, _, vt = np.linalg.svd(mat, full_matrices=mat.shape[0] < 3)
        vp = vt[2, :].copy()
        vp /= np.linalg.norm(vp, ord=2)
        vp *= np.sign(vp[2])
    except np.linalg.LinAlgError:
        vp = None
    return vp


def calc_lvsq_single(v, lp):
    """probability_functions.py:212-224 for the rows of lp against their own VP rows v (N,3)."""
    vx = v[:, 0] / v[:, 2]
    vy = v[:, 1] / v[:, 2]
    v1x = 0.5 * (lp[:, 0] + lp[:, 2]) - vx
    v1y = 0.5 * (lp[:, 1] + lp[:, 3]) - vy
    v2x = lp[:, 0] - lp[:, 2]
    v2y = lp[:, 1] - lp[:, 3]
    c = 1 - np.abs(dot2(v1x, v1y, v2x, v2y) / (norm2(v1x, v1y) * norm2(v2x, v2y)))
    return pow2_scalar(c)                                    # :222


def calc_vp_line_counts(vp, lp, s, metric, lweights, thresh):
    """vp_localisation.py:482-512 ("angle" branch)."""
    m_n = vp.shape[0]
    assoc = np.argmax(metric, axis=0)
    dist = calc_lvsq_single(vp[assoc], lp)
    out = (dist > thresh * np.sqrt(s[assoc])) | (lweights == 0)
    counts = np.zeros(m_n)
    counts_w = np.zeros(m_n)
    for n in np.nonzero(~out)[0]:                       # sequential accumulation order (:509-510)
        counts[assoc[n]] += 1
        counts_w[assoc[n]] += lweights[n]
    assoc = assoc.copy()
    assoc[out] = -1
    return counts, counts_w, assoc


def _variance(lvsq_col, pvl_row):
    """vp_localisation.py:301-304: exp(log sum(lvsq * p_vl) - log sum(p_vl))."""
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.exp(np.log(np.sum(lvsq_col * pvl_row)) - np.log(np.sum(pvl_row)))


# ----------------------------------------------------------------------------------------------
# the EM driver
# ----------------------------------------------------------------------------------------------
class _EM(object):
    """State of one run: ``cur`` = v[i], ``nxt`` = v[i+1] of the reference's (iter, M, 3)
    history array (only these two slices are ever read), ``s`` = per-VP variance."""

    def __init__(self, l, lp, cnn_response, sphere_image, init_vp, use_weights, wbias, num_init_vp,
                 outlier_thresh, trace):
        self.trace = trace
        self.lp = lp
        n = l.shape[0]
        self.n = n
        dist = pair_distance_closest(lp) if use_weights else None
        lsim = calc_lsim(lp, sigma=1, dist=dist) if use_weights else np.zeros((n, n))  # :177-180
        l /= np.sqrt(dot3(l[:, 0], l[:, 1], l[:, 2], l[:, 0], l[:, 1], l[:, 2]))[:, None]  # :185-186 (in place)
        v0 = find_initial_vps(sphere_image, cnn_response, num_init_vp)                 # :208
        self.pdfpar = pdf_params(cnn_response)                                         # :210
        if init_vp is not None:                                                        # :212-215
            v0 = init_vp.copy()
            v0 /= np.sqrt(dot3(v0[:, 0], v0[:, 1], v0[:, 2], v0[:, 0], v0[:, 1], v0[:, 2]))[:, None]
        self.langles = lines_angles(lp)                                                # :217
        l /= np.sqrt(dot3(l[:, 0], l[:, 1], l[:, 2], l[:, 0], l[:, 1], l[:, 2]))[:, None]  # :226 (again)
        self.l = l
        llen = line_lengths(lp)                                                        # :227
        if use_weights:
            lscore = np.clip(line_rating_knn(lp, k2=4, dist=dist), 0.2, 1)             # :230-231
            self.lweight = llen * lscore                                               # :232-233
        else:
            self.lweight = np.ones(n)
        self.smooth = Smoother(lsim, self.lweight, wbias)
        self.lsim = lsim
        self.cur = v0.copy()
        self.nxt = np.zeros_like(v0)
        self.thresh = outlier_thresh
        self.v0 = v0
        if trace is not None:
            trace.update(lsim=lsim, lweight=self.lweight.copy(), v0=v0.copy(),
                         pdf_weights=self.pdfpar.weights.copy(), langles=self.langles.copy())

    # helpers ---------------------------------------------------------------------------------
    def estep(self, v):
        return calc_probabilities(self.pdfpar, v, self.lp, self.s)

    def delete(self, idx):
        idx = np.asarray(idx, dtype=int)
        self.cur = np.delete(self.cur, idx, axis=0)
        self.nxt = np.delete(self.nxt, idx, axis=0)
        self.s = np.delete(self.s, idx, axis=0)

    def counts(self, v, metric):
        return calc_vp_line_counts(v, self.lp, self.s, metric, self.lweight, self.thresh)

    # vp_localisation.py:633-697 ---------------------------------------------------------------
    def merge(self, use_next, thresh, max_stdd=0.01):
        again = True
        while again and self.cur.shape[0] > 1:
            x = self.nxt if use_next else self.cur
            cosphi = np.clip(np.dot(x, x.T), -1, 1)
            ang = np.abs(np.arccos(np.clip(np.abs(cosphi), -1, 1)))
            np.fill_diagonal(ang, PI)
            j, k = np.unravel_index(ang.argmin(), ang.shape)
            if ang[j, k] < thresh:
                p = self.estep(x)
                w = self.smooth(p.vl)
                new = calc_new_vanishing_point(self.l, w[j, :] + w[k, :])
                pv = p.vl[k, :] + p.vl[j, :]
                with np.errstate(divide="ignore", invalid="ignore"):
                    self.s[k] = np.exp(np.log(np.sum(0.5 * (p.lvsq[:, j] + p.lvsq[:, k]) * pv))
                                       - np.log(np.sum(pv)))          # :663-666, written before the abort test
                if new is None or self.s[k] > max_stdd:
                    if self.trace is not None:
                        self.trace["merge_aborts"] = self.trace.get("merge_aborts", 0) + 1
                    again = False
                    continue
                x[k, :] = new
                self.delete([j])
                if self.trace is not None:
                    key = "final_merges" if thresh > 5e-3 else "periodic_merges"
                    self.trace[key] = self.trace.get(key, 0) + 1
            else:
                again = False

    # vp_localisation.py:527-630 ---------------------------------------------------------------
    def split(self, w, min_diff, num_clusters=2):
        import sklearn.cluster as cluster
        m_n, n = w.shape
        idx = w.argmax(axis=0)
        greedy = np.zeros(w.shape)
        greedy[idx, np.arange(n)] = w[idx, np.arange(n)]
        greedy /= w.max()
        stdd_phi = np.zeros(m_n)
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for m in range(m_n):
                    stdd_phi[m] = np.std(self.langles[greedy[m, :] > 0])   # NaN for empty VPs
        worst_order = np.argsort(stdd_phi)[::-1]
        worst = None
        for m in range(m_n):
            assoc_lines = np.where(idx == worst_order[m])[0]
            n_worst = assoc_lines.shape[0]
            vp = self.cur[m, :] / self.cur[m, 2]          # :557 tests VP m, not worst_order[m] (kept quirk)
            if n_worst > num_clusters * 4 and (vp[0] > -1 and vp[1] > -1 and vp[0] < 1 and vp[1] < 1):
                worst = worst_order[m]
                break
        if worst is None:
            return
        stdd = self.s[worst] / num_clusters
        lp = self.lp[assoc_lines]
        rows = np.repeat(np.arange(n_worst), n_worst).reshape(n_worst, n_worst)
        ldist = 1 - pair_cosangle(lp, 2, rows, rows.T)
        np.fill_diagonal(ldist, 0)
        model = cluster.AgglomerativeClustering(linkage="average", connectivity=ldist,
                                                n_clusters=num_clusters, metric="precomputed")
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model.fit_predict(ldist)
        labels = model.labels_
        lw = self.l[assoc_lines] * self.lweight[assoc_lines][:, None]
        new_vps = []
        for c in range(num_clusters):
            line_set = lw[labels == c]
            if line_set.shape[0] < 3:
                continue
            _, _, vt = np.linalg.svd(line_set, full_matrices=False)
            vp = vt[2, :].copy()
            vp /= np.linalg.norm(vp, ord=2)
            if vp[2] < 0:
                vp *= -1
            new_vps.append(vp)
        too_similar = True
        for c in range(len(new_vps)):
            for d in range(c + 1, len(new_vps)):
                cosphi = np.clip(np.dot(new_vps[c], new_vps[d]), -1, 1)
                if np.abs(np.arccos(np.clip(np.abs(cosphi), -1, 1))) > min_diff:
                    too_similar = False
        if too_similar:
            return
        for c, vp in enumerate(new_vps):
            if c == 0:
                self.cur[worst, :] = vp
                self.s[worst] = stdd
            else:
                self.cur = np.vstack([self.cur, vp[None, :]])
                self.nxt = np.vstack([self.nxt, np.zeros((1, 3))])
                self.s = np.append(self.s, stdd)
        if self.trace is not None:
            self.trace.setdefault("splits", []).append((int(worst), labels.copy(), assoc_lines.copy(), ldist.copy()))


def expectation_maximisation(l, lp, cnn_response, num_iter=100, sphere_image=None, init_vp=None,
                             do_merge=True, do_split=True, do_iterations=True, distance_measure="angle",
                             use_weights=True, wbias=1, num_init_vp=25, split_merge_freq=10,
                             merge_thresh=1e-3, outlier_thresh=1.96 ** 2, final_convergence=5e-3,
                             s_thresh=1e-200, num_min_lines=3, trace=None):
    """vp_localisation.py:168-450 (default "angle" distance only).  Normalises ``l`` in place.

    ``trace`` (optional dict) receives intermediates for the golden comparison."""
    assert distance_measure == "angle", "only the default distance measure is on the hot path"
    max_stdd = 1e-6
    s_init_factor = 1e-6
    merge_thresh_final = merge_thresh * 10
    split_merge_it = 100
    result = {"vp_assoc": None, "vp": None, "counts": None, "count_id": None,
              "decision_metric": None, "iterations": 0}

    em = _EM(l, lp, cnn_response, sphere_image, init_vp, use_weights, wbias, num_init_vp,
             outlier_thresh, trace)
    em.s = np.ones(em.cur.shape[0]) * (em.pdfpar.sigma * s_init_factor)

    p = em.estep(em.cur)                                                    # :245
    w = em.smooth(p.vl)
    counts, _, _ = em.counts(em.cur, w)
    if trace is not None:
        trace.update(p_v0=p.v.copy(), lvsq0=p.lvsq.copy(), p_vl0=p.vl.copy(), w0=w.copy(),
                     counts0=counts.copy())
    em.delete(np.where(counts < 3)[0])                                      # :250-251

    for i in range(num_iter):
        if em.cur.shape[0] == 0:                                            # :258-260
            return result
        events = 0
        if i % split_merge_freq == 0 and 0 < i < split_merge_it and do_split:   # :262-269
            mb = em.cur.shape[0]
            p = em.estep(em.cur)
            em.split(em.smooth(p.vl), merge_thresh)
            events += int(em.cur.shape[0] != mb)
        m_n = em.cur.shape[0]
        if trace is not None and trace.get("want_states"):
            trace.setdefault("states", []).append((em.cur.copy(), em.s.copy()))
        p = em.estep(em.cur)                                                # :273
        w = em.smooth(p.vl)                                                 # :282
        max_err = 0
        removed = []
        if do_iterations:
            for m in range(m_n):                                            # :284-322
                new = calc_new_vanishing_point(em.l, w[m, :])
                if new is None:
                    removed.append(m)
                    continue
                em.nxt[m, :] = new
                sm = _variance(p.lvsq[:, m], p.vl[m, :])
                sm = np.minimum(sm, max_stdd)
                sm = np.maximum(sm, s_thresh)
                em.s[m] = sm
                if np.isnan(sm):
                    removed.append(m)
                else:
                    err = np.arccos(np.minimum(np.abs(np.dot(em.cur[m], em.nxt[m])), 1.0))
                    max_err = np.maximum(max_err, err)
                    if err > 1.5:
                        removed.append(m)
        else:
            em.nxt[:, :] = em.cur                                           # :324-325
        em.delete(removed)                                                  # :329-331
        row = [em.cur.shape[0], float(max_err), 0, 0]
        if trace is not None:
            trace.setdefault("iters", []).append(row)
        # (:332 recomputes the E-step and discards it; its only side effect, flooring s at
        #  1e-200, is a no-op after the clamp at :307)

        if max_err < final_convergence or i == num_iter - 1 or not do_iterations:   # :335
            if do_merge:
                em.merge(True, merge_thresh_final)                          # :339 (index i+1)
            fin = [em.cur.shape[0]]
            p = em.estep(em.cur)                                            # :344 (stale index i)
            w = em.smooth(p.vl)
            removed = []
            assoc = np.argmax(w, axis=0)                                    # raises on M == 0, like the reference
            for m in range(em.cur.shape[0]):                                # :353-392
                sel = assoc == m
                if not sel.any():
                    continue
                wsel = w[m, sel] / np.max(w[m, sel])                        # :358
                new = calc_new_vanishing_point(em.l[sel, :], wsel)
                if new is None:
                    removed.append(m)
                    continue
                em.nxt[m, :] = new
                sm = np.minimum(_variance(p.lvsq[:, m], p.vl[m, :]), max_stdd)
                em.s[m] = sm
                if np.isnan(sm) or sm < s_thresh:
                    removed.append(m)
                else:
                    err = np.arccos(np.minimum(np.abs(np.dot(em.cur[m], em.nxt[m])), 1.0))
                    if err > 1.5:
                        removed.append(m)
            em.delete(removed)                                              # :394-396
            fin.append(em.cur.shape[0])
            p = em.estep(em.cur)                                            # :398 (still index i)
            metric = em.smooth(p.vl)
            if metric.size <= 0:                                            # :402-404
                return result
            good = np.unique(np.argmax(metric, axis=0))                     # :406-413
            em.cur, em.nxt, em.s = em.cur[good], em.nxt[good], em.s[good]
            fin.append(em.cur.shape[0])
            if trace is not None:
                trace["final"] = fin
            p = em.estep(em.nxt)                                            # :415 (index i+1 at last)
            metric = em.smooth(p.vl)
            counts, counts_w, vp_assoc = em.counts(em.nxt, metric)
            vidx = 0
            while vidx < em.cur.shape[0]:                                   # :423-437
                if counts[vidx] < num_min_lines:
                    em.delete([vidx])
                    p = em.estep(em.nxt)
                    metric = em.smooth(p.vl)
                    counts, counts_w, vp_assoc = em.counts(em.nxt, metric)
                else:
                    vidx += 1
            row[2], row[3] = em.cur.shape[0], events + 2
            return {"vp_assoc": vp_assoc, "vp": em.nxt, "counts": counts, "counts_weighted": counts_w,
                    "count_id": None, "decision_metric": metric, "iterations": i, "distribution": p,
                    "sigma": em.s}

        if i % split_merge_freq == 0 and 0 < i <= split_merge_it + split_merge_freq and do_merge:   # :444-448
            mb = em.cur.shape[0]
            em.merge(True, merge_thresh)
            events += 4 * int(em.cur.shape[0] != mb)
        row[2], row[3] = em.cur.shape[0], events
        em.cur = em.nxt
        em.nxt = np.zeros_like(em.cur)
    return result
