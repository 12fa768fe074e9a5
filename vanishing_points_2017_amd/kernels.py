"""Host-in / host-out wrappers of the fine-grained C-ABI entry points (unit parity against the
reference's individual functions).  Each call copies its inputs to HBM, launches one workgroup
and copies the result back -- for tests and debugging, not for throughput."""
import numpy as np

from .runtime import get_runtime


def _up(rt, a, dtype):
    return rt.torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(rt.tdev)


def pairwise(lp, device=0):
    """calc_lsim (vp_localisation.py:87-108) + line_rating_knn (:34-84) + lines_angles (:765-776)."""
    rt = get_runtime(device)
    t = rt.torch
    n = lp.shape[0]
    with rt.on_stream():
        d_lp = _up(rt, lp, np.float64)
        lsim = t.empty((n, n), dtype=t.float64, device=rt.tdev)
        lscore = t.empty((n,), dtype=t.float64, device=rt.tdev)
        langle = t.empty((n,), dtype=t.float64, device=rt.tdev)
        rt.check(rt.lib.vpk_pairwise(rt.h, n, rt.ptr(d_lp), rt.ptr(lsim), rt.ptr(lscore), rt.ptr(langle)))
    rt.synchronize()
    return lsim.cpu().numpy(), lscore.cpu().numpy(), langle.cpu().numpy()


def init_vps(cnn, sphere, num_max=25, device=0):
    """find_initial_vps (vp_localisation.py:111-165) + pdf_params weights (probability_functions.py:62-96)."""
    rt = get_runtime(device)
    t = rt.torch
    with rt.on_stream():
        d_cnn = _up(rt, np.asarray(cnn).reshape(400), np.float32)
        d_sp = _up(rt, sphere, np.uint8)
        v0 = t.zeros((64, 3), dtype=t.float64, device=rt.tdev)
        m0 = t.zeros((1,), dtype=t.int32, device=rt.tdev)
        w = t.zeros((400,), dtype=t.float32, device=rt.tdev)
        rt.check(rt.lib.vpk_init_vps(rt.h, rt.ptr(d_cnn), rt.ptr(d_sp), int(sphere.shape[0]), int(num_max),
                                     rt.ptr(v0), rt.ptr(m0), rt.ptr(w)))
    rt.synchronize()
    m = int(m0.cpu()[0])
    return v0.cpu().numpy()[:m], w.cpu().numpy()


def estep(lp, cnn, v, s, device=0):
    """calc_probabilities (probability_functions.py:99-120). Returns p_v, lvsq (N,M), p_vl (M,N), p_l, s."""
    rt = get_runtime(device)
    t = rt.torch
    n, m = lp.shape[0], v.shape[0]
    with rt.on_stream():
        d_lp = _up(rt, lp, np.float64)
        d_cnn = _up(rt, np.asarray(cnn).reshape(400), np.float32)
        d_v = _up(rt, v, np.float64)
        d_s = _up(rt, s, np.float64)
        pv = t.empty((m,), dtype=t.float64, device=rt.tdev)
        lvsq = t.empty((m, n), dtype=t.float64, device=rt.tdev)
        pvl = t.empty((m, n), dtype=t.float64, device=rt.tdev)
        pl = t.empty((n,), dtype=t.float64, device=rt.tdev)
        rt.check(rt.lib.vpk_estep(rt.h, n, m, rt.ptr(d_lp), rt.ptr(d_cnn), rt.ptr(d_v), rt.ptr(d_s), rt.ptr(pv),
                                  rt.ptr(lvsq), rt.ptr(pvl), rt.ptr(pl)))
    rt.synchronize()
    return pv.cpu().numpy(), lvsq.cpu().numpy().T.copy(), pvl.cpu().numpy(), pl.cpu().numpy(), d_s.cpu().numpy()


def weight_matrix(p_vl, lweight, lsim, bias=1.0, device=0):
    """weight_matrix (vp_localisation.py:515-524)."""
    rt = get_runtime(device)
    t = rt.torch
    m, n = p_vl.shape
    with rt.on_stream():
        d_p = _up(rt, p_vl, np.float64)
        d_lw = _up(rt, lweight, np.float64)
        d_ls = _up(rt, lsim, np.float64)
        w = t.empty((m, n), dtype=t.float64, device=rt.tdev)
        rt.check(rt.lib.vpk_weight_matrix(rt.h, n, m, rt.ptr(d_p), rt.ptr(d_lw), rt.ptr(d_ls), float(bias), rt.ptr(w)))
    rt.synchronize()
    return w.cpu().numpy()


def mstep(l, w, device=0):
    """calc_new_vanishing_point (vp_localisation.py:453-479) for every row of w (M,N)."""
    rt = get_runtime(device)
    t = rt.torch
    m, n = w.shape
    with rt.on_stream():
        d_l = _up(rt, l, np.float64)
        d_w = _up(rt, w, np.float64)
        vp = t.empty((m, 3), dtype=t.float64, device=rt.tdev)
        valid = t.empty((m,), dtype=t.int32, device=rt.tdev)
        rt.check(rt.lib.vpk_mstep(rt.h, n, m, rt.ptr(d_l), rt.ptr(d_w), rt.ptr(vp), rt.ptr(valid)))
    rt.synchronize()
    return vp.cpu().numpy(), valid.cpu().numpy()


def line_counts(lp, v, s, w, lweight, thresh=1.96 ** 2, device=0):
    """calc_vp_line_counts (vp_localisation.py:482-512): returns (counts, counts_weighted, assoc)."""
    rt = get_runtime(device)
    t = rt.torch
    m, n = w.shape
    with rt.on_stream():
        d_lp, d_v, d_s = _up(rt, lp, np.float64), _up(rt, v, np.float64), _up(rt, s, np.float64)
        d_w, d_lw = _up(rt, w, np.float64), _up(rt, lweight, np.float64)
        counts = t.empty((m,), dtype=t.float64, device=rt.tdev)
        counts_w = t.empty((m,), dtype=t.float64, device=rt.tdev)
        assoc = t.empty((n,), dtype=t.int64, device=rt.tdev)
        rt.check(rt.lib.vpk_line_counts(rt.h, n, m, rt.ptr(d_lp), rt.ptr(d_v), rt.ptr(d_s), rt.ptr(d_w), rt.ptr(d_lw),
                                        float(thresh), rt.ptr(counts), rt.ptr(counts_w), rt.ptr(assoc)))
    rt.synchronize()
    return counts.cpu().numpy(), counts_w.cpu().numpy(), assoc.cpu().numpy()


def cluster2(ldist, device=0):
    """The 2-cluster average-linkage agglomeration of split_best_vp (vp_localisation.py:568-578)."""
    rt = get_runtime(device)
    t = rt.torch
    n = ldist.shape[0]
    with rt.on_stream():
        d = _up(rt, ldist, np.float64)
        labels = t.empty((n,), dtype=t.int32, device=rt.tdev)
        flags = t.zeros((1,), dtype=t.int32, device=rt.tdev)
        rt.check(rt.lib.vpk_cluster2(rt.h, n, rt.ptr(d), rt.ptr(labels), rt.ptr(flags)))
    rt.synchronize()
    return labels.cpu().numpy(), int(flags.cpu()[0])


MATH_FUNCTIONS = ("exp", "acos", "asin", "atan", "sqrt", "sin", "cos", "log")


def math_probe(name, x, device=0):
    """The device's double-precision elementary function ``name`` (as the EM kernels call it) on every element of x."""
    rt = get_runtime(device)
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    dx = _up(rt, x, np.float64)
    with rt.on_stream():
        dy = rt.torch.empty_like(dx)
        rt.check(rt.lib.vpk_math_probe(rt.h, MATH_FUNCTIONS.index(name), int(x.shape[0]), rt.ptr(dx), rt.ptr(dy)))
    rt.synchronize()
    return dy.cpu().numpy()
