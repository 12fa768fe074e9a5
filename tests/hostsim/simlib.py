"""ctypes access to the TEST-ONLY host build of the EM device source (tests/hostsim/sim_em.cpp).

Builds tests/hostsim/_build/libvpk_hostsim.so with g++ on first use.  See hip_sim.hpp for what
this is (a single-lane logic check of the device code) and is not (a product path)."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
SO = os.path.join(BUILD, "libvpk_hostsim.so")
SRC = [os.path.join(HERE, "sim_em.cpp"), os.path.join(HERE, "hip_sim.hpp"),
       os.path.join(HERE, "..", "..", "vanishing_points_2017_amd", "csrc", "em_device.hpp"),
       os.path.join(HERE, "..", "..", "vanishing_points_2017_amd", "csrc", "em_layout.hpp")]


class EmParams(ctypes.Structure):
    _fields_ = [("num_iter", ctypes.c_int32), ("do_merge", ctypes.c_int32), ("do_split", ctypes.c_int32),
                ("do_iterations", ctypes.c_int32), ("use_weights", ctypes.c_int32),
                ("num_init_vp", ctypes.c_int32), ("split_merge_freq", ctypes.c_int32),
                ("num_min_lines", ctypes.c_int32), ("wbias", ctypes.c_double),
                ("merge_thresh", ctypes.c_double), ("outlier_thresh", ctypes.c_double),
                ("final_convergence", ctypes.c_double), ("s_thresh", ctypes.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    os.makedirs(BUILD, exist_ok=True)
    stale = (not os.path.exists(SO)) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRC)
    if stale:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                               "-Wno-unknown-pragmas", SRC[0], "-o", SO])
    _lib = ctypes.CDLL(SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def default_params(**kw):
    p = EmParams()
    lib().sim_default_params(ctypes.byref(p))
    for k, v in kw.items():
        if k == "init_vp":
            continue
        setattr(p, k, type(getattr(p, k))(v))
    return p


def em_single(l, lp, cnn, sphere, init_vp=None, max_vp=64, want_metric=False, sliced=False, lds_doubles=None, **kw):
    n = lp.shape[0]
    if lds_doubles is None:     # the LDS budget the phases plan with (the library's vpk_em_set_lds_panel)
        os.environ.pop("VPK_SIM_WT_DOUBLES", None)
    else:
        os.environ["VPK_SIM_WT_DOUBLES"] = str(int(lds_doubles))
    p = default_params(**kw)
    l = np.ascontiguousarray(l, dtype=np.float64)
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    cnn = np.ascontiguousarray(cnn, dtype=np.float32)
    sphere = np.ascontiguousarray(sphere, dtype=np.uint8)
    iv = np.ascontiguousarray(init_vp, dtype=np.float64) if init_vp is not None else None
    vp = np.zeros((max_vp, 3)); sigma = np.zeros(max_vp); counts = np.zeros(max_vp); cw = np.zeros(max_vp)
    num = np.zeros(1, np.int32); assoc = np.zeros(n, np.int64); it = np.zeros(1, np.int32)
    st = np.zeros(1, np.int32); fl = np.zeros(1, np.uint32)
    metric = np.zeros((n, max_vp)) if want_metric else None
    trace = np.zeros((p.num_iter + 1, 12))
    D, I, L, U = ctypes.c_double, ctypes.c_int32, ctypes.c_longlong, ctypes.c_uint32
    if sliced:      # suspend at every checkpoint, destroy the LDS image and the caller's arrays in between
        os.environ["VPK_SIM_SLICED"] = "1"
    else:
        os.environ.pop("VPK_SIM_SLICED", None)
    slices = lib().sim_em_single(n, _p(l, D), _p(lp, D), _p(cnn, ctypes.c_float), _p(sphere, ctypes.c_ubyte),
                        sphere.shape[0], _p(iv, D), 0 if iv is None else iv.shape[0], ctypes.byref(p),
                        max_vp, _p(vp, D), _p(sigma, D), _p(counts, D), _p(cw, D), _p(num, I), _p(assoc, L),
                        _p(it, I), _p(st, I), _p(fl, U), _p(metric, D), _p(trace, D))
    m = int(num[0])
    lib().sim_last_states.restype = ctypes.POINTER(ctypes.c_double)
    raw = np.ctypeslib.as_array(lib().sim_last_states(), shape=(p.num_iter, 1 + 4 * 64)).copy()
    states = [(raw[i, 65:65 + 3 * int(raw[i, 0])].reshape(-1, 3), raw[i, 1:1 + int(raw[i, 0])]) for i in range(p.num_iter)]
    os.environ.pop("VPK_SIM_SLICED", None)
    os.environ.pop("VPK_SIM_WT_DOUBLES", None)
    return {"slices": slices, "states": states, "status": int(st[0]), "flags": int(fl[0]), "iterations": int(it[0]), "vp": vp[:m],
            "sigma": sigma[:m], "counts": counts[:m], "counts_weighted": cw[:m], "vp_assoc": assoc,
            "l": l, "trace": trace, "decision_metric": None if metric is None else metric[:, :m].T}


def pairwise(lp):
    n = lp.shape[0]
    lp = np.ascontiguousarray(lp, dtype=np.float64)
    lsim = np.zeros((n, n)); lscore = np.zeros(n); langle = np.zeros(n)
    D = ctypes.c_double
    lib().sim_pairwise(n, _p(lp, D), _p(lsim, D), _p(lscore, D), _p(langle, D))
    return lsim, lscore, langle


def init_vps(cnn, sphere, num_max=25):
    cnn = np.ascontiguousarray(cnn, dtype=np.float32)
    sphere = np.ascontiguousarray(sphere, dtype=np.uint8)
    v0 = np.zeros((64, 3)); m0 = np.zeros(1, np.int32); w = np.zeros(400, np.float32)
    lib().sim_init_vps(_p(cnn, ctypes.c_float), _p(sphere, ctypes.c_ubyte), sphere.shape[0], num_max,
                       _p(v0, ctypes.c_double), _p(m0, ctypes.c_int32), _p(w, ctypes.c_float))
    return v0[:int(m0[0])], w


def estep(lp, cnn, v, s):
    n, m = lp.shape[0], v.shape[0]
    lp = np.ascontiguousarray(lp, dtype=np.float64); v = np.ascontiguousarray(v, dtype=np.float64)
    cnn = np.ascontiguousarray(cnn, dtype=np.float32)
    s = np.ascontiguousarray(s, dtype=np.float64).copy()
    pv = np.zeros(m); lvsq = np.zeros((m, n)); pvl = np.zeros((m, n))
    D = ctypes.c_double
    lib().sim_estep(n, m, _p(lp, D), _p(cnn, ctypes.c_float), _p(v, D), _p(s, D), _p(pv, D), _p(lvsq, D), _p(pvl, D))
    return pv, lvsq, pvl, s


def weight_matrix(p_vl, lweight, lsim, bias=1.0):
    m, n = p_vl.shape
    p_vl = np.ascontiguousarray(p_vl); lweight = np.ascontiguousarray(lweight); lsim = np.ascontiguousarray(lsim)
    w = np.zeros((m, n))
    D = ctypes.c_double
    lib().sim_weight_matrix(n, m, _p(p_vl, D), _p(lweight, D), _p(lsim, D), ctypes.c_double(bias), _p(w, D))
    return w


def cluster2(ldist):
    n = ldist.shape[0]
    ldist = np.ascontiguousarray(ldist, dtype=np.float64)
    labels = np.zeros(n, np.int32); fl = np.zeros(1, np.uint32)
    lib().sim_cluster2(n, _p(ldist, ctypes.c_double), _p(labels, ctypes.c_int32), _p(fl, ctypes.c_uint32))
    return labels, int(fl[0])


def mstep(l, w):
    m, n = w.shape
    l = np.ascontiguousarray(l, dtype=np.float64); w = np.ascontiguousarray(w, dtype=np.float64)
    vp = np.zeros((m, 3))
    D = ctypes.c_double
    lib().sim_mstep(n, m, _p(l, D), _p(w, D), _p(vp, D))
    return vp


# ---- the rasteriser's arithmetic (csrc/raster_device.hpp) -----------------------------------------------------------------
RASTER_SO = os.path.join(BUILD, "libvpk_hostsim_raster.so")
RASTER_SRC = [os.path.join(HERE, "sim_raster.cpp"),
              os.path.join(HERE, "..", "..", "vanishing_points_2017_amd", "csrc", "raster_device.hpp")]
_raster_lib = None


def raster_lib():
    global _raster_lib
    if _raster_lib is not None:
        return _raster_lib
    os.makedirs(BUILD, exist_ok=True)
    stale = (not os.path.exists(RASTER_SO)) or any(os.path.getmtime(s) > os.path.getmtime(RASTER_SO) for s in RASTER_SRC)
    if stale:
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", RASTER_SRC[0], "-o", RASTER_SO])
    _raster_lib = ctypes.CDLL(RASTER_SO)
    _raster_lib.sim_raster.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                       ctypes.c_int, ctypes.POINTER(ctypes.c_uint8)]
    _raster_lib.sim_edge_shares.argtypes = [ctypes.c_double] * 4 + [ctypes.c_int, ctypes.c_int]
    return _raster_lib


def sim_raster(lines, size=250, alpha=0.1, alternative=False):
    """The product's stroke / cell / blend arithmetic run serially on the host: uint8 [size, size] and the overflow flags."""
    l = np.ascontiguousarray(lines, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((size, size), dtype=np.uint8)
    flags = raster_lib().sim_raster(_p(l, ctypes.c_double), l.shape[0], size, float(alpha), int(bool(alternative)),
                                    _p(out, ctypes.c_uint8))
    return out, flags


def sim_edge_shares(x1, y1, x2, y2, size, k):
    return raster_lib().sim_edge_shares(x1, y1, x2, y2, size, k)
