"""Batched EM refinement on the GPU (C-ABI: vpk_em_batch).

Replaces the per-image loop of the reference's evaluation.run_em (evaluation.py:295-329): all
images of a batch are refined concurrently, one persistent workgroup per image."""
import ctypes

import numpy as np

from . import _lib
from .runtime import get_runtime

MAX_VP = 64


def _params(kw):
    kw = dict(kw)
    if kw.pop("distance_measure", "angle") != "angle":
        # reference vp_localisation.py:196-203: other measures are not on the hot path
        raise AssertionError("only distance_measure='angle' is supported")
    kw.pop("init_vp", None)
    kw.pop("sphere_image", None)
    return _lib.default_em_params(**kw)


def em_batch_device(rt, offsets, l, lp, cnn, sphere, init_vp=None, params=None, max_vp=MAX_VP,
                    want_metric=False, want_trace=False, want_distribution=False):
    """Run vpk_em_batch on device tensors.  offsets: host int64 (B+1).  l (sum N x 3, f64) is
    normalised in place.  Returns a dict of device tensors.  want_distribution adds "dist": the arrays of
    EM_result['distribution'] (include/vpk.h: vpk_em_set_distribution_out)."""
    torch = rt.torch
    offsets = _lib.host_i64(offsets)
    batch = offsets.shape[0] - 1
    total = int(offsets[-1])
    p = params if params is not None else _lib.default_em_params()
    ssize = int(sphere.shape[-1])
    dev = rt.tdev
    with rt.on_stream():
        out = {
            "vp": torch.empty((batch, max_vp, 3), dtype=torch.float64, device=dev),
            "sigma": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
            "counts": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
            "counts_weighted": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
            "num_vp": torch.empty((batch,), dtype=torch.int32, device=dev),
            "vp_assoc": torch.empty((max(total, 1),), dtype=torch.int64, device=dev),
            "iterations": torch.empty((batch,), dtype=torch.int32, device=dev),
            "status": torch.empty((batch,), dtype=torch.int32, device=dev),
            "flags": torch.empty((batch,), dtype=torch.int32, device=dev),
            "metric": torch.empty((max(total, 1), max_vp), dtype=torch.float64, device=dev) if want_metric else None,
            "trace": torch.empty((batch, p.num_iter + 1, 12), dtype=torch.float64, device=dev) if want_trace else None,
        }
        if want_distribution:
            z = lambda *shape: torch.zeros(shape, dtype=torch.float64, device=dev)
            out["dist"] = {"p_v": z(batch, max_vp), "angles": z(batch, max_vp, 2), "p_l": z(max(total, 1)),
                           "p_lv": z(max(total, 1), max_vp), "p_vl": z(max(total, 1), max_vp), "lvsq": z(max(total, 1), max_vp)}
            dd = _lib.EmDistOut(*[rt.ptr(out["dist"][k]) for k in ("p_v", "angles", "p_l", "p_lv", "p_vl", "lvsq")])
            rt.check(rt.lib.vpk_em_set_distribution_out(rt.h, ctypes.byref(dd)))
        n_init = 0 if init_vp is None else int(init_vp.shape[-2])
        rc = rt.lib.vpk_em_batch(
            rt.h, batch, offsets.ctypes.data_as(ctypes.c_void_p), rt.ptr(l), rt.ptr(lp), rt.ptr(cnn),
            rt.ptr(sphere), ssize, rt.ptr(init_vp), n_init, ctypes.byref(p), max_vp, rt.ptr(out["vp"]),
            rt.ptr(out["sigma"]), rt.ptr(out["counts"]), rt.ptr(out["counts_weighted"]), rt.ptr(out["num_vp"]),
            rt.ptr(out["vp_assoc"]), rt.ptr(out["iterations"]), rt.ptr(out["status"]), rt.ptr(out["flags"]),
            rt.ptr(out["metric"]), rt.ptr(out["trace"]))
        rt.check(rc)
    return out


def upload_batch(rt, scenes):
    """Concatenate per-image host arrays and copy them to HBM.  scenes: list of dicts with
    l (N x 3), lp (N x 4), cnn_response (20 x 20 f32), sphere_image (S x S u8)[, init_vp].  A scene whose
    sphere_image is None gets it from its lines first, as the reference's datum does (evaluation.py:175):
    sphere_mapping.attach_rasters -> vpk_sphere_raster."""
    torch = rt.torch
    if any(s.get("sphere_image") is None for s in scenes):
        from .sphere_mapping import attach_rasters
        attach_rasters(scenes, runtime=rt)
    counts = [int(s["lp"].shape[0]) for s in scenes]
    offsets = np.zeros(len(scenes) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    l = np.concatenate([np.asarray(s["l"], dtype=np.float64).reshape(-1, 3) for s in scenes], 0)
    lp = np.concatenate([np.asarray(s["lp"], dtype=np.float64).reshape(-1, 4) for s in scenes], 0)
    cnn = np.stack([np.asarray(s["cnn_response"], dtype=np.float32).reshape(400) for s in scenes], 0)
    sphere = np.stack([np.ascontiguousarray(s["sphere_image"], dtype=np.uint8) for s in scenes], 0)
    has_init = [s.get("init_vp") is not None for s in scenes]
    init = None
    if any(has_init):
        if not all(has_init):
            raise ValueError("init_vp must be given for every image of a batch or for none")
        init = np.stack([np.asarray(s["init_vp"], dtype=np.float64) for s in scenes], 0)
    with rt.on_stream():
        dev = {
            "offsets": offsets,
            "l": torch.from_numpy(np.ascontiguousarray(l)).to(rt.tdev),
            "lp": torch.from_numpy(np.ascontiguousarray(lp)).to(rt.tdev),
            "cnn": torch.from_numpy(np.ascontiguousarray(cnn)).to(rt.tdev),
            "sphere": torch.from_numpy(np.ascontiguousarray(sphere)).to(rt.tdev),
            "init_vp": None if init is None else torch.from_numpy(np.ascontiguousarray(init)).to(rt.tdev),
        }
    return dev


def em_batch(scenes, device=0, want_metric=False, want_trace=False, want_distribution=False, **kwargs):
    """Host-in / host-out convenience: refine a list of images, return one reference-style
    result dict per image (keys as vp_localisation.py:441-442 plus status/flags/l).  'distribution' is the
    reference's probability_functions.PDF tuple when want_distribution is set, None otherwise."""
    rt = get_runtime(device)
    p = _params(kwargs)
    d = upload_batch(rt, scenes)
    out = em_batch_device(rt, d["offsets"], d["l"], d["lp"], d["cnn"], d["sphere"], d["init_vp"], p,
                          want_metric=want_metric, want_trace=want_trace, want_distribution=want_distribution)
    rt.synchronize()
    dist = {k: v.cpu().numpy() for k, v in out.pop("dist").items()} if want_distribution else None
    host = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()}
    l_norm = d["l"].cpu().numpy()
    offs = d["offsets"]
    results = []
    for b in range(len(scenes)):
        lo, hi = int(offs[b]), int(offs[b + 1])
        m = int(host["num_vp"][b])
        status = int(host["status"][b])
        if status == 3:         # VPK_EM_NO_SLOT (include/vpk.h): the image was not refined -- an error, not a result
            raise _lib.VpkError("vpk_em_batch: image %d found no working-set slot (time-sliced launch)" % b)
        res = {"status": status, "flags": int(host["flags"][b]) & 0xffffffff, "l": l_norm[lo:hi]}
        if status == 0:
            res.update({
                "vp_assoc": host["vp_assoc"][lo:hi].copy(), "vp": host["vp"][b, :m].copy(),
                "counts": host["counts"][b, :m].copy(),
                "counts_weighted": host["counts_weighted"][b, :m].copy(), "count_id": None,
                "decision_metric": None if host["metric"] is None else host["metric"][lo:hi, :m].T.copy(),
                "iterations": int(host["iterations"][b]), "distribution": None,
                "sigma": host["sigma"][b, :m].copy()})
            if dist is not None:                        # probability_functions.py:120, shapes as the reference's
                from .probability_functions import PDF
                res["distribution"] = PDF(v=dist["p_v"][b, :m].copy(), lv=dist["p_lv"][lo:hi, :m].copy(),
                                          vl=dist["p_vl"][lo:hi, :m].T.copy(), l=dist["p_l"][lo:hi].copy(),
                                          lvsq=dist["lvsq"][lo:hi, :m].copy(), angles=dist["angles"][b, :m].copy())
        else:   # vp_localisation.py:205-206
            res.update({"vp_assoc": None, "vp": None, "counts": None, "count_id": None,
                        "decision_metric": None, "iterations": 0})
        if host["trace"] is not None:
            res["trace"] = host["trace"][b]
        results.append(res)
    return results
