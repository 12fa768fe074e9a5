"""Inverse-gnomonic sphere raster on the GPU (C-ABI: vpk_sphere_raster) behind the reference's
sphere_mapping.sphere_line_plot (sphere_mapping.py:36-72) / evaluation.get_sphere_image (:12-14)."""
import ctypes

import numpy as np

from ._lib import VpkError

from .runtime import get_runtime


def raster_batch_device(rt, l, offsets, size=500, alpha=0.1, out=None):
    """l: device tensor (sum N x 3, f64); offsets: host int64 (B+1).  Returns uint8 (B,size,size) -- ``out`` if given.
    Asynchronous on rt's stream; a call with the offsets of the previous call on this handle does not wait for anything."""
    t = rt.torch
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    batch = offsets.shape[0] - 1
    with rt.on_stream():
        if out is None:
            out = t.empty((batch, size, size), dtype=t.uint8, device=rt.tdev)
        assert out.dtype == t.uint8 and out.is_contiguous() and tuple(out.shape) == (batch, size, size)
        rt.check(rt.lib.vpk_sphere_raster(rt.h, rt.ptr(l), offsets.ctypes.data_as(ctypes.c_void_p), batch,
                                          int(size), float(alpha), rt.ptr(out)))
    return out


def raster_batch(lines_list, size=500, alpha=0.1, device=0, alternative=False, runtime=None):
    """Host convenience: list of (N_i x 3) arrays -> (B, size, size) uint8.  An image without lines gets the frame-only
    canvas (what the reference's figure shows when the loop at sphere_mapping.py:54 runs zero times)."""
    rt = runtime if runtime is not None else get_runtime(device)
    counts = [int(np.asarray(a).reshape(-1, 3).shape[0]) for a in lines_list]
    offsets = np.zeros(len(counts) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(counts)
    cat = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1, 3) for a in lines_list], 0))
    with rt.on_stream():
        d = rt.torch.from_numpy(cat).to(rt.tdev) if cat.shape[0] else None
    rt.check(rt.lib.vpk_sphere_raster_set_alternative(rt.h, 1 if alternative else 0))
    try:
        out = raster_batch_device(rt, d, offsets, size, alpha)
    finally:
        rt.check(rt.lib.vpk_sphere_raster_set_alternative(rt.h, 0))
    rt.synchronize()
    flags = raster_flags(rt, len(counts))
    if flags.any():                                          # never a silently incomplete raster
        raise VpkError("sphere raster: the kernel's buffers were too small for a line of image(s) %s (vertices per "
                       "outline or coverage bytes): the raster of those images is incomplete" % np.nonzero(flags)[0].tolist())
    return out.cpu().numpy()


def raster_flags(rt, batch):
    """Per-image flags of the last raster call on rt's handle (waits for it): bit 0 = a line was truncated / dropped."""
    flags = np.zeros(batch, dtype=np.uint32)
    rt.check(rt.lib.vpk_sphere_raster_flags(rt.h, int(batch), flags.ctypes.data_as(ctypes.c_void_p)))
    return flags


def attach_rasters(scenes, size=500, alpha=0.1, device=0, runtime=None):
    """What the reference does when it makes a datum (evaluation.py:175): every scene whose 'sphere_image' is None gets
    the raster of its lines (one vpk_sphere_raster call for all of them).  Returns ``scenes``."""
    todo = [s for s in scenes if s.get("sphere_image") is None]
    if todo:
        for s, r in zip(todo, raster_batch([s["l"] for s in todo], size=size, alpha=alpha, device=device, runtime=runtime)):
            s["sphere_image"] = r
    return scenes


def sphere_line_plot(lines, size, alpha=0.1, f=1.0, alternative=False, device=0):
    """sphere_mapping.py:36-72.  Scales lines[:, 0:2] by f IN PLACE like the reference (:55-56)."""
    lines[:, 0] *= f
    lines[:, 1] *= f
    return raster_batch([lines], size=size, alpha=alpha, device=device, alternative=alternative)[0]
