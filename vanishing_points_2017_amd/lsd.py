"""`lsd.detect_line_segments(image)` -- the call the reference makes into its un-vendored `lsdpython` submodule
(evaluation.py:7,238; .gitmodules:1-3).  Backed by the host-side detector in libvpk.so (csrc/vpk_lsd.cpp: the
published LSD algorithm with its default parameters; parity with the absent original is unpinned)."""
import ctypes

import numpy as np

from . import _lib


def detect_line_segments(image, scale=0.8):
    """image: 2-D array of grey levels 0..255 -> (N, 7) float64: x1, y1, x2, y2, width, p, -log10(NFA)
    in pixel coordinates (x = column, y = row)."""
    img = np.ascontiguousarray(image, dtype=np.float64)
    if img.ndim != 2:
        raise ValueError("detect_line_segments expects a 2-D grey-level image")
    lib = _lib.load()
    h, w = img.shape
    cap = 4096
    while True:
        out = np.zeros((cap, 7), dtype=np.float64)
        n = ctypes.c_int(0)
        rc = lib.vpk_lsd_detect(img.ctypes.data_as(ctypes.c_void_p), int(w), int(h), float(scale),
                                out.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n))
        if rc != 0:
            raise _lib.VpkError("vpk_lsd_detect failed with %d (image %d x %d)" % (rc, w, h))
        if n.value <= cap:
            return out[:n.value].copy()
        cap = n.value
