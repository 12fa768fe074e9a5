"""ctypes binding of libvpk.so (include/vpk.h).  There is no fallback: if the HIP library is
missing or no MI355X is present, using the package raises."""
import ctypes
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(PKG, "libvpk.so")

c_void = ctypes.c_void_p


class VpkError(RuntimeError):
    pass


class VpkRangeError(VpkError):
    """VPK_ERR_RANGE (include/vpk.h): the CNN's scaled fp16-pair activations reached fp16's largest finite number and were
    clamped -- the response maps since the last check are finite but not the net's.  ``flags``: bit li = the consuming layer
    (1 = conv2 ... 4 = conv5, 5 = fc6, 6 = fc7)."""

    def __init__(self, msg, flags):
        VpkError.__init__(self, msg)
        self.flags = flags


class EmParams(ctypes.Structure):
    """vpk_em_params: mirrors the keyword defaults of expectation_maximisation
    (reference vp_localisation.py:168-172)."""
    _fields_ = [("num_iter", ctypes.c_int32), ("do_merge", ctypes.c_int32), ("do_split", ctypes.c_int32),
                ("do_iterations", ctypes.c_int32), ("use_weights", ctypes.c_int32),
                ("num_init_vp", ctypes.c_int32), ("split_merge_freq", ctypes.c_int32),
                ("num_min_lines", ctypes.c_int32), ("wbias", ctypes.c_double),
                ("merge_thresh", ctypes.c_double), ("outlier_thresh", ctypes.c_double),
                ("final_convergence", ctypes.c_double), ("s_thresh", ctypes.c_double)]


class EmDistOut(ctypes.Structure):
    """vpk_em_dist_out (include/vpk.h): device buffers for EM_result['distribution']."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("p_v", "angles", "p_l", "p_lv", "p_vl", "lvsq")]


class StepArgs(ctypes.Structure):
    """vpk_step_args (include/vpk.h): the buffers of one vpk_pipeline_step call."""
    _fields_ = [("sphere", ctypes.c_void_p), ("batch", ctypes.c_int32), ("sphere_size", ctypes.c_int32),
                ("cnn_out", ctypes.c_void_p), ("offsets", ctypes.c_void_p), ("l_in", ctypes.c_void_p),
                ("l_work", ctypes.c_void_p), ("lp", ctypes.c_void_p), ("init_vp", ctypes.c_void_p),
                ("n_init", ctypes.c_int32), ("max_vp", ctypes.c_int32), ("params", ctypes.c_void_p),
                ("vp_out", ctypes.c_void_p), ("sigma_out", ctypes.c_void_p), ("counts_out", ctypes.c_void_p),
                ("counts_w_out", ctypes.c_void_p), ("num_vp_out", ctypes.c_void_p), ("assoc_out", ctypes.c_void_p),
                ("iterations_out", ctypes.c_void_p), ("status_out", ctypes.c_void_p), ("flags_out", ctypes.c_void_p),
                ("records", ctypes.c_void_p), ("image_ids", ctypes.c_void_p), ("events", ctypes.c_void_p),
                ("reuse_event", ctypes.c_void_p), ("em_prior", ctypes.c_void_p)]


EXPORTS = [
    "vpk_create", "vpk_destroy", "vpk_set_stream", "vpk_get_stream", "vpk_synchronize", "vpk_last_error", "vpk_version",
    "vpk_em_default_params", "vpk_device_info", "vpk_em_set_workgroups", "vpk_em_set_smoother", "vpk_em_set_lds_panel", "vpk_em_set_time_slice", "vpk_em_flush", "vpk_em_set_distribution_out", "vpk_cnn_load", "vpk_cnn_forward", "vpk_cnn_forward_tap",
    "vpk_cnn_set_profiling", "vpk_cnn_set_fusion", "vpk_cnn_set_precision", "vpk_cnn_set_algorithm", "vpk_cnn_last_layer_ms", "vpk_cnn_mean_layer_ms",
    "vpk_cnn_calibrate", "vpk_cnn_get_activation_scales", "vpk_cnn_set_activation_scales", "vpk_cnn_range_flags",
    "vpk_sphere_raster", "vpk_sphere_raster_flags", "vpk_sphere_raster_set_alternative", "vpk_em_batch", "vpk_em_workspace_bytes", "vpk_pairwise", "vpk_init_vps",
    "vpk_estep", "vpk_weight_matrix", "vpk_mstep", "vpk_line_counts", "vpk_cluster2", "vpk_horizon_batch", "vpk_lsd_detect",
    "vpk_pipeline_step", "vpk_build_records", "vpk_record_width", "vpk_math_probe",
]

_lib = None


def load():
    """Load libvpk.so; raises VpkError when it has not been built (python -m ...build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VpkError("libvpk.so is missing (%s): build it with `python -m vanishing_points_2017_amd.build`"
                       "; there is no CPU fallback" % SO_PATH)
    # torch first: the device buffers handed to the library are torch's, so the process must hold ONE HIP runtime -- the
    # one torch ships; loaded the other way round (libvpk.so's /opt/rocm runtime first) hipGetDeviceCount fails later
    import torch  # noqa: F401
    lib = ctypes.CDLL(SO_PATH)
    lib.vpk_last_error.restype = ctypes.c_char_p
    lib.vpk_last_error.argtypes = [c_void]
    lib.vpk_em_workspace_bytes.restype = ctypes.c_size_t
    lib.vpk_create.argtypes = [ctypes.c_int, ctypes.POINTER(c_void)]
    lib.vpk_destroy.argtypes = [c_void]
    lib.vpk_set_stream.argtypes = [c_void, c_void]
    lib.vpk_synchronize.argtypes = [c_void]
    lib.vpk_device_info.argtypes = [c_void, ctypes.POINTER(ctypes.c_int32)]
    lib.vpk_em_set_workgroups.argtypes = [c_void, ctypes.c_int]
    lib.vpk_em_set_smoother.argtypes = [c_void, ctypes.c_int]
    lib.vpk_em_set_lds_panel.argtypes = [c_void, ctypes.c_int]
    lib.vpk_em_set_time_slice.argtypes = [c_void, ctypes.c_double, ctypes.c_int]
    lib.vpk_em_set_distribution_out.argtypes = [c_void, ctypes.c_void_p]
    lib.vpk_em_flush.argtypes = [c_void]
    lib.vpk_pipeline_step.argtypes = [c_void, c_void, ctypes.POINTER(StepArgs)]
    lib.vpk_build_records.argtypes = [c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 6
    lib.vpk_horizon_batch.argtypes = [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void, ctypes.c_int,
                                      ctypes.c_double, ctypes.c_double, c_void, c_void]
    lib.vpk_get_stream.argtypes = [c_void]
    lib.vpk_get_stream.restype = ctypes.c_void_p
    lib.vpk_em_default_params.argtypes = [ctypes.POINTER(EmParams)]
    lib.vpk_em_batch.argtypes = [c_void, ctypes.c_int, c_void, c_void, c_void, c_void, c_void, ctypes.c_int,
                                 c_void, ctypes.c_int, ctypes.POINTER(EmParams), ctypes.c_int] + [c_void] * 11
    lib.vpk_em_workspace_bytes.argtypes = [c_void, ctypes.c_int, ctypes.c_int, ctypes.POINTER(EmParams),
                                           ctypes.c_int]
    lib.vpk_pairwise.argtypes = [c_void, ctypes.c_int, c_void, c_void, c_void, c_void]
    lib.vpk_init_vps.argtypes = [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void]
    lib.vpk_estep.argtypes = [c_void, ctypes.c_int, ctypes.c_int] + [c_void] * 8
    lib.vpk_weight_matrix.argtypes = [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void,
                                      ctypes.c_double, c_void]
    lib.vpk_mstep.argtypes = [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void]
    lib.vpk_cluster2.argtypes = [c_void, ctypes.c_int, c_void, c_void, c_void]
    lib.vpk_line_counts.argtypes = [c_void, ctypes.c_int, ctypes.c_int, c_void, c_void, c_void, c_void, c_void,
                                    ctypes.c_double, c_void, c_void, c_void]
    lib.vpk_cnn_load.argtypes = [c_void, ctypes.POINTER(c_void), c_void]
    lib.vpk_cnn_forward.argtypes = [c_void, c_void, ctypes.c_int, c_void]
    lib.vpk_cnn_forward_tap.argtypes = [c_void, c_void, ctypes.c_int, c_void, ctypes.c_int, c_void]
    lib.vpk_cnn_set_profiling.argtypes = [c_void, ctypes.c_int]
    lib.vpk_cnn_set_fusion.argtypes = [c_void, ctypes.c_int]
    lib.vpk_cnn_set_precision.argtypes = [c_void, ctypes.c_int]
    lib.vpk_cnn_set_algorithm.argtypes = [c_void, ctypes.c_int]
    lib.vpk_cnn_last_layer_ms.argtypes = [c_void, ctypes.POINTER(ctypes.c_float)]
    lib.vpk_cnn_mean_layer_ms.argtypes = [c_void, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
    lib.vpk_cnn_calibrate.argtypes = [c_void, c_void, ctypes.c_int]
    lib.vpk_cnn_get_activation_scales.argtypes = [c_void, ctypes.POINTER(ctypes.c_float)]
    lib.vpk_cnn_set_activation_scales.argtypes = [c_void, ctypes.POINTER(ctypes.c_float)]
    lib.vpk_cnn_range_flags.argtypes = [c_void, ctypes.POINTER(ctypes.c_uint32)]
    lib.vpk_sphere_raster.argtypes = [c_void, c_void, c_void, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void]
    lib.vpk_sphere_raster_flags.argtypes = [c_void, ctypes.c_int, c_void]
    lib.vpk_sphere_raster_set_alternative.argtypes = [c_void, ctypes.c_int]
    lib.vpk_math_probe.argtypes = [c_void, ctypes.c_int, ctypes.c_longlong, c_void, c_void]
    lib.vpk_lsd_detect.argtypes = [c_void, ctypes.c_int, ctypes.c_int, ctypes.c_double, c_void, ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_int)]
    _lib = lib
    return lib


def default_em_params(**overrides):
    p = EmParams()
    load().vpk_em_default_params(ctypes.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise TypeError("unknown EM parameter %r" % k)
        cur = getattr(p, k)
        setattr(p, k, int(v) if isinstance(cur, int) else float(v))
    return p


class Handle(object):
    """One vpk_handle (one per process / GPU).  Work is enqueued on ``stream`` (a raw hipStream_t,
    e.g. torch.cuda.current_stream().cuda_stream); None lets the library own a stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        h = c_void()
        rc = self.lib.vpk_create(int(device), ctypes.byref(h))
        if rc != 0:
            raise VpkError("vpk_create(device=%d) failed with %d: an MI355X (gfx950) is required, "
                           "there is no CPU fallback" % (device, rc))
        self.h = h
        self.device = device
        if stream:
            self.check(self.lib.vpk_set_stream(self.h, c_void(stream)))

    def check(self, rc):
        if rc != 0:
            raise VpkError("libvpk error %d: %s" % (rc, self.lib.vpk_last_error(self.h).decode()))

    def synchronize(self):
        self.check(self.lib.vpk_synchronize(self.h))

    def device_info(self):
        info = (ctypes.c_int32 * 4)()
        self.check(self.lib.vpk_device_info(self.h, info))
        return {"num_cu": info[0], "lds_per_block": info[1], "arch": info[2], "hbm_gib": info[3]}

    def em_set_workgroups(self, n):
        self.check(self.lib.vpk_em_set_workgroups(self.h, int(n)))

    def em_set_smoother(self, mode):
        self.check(self.lib.vpk_em_set_smoother(self.h, int(mode)))

    def em_set_lds_panel(self, doubles):
        self.check(self.lib.vpk_em_set_lds_panel(self.h, int(doubles)))

    def em_set_time_slice(self, slice_ms, n_max=0):
        """Time-sliced EM launches (include/vpk.h: vpk_em_set_time_slice); 0 switches back."""
        self.check(self.lib.vpk_em_set_time_slice(self.h, float(slice_ms), int(n_max)))

    def em_flush(self):
        self.check(self.lib.vpk_em_flush(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.vpk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_handles = {}


def get_handle(device=0):
    """Process-wide handle for ``device``.  The library owns a non-blocking HIP stream; callers
    that mix torch work with library calls synchronise through Handle.synchronize() /
    torch.cuda.synchronize() (see em.py)."""
    import torch
    if not torch.cuda.is_available():
        raise VpkError("no GPU visible: the vanishing-point hot path runs on MI355X only (no CPU fallback)")
    key = int(device)
    if key not in _handles:
        _handles[key] = Handle(key)
    return _handles[key]


def host_i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)
