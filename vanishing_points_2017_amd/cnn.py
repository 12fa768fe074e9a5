"""AlexNet-500 forward on the GPU (C-ABI: vpk_cnn_load / vpk_cnn_forward) behind the reference's
evaluation.py CNN surface: init_caffe (:17), read_mean_blob (:25), caffe_forward (:34)."""
import ctypes

import numpy as np

from ._lib import VpkRangeError
from .runtime import get_runtime

# (name, weight shape in Caffe layout, fan-in) per cnn/deploy.prototxt
LAYER_SHAPES = [
    ("conv1", (96, 1, 11, 11)), ("conv2", (256, 48, 5, 5)), ("conv3", (384, 256, 3, 3)),
    ("conv4", (384, 192, 3, 3)), ("conv5", (256, 192, 3, 3)), ("fc6", (4096, 57600)),
    ("fc7", (4096, 4096)), ("fc8", (400, 4096)),
]
TAP_SHAPES = [(96, 123, 123), (96, 61, 61), (256, 61, 61), (256, 30, 30), (384, 30, 30), (384, 30, 30),
              (256, 30, 30), (256, 15, 15), (4096,), (4096,), (400,)]


def synthetic_weights(seed=0, scale=1.0):
    """Seeded random-init weights of the deploy architecture (the trained caffemodel is a download
    the offline build cannot reach).  He-style std = sqrt(2 / fan_in) keeps activations O(1)
    through the 8 layers so that numerics tests are meaningful; biases as train_val.prototxt."""
    rs = np.random.RandomState(seed)
    bias_val = {"conv1": 0.0, "conv2": 0.1, "conv3": 0.0, "conv4": 0.1, "conv5": 0.1, "fc6": 0.1, "fc7": 0.1,
                "fc8": 0.0}
    out = {}
    for name, shape in LAYER_SHAPES:
        fan_in = int(np.prod(shape[1:]))
        std = scale * np.sqrt(2.0 / fan_in)
        if name == "conv1":
            std /= 40.0          # input is raw 0..255 grey minus mean (no scaling, evaluation.py:35)
        w = rs.standard_normal(size=shape).astype(np.float32)
        w *= np.float32(std)
        out[name] = (w, np.full((shape[0],), bias_val[name], dtype=np.float32))
    return out


def synthetic_mean(seed=0):
    rs = np.random.RandomState(seed + 1)
    return (12.0 + 4.0 * rs.random_sample((500, 500))).astype(np.float32)


class Net(object):
    """Counterpart of caffe.Net(model_def, model_weights, caffe.TEST) for this one topology."""

    def __init__(self, weights, mean, device=0, runtime=None):
        self.rt = runtime if runtime is not None else get_runtime(device)
        blobs = []
        keep = []
        for name, shape in LAYER_SHAPES:
            w, b = weights[name]
            w = np.ascontiguousarray(w, dtype=np.float32)
            b = np.ascontiguousarray(b, dtype=np.float32)
            if w.shape != shape or b.shape != (shape[0],):
                raise ValueError("%s: expected weight %s / bias (%d,), got %s / %s" % (name, shape, shape[0], w.shape, b.shape))
            keep += [w, b]
            blobs += [w.ctypes.data, b.ctypes.data]
        mean = np.ascontiguousarray(np.asarray(mean, dtype=np.float32).reshape(500, 500))
        arr = (ctypes.c_void_p * 16)(*blobs)
        self.rt.check(self.rt.lib.vpk_cnn_load(self.rt.h, arr, ctypes.c_void_p(mean.ctypes.data)))

    def forward_device(self, sphere, tap=None):
        """sphere: uint8 device tensor (B,500,500) -> float32 device tensor (B,20,20) [, tap tensor]."""
        rt = self.rt
        t = rt.torch
        batch = int(sphere.shape[0])
        with rt.on_stream():
            out = t.empty((batch, 20, 20), dtype=t.float32, device=rt.tdev)
            if tap is None:
                rt.check(rt.lib.vpk_cnn_forward(rt.h, rt.ptr(sphere), batch, rt.ptr(out)))
                return out
            tp = t.empty((batch,) + TAP_SHAPES[tap], dtype=t.float32, device=rt.tdev)
            rt.check(rt.lib.vpk_cnn_forward_tap(rt.h, rt.ptr(sphere), batch, rt.ptr(out), int(tap), rt.ptr(tp)))
            return out, tp

    LAYER_NAMES = ["conv1", "norm1", "pool1", "conv2", "norm2", "pool2", "conv3", "conv4", "conv5", "pool5",
                   "fc6", "fc7", "fc8"]
    # 2*MAC per image of the MFMA layers, for roofline accounting
    LAYER_FLOP = {"conv1": 2 * 96 * 15129 * 121, "conv2": 2 * 256 * 3721 * 1200, "conv3": 2 * 384 * 900 * 2304,
                  "conv4": 2 * 384 * 900 * 1728, "conv5": 2 * 256 * 900 * 1728, "fc6": 2 * 57600 * 4096,
                  "fc7": 2 * 4096 * 4096, "fc8": 2 * 4096 * 400}

    @classmethod
    def executed_flop(cls, layer, fusion=3, precision=0, algorithm=4, batch=102):
        """(matrix-core flops the kernels EXECUTE for `layer` per image, which pipe: "f32" or "bf16") under a setting -- tile padding,
        Winograd's product count and the six (conv1: three) bf16 products -- or three fp16 products, algorithm 4 -- per f32 product
        included; the roofline's numerator.  Pipes: "f32", "bf16", "f16" (the last two: the same dense peak)."""
        alg = cls.LAYER_FLOP[layer]
        if layer == "conv1":
            if fusion == 3:      # 168 tiles x 12 waves x 72 v_mfma_f32_16x16x32_bf16 (cnn_conv1_pieces.hpp)
                return 168 * 12 * 72 * 2.0 * 16 * 16 * 32, "bf16"
            if fusion == 4:      # ... x 48 v_mfma_f32_16x16x32_f16
                return 168 * 12 * 48 * 2.0 * 16 * 16 * 32, "f16"
            if fusion == 1:      # 168 tiles x 8 waves x 186 v_mfma_f32_16x16x4_f32
                return 168 * 8 * 186 * 2.0 * 16 * 16 * 4, "f32"
            return alg * 128.0 / 121.0, "f32"
        if layer in ("conv2", "conv3", "conv4", "conv5"):
            if precision == 1:
                return 6.0 * alg, "bf16"
            if algorithm == 4:      # fp16 pairs: tiles of 128 channels x 4 rows (conv4: 64 x 8) x 32 columns, 4 waves x 12 v_mfma_f32_32x32x16_f16 per K16 step
                tiles, steps = {"conv2": (2 * 16 * 2 * 1, 75), "conv3": (1 * 8 * 1 * 3, 16 * 9), "conv4": (2 * 4 * 1 * 3, 12 * 9),
                                "conv5": (2 * 8 * 1 * 1, 12 * 9)}[layer]       # (groups x row tiles x column tiles x channel tiles, K16 steps)
                return tiles * steps * 4 * 12 * 2.0 * 32 * 32 * 16, "f16"
            if layer == "conv2" and algorithm >= 2:      # 64 tiles x 4 waves x 75 steps x 24 v_mfma_f32_32x32x16_bf16
                return 64 * 4 * 75 * 24 * 2.0 * 32 * 32 * 16, "bf16"
            if algorithm >= 1:
                return alg * (36.0 / 100.0 if layer == "conv2" else 16.0 / 36.0), "f32"
            return alg, "f32"
        pad = (-(-batch // 128) * 128) / float(batch)     # dense layers: 128-column tiles
        if precision == 0 and ((layer == "fc6" and algorithm >= 2) or (layer == "fc7" and algorithm == 4)):
            return alg * pad * (3.0 if algorithm == 4 else 6.0), "f16" if algorithm == 4 else "bf16"      # cnn_dense_pieces.hpp
        return alg * pad, "f32"

    def set_fusion(self, on=3):
        """conv1 + norm1 + pool1: 3 (default) one kernel on the bf16 matrix cores with exact operands, 4 the same kernel on scaled fp16
        pairs of the weights, 1 one kernel on the f32 matrix cores, 2 the implicit-GEMM kernel with the fused epilogue, 0 separate
        kernels (include/vpk.h)."""
        self.rt.check(self.rt.lib.vpk_cnn_set_fusion(self.rt.h, int(on)))

    def set_algorithm(self, mode):
        """conv2..5 and fc6: 4 (default) = direct convolutions / weight stream on SCALED fp16 PAIRS of the f32 operands (three exact
        products per f32 product), 2 = conv2 and fc6 on exact bf16 triples (six products) + conv3..5 Winograd F(2 x 2, 3 x 3) on the
        f32 matrix cores (round 5's first default), 3 = as 2 with conv3 and conv5 on triples too, 1 = Winograd everywhere
        (conv2: F(2 x 2, 5 x 5)), 0 = direct implicit GEMM on the f32 matrix cores (include/vpk.h)."""
        self.rt.check(self.rt.lib.vpk_cnn_set_algorithm(self.rt.h, int(mode)))

    def set_precision(self, mode):
        """conv2..5 under set_algorithm(0): 0 = f32-input matrix instructions; 1 = implicit GEMM on exact bf16 pieces (include/vpk.h).
        Mode 1 overrides the algorithm setting."""
        self.rt.check(self.rt.lib.vpk_cnn_set_precision(self.rt.h, int(mode)))

    RANGE_LAYERS = ["conv2", "conv3", "conv4", "conv5", "fc6", "fc7"]      # the layers whose INPUT is a scaled fp16 pair

    def activation_scales(self):
        """The six powers of two the inputs of conv2..5, fc6, fc7 are multiplied by under set_algorithm(4) (calibrated at load)."""
        sc = (ctypes.c_float * 6)()
        self.rt.check(self.rt.lib.vpk_cnn_get_activation_scales(self.rt.h, sc))
        return np.array([float(x) for x in sc], dtype=np.float32)

    def set_activation_scales(self, scales):
        sc = (ctypes.c_float * 6)(*[float(x) for x in scales])
        self.rt.check(self.rt.lib.vpk_cnn_set_activation_scales(self.rt.h, sc))

    def calibrate(self, rasters=None):
        """Set the activation scales from the blob maxima of `rasters` (n x 500 x 500 uint8, host or device) -- None: the built-in
        set of vpk_cnn_load (include/vpk.h: vpk_cnn_calibrate)."""
        rt = self.rt
        if rasters is None:
            rt.check(rt.lib.vpk_cnn_calibrate(rt.h, None, 0))
            return
        t = rt.torch
        if not isinstance(rasters, t.Tensor):
            rasters = t.from_numpy(np.ascontiguousarray(rasters, dtype=np.uint8).reshape(-1, 500, 500))
        with rt.on_stream():
            d = rasters.to(rt.tdev).contiguous()
        rt.synchronize()
        rt.check(rt.lib.vpk_cnn_calibrate(rt.h, rt.ptr(d), int(d.shape[0])))

    def range_flags(self):
        """Waits for the handle's stream; the range word of every forward since the last call (and clears it): 0 = every scaled
        activation stayed inside fp16's range.  Does not raise."""
        w = ctypes.c_uint32(0)
        rc = self.rt.lib.vpk_cnn_range_flags(self.rt.h, ctypes.byref(w))
        if rc not in (0, -6):
            self.rt.check(rc)
        return int(w.value)

    def check_range(self):
        """Raise VpkRangeError if a forward since the last check clamped an activation (include/vpk.h: vpk_cnn_range_flags)."""
        w = ctypes.c_uint32(0)
        rc = self.rt.lib.vpk_cnn_range_flags(self.rt.h, ctypes.byref(w))
        if rc == -6:
            raise VpkRangeError("libvpk error -6: %s" % self.rt.lib.vpk_last_error(self.rt.h).decode(), int(w.value))
        self.rt.check(rc)

    def set_profiling(self, on=True):
        self.rt.check(self.rt.lib.vpk_cnn_set_profiling(self.rt.h, int(bool(on))))

    def last_layer_ms(self):
        ms = (ctypes.c_float * 13)()
        self.rt.check(self.rt.lib.vpk_cnn_last_layer_ms(self.rt.h, ms))
        return dict(zip(self.LAYER_NAMES, [float(x) for x in ms]))

    def mean_layer_ms(self):
        """({layer: ms averaged over the profiled passes since set_profiling(True)}, number of passes)."""
        ms = (ctypes.c_float * 13)()
        n = ctypes.c_int(0)
        self.rt.check(self.rt.lib.vpk_cnn_mean_layer_ms(self.rt.h, ms, ctypes.byref(n)))
        return dict(zip(self.LAYER_NAMES, [float(x) for x in ms])), int(n.value)

    def forward_batch(self, sphere_u8, mean_arr=None):
        return self.forward(sphere_u8)

    def forward_single(self, image, mean_arr=None):
        return self.forward(np.asarray(image)[None])[0]

    def forward(self, sphere_u8, tap=None):
        rt = self.rt
        sphere_u8 = np.ascontiguousarray(sphere_u8, dtype=np.uint8).reshape(-1, 500, 500)
        with rt.on_stream():
            d = rt.torch.from_numpy(sphere_u8).to(rt.tdev)
        res = self.forward_device(d, tap)
        self.check_range()          # (waits for the stream) a clamped activation is an error here, never a silently wrong map
        if tap is None:
            return res.cpu().numpy()
        return res[0].cpu().numpy(), res[1].cpu().numpy()


class LazyNet(object):
    """What evaluation.init_caffe returns: weights are known at construction, the mean blob only at
    the first forward (the reference passes it to caffe_forward, evaluation.py:34-35).  The HIP net is
    (re)built when the mean changes, because `image - mean` is fused into conv1's input load."""

    def __init__(self, weights, device=0, mean=None):
        self.weights = weights
        self.device = device
        self._mean = None
        self._net = None
        if mean is not None:
            self._bind(mean)

    def _bind(self, mean_arr):
        mean = np.ascontiguousarray(np.asarray(mean_arr, dtype=np.float32).reshape(500, 500))
        if self._net is None or not np.array_equal(mean, self._mean):
            self._net = Net(self.weights, mean, device=self.device)
            self._mean = mean
        return self._net

    def forward_batch(self, sphere_u8, mean_arr=None):
        net = self._bind(mean_arr) if mean_arr is not None else self._net
        if net is None:
            raise ValueError("no mean blob bound: pass mean_arr")
        return net.forward(sphere_u8)

    def forward_single(self, image, mean_arr=None):
        return self.forward_batch(np.asarray(image)[None], mean_arr)[0]


def caffe_forward(net, image, mean_arr=None):
    """evaluation.py:34-38: one 500x500 uint8 raster -> (20,20) float32.  The mean blob was bound
    at load time (it is fused into conv1's input load); mean_arr is accepted for signature parity."""
    if isinstance(net, LazyNet):
        return net.forward_single(image, mean_arr)
    return net.forward(np.asarray(image)[None])[0]
