"""Shader-clock breakdown of conv5x5_winograd_kernel's phases (dev tool).  Needs a library built with -DW5_TIME:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DW5_TIME -c vanishing_points_2017_amd/csrc/vpk_cnn.hip -o /tmp/cnn_t.o
    hipcc --offload-arch=gfx950 -shared -fPIC <the other objects of csrc/_obj> /tmp/cnn_t.o -o scripts/libvpk_w5time.so
(scripts/build_w5time.sh does both)."""
import sys, ctypes, os, numpy as np
sys.path.insert(0, ".")
from vanishing_points_2017_amd import _lib
_lib.SO_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libvpk_w5time.so")
import torch
from vanishing_points_2017_amd import cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
net.set_profiling(True)
x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt.tdev)
for _ in range(3):
    net.forward_device(x)
rt.synchronize()
print("conv2 ms", net.last_layer_ms()["conv2"])
lib = ctypes.CDLL(_lib.SO_PATH)
buf = np.zeros(256 * 12 * 8, dtype=np.int64)
lib.vpk_dbg_w5(buf.ctypes.data_as(ctypes.c_void_p))
b = buf.reshape(256, 12, 8).astype(np.float64)
names = ["transform", "operands + MFMAs", "chunk-end wait + barrier", "prologue", "epilogue: write", "epilogue: barriers", "epilogue: gather",
         "tile setup"]
tot = b.sum(axis=2).mean()
print("per wave, mean over workgroups: cycles (share); per-wave means w0..w11")
for i, n in enumerate(names):
    print("%-26s %10.0f  %5.1f%%   " % (n, b[:, :, i].mean(), 100 * b[:, :, i].mean() / tot), " ".join("%7.0f" % b[:, w, i].mean() for w in range(12)))
print("total", tot)

# the 3 x 3 kernel (its last launch of the pass: conv5, 24 chunks of 8 channels; same slots)
buf3 = np.zeros(256 * 8 * 8, dtype=np.int64)
lib.vpk_dbg_w3(buf3.ctypes.data_as(ctypes.c_void_p))
b3 = buf3.reshape(256, 8, 8).astype(np.float64)
tot3 = b3.sum(axis=2).mean()
print("conv3x3_winograd_kernel (conv5), ms:", net.last_layer_ms()["conv5"])
for i, n in enumerate(names):
    print("%-26s %10.0f  %5.1f%%   " % (n, b3[:, :, i].mean(), 100 * b3[:, :, i].mean() / tot3), " ".join("%7.0f" % b3[:, w, i].mean() for w in range(8)))
print("total", tot3)
