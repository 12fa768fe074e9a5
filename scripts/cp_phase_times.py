"""Shader-clock breakdown of conv_pieces_kernel's phases (dev tool): run under scripts/cnn_variant.sh "-DCP_TIME" on the GPU box."""
import sys, ctypes, os, numpy as np
sys.path.insert(0, ".")
from vanishing_points_2017_amd import _lib
import torch
from vanishing_points_2017_amd import cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
net.set_profiling(True)
net.set_algorithm(2)
x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt.tdev)
for _ in range(3):
    net.forward_device(x)
rt.synchronize()
ms = net.last_layer_ms()
lib = ctypes.CDLL(_lib.SO_PATH)
buf = np.zeros(2 * 256 * 8 * 8, dtype=np.int64)
lib.vpk_dbg_cp(buf.ctypes.data_as(ctypes.c_void_p))
names = ["compute (MFMA issue)", "barrier after compute", "add", "load: DMA + ds_read issue", "load: vmcnt wait", "load: lgkmcnt wait",
         "barrier after load", "tile prologue + epilogue"]
for k, layer in enumerate(("conv2", "conv5")):
    b = buf.reshape(2, 256, 8, 8)[k].astype(np.float64)
    tot = b.sum(axis=2).mean()
    print(layer, "ms", ms[layer], "cycles per wave", tot)
    for i, n in enumerate(names):
        print("  %-28s %10.0f  %5.1f%%   " % (n, b[:, :, i].mean(), 100 * b[:, :, i].mean() / tot), " ".join("%8.0f" % b[:, w, i].mean() for w in range(8)))
