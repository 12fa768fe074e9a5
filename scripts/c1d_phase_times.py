"""Device-clock breakdown of conv1_direct_kernel's tile phases (dev tool).  Needs a library built with -DC1D_TIME:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DC1D_TIME -c vanishing_points_2017_amd/csrc/vpk_cnn.hip -o /tmp/cnn_t.o
and linked in place of csrc/_obj/vpk_cnn.o (the timing build exports vpk_dbg_c1d)."""
import sys, ctypes, numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn, _lib
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
net.set_profiling(True)
x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt.tdev)
for _ in range(5):
    net.forward_device(x)
rt.synchronize()
print("conv1 ms", net.last_layer_ms()["conv1"])
lib = ctypes.CDLL(_lib.SO_PATH)
buf = np.zeros(256 * 8 * 8, dtype=np.int64)
rc = lib.vpk_dbg_c1d(buf.ctypes.data_as(ctypes.c_void_p))
b = buf.reshape(256, 8, 8).astype(np.float64)
names = ["A(mfma+lrn)", "barA", "B(mfma+pool)", "barB", "Cs write", "barC", "raw read", "barR"]
tiles = 17136 / 256.0
print("per tile cycles, mean over waves; per-wave means (w0..w7)")
for i, n in enumerate(names):
    print("%-14s %8.0f   " % (n, b[:, :, i].mean() / tiles), " ".join("%6.0f" % (b[:, w, i].mean() / tiles) for w in range(8)))
print("total", b.sum(axis=2).mean() / tiles)
