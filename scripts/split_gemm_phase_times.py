"""Device-clock breakdown of conv_gemm_split_kernel's phases (dev tool): stage issue, LDS reads + MFMAs, counted wait,
barrier, epilogue.  Needs a library built with -DSG_TIME (the timing build exports vpk_dbg_sg):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSG_TIME -c vanishing_points_2017_amd/csrc/vpk_cnn.hip -o /tmp/cnn_t.o
linked in place of csrc/_obj/vpk_cnn.o.  The figures are those of the LAST split-kernel launch of a forward pass (conv5)."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
from vanishing_points_2017_amd import _lib, cnn  # noqa: E402
from vanishing_points_2017_amd.runtime import get_runtime  # noqa: E402

rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
net.set_profiling(True)
net.set_precision(1)
x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt.tdev)
for _ in range(3):
    net.forward_device(x)
rt.synchronize()
print({k: round(v, 3) for k, v in net.last_layer_ms().items() if k.startswith("conv")})
lib = ctypes.CDLL(_lib.SO_PATH)
buf = np.zeros(256 * 8 * 8, dtype=np.int64)
lib.vpk_dbg_sg(buf.ctypes.data_as(ctypes.c_void_p))
b = buf.reshape(256, 8, 8).astype(np.float64)
names = ["prologue", "issue", "reads+mfma", "wait_stage", "barrier", "epilogue", "tile setup"]
tot = b.sum(axis=2).mean()
print("cycles per wave (mean over workgroups), share; per-wave means")
for i, n in enumerate(names):
    print("%-12s %9.0f %5.1f%%  " % (n, b[:, :, i].mean(), 100 * b[:, :, i].mean() / tot),
          " ".join("%8.0f" % b[:, w, i].mean() for w in range(8)))
