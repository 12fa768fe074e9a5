"""CNN forward time while a synthetic kernel occupies a few CUs on another stream (dev tool)."""
import sys, ctypes, numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt_c = get_runtime(0, "cnn")
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), device=0, runtime=rt_c)
x = torch.randint(0, 60, (102, 500, 500), dtype=torch.uint8, device=rt_c.tdev)
net.forward_device(x); rt_c.synchronize()
lib = ctypes.CDLL("scripts/ubench/libspin.so")
lib.spin.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
st = torch.cuda.Stream()
per = 1 << 17   # doubles per workgroup (1 MiB)
buf = torch.zeros(256 * per, dtype=torch.float64, device=rt_c.tdev)
def cnn_ms(reps=2):
    with rt_c.on_stream():
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): net.forward_device(x)
        e1.record()
    return e0, e1
e0, e1 = cnn_ms(); rt_c.synchronize(); print("CNN alone: %.2f ms" % (e0.elapsed_time(e1) / 2))
for mode, iters, name in ((0, 2500000, "alu"), (1, 40000, "rmw L2"), (2, 80000, "read L2")):
    for wgs, lds in ((8, 0), (8, 150 * 1024), (64, 0), (64, 150 * 1024)):
        with torch.cuda.stream(st):
            a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
            a0.record()
            lib.spin(ctypes.c_void_p(st.cuda_stream), wgs, iters, mode, ctypes.c_void_p(buf.data_ptr()), per, lds)
            a1.record()
        e0, e1 = cnn_ms()
        torch.cuda.synchronize()
        print("%-8s wgs=%3d lds=%6d spin %.1f ms   CNN co-running %.2f ms" % (name, wgs, lds, a0.elapsed_time(a1), e0.elapsed_time(e1) / 2))
