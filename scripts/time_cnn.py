import sys, numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
net.set_profiling(True)
import os
net.set_fusion(int(os.environ.get("VPK_FUSION", "3")))
net.set_precision(int(os.environ.get("VPK_PRECISION", "0")))
net.set_algorithm(int(os.environ.get("VPK_ALGORITHM", "4")))
args = sys.argv[1:]
passes = 3
if "--passes" in args:
    passes = int(args[args.index("--passes") + 1])
    del args[args.index("--passes"):args.index("--passes") + 2]
for B in ([int(a) for a in args] or [102, 512]):
    x = torch.randint(0, 60, (B, 500, 500), dtype=torch.uint8, device=rt.tdev)
    for _ in range(passes):
        net.forward_device(x)
    rt.synchronize()
    ms = net.last_layer_ms()
    tot = sum(ms.values())
    print("B=%d total %.2f ms (%.0f img/s) " % (B, tot, B / tot * 1e3), {k: round(v, 3) for k, v in ms.items()})
    print("   TF:", {k: round(cnn.Net.LAYER_FLOP[k] * B / (ms[k] * 1e-3) / 1e12, 1) for k in cnn.Net.LAYER_FLOP})
