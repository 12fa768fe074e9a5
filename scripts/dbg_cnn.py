import sys, numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
x = torch.randint(0, 60, (2, 500, 500), dtype=torch.uint8, device=rt.tdev)
net.forward_device(x); rt.synchronize(); print("ok")
