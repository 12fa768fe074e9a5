"""CPU oracle for the CNN: torch-CPU fp32 functional restatement of cnn/deploy.prototxt.

TEST INFRASTRUCTURE (see oracle/em_numpy.py for the rules).  PARITY UNPINNED at the Caffe
boundary: the reference runs this net inside third-party BVLC Caffe 1.0-RC5 (README.md:5;
call sites evaluation.py:17-38), which is not installable here, the trained weights are a
download (README.md:24-25) and the reference has no tests -- so there are no golden vectors
for the CNN.  This restatement follows deploy.prototxt layer by layer with Caffe's published
layer semantics (cross-correlation, OIHW, contiguous channel groups, ceil-mode pooling with
clipped windows, LRN x*(1+alpha/n*sum x^2)^-beta, InnerProduct over C*H*W, sigmoid) and is
checked for shapes (123 -> 61 -> 30 -> 15) only.  Weights are seeded synthetic.
"""
import numpy as np
import torch
import torch.nn.functional as F

LAYERS = ["conv1", "conv2", "conv3", "conv4", "conv5", "fc6", "fc7", "fc8"]
TAPS = ["conv1", "pool1", "conv2", "pool2", "conv3", "conv4", "conv5", "pool5", "fc6", "fc7", "fc8"]


def forward(weights, mean, sphere_u8, want_taps=False, threads=None, dtype=np.float32):
    """weights: {name: (W, b)} in Caffe layout; mean: (500,500) f32; sphere_u8: (B,500,500) uint8.
    Returns (B,20,20) [, taps dict] in `dtype` (float32 like Caffe; float64 = the exact-arithmetic yardstick the
    two GPU precisions are measured against)."""
    if threads:
        torch.set_num_threads(threads)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=dtype)))
    w = {k: (t(v[0]), t(v[1])) for k, v in weights.items()}
    x = t(sphere_u8) - t(mean)[None]                                            # evaluation.py:35
    x = x[:, None]
    taps = {}
    with torch.no_grad():
        x = F.relu(F.conv2d(x, w["conv1"][0], w["conv1"][1], stride=4)); taps["conv1"] = x     # :9-33
        x = F.local_response_norm(x, 5, alpha=1e-4, beta=0.75, k=1.0)                        # :34-44
        x = F.max_pool2d(x, 3, 2, ceil_mode=True); taps["pool1"] = x                          # :45-55
        x = F.relu(F.conv2d(x, w["conv2"][0], w["conv2"][1], padding=2, groups=2)); taps["conv2"] = x
        x = F.local_response_norm(x, 5, alpha=1e-4, beta=0.75, k=1.0)
        x = F.max_pool2d(x, 3, 2, ceil_mode=True); taps["pool2"] = x
        x = F.relu(F.conv2d(x, w["conv3"][0], w["conv3"][1], padding=1)); taps["conv3"] = x
        x = F.relu(F.conv2d(x, w["conv4"][0], w["conv4"][1], padding=1, groups=2)); taps["conv4"] = x
        x = F.relu(F.conv2d(x, w["conv5"][0], w["conv5"][1], padding=1, groups=2)); taps["conv5"] = x
        x = F.max_pool2d(x, 3, 2, ceil_mode=True); taps["pool5"] = x
        x = x.flatten(1)
        x = F.relu(F.linear(x, w["fc6"][0], w["fc6"][1])); taps["fc6"] = x                   # drop6 = identity
        x = F.relu(F.linear(x, w["fc7"][0], w["fc7"][1])); taps["fc7"] = x
        x = F.linear(x, w["fc8"][0], w["fc8"][1]); taps["fc8"] = x
        out = torch.sigmoid(x).reshape(-1, 20, 20)                                           # :283-304
    out = out.numpy()
    if want_taps:
        return out, {k: v.numpy() for k, v in taps.items()}
    return out
