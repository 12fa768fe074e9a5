"""Error of the HIP net against the same net evaluated in float64 (oracle/cnn_torch.py), per tap, for a list of
(fusion, precision, algorithm) settings: python scripts/cnn_accuracy.py 1,0,0 3,0,1 3,0,2   (dev tool, GPU box)"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import cnn_torch
from vanishing_points_2017_amd import cnn, sphere_mapping, synth
w, mean = cnn.synthetic_weights(0), cnn.synthetic_mean(0)
net = cnn.Net(w, mean)
B = 3
sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=B, start=10)])
ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True, dtype=np.float64)
for spec in sys.argv[1:]:
    f, p, a = [int(x) for x in spec.split(",")]
    net.set_fusion(f); net.set_precision(p); net.set_algorithm(a)
    row = []
    for tap in (1, 2, 3, 4, 5, 6, 8):
        want = taps[cnn_torch.TAPS[tap]]
        out, got = net.forward(sphere, tap=tap)
        row.append("%s %.2e" % (cnn_torch.TAPS[tap], np.abs(got.reshape(want.shape) - want).max() / np.abs(want).max()))
    print(spec, "|", "  ".join(row), "| out %.2e" % np.abs(out - ref).max())
