import sys
sys.path.insert(0, ".")
import numpy as np
from oracle import cnn_torch
from vanishing_points_2017_amd import cnn, synth, sphere_mapping
w = cnn.synthetic_weights(0); mean = cnn.synthetic_mean(0)
sphere = sphere_mapping.raster_batch([s["l"] for s in synth.config_scenes(2, count=102)])
ref, taps = cnn_torch.forward(w, mean, sphere, want_taps=True)
net = cnn.Net(w, mean)
base = None
for it in range(12):
    out = net.forward(sphere)
    e = np.abs(out - ref).max()
    if base is None: base = out.copy()
    same = np.array_equal(out, base)
    bad = np.argwhere(np.abs(out - ref).max(axis=(1, 2)) > 2e-5).ravel()
    print(it, "err %.2e" % e, "same bits as first" if same else "DIFFERENT", bad[:10])
for tap in (1, 2, 3, 4, 5, 6, 7, 8):
    outs = []
    for it in range(4):
        o, g = net.forward(sphere, tap=tap)
        outs.append(g.copy())
    want = taps[cnn_torch.TAPS[tap]].reshape(outs[0].shape)
    print(cnn_torch.TAPS[tap], ["%.2e" % (np.abs(g - want).max() / np.abs(want).max()) for g in outs], [np.array_equal(outs[0], g) for g in outs])
