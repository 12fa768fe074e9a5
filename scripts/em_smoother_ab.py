"""Row-sliced smoother against the round-2 kernels on the bench's YUD-shape batch (dev tool): per-phase device time
summed over the 102 images, the slowest image, and the kernel time, for vpk_em_set_smoother(0) and (1)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, em as gem, cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
from vanishing_points_2017_amd import sphere_mapping
scenes = sphere_mapping.attach_rasters(list(synth.config_scenes(2, count=102)))
if "--cnn" in sys.argv:     # the bench's situation: the random-weight CNN's response maps as the prior
    net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
    resp = net.forward(np.stack([s["sphere_image"] for s in scenes]))
    for s, r in zip(scenes, resp):
        s["cnn_response"] = r
p = gem._params({})
d = gem.upload_batch(rt, scenes)
l0 = d["l"].clone()
for mode in (0, 2, 0, 2):
    rt.handle.em_set_smoother(mode)
    best = None
    for rep in range(3):
        d["l"].copy_(l0)
        rt.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        with rt.on_stream():
            e0.record()
            out = gem.em_batch_device(rt, d["offsets"], d["l"], d["lp"], d["cnn"], d["sphere"], None, p, want_trace=True)
            e1.record()
        rt.synchronize()
        ms = e0.elapsed_time(e1)
        if best is None or ms < best[0]:
            best = (ms, out["trace"].cpu().numpy(), out["iterations"].cpu().numpy())
    ms, tr, it = best
    tot = {k: 0.0 for k in ("estep", "smooth", "mstep", "total")}
    for b in range(len(scenes)):
        tot["total"] += tr[b, -1, 2]
        for name, col in (("estep", 4), ("smooth", 5), ("mstep", 6)):
            tot[name] += tr[b, :it[b] + 1, col].sum()
    slow = int(np.argmax(tr[:, -1, 2]))
    print("smoother %d: kernel %.2f ms | batch sums (ms): %s | slowest image %.2f ms (N=%d, %d iterations; smooth %.0f us/iter; "
          "smoother split of that image, ms: rows %.2f, staging + reduction + wait %.2f)" % (
        mode, ms, {k: round(float(v) / 1e3, 1) for k, v in tot.items()}, tr[slow, -1, 2] / 1e3,
        int(d["offsets"][slow + 1] - d["offsets"][slow]), it[slow], tr[slow, :it[slow] + 1, 5].mean(),
        tr[slow, -1, 7] / 1e3, tr[slow, -1, 6] / 1e3))
rt.handle.em_set_smoother(0)
