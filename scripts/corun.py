"""CNN forward time while an EM batch of b images runs on another stream (dev tool)."""
import sys, numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn, em as gem, synth
from vanishing_points_2017_amd.runtime import get_runtime
rt_c = get_runtime(0, "cnn"); rt_e = get_runtime(0, "em0")
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), device=0, runtime=rt_c)
scenes = list(synth.config_scenes(2, count=102))
x = torch.from_numpy(np.stack([s["sphere_image"] for s in scenes])).to(rt_c.tdev)
resp = net.forward_device(x); rt_c.synchronize()
r = resp.cpu().numpy()
for s, q in zip(scenes, r): s["cnn_response"] = q
params = gem._params({})
def cnn_ms(reps=3):
    with rt_c.on_stream():
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): net.forward_device(x)
        e1.record()
    return e0, e1
e0, e1 = cnn_ms(); rt_c.synchronize(); print("CNN alone: %.2f ms" % (e0.elapsed_time(e1) / 3))
for b in (8, 16, 32, 64, 102):
    # the b slowest-ish images: take scenes with most lines
    order = np.argsort([-s["lp"].shape[0] for s in scenes])[:b]
    d = gem.upload_batch(rt_e, [scenes[i] for i in order]); rt_e.synchronize()
    gem.em_batch_device(rt_e, d["offsets"], d["l"].clone(), d["lp"], d["cnn"], d["sphere"], None, params); rt_e.synchronize()
    with rt_e.on_stream():
        a0 = torch.cuda.Event(enable_timing=True); a1 = torch.cuda.Event(enable_timing=True)
        a0.record()
        gem.em_batch_device(rt_e, d["offsets"], d["l"].clone(), d["lp"], d["cnn"], d["sphere"], None, params)
        a1.record()
    e0, e1 = cnn_ms(2)
    rt_c.synchronize(); rt_e.synchronize()
    print("EM batch %3d (%.1f ms)  CNN co-running: %.2f ms per forward" % (b, a0.elapsed_time(a1), e0.elapsed_time(e1) / 2))
