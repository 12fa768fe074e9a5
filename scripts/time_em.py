"""Quick EM-only timing on the GPU: YUD-shape batch and a stress subset (dev tool)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, em as gem, _lib
from vanishing_points_2017_amd.runtime import get_runtime

rt = get_runtime(0)
print(rt.handle.device_info())

def run(scenes, reps, **kw):
    p = gem._params(kw)
    d = gem.upload_batch(rt, scenes)
    l0 = d["l"].clone()
    ts = []
    for r in range(reps + 1):
        d["l"].copy_(l0)
        rt.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        with rt.on_stream():
            e0.record()
            out = gem.em_batch_device(rt, d["offsets"], d["l"], d["lp"], d["cnn"], d["sphere"], d["init_vp"], p)
            e1.record()
        rt.synchronize()
        ts.append(e0.elapsed_time(e1))
    it = out["iterations"].cpu().numpy()
    print("  times ms:", ["%.2f" % t for t in ts], "iters mean %.1f max %d" % (it.mean(), it.max()),
          "status", np.bincount(out["status"].cpu().numpy()))
    return min(ts[1:])

t0 = time.time()
scenes = list(synth.config_scenes(2))
print("yud gen %.1fs, N mean %.0f" % (time.time() - t0, np.mean([s["lp"].shape[0] for s in scenes])))
t = run(scenes, 3)
print("YUD-shape 102 images: %.2f ms -> %.0f img/s" % (t, 102 / t * 1e3))

base = [synth.make_scene(5000 + i, 1000, 8) for i in range(8)]
for i, s in enumerate(base):
    s["init_vp"] = synth.stress_init_vps(5000 + i)
for nb in (64, 512):
    sc = [base[i % 8] for i in range(nb)]
    t = run(sc, 2, num_iter=50, do_split=False, do_merge=False, final_convergence=-1)
    n, m, I = 1000, 8, 53
    bem = 8 * n * n * (I + 1) + I * (64 * n + 16 * m * n)
    print("stress %d images: %.2f ms -> %.1f img/s, %.2f TB/s algorithmic" % (nb, t, nb / t * 1e3, bem * nb / t * 1e3 / 1e12))

# phase breakdown from the device-side trace (one stress image, one YUD image)
for label, sc, kw in (("stress", [base[0]], dict(num_iter=50, do_split=False, do_merge=False, final_convergence=-1)),
                      ("yud", scenes[:8], {})):
    res = gem.em_batch(sc, want_trace=True, **kw)
    for r in res[:3]:
        tr = r["trace"]
        it = r["iterations"]
        print(label, "N", r["l"].shape[0], "iters", it, "M", r["vp"].shape[0], "setup us (pairwise, rest, total):", tr[-1, :3].round(0))
        print("   per-iter mean us: estep %.1f smooth %.1f mstep %.1f iter %.1f | last-iter total %.1f" % (
            tr[:it, 4].mean() if it else 0, tr[:it, 5].mean() if it else 0, tr[:it, 6].mean() if it else 0,
            tr[:it, 7].mean() if it else 0, tr[it, 7]))
