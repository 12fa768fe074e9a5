"""Per-phase device timing of the EM kernel on the bench's YUD-shape workload (dev tool)."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, em as gem, cnn
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
from vanishing_points_2017_amd import sphere_mapping
scenes = sphere_mapping.attach_rasters(list(synth.config_scenes(2, count=102)))
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0))
resp = net.forward(np.stack([s["sphere_image"] for s in scenes]))
if "--fixture" not in sys.argv:                       # default: the random-weight CNN's response maps (bench.py's `value`)
    for s, r in zip(scenes, resp):
        s["cnn_response"] = r
res = gem.em_batch(scenes, want_trace=True)
rows = []
extra = []
for s, r in zip(scenes, res):
    tr = r["trace"]; it = r["iterations"]
    extra.append((tr[-1, 6], tr[-1, 7], int((tr[:it + 1, 3].astype(int) & 1).sum()), int((tr[:it + 1, 3].astype(int) & 4).sum() // 4)))
    rows.append((tr[-1, 2], s["lp"].shape[0], it, r["vp"].shape[0] if r["vp"] is not None else 0, tr[-1, 0], tr[-1, 1],
                 tr[:it + 1, 4].sum(), tr[:it + 1, 5].sum(), tr[:it + 1, 6].sum(), tr[:it + 1, 7].sum(), tr[:it + 1, 0].mean()))
order = np.argsort([-r[0] for r in rows])
print("smoother staging+reduce_us / main_loop_us and split/merge events for the slowest:", [tuple(np.round(extra[i], 0)) for i in order[:4]])
rows.sort(reverse=True)
print("total_us   N  iters Mfinal | pairwise  setup_rest | estep   smooth   mstep   iter_total | M_mean")
for r in rows[:8] + rows[50:52]:
    print("%8.0f %4d %4d %4d | %8.0f %8.0f | %7.0f %8.0f %7.0f %9.0f | %5.1f" % r)
tot = np.array([r[0] for r in rows])
print("sum of image times %.1f ms, max %.1f ms, mean %.2f ms" % (tot.sum() / 1e3, tot.max() / 1e3, tot.mean() / 1e3))
# per-iteration detail of the slowest image
i = int(np.argmax([r["trace"][-1, 2] if r.get("trace") is not None else 0 for r in res]))
tr = res[i]["trace"]; it = res[i]["iterations"]
print("slowest image: N=%d iterations=%d" % (scenes[i]["lp"].shape[0], it))
print("iter  M  events  estep smooth mstep total other | split: select cluster fit | merge")
for k in range(0, min(it + 1, 45)):
    print("%3d %3d %5d %7.0f %6.0f %6.0f %6.0f %6.0f | %6.0f %6.0f %6.0f | %6.0f" % (
        k, tr[k, 0], tr[k, 3], tr[k, 4], tr[k, 5], tr[k, 6], tr[k, 7], tr[k, 7] - tr[k, 4] - tr[k, 5] - tr[k, 6],
        tr[k, 8], tr[k, 9], tr[k, 10], tr[k, 11]))
# where the time of the whole batch goes (sum over images, ms)
tot = {k: 0.0 for k in ("pairwise", "setup", "estep", "smooth", "mstep", "split_select", "split_cluster", "split_fit", "merge", "total")}
for s_, r in zip(scenes, res):
    tr = r["trace"]; it = r["iterations"]
    tot["pairwise"] += tr[-1, 0]; tot["setup"] += tr[-1, 1]; tot["total"] += tr[-1, 2]
    for name, col in (("estep", 4), ("smooth", 5), ("mstep", 6), ("split_select", 8), ("split_cluster", 9), ("split_fit", 10), ("merge", 11)):
        tot[name] += tr[:it + 1, col].sum()
print("batch totals (ms):", {k: round(v / 1e3, 1) for k, v in tot.items()})
print("E-step split over the batch (ms): prior %.1f, lines %.1f" % (sum(r["trace"][-1, 8] for r in res) / 1e3, sum(r["trace"][-1, 9] for r in res) / 1e3))
