"""HBM traffic of the EM kernel at the stress shape, by part (dev tool; run under rocprofv3 --pmc by scripts/pmc_em_traffic.sh):
256 images x 1000 lines x 8 VPs, `num_iter` forced iterations (1: the set-up dominates), smoother mode 0 / 1 (1 = the row-by-row
pair pass and the other round-5 forms).      python3 scripts/em_traffic_probe.py <num_iter> <mode>"""
import sys
sys.path.insert(0, ".")
from vanishing_points_2017_amd import synth, em as gem
from vanishing_points_2017_amd.runtime import get_runtime
num_iter, mode = int(sys.argv[1]), int(sys.argv[2])
rt = get_runtime(0)
rt.handle.em_set_smoother(mode)
kw = dict(num_iter=num_iter, do_split=False, do_merge=False, final_convergence=-1)
base = []
for i in range(16):
    s = synth.make_scene(5000 + i, 1000, 8)
    s["init_vp"] = synth.stress_init_vps(5000 + i)
    base.append(s)
scenes = [base[i % 16] for i in range(256)]
p = gem._params(kw)
d = gem.upload_batch(rt, scenes)
for rep in range(2):
    l = d["l"].clone()
    with rt.on_stream():
        out = gem.em_batch_device(rt, d["offsets"], l, d["lp"], d["cnn"], d["sphere"], d["init_vp"], p)
    rt.synchronize()
print("iterations", int(out["iterations"].max()), "status ok", int((out["status"] == 0).sum()))
