"""One image of a config through the HIP EM with its trace, next to the CPU oracle's (dev tool):  python scripts/trace_one.py <config> <image>"""
import sys
import numpy as np
sys.path.insert(0, ".")
from vanishing_points_2017_amd import em as gem, synth
from oracle import em_numpy
cfg, idx = int(sys.argv[1]), int(sys.argv[2])
sc = next(synth.config_scenes(cfg, count=1, start=idx))
r = gem.em_batch([sc], want_trace=True)[0]
tr = {}
o = em_numpy.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(), sphere_image=sc["sphere_image"], trace=tr)
t = r["trace"]
print("gpu: status %d iterations %d M %d | oracle: iterations %d M %d" % (r["status"], r["iterations"], r["vp"].shape[0], o["iterations"], o["vp"].shape[0]))
print("iteration: M gpu/oracle, max_err gpu/oracle, M_end gpu/oracle, events gpu/oracle")
for it, row in enumerate(tr["iters"]):
    print("%3d: %2d %2d  %.15e %.15e  %2d %2d  %d %d" % (it, t[it, 0], row[0], t[it, 1], row[1], t[it, 2], row[2], t[it, 3], row[3]))
print("finalisation (gpu): M after merge %d, after the hard M-step %d, after winner selection %d" % tuple(t[-1, 3:6]))
if "final" in tr:
    print("finalisation (oracle):", tr["final"])
print("sigma gpu   ", np.sort(r["sigma"]))
print("sigma oracle", np.sort(o["sigma"]))
print("counts gpu   ", r["counts"])
print("counts oracle", o["counts"])
