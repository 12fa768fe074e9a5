"""Host time of one vpk_pipeline_step call (dev tool): enqueue on idle streams, with / without per-layer profiling events."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import cnn, em as gem, pipeline, synth
from vanishing_points_2017_amd.runtime import get_runtime
rt_cnn, rt_em = get_runtime(0, "cnn"), get_runtime(0, "em0")
from vanishing_points_2017_amd import sphere_mapping
scenes = sphere_mapping.attach_rasters(list(synth.config_scenes(2, count=102)))
net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), runtime=rt_cnn)
params = gem._params({})
d = gem.upload_batch(rt_em, scenes)
rt_em.handle.em_set_workgroups(30)
ring = [pipeline.Step(rt_cnn, rt_em, d, params, l_in=d["l"].clone(), timing=False) for _ in range(4)]
for prof in (True, False):
    net.set_profiling(prof)
    for st in ring:
        st.enqueue()
    torch.cuda.synchronize()
    for n in (1, 2, 4, 8, 16):
        t0 = time.perf_counter()
        for k in range(n):
            ring[k % 4].enqueue()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("profiling %-5s: %2d steps enqueued in %.3f ms (%.3f ms per step); GPU done after %.1f ms" % (prof, n, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / n, (t2 - t0) * 1e3))
# the pieces
torch.cuda.synchronize()
t0 = time.perf_counter(); resp = net.forward_device(d["sphere"]); t1 = time.perf_counter()
torch.cuda.synchronize()
l = d["l"].clone(); torch.cuda.synchronize()
t2 = time.perf_counter(); out = gem.em_batch_device(rt_em, d["offsets"], l, d["lp"], resp.reshape(-1, 400), d["sphere"], None, params); t3 = time.perf_counter()
torch.cuda.synchronize()
print("vpk_cnn_forward (python wrapper) %.3f ms; em_batch_device %.3f ms" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3))
