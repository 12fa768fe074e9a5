"""CU-time of the EM per YUD-shape batch at different concurrencies (dev tool): the sum of the per-image device
times (one workgroup = one CU each) for `copies` copies of the 102-image batch run by `wgs` workgroups."""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, em as gem
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
from vanishing_points_2017_amd import sphere_mapping
scenes = sphere_mapping.attach_rasters(list(synth.config_scenes(2, count=102)))
for copies, wgs in ((1, 34), (1, 102), (3, 102), (3, 256), (5, 256)):
    rt.handle.em_set_workgroups(wgs)
    sc = scenes * copies
    p = gem._params({})
    d = gem.upload_batch(rt, sc)
    l0 = d["l"].clone()
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
        tr = out["trace"].cpu().numpy()
        cu_s = tr[:, -1, 2].sum() * 1e-6
        ms = e0.elapsed_time(e1)
        if best is None or ms < best[0]:
            best = (ms, cu_s)
    print("copies %d wgs %3d: kernel %.2f ms, CU-time %.3f CU-s per batch (%.3f total), balanced time %.2f ms per batch" % (
        copies, wgs, best[0], best[1] / copies, best[1], best[1] / copies / min(wgs, len(sc)) * 1e3))
