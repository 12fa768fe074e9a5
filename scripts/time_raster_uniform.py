"""blend tail experiment (dev tool): same total lines, every image the same count vs the bench's mix."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, sphere_mapping
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
scenes = list(synth.config_scenes(2, count=102, raster=None))
allc = np.concatenate([s["l"] for s in scenes])
for name, counts in (("mixed", [s["l"].shape[0] for s in scenes]), ("uniform", [len(allc) // 102] * 102)):
    offsets = np.zeros(len(counts) + 1, dtype=np.int64); offsets[1:] = np.cumsum(counts)
    cat = torch.from_numpy(np.ascontiguousarray(allc[:offsets[-1]])).to(rt.tdev)
    for rep in range(3):
        rt.synchronize(); t0 = time.perf_counter()
        out = sphere_mapping.raster_batch_device(rt, cat, offsets, 500, 0.1)
        rt.synchronize(); dt = time.perf_counter() - t0
    print(name, "max lines", max(counts), "%.2f ms" % (dt * 1e3))
