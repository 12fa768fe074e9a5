"""Time the GPU rasteriser (dev tool): the bench's 102 YUD-shape line sets in one call."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from vanishing_points_2017_amd import synth, sphere_mapping
from vanishing_points_2017_amd.runtime import get_runtime
rt = get_runtime(0)
only = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for cfg, count in ((2, 102), (4, 512)):
    if only and cfg != only:
        continue
    scenes = list(synth.config_scenes(cfg, count=count, raster=None))
    counts = [s["l"].shape[0] for s in scenes]
    offsets = np.zeros(len(counts) + 1, dtype=np.int64); offsets[1:] = np.cumsum(counts)
    cat = torch.from_numpy(np.ascontiguousarray(np.concatenate([s["l"] for s in scenes]))).to(rt.tdev)
    for rep in range(3):
        rt.synchronize(); t0 = time.perf_counter()
        out = sphere_mapping.raster_batch_device(rt, cat, offsets, 500, 0.1)
        rt.synchronize(); dt = time.perf_counter() - t0
    print("config %d: %d images, %d lines: %.2f ms (%.0f images/s, %.2f us per line)" % (cfg, count, offsets[-1], dt * 1e3, count / dt, dt * 1e6 / offsets[-1]))
