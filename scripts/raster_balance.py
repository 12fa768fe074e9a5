"""How evenly the coverage kernel's (edge, share) items load its threads (dev tool; host build of the raster arithmetic)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests/hostsim")
import simlib
from vanishing_points_2017_amd import synth
lib = simlib.raster_lib()
lib.sim_polygon_balance.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
rows = []
for s in synth.config_scenes(2, count=4, raster=None):
    for l in s["l"][:60]:
        l = np.ascontiguousarray(l, dtype=np.float64); out = np.zeros(8)
        if lib.sim_polygon_balance(l.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 500, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == 0:
            rows.append(out.copy())
r = np.array(rows)
print("polygons %d: vertices mean %.0f (simplified points %.0f), K mean %.1f, cells mean %.0f" % (len(r), r[:, 0].mean(), r[:, 6].mean(), r[:, 1].mean(), r[:, 2].mean()))
ideal = r[:, 2] / 512
print("cells per thread if even: %.1f; slowest wave's sum of per-round maxima (setup = 8 cells): mean %.0f, the largest single item mean %.0f max %.0f" % (ideal.mean(), r[:, 3].mean(), r[:, 5].mean(), r[:, 5].max()))
print("ratio slowest wave / even share: %.1f" % (r[:, 3].mean() / ideal.mean()))
