import sys, time, os; sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
import numpy as np
from vanishing_points_2017_amd import sphere_mapping, synth
from vanishing_points_2017_amd._lib import VpkError
from oracle import agg_raster
bad = np.array([[-0.26247709, -1.59752339, -74.31417043]])
one = sphere_mapping.raster_batch([bad], size=500, alpha=0.1)[0]
print("wide-gap line alone == oracle:", np.array_equal(one, agg_raster.raster(bad)))
rng = np.random.default_rng(99)
sets = []
for k in range(4):
    l = rng.normal(size=(100, 3))
    if k == 1: l[:, 1] *= 0.01
    if k == 2: l[:, 2] *= 50
    if k == 3: l = synth.make_scene(500 + k, 100, 3, raster=None)["l"]
    sets.append(l)
for rep in range(2):
    try:
        got = sphere_mapping.raster_batch(sets, size=500, alpha=0.1)
        print("batch:", [bool(np.array_equal(g, agg_raster.raster(l))) for g, l in zip(got, sets)])
    except VpkError as e:
        print("batch refused:", str(e)[:150])
for i, l in enumerate(sets):
    try:
        g = sphere_mapping.raster_batch([l], size=500, alpha=0.1)[0]
        print(i, "alone == oracle:", np.array_equal(g, agg_raster.raster(l)))
    except VpkError as e:
        print(i, "alone refused:", str(e)[:120])
