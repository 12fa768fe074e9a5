"""What the GPU rasteriser's differences from the reference's matplotlib raster do downstream (dev tool):
find_initial_vps on both rasters, and the full EM on both."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden_cases
from golden_util import load
from vanishing_points_2017_amd import kernels, sphere_mapping, em as gem

for name in golden_cases():
    g = load(name)
    if "init_vp" in g or name.startswith("hard1row"):
        continue
    ras = sphere_mapping.sphere_line_plot(g["l"].copy(), 500, alpha=0.1)
    v_ref, _ = kernels.init_vps(g["cnn_response"], g["sphere_image"])
    v_gpu, _ = kernels.init_vps(g["cnn_response"], ras)
    line = "%-16s v0 %2d/%2d" % (name, v_gpu.shape[0], v_ref.shape[0])
    if v_gpu.shape == v_ref.shape:
        ang = np.degrees(np.arccos(np.clip(np.abs((v_ref * v_gpu).sum(1)), 0, 1)))
        line += " angle max %.3f deg mean %.3f, >1deg: %d" % (ang.max(), ang.mean(), (ang > 1).sum())
    kw = {k[3:]: g[k].item() for k in g if k.startswith("kw_")}
    a = gem.em_batch([{"l": g["l"].copy(), "lp": g["lp"], "cnn_response": g["cnn_response"], "sphere_image": g["sphere_image"]}], **kw)[0]
    b = gem.em_batch([{"l": g["l"].copy(), "lp": g["lp"], "cnn_response": g["cnn_response"], "sphere_image": ras}], **kw)[0]
    if a["vp"] is not None and b["vp"] is not None:
        # match each strong VP of the reference-raster run to the nearest VP of the GPU-raster run
        strong = np.argsort(a["counts"])[::-1][:3]
        d = np.degrees(np.arccos(np.clip(np.abs(a["vp"][strong] @ b["vp"].T), 0, 1))).min(1)
        # same partition of the lines?  (labels differ, so compare co-membership through the matched VPs)
        match = np.degrees(np.arccos(np.clip(np.abs(a["vp"] @ b["vp"].T), 0, 1))).argmin(1)
        mapped = np.where(a["vp_assoc"] >= 0, match[np.maximum(a["vp_assoc"], 0)], -1)
        agree = (mapped == b["vp_assoc"]).mean()
        line += " | EM: M %d/%d iters %d/%d top-3 VP angle %s deg, assignment agreement %.3f" % (
            b["vp"].shape[0], a["vp"].shape[0], b["iterations"], a["iterations"], np.round(d, 4), agree)
    print(line, flush=True)
