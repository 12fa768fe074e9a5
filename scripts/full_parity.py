"""GPU EM vs the stored results of the reference (tests/golden/full_c<cfg>.npz), image by image.

    python scripts/full_parity.py <config> [--trace IMAGE]

From the lines alone (the rasters are made by vpk_sphere_raster and hash-compared with the reference's).  Prints every image
that misses the parity bar with its deltas, writes gpurun_out/full_parity_c<cfg>.json and adds the config's failing images
to gpurun_out/parity_failures.json -- the input of oracle/make_instability_certificates.py (build container).
--trace IMAGE also prints the per-iteration (M, max VP change) trajectory of that image on the GPU next to
the CPU oracle's, to locate the iteration where the two runs part."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vanishing_points_2017_amd import em as gem, parity, synth  # noqa: E402


def main(argv):
    cfg = int(argv[0])
    trace_img = int(argv[argv.index("--trace") + 1]) if "--trace" in argv else None
    ref = parity.ReferenceResults(cfg)
    scenes = [next(synth.config_scenes(cfg, count=1, start=int(i))) for i in ref.index]
    differ = [int(i) for i, s in zip(ref.index, scenes) if parity.input_sha(s) != ref.get(i)["input_sha"]]
    print("config %d: %d stored images, %d with different regenerated inputs %s" % (cfg, len(ref), len(differ), differ[:10]))
    res = gem.em_batch(scenes, want_trace=trace_img is not None)
    ras = [int(i) for i, s in zip(ref.index, scenes) if parity.raster_sha(s["sphere_image"]) != ref.get(i)["raster_sha"]]
    print("rasters equal to the reference's: %d of %d %s" % (len(ref) - len(ras), len(ref), ras[:10]))
    comps = {}
    for i, r in zip(ref.index, res):
        if int(i) in differ:
            continue
        c = parity.compare_one(r, ref.get(i))
        comps[int(i)] = c
        if not parity.passes(c):
            g = ref.get(i)
            print("MISS image %d N=%d: status %s iters gpu %s ref %s | M gpu %s ref %s | assoc_diff %s | vp_err %.3g | flags %s | ref events %s"
                  % (i, len(g["vp_assoc"]), c["status"], r.get("iterations"), g["iterations"],
                     None if r["vp"] is None else r["vp"].shape[0], g["vp"].shape[0], c["assoc_diff"], c["vp_err"],
                     r["flags"], g["events"]))
    summ = parity.summarise(comps)
    print(json.dumps(summ))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/full_parity_c%d.json" % cfg, "w") as fh:
        json.dump({"summary": summ, "images": {str(k): {a: (b if not isinstance(b, (np.bool_,)) else bool(b)) for a, b in v.items()}
                                                for k, v in comps.items()}}, fh, default=float)
    fails = {}
    if os.path.isfile("gpurun_out/parity_failures.json"):
        fails = json.load(open("gpurun_out/parity_failures.json"))
    fails[str(cfg)] = summ["failing_images"]
    with open("gpurun_out/parity_failures.json", "w") as fh:
        json.dump(fails, fh)
    if trace_img is not None:
        from oracle import em_numpy
        k = list(ref.index).index(trace_img)
        sc = scenes[k]
        tr = {}
        em_numpy.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                          sphere_image=sc["sphere_image"], trace=tr)
        t = res[k]["trace"]
        print("iteration: M gpu/oracle, max_err gpu/oracle, events gpu/oracle")
        for it, row in enumerate(tr["iters"]):
            print("%3d: %2d %2d  %.12e %.12e  %d %d" % (it, t[it, 0], row[0], t[it, 1], row[1], t[it, 3], row[3]))


if __name__ == "__main__":
    main(sys.argv[1:])
