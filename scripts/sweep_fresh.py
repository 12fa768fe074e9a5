"""HIP EM vs the CPU oracle on scenes OUTSIDE the stored reference tables (other seeds of the same generators): config 2
images 102.., config 3 images 103.., config 4 images beyond the 64 stored ones.  The oracle runs in a process pool
(the GPU box has many host cores; the pool is forked before anything touches the GPU).  The scenes are LINES: their rasters
are made by vpk_sphere_raster first (evaluation.py:175) and the oracle is given the same rasters.  Any mismatch is dumped to
gpurun_out/mismatch_fresh/ and listed in gpurun_out/fresh_failures.json -- the input of
oracle/make_instability_certificates.py --from (build container).

    python scripts/sweep_fresh.py [images per config [offset [configs]]]     e.g.  400 200 2,4
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")


def oracle_one(sc):
    from oracle import em_numpy
    try:
        r = em_numpy.expectation_maximisation(sc["l"].copy(), sc["lp"].copy(), sc["cnn_response"].copy(),
                                              sphere_image=sc["sphere_image"])
    except ValueError:
        return {"vp": "valueerror"}
    return {k: r[k] for k in ("vp", "vp_assoc", "iterations")}


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    offset = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 3, 4]
    import json
    pool = mp.get_context("fork").Pool(min(64, os.cpu_count() or 8))
    from vanishing_points_2017_amd import em as gem, sphere_mapping, synth
    os.makedirs("gpurun_out/mismatch_fresh", exist_ok=True)
    total = bad = 0
    failures = {}
    for cfg, start in ((2, 102 + offset), (3, 103 + offset), (4, 2000 + offset)):
        if cfg not in only:
            continue
        scenes = [next(synth.config_scenes(cfg, count=1, start=start + i)) for i in range(per)]
        sphere_mapping.attach_rasters(scenes)
        t0 = time.time()
        refs = pool.map(oracle_one, scenes, chunksize=1)
        to = time.time() - t0
        res = gem.em_batch(scenes)
        worst = 0.0
        for i, (sc, r, ref) in enumerate(zip(scenes, res, refs)):
            total += 1
            if isinstance(ref["vp"], str):
                ok = r["status"] == 2
            elif ref["vp"] is None:
                ok = r["vp"] is None
            else:
                ok = (r["vp"] is not None and r["vp"].shape == ref["vp"].shape and np.array_equal(r["vp_assoc"], ref["vp_assoc"])
                      and np.abs(r["vp"] - ref["vp"]).max() <= 1e-4 and r["iterations"] == ref["iterations"])
                if ok:
                    worst = max(worst, float(np.abs(r["vp"] - ref["vp"]).max()))
            if not ok:
                bad += 1
                failures.setdefault(str(cfg), []).append(start + i)
                print("MISMATCH config", cfg, "image", start + i, "N", sc["lp"].shape[0], "iterations (oracle)",
                      ref.get("iterations"), "(hip)", r.get("iterations"))
                np.savez_compressed("gpurun_out/mismatch_fresh/c%d_%d.npz" % (cfg, start + i), l=sc["l"], lp=sc["lp"],
                                    cnn_response=sc["cnn_response"], sphere_image=sc["sphere_image"])
        print("config %d images %d..%d: oracle %.0f s, largest VP difference among the matching ones %.1e" % (
            cfg, start, start + per - 1, to, worst), flush=True)
    print("TOTAL %d fresh scenes, %d mismatches" % (total, bad))
    with open("gpurun_out/fresh_failures.json", "w") as fh:
        json.dump(failures, fh)


if __name__ == "__main__":
    main()
