"""Horizon line from the best orthogonal VP triplet -- host-side port of the reference's
calc_horizon.calculate_horizon_and_ortho_vp (calc_horizon.py:19-225).  Post-processing of the EM
result: needed to report the horizon-AUC half of the metric.
``calculate_horizon_batch`` runs the same selection for a whole dataset on the GPU (vpk_horizon_batch):
the reference's per-image Python loop over C(20,3) triplets is otherwise the slowest stage of the scoring
loop once the EM runs on the GPU."""
import itertools

import numpy as np

_EZ = np.array([0.0, 0.0, 1.0])


def num_combo3(n):
    """calc_horizon.py:3-8 (n choose 3 via the reference's recurrence)."""
    return n * (n - 1) * (n - 2) // 6 if n >= 3 else 0


def vp_in_image(vp):
    """calc_horizon.py:11-16."""
    q = vp / vp[2]
    return bool(q[0] <= 1 and q[0] >= -1 and q[1] <= 1 and q[1] >= -1)


def _end_points(hlin):
    p1 = np.cross(hlin, np.array([1, 0, 1]))
    p2 = np.cross(hlin, np.array([-1, 0, 1]))
    return p1 / p1[2], p2 / p2[2]


def calculate_horizon_and_ortho_vp(em_result, maxbest=10, theta_vmin=np.pi / 10., theta_z=np.pi / 4.):
    """Returns (hP1, hP2, zVP, hVP1, hVP2, best_combo) exactly like the reference."""
    vps = em_result['vp'].copy()
    counts = em_result['counts']
    num_best = int(np.minimum(maxbest, vps.shape[0]))
    zenith_set = set(np.where(np.abs(vps[:, 1]) > np.sin(theta_z))[0].tolist())          # :31
    best_vps = np.argsort(counts)[::-1][0:num_best]                                       # :34-36
    costh = np.cos(theta_vmin)
    hlin = None
    best_combo = 0
    if num_best > 2:
        combos = list(itertools.combinations(range(num_best), 3))                         # :45-50 (i<j<k order)
        best_score = -1
        best_idx = 0
        for idx, (a, b, c) in enumerate(combos):
            ia, ib, ic = best_vps[a], best_vps[b], best_vps[c]
            va, vb, vc = vps[ia], vps[ib], vps[ic]
            ab, bc, ac = np.abs(np.dot(va, vb)), np.abs(np.dot(vb, vc)), np.abs(np.dot(va, vc))
            num_zenith = 0
            zenith = None
            for i_, v_ in ((ia, va), (ib, vb), (ic, vc)):                                 # :82-91 (last one wins)
                if i_ in zenith_set:
                    num_zenith += 1
                    zenith = v_
            num_central = int(vp_in_image(va)) + int(vp_in_image(vb)) + int(vp_in_image(vc))
            ya, yb, yc = np.abs(va[1]), np.abs(vb[1]), np.abs(vc[1])
            if ya > yb and ya > yc:                                                       # :105-125
                h1, h2, zv, c1, c2 = vb, vc, va, counts[ib], counts[ic]
            elif yb > ya and yb > yc:
                h1, h2, zv, c1, c2 = va, vc, vb, counts[ia], counts[ic]
            else:
                h1, h2, zv, c1, c2 = va, vb, vc, counts[ia], counts[ib]
            zlin = np.cross(zv, _EZ)
            zlin = zlin / np.linalg.norm(zlin[0:2])
            l1, l2 = zlin[0], zlin[1]
            d1 = np.linalg.norm(_EZ - h1 / h1[2])
            d2 = np.linalg.norm(_EZ - h2 / h2[2])
            h3 = ((h1[0] * l2 - h1[1] * l1) / h1[2] * (d2 * c1) + (h2[0] * l2 - h2[1] * l1) / h2[2] * (d1 * c2)) \
                / ((d1 * c2) + (d2 * c1))                                                 # :147
            hl = np.array([-l2, l1, h3])
            hvec = (h1 / h1[2]) - (h2 / h2[2])
            hang = np.arccos(np.abs(np.dot(hvec, np.array([1, 0, 0]))) / np.linalg.norm(hvec))
            p1, p2 = _end_points(hl)
            ortho = 0
            if num_zenith == 1:                                                           # :164-167
                cosphi = np.abs(np.dot(hvec / np.linalg.norm(hvec), zenith / np.linalg.norm(zenith)))
                ortho = 1 - np.clip(1.0 * cosphi, 0, 1)
            zenith_pos = 1 if zv[1] > 0 else -1
            hor_pos = 1 if (p1[1] + p2[1]) / 2 < 0 else -1
            ok = (ab < costh and bc < costh and ac < costh and num_zenith == 1 and num_central <= 1
                  and hang < 30 * np.pi / 180 and zenith_pos * hor_pos == 1)              # :176-179
            score = (1 if ok else 0) * (counts[ia] + counts[ib] + counts[ic]) * ortho     # :182-185
            if score > best_score:                                                        # :190-196
                best_idx, best_score = idx, score
                hvp1, hvp2, zvp, hlin = h1, h2, zv, hl
        best_combo = best_vps[np.array(combos[best_idx])]
    elif num_best > 1:                                                                    # :200-205
        hvp1, hvp2, zvp = vps[0, :], vps[1, :], np.array([0, 1, 0])
        best_combo = np.array([0, 1])
        hlin = np.cross(hvp1, hvp2)
    elif num_best > 0:                                                                    # :206-211
        hvp1, hvp2, zvp = vps[0, :], vps[0, :], np.array([0, 1, 0])
        best_combo = np.array([0, 0])
        hlin = np.cross(np.array([0, 0, 1]), np.array([1, 0, 1]))
    else:                                                                                 # :212-217
        hvp1, hvp2, zvp = np.array([-1, 0, 0]), np.array([1, 0, 0]), np.array([0, 1, 0])
        best_combo = np.array([0, 0])
        hlin = np.cross(np.array([0, 0, 1]), np.array([1, 0, 1]))
    hp1, hp2 = _end_points(hlin)
    return (hp1, hp2, zvp, hvp1, hvp2, best_combo)


def horizon_error(hp1, hp2, true_horizon, image_shape):
    """benchmark.py:245-253: max vertical deviation at x = +-1, relative to the image height."""
    height, width = image_shape[0], image_shape[1]
    scale = np.maximum(width, height)
    t1 = np.cross(true_horizon, np.array([1, 0, 1]))
    t2 = np.cross(true_horizon, np.array([-1, 0, 1]))
    t1 = t1 / t1[2]
    t2 = t2 / t2[2]
    return np.maximum(np.abs(hp1[1] - t1[1]), np.abs(hp2[1] - t2[1])) / 2 * scale * 1.0 / height


def calculate_horizon_batch(em_results, maxbest=10, theta_vmin=np.pi / 10., theta_z=np.pi / 4., device=0):
    """calculate_horizon_and_ortho_vp for a list of EM results in one launch of vpk_horizon_batch.
    Returns one (hP1, hP2, zVP, hVP1, hVP2, best_combo) tuple per result, like the per-image function."""
    import ctypes
    from .runtime import get_runtime
    rt = get_runtime(device)
    torch = rt.torch
    batch = len(em_results)
    max_vp = max([1] + [r['vp'].shape[0] for r in em_results])
    if max_vp > 64 or maxbest > 64:
        raise ValueError("vpk_horizon_batch handles at most 64 VPs per image")
    vp = np.zeros((batch, max_vp, 3))
    counts = np.zeros((batch, max_vp))
    num = np.zeros(batch, dtype=np.int32)
    order = np.zeros((batch, maxbest), dtype=np.int32)
    for b, r in enumerate(em_results):
        m = r['vp'].shape[0]
        num[b] = m
        vp[b, :m] = r['vp']
        counts[b, :m] = r['counts']
        nb = min(maxbest, m)
        order[b, :nb] = np.argsort(r['counts'])[::-1][0:nb]          # calc_horizon.py:34-36 (the caller's tie order)
    with rt.on_stream():
        d_vp, d_cnt = torch.from_numpy(vp).to(rt.tdev), torch.from_numpy(counts).to(rt.tdev)
        d_num, d_ord = torch.from_numpy(num).to(rt.tdev), torch.from_numpy(order).to(rt.tdev)
        out = torch.empty((batch, 15), dtype=torch.float64, device=rt.tdev)
        combo = torch.empty((batch, 3), dtype=torch.int32, device=rt.tdev)
        rt.check(rt.lib.vpk_horizon_batch(rt.h, batch, max_vp, rt.ptr(d_vp), rt.ptr(d_cnt), rt.ptr(d_num), rt.ptr(d_ord),
                                          int(maxbest), ctypes.c_double(theta_vmin), ctypes.c_double(theta_z), rt.ptr(out),
                                          rt.ptr(combo)))
    rt.synchronize()
    out, combo = out.cpu().numpy(), combo.cpu().numpy()
    res = []
    for b in range(batch):
        o = out[b]
        c = combo[b]
        res.append((o[0:3].copy(), o[3:6].copy(), o[6:9].copy(), o[9:12].copy(), o[12:15].copy(),
                    c.astype(np.int64) if c[2] >= 0 else c[:2].astype(np.int64)))
    return res
