"""Image sharding across the GPUs of one node.  Images are independent units, so
the only collective of the path is one all_gather of fixed-size result records at the end."""
import numpy as np

REC_VPS = 20                       # VPs kept per record (calc_horizon uses at most maxbest = 20)
REC_WIDTH = 3 + REC_VPS * 4 + 1    # image id, status, num_vp, (vp xyz + count) x 20, horizon error


def shard_range(n_items, rank, world):
    """Contiguous block of rank `rank`: sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_balanced(costs, world):
    """Greedy longest-processing-time assignment by cost (N^2 varies ~10x between images).
    Returns a list of index arrays, one per rank; deterministic."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    buckets = [[] for _ in range(world)]
    for i in order:
        r = int(np.argmin(load))
        buckets[r].append(int(i))
        load[r] += costs[i]
    return [np.array(sorted(b), dtype=np.int64) for b in buckets]


def pack_records(image_ids, results, errors=None):
    """Per-image EM results -> float64 (n, REC_WIDTH) records."""
    rec = np.zeros((len(results), REC_WIDTH))
    for k, (iid, r) in enumerate(zip(image_ids, results)):
        rec[k, 0] = iid
        rec[k, 1] = r.get("status", 0 if r.get("vp") is not None else 1)
        if r.get("vp") is not None:
            m = min(REC_VPS, r["vp"].shape[0])
            order = np.argsort(r["counts"])[::-1][:m]
            rec[k, 2] = m
            rec[k, 3:3 + 3 * m] = r["vp"][order].reshape(-1)
            rec[k, 3 + 3 * REC_VPS:3 + 3 * REC_VPS + m] = r["counts"][order]
        rec[k, -1] = np.nan if errors is None else errors[k]
    return rec


def unpack_record(row):
    m = int(row[2])
    return {"image": int(row[0]), "status": int(row[1]), "vp": row[3:3 + 3 * m].reshape(m, 3).copy(),
            "counts": row[3 + 3 * REC_VPS:3 + 3 * REC_VPS + m].copy(), "error": float(row[-1])}


def device_records(torch, image_ids, out):
    """The same records as pack_records, built from vpk_em_batch's DEVICE outputs without a host round
    trip (bench.py gathers once per step inside its timed region).  image_ids: int64 tensor (B) on the
    device; out: the dict returned by em.em_batch_device.  VPs are ordered by descending line count like
    calc_horizon.py:34-36; the order among EQUAL counts is torch's stable sort here and NumPy's
    reversed argsort in pack_records -- the set of VPs kept is the same."""
    counts, nv = out["counts"], out["num_vp"].to(torch.int64)
    b, max_vp = counts.shape
    dev = counts.device
    valid = torch.arange(max_vp, device=dev)[None, :] < nv[:, None]
    key = torch.where(valid, counts, torch.full_like(counts, -1.0))
    width = min(REC_VPS, max_vp)
    order = torch.argsort(key, dim=1, descending=True, stable=True)[:, :width]
    m = torch.clamp(nv, max=width)
    keep = (torch.arange(width, device=dev)[None, :] < m[:, None]).to(torch.float64)
    vp = torch.gather(out["vp"], 1, order[:, :, None].expand(-1, -1, 3)) * keep[:, :, None]
    cnt = torch.gather(counts, 1, order) * keep
    rec = torch.zeros((b, REC_WIDTH), dtype=torch.float64, device=dev)
    rec[:, 0] = image_ids.to(torch.float64)
    rec[:, 1] = out["status"].to(torch.float64)
    rec[:, 2] = m.to(torch.float64)
    rec[:, 3:3 + 3 * width] = vp.reshape(b, -1)
    rec[:, 3 + 3 * REC_VPS:3 + 3 * REC_VPS + width] = cnt
    rec[:, -1] = float("nan")
    return rec


def gather_device(dist, rec):
    """all_gather of equally sized per-rank record blocks that stay on the device (RCCL over xGMI with the
    nccl backend; gloo on CPU tensors): (B, W) per rank -> (world * B, W), rank-major."""
    blocks = [rec.new_empty(rec.shape) for _ in range(dist.get_world_size())]
    dist.all_gather(blocks, rec)
    import torch
    return torch.cat(blocks, 0)


def gather_records(dist, rec, device=None):
    """all_gather of ragged per-rank record blocks (padded to the largest block).  Works with the
    nccl (= RCCL) backend on GPU tensors and with gloo on CPU tensors."""
    import torch
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float64))
    n = torch.tensor([t.shape[0]], dtype=torch.int64)
    if device is not None:
        t, n = t.to(device), n.to(device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = int(max(int(c.item()) for c in counts))
    pad = torch.zeros((nmax, t.shape[1] if t.ndim == 2 else REC_WIDTH), dtype=torch.float64, device=t.device)
    pad[:t.shape[0]] = t
    blocks = gather_device(dist, pad).reshape(world, nmax, -1)
    out = [b[:int(c.item())].cpu().numpy() for b, c in zip(blocks, counts)]
    allrec = np.concatenate(out, 0) if out else np.zeros((0, REC_WIDTH))
    return allrec[np.argsort(allrec[:, 0], kind="stable")]
