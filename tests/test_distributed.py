"""The N > 1 path on CPU: image sharding + the final record gather with world_size-2 gloo."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from vanishing_points_2017_amd import sharding


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 102, 2018):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_shard_balanced():
    rs = np.random.RandomState(0)
    costs = rs.randint(100, 1000, 2018).astype(float) ** 2
    parts = sharding.shard_balanced(costs, 8)
    assert sorted(np.concatenate(parts).tolist()) == list(range(2018))
    loads = np.array([costs[p].sum() for p in parts])
    assert loads.max() / loads.mean() < 1.01


def _fake_result(i):
    rs = np.random.RandomState(i)
    m = rs.randint(0, 6)
    if m == 0:
        return {"vp": None, "status": 1}
    v = rs.randn(m, 3)
    return {"vp": v / np.linalg.norm(v, axis=1)[:, None], "counts": rs.randint(3, 50, m).astype(float), "status": 0}


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sharding.shard_range(n_items, rank, world)
    ids = list(range(lo, hi))
    rec = sharding.pack_records(ids, [_fake_result(i) for i in ids], errors=[0.01 * i for i in ids])
    allrec = sharding.gather_records(dist, rec)
    if rank == 0:
        q.put(allrec)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_records_gloo_world2():
    world, n_items = 2, 11
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    allrec = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert allrec.shape == (n_items, sharding.REC_WIDTH)
    for i in range(n_items):
        got = sharding.unpack_record(allrec[i])
        want = _fake_result(i)
        assert got["image"] == i and abs(got["error"] - 0.01 * i) < 1e-15
        if want["vp"] is None:
            assert got["status"] == 1 and got["vp"].shape[0] == 0
        else:
            order = np.argsort(want["counts"])[::-1]
            assert np.array_equal(got["vp"], want["vp"][order]) and np.array_equal(got["counts"], want["counts"][order])
