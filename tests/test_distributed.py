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


def test_device_records_match_pack_records():
    """bench.py's on-device records (sharding.device_records) carry what pack_records carries."""
    rs = np.random.RandomState(3)
    b, max_vp = 7, 64
    out = {"vp": torch.zeros((b, max_vp, 3), dtype=torch.float64), "counts": torch.zeros((b, max_vp), dtype=torch.float64),
           "num_vp": torch.zeros(b, dtype=torch.int32), "status": torch.zeros(b, dtype=torch.int32)}
    results = []
    for i in range(b):
        m = [0, 1, 3, 20, 21, 40, 5][i]
        v = rs.randn(m, 3)
        c = rs.permutation(200)[:m].astype(float)          # distinct counts: the order is unambiguous
        out["vp"][i, :m] = torch.from_numpy(v)
        out["counts"][i, :m] = torch.from_numpy(c)
        out["vp"][i, m:] = 777.0                           # garbage beyond num_vp must not leak into the record
        out["counts"][i, m:] = 999.0
        out["num_vp"][i] = m
        out["status"][i] = 0 if m else 1
        results.append({"vp": v if m else None, "counts": c, "status": 0 if m else 1})
    ids = torch.arange(100, 100 + b)
    rec = sharding.device_records(torch, ids, out).numpy()
    want = sharding.pack_records(list(range(100, 100 + b)), results)
    assert np.array_equal(rec[:, :-1], want[:, :-1]) and np.isnan(rec[:, -1]).all()


def _bench_worker(rank, world, port, tmp, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from vanishing_points_2017_amd import benchmark
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dataset = _stub_dataset(tmp, write=False)
    auc, errs, allrec = benchmark.run_sharded(dataset, rank, world, dist, em_fn=_stub_em, horizon_fn=_host_horizons)
    if rank == 0:
        q.put((auc, errs))
    dist.barrier()
    dist.destroy_process_group()


def _stub_dataset(tmp, write=True, n=13):
    """Reference-schema pickles of a small synthetic HLW-shape set (no EM result yet)."""
    from vanishing_points_2017_amd import evaluation, synth
    scenes = list(synth.config_scenes(4, count=n, raster=None))
    files = [os.path.join(tmp, "img%03d.data.pkl" % i) for i in range(n)]
    if write:
        for s, f in zip(scenes, files):
            datum = {"dataset": "stub", "image_file": "synthetic", "image_shape": s["image_shape"], "image": None,
                     "line_segments": s["lp"], "lines": s["l"]}
            evaluation._dump_pickle({'lines': datum, 'sphere_image': None, 'cnn_prediction': s["cnn_response"],
                                     'true_vps': s["true_vps"]}, f)
    return {'pickle_files': files, 'true_horizon': [s["true_horizon"] for s in scenes],
            'image_shape': [s["image_shape"] for s in scenes], 'line_counts': [s["lp"].shape[0] for s in scenes],
            'distance_measure': "angle", 'use_weights': True, 'do_split': True, 'do_merge': True}


def _stub_em(dataset, indices):
    """Stands in for evaluation.run_em on a box without a GPU: the 'EM result' is the scene's true VPs."""
    from vanishing_points_2017_amd import evaluation
    for i in indices:
        f = dataset['pickle_files'][int(i)]
        d = evaluation._load_pickle(f)
        vps = d['true_vps']
        d['EM_result'] = {"vp": vps, "counts": np.arange(vps.shape[0], 0, -1).astype(float) * 10}
        evaluation._dump_pickle(d, f)


def _host_horizons(results):
    from vanishing_points_2017_amd import calc_horizon as ch
    return [ch.calculate_horizon_and_ortho_vp(r, maxbest=20, theta_vmin=np.pi / 10) for r in results]


def test_sharded_benchmark_flow_gloo_world2(tmp_path):
    """benchmark.run_sharded (config 4: cost-balanced image shards, per-rank EM + scoring, ONE record gather,
    AUC on rank 0) with two gloo ranks and a stubbed EM gives the single-process AUC."""
    from vanishing_points_2017_amd import benchmark
    tmp = str(tmp_path)
    dataset = _stub_dataset(tmp)
    auc1, errs1, _ = benchmark.run_sharded(dataset, 0, 1, None, em_fn=_stub_em, horizon_fn=_host_horizons)
    assert 0.0 < auc1 <= 1.0
    for f in dataset['pickle_files']:                       # wipe the results: the two ranks must produce them
        from vanishing_points_2017_amd import evaluation
        d = evaluation._load_pickle(f)
        d.pop('EM_result')
        evaluation._dump_pickle(d, f)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
    for p in procs:
        p.start()
    auc2, errs2 = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert auc2 == auc1 and np.array_equal(errs1, errs2)


def test_bench_gpus_flag_is_honoured():
    """`python bench.py --gpus 2` must either run two ranks or fail loudly -- never silently run one."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    if torch.cuda.device_count() < 2:
        assert r.returncode == 2 and "--gpus 2" in r.stderr
    else:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
