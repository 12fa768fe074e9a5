"""Benchmark driver with the reference's CLI (benchmark.py:16-26):

    python -m vanishing_points_2017_amd.benchmark --yud --result_dir /tmp/vp --update_datalist \\
        --update_datafiles --run_cnn --run_em [--synthetic [--count N]]

``--synthetic`` replaces the dataset directory (YUD / ECD / HLW are not reachable offline,
config.py:3-5) by the seeded synthetic set of the same shape (synth.CONFIGS): line segments and
ground-truth horizons come from the generator, rasters from the GPU rasteriser, and -- unless
--run_cnn is given together with real weight files -- the generator's response map stands in for a
trained CNN's prediction.  The scoring loop is the reference's (:108-266): horizon from the best
orthogonal triplet, error = max end-point deviation / image height, AUC up to 0.25.
Ground-truth loaders for the real datasets (.mat / .csv, :142-220) are out of scope."""
import argparse
import os
import time

import numpy as np

from . import auc as auc_mod, calc_horizon as ch, config, evaluation, sharding, sphere_mapping, synth

SHAPES = {"york": 2, "eurasian": 3, "horizon": 4}


def build_parser():
    p = argparse.ArgumentParser(description='')
    p.add_argument('--yud', dest='yud', action='store_true', help='Run benchmark on YUD')
    p.add_argument('--ecd', dest='ecd', action='store_true', help='Run benchmark on ECD')
    p.add_argument('--hlw', dest='hlw', action='store_true', help='Run benchmark on HLW')
    p.add_argument('--result_dir', default='/tmp/', type=str, help='Directory to store (intermediate) results')
    p.add_argument('--gpu', default=0, type=int, help='GPU ID to use')
    p.add_argument('--update_datalist', dest='update_datalist', action='store_true', help='Update the dataset list')
    p.add_argument('--update_datafiles', dest='update_datafiles', action='store_true', help='Update the dataset files')
    p.add_argument('--run_cnn', dest='run_cnn', action='store_true', help='Evaluate CNN on the data')
    p.add_argument('--run_em', dest='run_em', action='store_true', help='Run EM refinement on the data')
    p.add_argument('--synthetic', action='store_true', help='use the seeded synthetic set of the dataset shape')
    p.add_argument('--count', type=int, default=None, help='number of synthetic images (default: dataset size)')
    p.add_argument('--force-dist', action='store_true',
                   help='initialise torch.distributed (RCCL unless VPK_DIST_BACKEND says otherwise) even with one rank: the N > 1 '
                        'code path -- process group, barriers, the record gather on the GPU -- on a one-GPU box')
    p.add_argument('--gpus', type=int, default=1,
                   help='GPUs of this node to shard the images over (one process per GPU; results gathered with one '
                        'RCCL all_gather).  Without a launcher environment the ranks are started as child processes.')
    return p


def synthetic_dataset(name, result_dir, count, update):
    """Writes reference-schema pickles for the synthetic set and returns the dataset dict."""
    cfg = SHAPES[name]
    dest = os.path.join(result_dir, name, "synthetic_angle_weights_split_merge")
    os.makedirs(dest, exist_ok=True)
    scenes = list(synth.config_scenes(cfg, count=count, raster=None))
    files = [os.path.join(dest, "img%05d.data.pkl" % i) for i in range(len(scenes))]
    dataset = {'source_folder': "synthetic:%s" % name, 'destination_folder': dest, 'use_weights': True,
               'distance_measure': "angle", 'do_split': True, 'do_merge': True, 'name': "synthetic_" + name,
               'image_files': ["synthetic:%s:%d" % (name, s["seed"]) for s in scenes], 'pickle_files': files,
               'true_horizon': [s["true_horizon"] for s in scenes], 'image_shape': [s["image_shape"] for s in scenes],
               'line_counts': [int(s["lp"].shape[0]) for s in scenes]}
    if update or not all(os.path.isfile(f) for f in files):
        rasters = sphere_mapping.raster_batch([s["l"] for s in scenes], size=500, alpha=0.1)
        for s, f, r in zip(scenes, files, rasters):
            datum = {"dataset": dataset["name"], "image_file": "synthetic", "image_shape": s["image_shape"],
                     "image": None, "line_segments": s["lp"], "lines": s["l"]}
            evaluation._dump_pickle({'lines': datum, 'sphere_image': r, 'cnn_prediction': s["cnn_response"]}, f)
    return dataset


def horizon_errors(dataset, indices, n_vp=20, theta_vmin=np.pi / 10, horizon_fn=None, device=0):
    """Per-image part of the scoring loop (benchmark.py:229-257) for the images in ``indices``: returns
    (em_results, errors).  ``horizon_fn`` defaults to the batched GPU selection."""
    todo = []
    for idx in indices:
        datum = evaluation._load_pickle(dataset['pickle_files'][int(idx)])
        em_result = datum.get('EM_result')
        assert em_result is not None, "no EM result!"                        # :231
        if em_result['vp'] is None:
            em_result = {'vp': np.zeros((0, 3)), 'counts': np.zeros(0)}
        todo.append(em_result)
    fn = horizon_fn or (lambda rs: ch.calculate_horizon_batch(rs, maxbest=n_vp, theta_vmin=theta_vmin, device=device))
    horizons = fn(todo) if todo else []
    errors = [ch.horizon_error(h[0], h[1], dataset['true_horizon'][int(idx)], dataset['image_shape'][int(idx)])
              for idx, h in zip(indices, horizons)]
    return todo, np.array(errors)


def run_sharded(dataset, rank=0, world=1, dist=None, device=0, start=0, run_em=True, em_fn=None,
                horizon_fn=None, err_cutoff=0.25, gather_on=None):
    """BASELINE configs[3]: the images of a dataset sharded over the ranks of one node.

    The reference's only multi-process facility is the start/end slice of run_em (evaluation.py:295-307).
    Here every rank (one process per GPU) takes one part of a cost-balanced partition (cost = N^2: the
    pairwise and smoothing work), refines and scores its images, and ONE all_gather of fixed-size records
    (sharding.pack_records / gather_records: RCCL over xGMI with the nccl backend, gloo on CPU) brings VPs,
    counts and horizon errors to every rank; rank 0 reports the AUC (benchmark.py:264).  No other exchange.
    ``em_fn(dataset, indices)`` defaults to evaluation.run_em on GPU ``device``; ``gather_on`` = torch device
    the records are gathered on (the rank's GPU for nccl, None = CPU tensors for gloo)."""
    n = len(dataset['pickle_files'])
    costs = np.asarray(dataset.get('line_counts', np.ones(n)), dtype=np.float64) ** 2
    mine = sharding.shard_balanced(costs, world)[rank]
    if run_em and len(mine):
        (em_fn or (lambda ds, idx: evaluation.run_em(ds, indices=idx, device=device)))(dataset, mine)
    results, errors = horizon_errors(dataset, mine, horizon_fn=horizon_fn, device=device)
    rec = sharding.pack_records(mine, results, errors)
    allrec = sharding.gather_records(dist, rec, device=gather_on) if dist is not None else rec
    assert allrec.shape[0] == n and np.array_equal(allrec[:, 0], np.arange(n)), "records lost in the gather"
    errs = allrec[start:, -1]
    auc, pts = auc_mod.calc_auc(errs.copy(), cutoff=err_cutoff)
    return auc, errs, allrec


def score(dataset, start=0, n_vp=20, theta_vmin=np.pi / 10, err_cutoff=0.25):
    """benchmark.py:108-266 with the ground truth supplied by the dataset dict."""
    errors = []
    todo = []
    for idx, data_file in enumerate(dataset['pickle_files']):
        if idx < start:
            continue
        datum = evaluation._load_pickle(data_file)
        em_result = datum.get('EM_result')
        assert em_result is not None, "no EM result!"                        # :231
        if em_result['vp'] is None:
            em_result = {'vp': np.zeros((0, 3)), 'counts': np.zeros(0)}
        todo.append((idx, em_result))
    # the reference calls calculate_horizon_and_ortho_vp image by image (:237-243); one batched launch here
    horizons = ch.calculate_horizon_batch([r for _, r in todo], maxbest=n_vp, theta_vmin=theta_vmin)
    for (idx, _), (hp1, hp2, _, _, _, _) in zip(todo, horizons):
        errors.append(ch.horizon_error(hp1, hp2, dataset['true_horizon'][idx], dataset['image_shape'][idx]))
    errors = np.array(errors)
    auc, pts = auc_mod.calc_auc(errors, cutoff=err_cutoff)
    return auc, errors, pts


def _spawn(args_list, gpus):
    """Start one process per GPU (torch.distributed.run) as children of this not-yet-GPU-initialised process."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "-m", "vanishing_points_2017_amd.benchmark"] + args_list
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    import sys
    args = build_parser().parse_args(argv)
    if args.yud:
        name = "york"
    elif args.ecd:
        name = "eurasian"
    elif args.hlw:
        name = "horizon"
    else:
        assert False                                                         # benchmark.py:49
    if not args.synthetic:
        raise NotImplementedError(
            "the real %s dataset and its ground-truth loaders are out of scope (no dataset offline, LSD not "
            "vendored); run with --synthetic, or use evaluation.run_cnn / run_em on existing reference pickles" % name)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return _spawn(list(sys.argv[1:] if argv is None else argv), args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    device = args.gpu
    backend = os.environ.get("VPK_DIST_BACKEND", "nccl")      # "gloo": ranks may share a GPU (tests on a one-GPU box)
    if world > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        device = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(device)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend=backend)
    if rank == 0:    # one rank writes the input pickles; the others read them after the barrier
        dataset = synthetic_dataset(name, args.result_dir, args.count, args.update_datafiles or args.update_datalist)
    if dist is not None:
        dist.barrier()
    if rank != 0:
        dataset = synthetic_dataset(name, args.result_dir, args.count, False)
    if args.run_cnn and rank == 0:
        if not all(os.path.isfile(f) for f in (config.cnn_weights_path, config.cnn_mean_path)):
            print("no trained weights at %s: keeping the generator's response maps as cnn_prediction"
                  % config.cnn_weights_path)
        else:
            evaluation.run_cnn(dataset, mean_file=config.cnn_mean_path, model_def=config.cnn_config_path,
                               model_weights=config.cnn_weights_path, gpu=device)
    if dist is not None:
        dist.barrier()
    start = 25 if (args.yud or args.ecd) else 0                              # :69
    t0 = time.time()
    tdev = None
    if dist is not None and backend == "nccl":
        import torch
        tdev = torch.device("cuda", device)
    auc, errors, _ = run_sharded(dataset, rank, world, dist, device=device, start=start, run_em=args.run_em,
                                 gather_on=tdev)
    if rank == 0:
        print("EM + scoring time (%d rank(s)%s): " % (world, "" if dist is None else ", records gathered over " + backend), time.time() - t0)
        print("AUC: ", auc)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return auc


if __name__ == "__main__":
    main()
