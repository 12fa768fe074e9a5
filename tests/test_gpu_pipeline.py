"""End-to-end parity of the metric: horizon-line AUC of the GPU path vs the CPU oracle on the same
seeded synthetic YUD-shape inputs (BASELINE.json: 'identical horizon-line AUC', errors within 1e-4),
through the reference's call surface (evaluation.run_em on reference-schema pickles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_horizon_auc_parity_yud_shape(tmp_path):
    from oracle import em_numpy
    from vanishing_points_2017_amd import auc, benchmark, calc_horizon as ch, evaluation
    ds = benchmark.synthetic_dataset("york", str(tmp_path), 40, True)
    evaluation.run_em(ds)
    auc_gpu, err_gpu, _ = benchmark.score(ds, start=25)
    err_ref = []
    for idx, f in enumerate(ds['pickle_files']):
        if idx < 25:
            continue
        d = evaluation._load_pickle(f)
        lines = d['lines']
        # the pickled lines were normalised in place by run_em (like the reference): equivalent input
        ref = em_numpy.expectation_maximisation(lines['lines'].copy(), lines['line_segments'].copy(),
                                                d['cnn_prediction'].copy(), sphere_image=d['sphere_image'])
        got = d['EM_result']
        assert np.array_equal(got['vp_assoc'], ref['vp_assoc'])              # bit-exact assignments
        assert np.abs(got['vp'] - ref['vp']).max() <= 1e-4
        hp1, hp2, _, _, _, _ = ch.calculate_horizon_and_ortho_vp(ref, maxbest=20, theta_vmin=np.pi / 10)
        err_ref.append(ch.horizon_error(hp1, hp2, ds['true_horizon'][idx], ds['image_shape'][idx]))
    err_ref = np.array(err_ref)
    assert np.abs(err_gpu - err_ref).max() <= 1e-4
    auc_ref, _ = auc.calc_auc(err_ref, cutoff=0.25)
    assert abs(auc_gpu - auc_ref) <= 1e-4
    assert auc_gpu > 0.5                                                     # the scenes are solvable


def test_run_cnn_surface_with_written_model_files(tmp_path):
    """init_caffe / read_mean_blob / run_cnn on caffemodel + binaryproto files written without Caffe."""
    from oracle import cnn_torch
    from vanishing_points_2017_amd import benchmark, caffe_io, cnn, evaluation
    w = cnn.synthetic_weights(5)
    mean = cnn.synthetic_mean(5)
    layers = {("fc8_20x20" if k == "fc8" else k): [v[0], v[1]] for k, v in w.items()}
    caffe_io.write_caffemodel(str(tmp_path / "weights.caffemodel"), layers)
    caffe_io.write_binaryproto(str(tmp_path / "mean.binaryproto"), mean.reshape(1, 1, 500, 500))
    ds = benchmark.synthetic_dataset("york", str(tmp_path), 3, True)
    evaluation.run_cnn(ds, None, str(tmp_path / "weights.caffemodel"), str(tmp_path / "mean.binaryproto"), gpu=0)
    sphere = np.stack([evaluation._load_pickle(f)['sphere_image'] for f in ds['pickle_files']])
    ref = cnn_torch.forward(w, mean, sphere)
    for f, r in zip(ds['pickle_files'], ref):
        pred = evaluation._load_pickle(f)['cnn_prediction']
        assert pred.shape == (20, 20) and pred.dtype == np.float32
        assert np.abs(pred - r).max() <= 2e-5


def test_horizon_batch_matches_host_selection():
    """vpk_horizon_batch (calc_horizon.py:19-225 on the GPU) against the host port on the EM results of the
    goldens and of fresh scenes, plus the < 3 VP fallbacks: same triplet, end points to 1e-12."""
    from conftest import golden_cases
    from golden_util import load
    from vanishing_points_2017_amd import calc_horizon as ch, em as gem, synth
    results = []
    for name in golden_cases():
        g = load(name)
        if "o_vp" in g:
            results.append({"vp": g["o_vp"], "counts": g["o_counts"]})
    scenes = list(synth.config_scenes(3, count=24, start=7)) + list(synth.config_scenes(2, count=24, start=60))
    for r in gem.em_batch(scenes):
        if r["status"] == 0:
            results.append({"vp": r["vp"], "counts": r["counts"]})
    rs = np.random.RandomState(11)
    for m in (0, 1, 2, 3, 30):            # fallbacks, the smallest triplet case, more VPs than maxbest
        v = rs.normal(size=(m, 3))
        v /= np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
        results.append({"vp": v, "counts": np.floor(rs.uniform(3, 40, m))})
    got = ch.calculate_horizon_batch(results, maxbest=20, theta_vmin=np.pi / 10)
    assert len(got) == len(results) and len(results) > 50
    for r, g_ in zip(results, got):
        ref = ch.calculate_horizon_and_ortho_vp(r, maxbest=20, theta_vmin=np.pi / 10)
        assert np.array_equal(np.asarray(ref[5]).ravel(), np.asarray(g_[5]).ravel())
        for a, b in zip(ref[:5], g_[:5]):
            assert np.allclose(np.asarray(a, dtype=float), b, rtol=0, atol=1e-12, equal_nan=True)


def test_image_files_through_the_whole_call_surface(tmp_path):
    """example.py's flow (example.py:30-39) from image FILES: get_data_list -> create_data_pickles (this package's
    front end: Pillow, LSD, normalisation, GPU raster; target_size 640) -> run_cnn -> run_em, reference pickle schema."""
    from PIL import Image
    from test_frontend import _render
    from vanishing_points_2017_amd import cnn, evaluation
    src, dst = tmp_path / "images", tmp_path / "results"
    src.mkdir(); dst.mkdir()
    rs = np.random.RandomState(3)
    for k in range(2):                                  # strokes towards two vanishing points, rendered at 1280 x 960
        segs = []
        for vp in ((2000.0, 300.0), (-900.0, 500.0)):
            for _ in range(14):
                x, y = rs.uniform(100, 1180), rs.uniform(100, 860)
                d = np.array([vp[0] - x, vp[1] - y]); d /= np.linalg.norm(d)
                ln = rs.uniform(120, 300)
                segs.append((x, y, x + ln * d[0], y + ln * d[1]))
        img = _render(segs, 960, 1280, width=4.0).astype(np.uint8)
        Image.fromarray(np.repeat(img[:, :, None], 3, 2)).save(str(src / ("scene%d.png" % k)))
    dataset = evaluation.get_data_list(str(src), str(dst), 'default_net', "", "0", update=True)
    assert len(dataset['image_files']) == 2
    evaluation.create_data_pickles(dataset, update=True, cnn_input_size=500, target_size=640)
    evaluation.run_cnn(dataset, None, None, None, net=cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0)))
    evaluation.run_em(dataset)
    for f in dataset['pickle_files']:
        d = evaluation._load_pickle(f)
        assert d['lines']['image'].shape == (480, 640, 3)                      # resized to fit 640
        seg = d['lines']['line_segments']
        assert seg.shape[0] >= 40 and np.abs(seg).max() <= 1.0                # two edges per stroke, normalised coordinates
        assert d['sphere_image'].shape == (500, 500) and d['sphere_image'].max() > 0
        assert d['cnn_prediction'].shape == (20, 20)
        res = d['EM_result']
        assert res is not None and res['vp'] is not None and res['vp_assoc'].shape[0] == seg.shape[0]


@pytest.mark.parametrize("mode", ["lanes", "slice", "serial"])
def test_bench_modes_produce_a_valid_line(mode):
    """bench.py end to end (small batch): one JSON line with the contract's fields, every image refined, and the EM
    kernel's results on the first scenes equal to the reference's stored results -- in each scheduling mode."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--images", "12",
                        "--no-cpu-baseline", "--em-mode", mode], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "parity", "ranks_seen"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["value"] > 0
    assert line["em_stats"]["ok_images"] == 12
    assert line["parity"]["images"] == 12 and line["parity"]["all_criteria"] == 12
    assert line["roofline"]["bound"] in ("mfma", "hbm") and 0 < line["roofline"]["frac"] < 1
    assert line["dtype"].startswith("f32 (CNN") and "fp16 PAIRS" in line["dtype"]      # the line says what multiplies ...
    if mode == "lanes":                                      # ... and the exact-operand configuration is reported beside it
        alt = line["alt_exact_operands"]
        assert alt["value"] > 0 and alt["steps"] == 3 and "bf16" in alt["cnn"]
        fl = line["from_lines"]                              # ... and the same steps starting from the line sets:
        assert fl["value"] > 0 and fl["steps"] == 3          # raster (own stream) -> CNN -> EM, pipelined, gives what the
        assert fl["results_equal_unpipelined_pass"] is True  # three stages give one after the other


@pytest.mark.parametrize("ranks", [2, 8])
def test_sharded_benchmark_ranks_sharing_one_gpu(tmp_path, ranks):
    """benchmark.py --hlw --synthetic with TWO / EIGHT ranks (torch.distributed.run, as on an 8-GPU node) that share this
    box's GPU: cost-balanced shards, the real EM on each rank, records gathered (gloo here: RCCL refuses several ranks on
    one device), AUC on rank 0 -- and the same AUC as the one-process run, a clean exit of every rank."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    args = ["--hlw", "--synthetic", "--count", "48" if ranks == 2 else "72", "--update_datafiles", "--run_em"]
    one = subprocess.run([sys.executable, "-m", "vanishing_points_2017_amd.benchmark", "--result_dir", str(tmp_path / "one")] + args,
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    env2 = dict(env, VPK_DIST_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                          "--master-addr", "127.0.0.1", "--master-port", str(29641 + ranks), "-m", "vanishing_points_2017_amd.benchmark",
                          "--result_dir", str(tmp_path / "two")] + args,
                         capture_output=True, text=True, env=env2, timeout=900, cwd=root)
    assert two.returncode == 0, two.stderr[-2000:]
    auc1 = float(re.search(r"AUC:\s+([0-9.eE+-]+)", one.stdout).group(1))
    auc2 = float(re.search(r"AUC:\s+([0-9.eE+-]+)", two.stdout).group(1))
    assert "%d rank(s)" % ranks in two.stdout
    assert auc1 == auc2 and 0.5 < auc1 <= 1.0


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_ranks_sharing_one_gpu(ranks):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, RANK / WORLD_SIZE in the
    environment), two / eight ranks sharing this box's GPU over gloo: image shards per rank, per-step record gather,
    barrier + max-over-ranks timing, ONE JSON line from rank 0 with n_gpus = ranks_seen = N, no 'workloads' part (that
    is the one-GPU default run's), host threads divided among the ranks."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VPK_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                        "--master-addr", "127.0.0.1", "--master-port", str(29651 + ranks), os.path.join(root, "bench.py"),
                        "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--images", "12", "--no-alt"],
                       capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["ranks_seen"] == ranks and line["scaling"] == "weak"
    assert line["config"]["images_per_gpu"] == 12 and line["value"] > 0
    assert "workloads" not in line and "cpu_baseline" not in line
    assert line["host_threads_per_rank"] * ranks <= (os.cpu_count() or 1) or line["host_threads_per_rank"] == 1


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_on_the_rccl_backend_one_rank():
    """The code path the driver's N > 1 runs take, with the backend they take it with: bench.py --force-dist initialises
    torch.distributed with the default backend "nccl" (= RCCL) on this box's one GPU -- process group bound to the device,
    dist.barrier() inside the runtime's stream context, the per-step all_gather of the device-resident records on an EM
    lane's stream, the all_gather of the ranks' elapsed times, RCCL's banner flushed before the JSON line -- as a child
    process.  (Every multi-rank test on this box forces gloo: RCCL refuses several ranks on one device.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VPK_DIST_BACKEND")}
    env["MASTER_ADDR"], env["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "3", "--warmup", "1",
                        "--images", "12", "--no-cpu-baseline", "--no-extra"], capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0]     # ONE JSON line, and it is the last line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["steps"] == 3 and line["value"] > 0
    g = line["gather"]
    assert g["backend"] == "nccl" and g["records_per_step"] == 12 and g["records"] % 12 == 0 and g["image_ids_complete"] is True
    from vanishing_points_2017_amd import sharding
    assert g["width"] == sharding.REC_WIDTH
    assert line["em_stats"]["ok_images"] == 12 and line["parity"]["all_criteria"] == 12
    assert line["alt_exact_operands"]["value"] > 0 and line["from_lines"]["results_equal_unpipelined_pass"] is True   # their gathers too


def test_sharded_benchmark_on_the_rccl_backend_one_rank(tmp_path):
    """benchmark.py --hlw --synthetic launched the way an 8-GPU run launches it (torch.distributed.run) with ONE rank and
    the default nccl backend: process group on the device, barriers, the ragged record gather on GPU tensors
    (sharding.gather_records), AUC on rank 0 -- the same AUC as without torch.distributed, and a clean exit."""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "VPK_DIST_BACKEND")}
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = ["--hlw", "--synthetic", "--count", "40", "--update_datafiles", "--run_em"]
    one = subprocess.run([sys.executable, "-m", "vanishing_points_2017_amd.benchmark", "--result_dir", str(tmp_path / "one")] + args,
                         capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "-m", "vanishing_points_2017_amd.benchmark",
                          "--result_dir", str(tmp_path / "two"), "--force-dist"] + args,
                         capture_output=True, text=True, env=env, timeout=900, cwd=root)
    assert two.returncode == 0, two.stderr[-3000:]
    assert "records gathered over nccl" in two.stdout
    auc1 = float(re.search(r"AUC:\s+([0-9.eE+-]+)", one.stdout).group(1))
    auc2 = float(re.search(r"AUC:\s+([0-9.eE+-]+)", two.stdout).group(1))
    assert auc1 == auc2 and 0.5 < auc1 <= 1.0


def test_pipeline_step_equals_separate_calls():
    """vpk_pipeline_step (CNN -> EM of one batch enqueued by one host call, on two streams) against vpk_cnn_forward
    followed by vpk_em_batch: the same response maps and the same EM outputs, bit for bit; its records are
    sharding.pack_records' (the host form) and sharding.device_records' (the torch form); a ring of two buffer sets
    re-enqueued three times without any host synchronisation in between keeps giving the same bits (the guard event
    orders the reuse on the device), and the resident input lines are never modified."""
    import torch
    from vanishing_points_2017_amd import cnn, em as gem, pipeline, sharding, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    rt_cnn, rt_em = get_runtime(0, "pipe_cnn"), get_runtime(0, "pipe_em")
    scenes = list(synth.config_scenes(2, count=9, start=60))
    net = cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), runtime=rt_cnn)
    params = gem._params({})
    d = gem.upload_batch(rt_em, scenes)
    rt_em.synchronize()
    l0 = d["l"].clone()
    # separate calls
    resp = net.forward_device(d["sphere"])
    rt_cnn.synchronize()
    ref = gem.em_batch_device(rt_em, d["offsets"], l0.clone(), d["lp"], resp.reshape(-1, 400), d["sphere"], None, params)
    rt_em.synchronize()
    ids = torch.arange(100, 100 + len(scenes), dtype=torch.int64, device=rt_em.tdev)
    ring = [pipeline.Step(rt_cnn, rt_em, d, params, l_in=l0, records=True, image_ids=ids) for _ in range(2)]
    for rep in range(3):
        for st in ring:
            st.enqueue()
    rt_cnn.synchronize()
    rt_em.synchronize()
    for st in ring:
        assert torch.equal(st.resp, resp)
        for k in ("vp_assoc", "iterations", "status", "num_vp", "flags"):
            assert torch.equal(st.out[k], ref[k]), k
        m = ref["num_vp"].cpu().numpy()
        for b in range(len(scenes)):
            for k in ("vp", "sigma", "counts", "counts_weighted"):
                assert torch.equal(st.out[k][b, :m[b]], ref[k][b, :m[b]]), (k, b)
        want = sharding.device_records(torch, ids, ref).cpu().numpy()
        got = st.records.cpu().numpy()
        assert np.array_equal(got[:, :-1], want[:, :-1]) and np.isnan(got[:, -1]).all()
        cnn_ms, em_ms = st.stage_ms()
        assert cnn_ms > 0 and em_ms > 0
    assert torch.equal(l0, d["l"])                      # the resident lines are inputs only
    # the asynchronous steps' value-range check (include/vpk.h: vpk_cnn_range_flags): clean here, and an error -- not a silently
    # different response map -- when a layer's activation scale is far too high for these rasters
    ring[0].check_cnn_range()
    from vanishing_points_2017_amd._lib import VpkRangeError
    good = net.activation_scales()
    bad = good.copy()
    bad[2] = good[2] * np.float32(2.0 ** 16)
    try:
        net.set_activation_scales(bad)
        ring[0].enqueue()
        with pytest.raises(VpkRangeError) as ei:
            ring[0].check_cnn_range()
        assert ei.value.flags == 1 << 3                 # conv4's input
        rt_em.synchronize()
        assert bool(torch.isfinite(ring[0].resp).all().item())
    finally:
        net.set_activation_scales(good)
    ring[0].enqueue()
    ring[0].check_cnn_range()
    rt_em.synchronize()
    assert torch.equal(ring[0].resp, resp)
    host = {k: v.cpu().numpy() for k, v in ref.items() if v is not None}
    res = [{"status": int(host["status"][b]), "vp": host["vp"][b, :host["num_vp"][b]], "counts": host["counts"][b, :host["num_vp"][b]]}
           for b in range(len(scenes))]
    packed = sharding.pack_records(ids.cpu().numpy(), res)
    rec = ring[0].records.cpu().numpy()
    assert np.array_equal(rec[:, :3], packed[:, :3])
    for b in range(len(scenes)):                        # same VPs kept; the order among EQUAL counts is each form's own
        a, c = sharding.unpack_record(rec[b]), sharding.unpack_record(packed[b])
        assert np.array_equal(np.sort(a["counts"]), np.sort(c["counts"]))
