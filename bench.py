#!/usr/bin/env python3
"""bench.py -- images/s through the vanishing-point hot path (CNN forward -> EM refinement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload yud|stress]

One "step" = one pass of the hot path over one batch of synthetic images whose inputs (sphere
rasters, line segments) are already resident in HBM: AlexNet-500 forward on the B x 500 x 500
uint8 rasters -> 20 x 20 response maps -> EM refinement of every image (the CNN's own output is
the EM's prior).  Default workload = BASELINE.json configs[1]: YUD-shape, 102 images per GPU,
N ~ U{100..400} lines, 3 VPs (seeded synthetic; random-init weights: the datasets and the trained
caffemodel are not reachable offline).

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment), or -- with no launcher environment -- bench.py
itself starts the N ranks as child processes before anything here touches the GPU and relays rank 0's
line.  Images shard across ranks (weak scaling: 102 images per rank), no data-path collective, one RCCL
all_gather of the fixed-size result records per step (sharding.device_records / gather_device).
Rank 0 prints ONE JSON line.

With no --workload given (the driver's invocation) and one GPU, the line also carries "workloads": a short run of the
HBM-roofline workload (configs[4] stress: 512 images x 1000 lines x 8 VPs x 50 iterations, 3 steps) and one pass over the
2 018-image HLW-shape set (configs[3]: EM + horizon selection, AUC, parity of the stored subsample with the reference).

What is timed: CNN forward -> EM on inputs resident in HBM.  LSD, the rasteriser and the horizon / AUC
stage are not in the timed region.  The workload generator yields LINES only; the resident rasters are made from them
by vpk_sphere_raster before the clock starts, as the reference makes a datum's raster from its lines
(evaluation.py:175) -- and they are the reference's own, pixel for pixel: the "parity" object checks every raster's
hash against the hash of the reference's sphere_line_plot output stored in tests/golden/full_c2.npz, and compares the
EM's results on these rasters (with the generator's response maps, which is what the reference was given) with the
reference's stored results, outside the timed region.  "value" feeds the EM the random-weight CNN's own response maps
(the reference's data flow); "value_fixture_prior" repeats the same K steps -- CNN executed all the same -- with the EM
fed the response maps the parity object uses, i.e. the workload whose parity the line quotes.

How a step is scheduled (--em-mode slice, the default): the CNN of step k runs on its stream; the EM launch of step k (a second
stream, waiting for that forward on the device) holds --em-wgs CUs for at most --em-slice-ms, parks the images it has not finished
and resumes them in the next launch; the last step is followed by a flush that finishes the parked images -- the flush is inside
the timed region (the job is not done before it), so the run's tail (~12 ms: the slowest image of the last batches) is part of
`ms_per_step`.  --em-mode lanes is the round-1..4 scheme (three whole-batch EM launches in flight).

CNN arithmetic (--cnn-algorithm 4, the library's default): f32 operands, f32 accumulation and results; conv2..5, fc6 and fc7 multiply
scaled fp16 PAIRS of the operands (three exact products per f32 product), conv1 exact bf16 pieces -- `dtype` spells it out, and the rule
behind it (error against the float64 net no larger than the f32 direct kernels' at every tap) is tests/test_gpu_cnn.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# one hardware queue per stream (1 CNN + the EM lanes + torch's own): with the default of 4 a fifth stream shares
# a queue, and a CNN kernel can end up queued behind a 20 ms EM kernel.  Must be set before HIP initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: bf16 / fp16 MFMA dense peak
# measured on this pool with operands that change between instructions (random bits; scripts/ubench/mfma_f16_pairs.hip): power-limited
MFMA_SUSTAINED_TFLOPS = {"bf16": 1770.0, "f16": 1675.0}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=None, choices=["yud", "stress"],
                    help="default: yud, followed (one GPU only) by a short stress run and the HLW-shape pass, reported under 'workloads'")
    ap.add_argument("--no-extra", action="store_true", help="skip the 'workloads' part of the default run")
    ap.add_argument("--images", type=int, default=0, help="images per GPU (default: 102 yud / 512 stress)")
    ap.add_argument("--em-mode", default="slice", choices=["lanes", "slice", "serial"],
                    help="slice: one CNN stream + one EM stream whose launches are time-sliced (vpk_em_set_time_slice): a launch "
                         "holds --em-wgs CUs for at most --em-slice-ms, images unfinished by then are parked and resumed by the "
                         "next launch, so no launch waits for a 99-iteration straggler; serial: ONE stream, CNN(k) then a sliced EM "
                         "launch on all CUs; lanes: round 1's scheme, whole EM batches on --em-lanes streams beside the CNN stream")
    ap.add_argument("--em-slice-ms", type=float, default=-1.0, help="time budget of one sliced EM launch (default 2.5 slice / 1.6 serial)")
    ap.add_argument("--em-lanes", type=int, default=3,
                    help="EM batches in flight (HIP streams); the EM of a YUD-size batch fills <half of the CUs")
    ap.add_argument("--em-wgs", type=int, default=-1,
                    help="workgroups (CUs) per EM launch; default: 0.29 x images for yud (the launch lasts as long as its "
                         "slowest image either way, and the CNN keeps the other CUs), one per image for stress")
    ap.add_argument("--cnn-precision", type=int, default=0, choices=[0, 1],
                    help="0: native f32 matrix instructions (default); 1: conv2..5 as six bf16 matrix products per f32 "
                         "product (vpk_cnn_set_precision, same error class; reported as its own dtype)")
    ap.add_argument("--cnn-algorithm", type=int, default=4, choices=[0, 1, 2, 3, 4],
                    help="conv2..5 / fc6: 4 = direct on scaled fp16 pairs of the f32 operands, three exact products per f32 product (default, "
                         "the library's default); 2 = conv2 / fc6 on exact bf16 triples (six products) + conv3..5 Winograd F(2x2,3x3) on the "
                         "f32 matrix cores; 1 = Winograd everywhere (conv2: F(2x2,5x5)); 0 = direct implicit GEMM on the f32 matrix cores "
                         "(vpk_cnn_set_algorithm)")
    ap.add_argument("--cnn-fusion", type=int, default=3, choices=[0, 1, 2, 3, 4],
                    help="conv1 + norm1 + pool1: 3 = direct convolution on the bf16 matrix cores with exact operands (uint8 raster = one "
                         "bf16 piece, weights = three; default), 1 = direct convolution on the f32 matrix cores, 2 = implicit-GEMM "
                         "kernel with the fused epilogue, 0 = separate kernels")
    ap.add_argument("--cnn-priority", type=int, default=0, choices=[-1, 0],
                    help="HIP stream priority of the CNN stream (-1 = high: its kernels' workgroups are dispatched ahead of the EM lanes')")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (RCCL) even with one rank: exercises the N > 1 code path on one GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra timed legs (alt_exact_operands, fixture prior, from the lines)")
    ap.add_argument("--no-from-lines", action="store_true", help="skip the from_lines leg (raster -> CNN -> EM per step)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="images timed on the CPU (default 6 yud / 1 stress)")
    return ap.parse_args()


def visible_gpus():
    """GPUs this process would see, counted WITHOUT touching the HIP runtime (a process that has initialised HIP must
    not start the ranks): KFD's topology nodes with SIMDs, narrowed by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.  None if the
    topology cannot be read (then the children validate the count themselves)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:       # no KFD topology (no GPU driver, or a sandbox): ask a throw-away child process instead
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               capture_output=True, text=True, timeout=300)
            return int(r.stdout.strip().splitlines()[-1])
        except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
            return None
    for f in nodes:
        try:
            props = dict(line.split()[:2] for line in open(f) if len(line.split()) >= 2)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            # a GPU node; containers restrict GPUs by withholding the render node, so count only what can be opened
            minor = props.get("drm_render_minor")
            if minor is None or os.access("/dev/dri/renderD%s" % minor, os.R_OK | os.W_OK):
                n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(args):
    """--gpus N with no launcher environment: run N ranks as CHILD processes (torch.distributed.run, one per
    GPU) and return their exit code.  This parent never touches the GPU: it counts devices from sysfs (visible_gpus),
    not through torch / HIP, and every child checks its own device."""
    import socket
    import subprocess
    have = visible_gpus()
    if have is not None and have < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible\n" % (args.gpus, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    return subprocess.call(cmd, env=env)


def make_workload(kind, rank, count):
    from vanishing_points_2017_amd import synth
    if kind == "yud":
        scenes = list(synth.config_scenes(2, count=count, start=rank * count))
        kw = {}
    else:
        distinct = min(count, 16)
        base = []
        for i in range(distinct):
            s = synth.make_scene(5000 + 16 * rank + i, 1000, 8)
            s["init_vp"] = synth.stress_init_vps(5000 + 16 * rank + i)
            base.append(s)
        from vanishing_points_2017_amd import sphere_mapping
        sphere_mapping.attach_rasters(base)               # (once per distinct scene, not per copy)
        scenes = [base[i % distinct] for i in range(count)]
        kw = dict(num_iter=50, do_split=False, do_merge=False, final_convergence=-1)
    return scenes, kw


def cpu_baseline(scenes, kw, weights, mean, sample):
    """The oracle (numpy EM port + torch-CPU CNN restatement) timed on this host's cores on a
    bounded sample of the same workload.  Reported beside the GPU number, never the target."""
    import torch
    from oracle import cnn_torch, em_numpy
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    sub = scenes[:sample]
    t0 = time.time()
    sphere = np.stack([s["sphere_image"] for s in sub])
    resp = cnn_torch.forward(weights, mean, sphere)
    for s, r in zip(sub, resp):
        try:
            em_numpy.expectation_maximisation(s["l"].copy(), s["lp"].copy(), r.copy(), sphere_image=s["sphere_image"],
                                              init_vp=s.get("init_vp"), **kw)
        except ValueError:
            pass
    dt = time.time() - t0
    return {"value": len(sub) / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d images of the same workload (torch-CPU fp32 CNN + vectorised numpy EM oracle), %.1f s"
                      % (len(sub), dt)}


def reference_parity(rt, gem, scenes, d, l_pristine, params, max_vp, first_index):
    """EM kernel vs the REFERENCE's stored results on the same scenes (configs[1] = config 2), outside the
    timed region.  The timed steps feed the EM with the random-weight CNN's response; the reference saw the
    generator's response map, so this pass runs the same kernel on that input."""
    from vanishing_points_2017_amd import calc_horizon as ch, auc as auc_mod, parity
    if not os.path.isfile(parity.golden_path(2)):
        return None
    ref = parity.ReferenceResults(2)
    l = l_pristine.clone()                          # the EM normalises l in place
    out = gem.em_batch_device(rt, d["offsets"], l, d["lp"], d["cnn"], d["sphere"], None, params, max_vp=max_vp)
    with rt.on_stream():
        rt.handle.em_flush()                        # time-sliced mode: finish what the launch parked (no-op otherwise)
    rt.synchronize()
    host = {k: v.cpu().numpy() for k, v in out.items() if v is not None}
    offs = d["offsets"]
    comps, err_gpu, err_ref, skipped, raster_ok, raster_n = {}, [], [], 0, 0, 0
    sphere_host = d["sphere"].cpu().numpy()             # the resident rasters of the timed run (vpk_sphere_raster)
    for b, sc in enumerate(scenes):
        idx = first_index + b
        if not ref.has(idx):
            continue
        r = ref.get(idx)
        if parity.input_sha(sc) != r["input_sha"]:      # the generator produced other inputs on this host
            skipped += 1
            continue
        if r["raster_sha"] is not None:                 # the reference's own sphere_line_plot output for these lines
            raster_n += 1
            raster_ok += int(parity.raster_sha(sphere_host[b]) == r["raster_sha"])
        m = int(host["num_vp"][b])
        res = {"status": int(host["status"][b]), "iterations": int(host["iterations"][b]),
               "vp_assoc": host["vp_assoc"][offs[b]:offs[b + 1]], "vp": host["vp"][b, :m], "counts": host["counts"][b, :m]}
        comps[idx] = parity.compare_one(res, r)
        if idx >= 25 and r["status"] == 0 and res["status"] == 0:       # benchmark.py:69 skips the first 25 images
            hp = ch.calculate_horizon_and_ortho_vp(res, maxbest=20, theta_vmin=np.pi / 10)
            err_gpu.append(ch.horizon_error(hp[0], hp[1], sc["true_horizon"], sc["image_shape"]))
            err_ref.append(ch.horizon_error(r["hP1"], r["hP2"], sc["true_horizon"], sc["image_shape"]))
    if not comps:
        return {"images": 0, "inputs_differ": skipped}
    out = parity.summarise(comps)
    out["inputs_differ"] = skipped
    out["rasters_equal_reference"] = "%d/%d" % (raster_ok, raster_n)
    out["path"] = ("lines -> vpk_sphere_raster -> vpk_em_batch with the generator's response maps (the inputs the reference "
                   "was given); the same resident rasters feed the timed steps")
    if err_gpu:
        out["horizon_auc"] = float(auc_mod.calc_auc(np.array(err_gpu), cutoff=0.25)[0])
        out["horizon_auc_reference"] = float(auc_mod.calc_auc(np.array(err_ref), cutoff=0.25)[0])
        out["horizon_auc_images"] = len(err_gpu)
    out["reference_seconds_per_image"] = float(np.mean(ref.g["ref_seconds"]))
    out["source"] = ("tests/golden/full_c2.npz: the reference's own sphere_line_plot + EM + calc_horizon on the same %d "
                     "line sets" % len(ref))
    return out


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and int(os.environ.get("RANK", "0")) == 0:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s); reporting n_gpus = %d\n"
                         % (args.gpus, world, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("VPK_DIST_BACKEND", "nccl")    # "gloo": ranks may share a GPU (tests on a one-GPU box)
        local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    else:
        dist = None
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    if world > 1:       # N ranks share one host: divide its cores (torch's intra-op pool, OpenMP / MKL in NumPy)
        torch.set_num_threads(max(1, (os.cpu_count() or 1) // world))
    extra = args.workload is None and not args.no_extra and world == 1
    if args.workload is None:
        args.workload = "yud"
    line = run_workload(args, dist, rank, local_rank, world)
    if rank == 0 and extra:
        line["workloads"] = extra_workloads(args, local_rank)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def hlw_pass(local_rank, n_batches=10, em_wgs=128):
    """BASELINE configs[3] on this GPU: the 2 018 HLW-shape line sets (100..1000 lines) through the whole path as a PIPELINE of
    `n_batches` cost-balanced batches (sharding.shard_balanced on N^2, what the 8-GPU form does across ranks): vpk_sphere_raster on a
    stream of its own -> vpk_pipeline_step (CNN stream, then an EM lane; three lanes) -> one vpk_horizon_batch over all results --
    raster, CNN and EM of consecutive batches overlap.  The line sets and the response maps are resident in HBM before the clock
    starts.  The EM's prior is the generator's response map of each image (the random-weight CNN's output is computed and
    discarded: the AUC is meant to be the fixture workload's).  Then the 64 images the reference's own results are stored for
    (tests/golden/full_c4.npz) as a second small batch from their lines alone, compared with them.  Returns "workloads.hlw"."""
    import torch
    from vanishing_points_2017_amd import _lib, auc as auc_mod, calc_horizon as ch, em as gem, parity, pipeline, sharding, sphere_mapping, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    lanes = [get_runtime(local_rank, "em%d" % i) for i in range(3)]
    rt = lanes[0]
    rt_cnn = get_runtime(local_rank, "cnn")
    rt_r = get_runtime(local_rank, "raster")
    net = get_net(local_rank, rt_cnn)[0]
    net.set_precision(0)
    net.set_profiling(False)
    for r in lanes:
        r.handle.em_set_workgroups(em_wgs)
    t0 = time.time()
    scenes = list(synth.config_scenes(4))
    params = gem._params({})
    n_all = np.array([s["lp"].shape[0] for s in scenes], dtype=np.float64)
    groups = sharding.shard_balanced(n_all ** 2, n_batches)               # index arrays, N^2-balanced
    batches = []
    for k, idx in enumerate(groups):
        sub = [scenes[int(i)] for i in idx]
        d = gem.upload_batch(rt, sub)                     # (also makes the rasters once, outside the timing: compared below)
        with rt_r.on_stream():
            sphere_t = torch.empty_like(d["sphere"])
            ev = torch.cuda.Event(enable_timing=False)
        l_in = d["l"].clone()
        st = pipeline.Step(rt_cnn, lanes[k % len(lanes)], dict(d, sphere=sphere_t), params, l_in=l_in, max_vp=64, timing=False,
                           em_prior=d["cnn"])
        batches.append({"idx": idx, "d": d, "sphere": sphere_t, "ev": ev, "l_in": l_in, "offs": _lib.host_i64(d["offsets"]), "step": st})
    setup_s = time.time() - t0

    def sync_all():
        rt_r.synchronize()
        rt_cnn.synchronize()
        for r in lanes:
            r.synchronize()

    def one_pass():
        """every batch: raster (own stream) -> CNN -> EM, enqueued back to back; the host waits once, at the end"""
        sync_all()
        t = time.perf_counter()
        for b in batches:
            rt_r.stream.wait_event(b["step"].guard)       # (the EM that read this raster buffer last has finished)
            sphere_mapping.raster_batch_device(rt_r, b["l_in"], b["offs"], 500, 0.1, out=b["sphere"])
            b["ev"].record(rt_r.stream)
            rt_cnn.stream.wait_event(b["ev"])
            b["out"] = b["step"].enqueue()
        sync_all()
        return time.perf_counter() - t

    # one untimed pass first: the handles' workspaces (raster pools, CNN arena, EM slots, pinned header buffers) grow to these
    # batches' sizes on first use, then the timed pass
    first_s = one_pass()
    pipe_s = one_pass()
    # each stage of the same work alone (every batch waited for), for the record: what the pipeline overlaps
    def stage_alone(fn):
        sync_all()
        t = time.perf_counter()
        for b in batches:
            fn(b)
        sync_all()
        return time.perf_counter() - t
    raster_s = stage_alone(lambda b: sphere_mapping.raster_batch_device(rt_r, b["l_in"], b["offs"], 500, 0.1, out=b["sphere"]))
    cnn_s = stage_alone(lambda b: net.forward_device(b["sphere"]))

    def em_only(b):
        b["l_tmp"] = b["l_in"].clone()
        lane = b["step"].rt_em
        with lane.on_stream():
            gem.em_batch_device(lane, b["d"]["offsets"], b["l_tmp"], b["d"]["lp"], b["d"]["cnn"], b["sphere"], None, params, max_vp=64)
    em_s = stage_alone(em_only)
    rasters_same = all(bool(torch.equal(b["sphere"], b["d"]["sphere"])) for b in batches)
    # the CNN's maps are not the EM's prior in this pass, but they are computed on 1000-line rasters: assert that the default
    # arithmetic stayed inside its calibrated range on them and produced finite maps (include/vpk.h: vpk_cnn_range_flags)
    cnn_range = net.range_flags()
    cnn_finite = all(bool(torch.isfinite(b["step"].resp).all().item()) for b in batches)
    assert cnn_range == 0 and cnn_finite, "HLW pass: CNN left the fp16-pair range (flags %#x) or produced non-finite maps" % cnn_range
    results = [None] * len(scenes)
    iters = np.zeros(len(scenes)); nvp = np.zeros(len(scenes)); status = np.zeros(len(scenes), dtype=np.int64)
    for b in batches:
        host = {k: v.cpu().numpy() for k, v in b["out"].items() if v is not None}
        assert not (host["status"] == 3).any()
        for j, i in enumerate(b["idx"]):
            m = int(host["num_vp"][j]) if host["status"][j] == 0 else 0
            results[int(i)] = {"vp": host["vp"][j, :m], "counts": host["counts"][j, :m]}
            iters[int(i)], nvp[int(i)], status[int(i)] = host["iterations"][j], host["num_vp"][j], host["status"][j]
    t2 = time.perf_counter()
    horizons = ch.calculate_horizon_batch(results, maxbest=20, theta_vmin=np.pi / 10, device=local_rank)
    hor_s = time.perf_counter() - t2
    errs = np.array([ch.horizon_error(h[0], h[1], s["true_horizon"], s["image_shape"]) for s, h in zip(scenes, horizons)])
    evals = iters + 5
    b_em = float(np.sum(8.0 * n_all ** 2 * (evals + 1) + evals * (64.0 * n_all + 16.0 * np.maximum(nvp, 1) * n_all)))
    total_s = pipe_s + hor_s
    res = {"config": "configs[3] HLW-shape: 2018 images, N~U{100..1000} lines: sphere raster -> CNN -> EM -> horizon selection on one "
                     "GPU (the 8-GPU sharded form is benchmark.py --hlw --synthetic --gpus 8)",
           "images": len(scenes), "images_per_s": len(scenes) / total_s, "batches": len(batches),
           "pipeline_ms": pipe_s * 1e3, "horizon_ms": hor_s * 1e3, "first_pass_ms": first_s * 1e3,
           "stages_alone_ms": {"raster": raster_s * 1e3, "cnn": cnn_s * 1e3, "em": em_s * 1e3, "sum": (raster_s + cnn_s + em_s) * 1e3,
                               "note": "the same batches through one stage at a time (host clock).  Each stage by itself keeps the "
                                       "GPU's CUs busy at this shape (the EM of 2018 images of up to 1000 lines is 30 CU-seconds, the "
                                       "raster 40), so overlapping them fills tails but adds no capacity: pipeline_ms is close to the sum"},
           "em_images_per_s": len(scenes) / em_s,
           "raster_images_per_s": len(scenes) / raster_s, "cnn_images_per_s": len(scenes) / cnn_s,
           "ok_images": int((status == 0).sum()),
           "iterations_mean": float(iters.mean()), "lines_mean": float(n_all.mean()),
           "em_algorithmic_gb": b_em / 1e9,
           "cnn_roofline": hlw_cnn_roofline(net, len(scenes), cnn_s),
           "horizon_auc": float(auc_mod.calc_auc(errs.copy(), cutoff=0.25)[0]),
           "rasters_equal_untimed_pass": rasters_same,
           "cnn_range_check": {"range_flags": cnn_range, "response_maps_finite": cnn_finite},
           "setup_s_outside_timing": setup_s,
           "note": "timed (host clock around the whole pass): per batch vpk_sphere_raster on its own stream -> vpk_pipeline_step (CNN "
                   "stream -> one of three EM lanes, %d workgroups each; priors = the generator's response maps, the CNN's output is "
                   "discarded), all %d batches enqueued back to back and waited for once; then vpk_horizon_batch over the 2018 results; "
                   "not timed: generator, upload of lines / priors" % (em_wgs, len(batches))}
    if os.path.isfile(parity.golden_path(4)):
        ref = parity.ReferenceResults(4)
        stored = [next(synth.config_scenes(4, count=1, start=int(i))) for i in ref.index]
        same = [parity.input_sha(sc) == ref.get(i)["input_sha"] for sc, i in zip(stored, ref.index)]
        got = gem.em_batch(stored, device=local_rank)   # from the lines alone: the rasters are made on the way
        ras_ok = sum(int(ref.get(i)["raster_sha"] is not None and parity.raster_sha(sc["sphere_image"]) == ref.get(i)["raster_sha"])
                     for sc, i in zip(stored, ref.index))
        comps, e_gpu, e_ref = {}, [], []
        for sc, i, r, ok in zip(stored, ref.index, got, same):
            if not ok:
                continue
            g = ref.get(i)
            comps[int(i)] = parity.compare_one(r, g)
            if r["status"] == 0 and g["status"] == 0:
                hp = ch.calculate_horizon_and_ortho_vp(r, maxbest=20, theta_vmin=np.pi / 10)
                e_gpu.append(ch.horizon_error(hp[0], hp[1], sc["true_horizon"], sc["image_shape"]))
                e_ref.append(ch.horizon_error(g["hP1"], g["hP2"], sc["true_horizon"], sc["image_shape"]))
        par = parity.summarise(comps)
        par["inputs_differ"] = int(len(same) - sum(same))
        par["rasters_equal_reference"] = "%d/%d" % (ras_ok, len(stored))
        if e_gpu:
            par["horizon_auc"] = float(auc_mod.calc_auc(np.array(e_gpu), cutoff=0.25)[0])
            par["horizon_auc_reference"] = float(auc_mod.calc_auc(np.array(e_ref), cutoff=0.25)[0])
        par["source"] = "tests/golden/full_c4.npz: the reference's own sphere_line_plot + EM + calc_horizon on %d of the 2018 line sets" % len(ref)
        res["parity"] = par
    return res


def hlw_cnn_roofline(net, images, cnn_s):
    """Whole-net figure of the HLW pass's forward (library defaults): the time the matrix pipes would need at their peaks for what
    the kernels execute / the forward's wall time -- a fraction of a bound, never above 1; the algorithmic rate beside it."""
    pipe_s, ex = 0.0, {"f32": 0.0, "bf16": 0.0, "f16": 0.0}
    for lname in net.LAYER_FLOP:
        f_, p_ = net.executed_flop(lname, batch=101)           # (the forward runs in chunks of ~100 images)
        ex[p_] += f_ * images / 1e12
        pipe_s += f_ * images / ((MFMA_F32_PEAK_TFLOPS if p_ == "f32" else MFMA_BF16_PEAK_TFLOPS) * 1e12)
    return {"bound": "mfma", "frac": pipe_s / cnn_s, "executed_tflop_f32_mfma": ex["f32"], "executed_tflop_bf16_mfma": ex["bf16"],
            "executed_tflop_f16_mfma": ex["f16"],
            "direct_equivalent_tflops": sum(net.LAYER_FLOP.values()) * images / cnn_s / 1e12, "unit": "TFLOP/s",
            "note": "frac = (executed f32-MFMA flops / 157.3 TF + executed bf16- and fp16-MFMA flops / 2500 TF) / the forward's wall time, nothing "
                    "beside it; direct_equivalent_tflops = 6.73 GFLOP per image (SURVEY 8d) x images / the same time"}


def extra_workloads(args, local_rank):
    """The other two BASELINE workloads, after the headline run (one GPU, default invocation only)."""
    import copy
    out = {}
    a = copy.copy(args)
    a.workload, a.steps, a.warmup, a.images = "stress", 3, 1, 0
    a.no_alt = a.no_cpu_baseline = True
    a.em_wgs, a.cnn_precision = -1, 0
    s = run_workload(a, None, 0, local_rank, 1)
    out["stress"] = {k: s[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "config", "stage_ms",
                                       "cnn_layer_ms", "em_stats", "roofline", "cnn_range_check")}
    out["stress"]["roofline_secondary"] = {
        "kernel": s["roofline_secondary"]["kernel"], "achieved": None, "frac": None,
        "note": "not reported for this shape: the EM holds every CU, so the CNN's kernels run parked behind it (in a kernel trace of this run "
                "conv2's launch lasts as long as the EM launch it waits for) and a per-layer rate from the stream's events describes the "
                "moments the layer had the CUs, not the step; the CNN's rates are the headline workload's (roofline / cnn_stream above)"}
    layers = sum(s["cnn_layer_ms"].values())
    out["stress"]["overlap"] = {
        "cnn_layers_sum_ms": layers, "em_kernel_ms": s["stage_ms"]["em"], "step_ms": s["ms_per_step"],
        "note": "no overlap to speak of at this shape: an EM workgroup holds a whole CU (254 VGPRs x 512 threads, the whole LDS) and "
                "512 of them per launch leave no CU idle, so the step is the sum of the two kernels' CU-times (EM launch + the "
                "CNN's ~22 ms for 512 images alone).  The CNN stream waits for CUs the EM lanes hold -- in front of the forward "
                "(stage_ms.cnn, from the forward's enqueue to its end, includes that wait; cnn_layers_sum_ms is the kernels' own time) "
                "or inside whichever layer is in flight when an EM launch starts (a profiled run shifts it there: "
                "profiles/r04_stress_kernel_stats.csv has it in conv2's kernel) -- so roofline_secondary's figure for conv2 is that "
                "layer's rate when it had the CUs, not the CNN's share of the step"}
    out["hlw"] = hlw_pass(local_rank)
    return out


def cnn_algorithm_text(args):
    c1 = {3: "conv1: direct on the bf16 matrix cores, exact operands (uint8 raster = 1 bf16 piece, weights = 3 pieces: 3 products per f32 product)",
          4: "conv1: direct on the fp16 matrix cores (uint8 raster = 1 exact fp16 piece, weights = scaled fp16 pairs: 2 products per f32 product)",
          1: "conv1: direct on v_mfma_f32", 2: "conv1: implicit GEMM on v_mfma_f32", 0: "conv1: implicit GEMM on v_mfma_f32, separate LRN / pool"}[args.cnn_fusion]
    fc = "fc6-8: v_mfma_f32"
    if args.cnn_precision == 1:
        rest = "conv2-5: implicit GEMM on exact bf16 pieces (6 bf16 products per f32 product)"
    elif args.cnn_algorithm == 4:
        rest = ("conv2-5: direct on scaled fp16 PAIRS of the f32 operands (h0 = fp16(s x), h1 = fp16(s x - h0): 22 of 24 significand bits; 3 "
                "exact fp16 products per f32 product; s a power of two per layer; block sums rounded once per kernel row x 16 channels)")
        fc = "fc6, fc7: the same pairs, weights split in registers from the f32 stream; fc8: v_mfma_f32"
    elif args.cnn_algorithm >= 2:
        rest = ("conv2: direct on exact bf16 pieces (3 + 3 pieces, 6 bf16 products per f32 product, block sums rounded once per kernel row x 16 "
                "channels); conv3-5: Winograd F(2x2,3x3) on v_mfma_f32")
        fc = "fc6: exact bf16 pieces (6 products); fc7-8: v_mfma_f32"
    elif args.cnn_algorithm == 1:
        rest = "conv2: Winograd F(2x2,5x5), conv3-5: Winograd F(2x2,3x3), on v_mfma_f32"
    else:
        rest = "conv2-5: implicit GEMM on v_mfma_f32"
    return c1 + "; " + rest + "; " + fc + "; f32 operands, f32 accumulation and f32 results throughout"


def dtype_text(args):
    bf, hf = [], []
    if args.cnn_fusion == 3:
        bf.append("conv1")
    elif args.cnn_fusion == 4:
        hf.append("conv1")
    if args.cnn_precision == 1:
        bf.append("conv2-5")
    elif args.cnn_algorithm == 4:
        hf.append("conv2-5, fc6 and fc7")
    elif args.cnn_algorithm >= 2:
        bf.append("conv2 and fc6")
    if not bf and not hf:
        return "f32 (CNN, MFMA) + f64 (EM)"
    parts = []
    if bf:
        parts.append("%s multiply EXACT bf16 pieces of the f32 operands (three per operand) on the bf16 matrix cores" % " and ".join(bf))
    if hf:
        parts.append("%s multiply scaled fp16 PAIRS of the f32 operands (22 of 24 significand bits, three exact products per f32 product) on "
                     "the fp16 matrix cores" % " and ".join(hf))
    return ("f32 (CNN: f32 operands / accumulation / results; %s -- error against the float64 net no larger than the f32-input direct "
            "kernels' at any tap, measured 2-4 x smaller, tests/test_gpu_cnn.py; the other layers on v_mfma_f32) + f64 (EM)" % "; ".join(parts))


_NETS = {}


def get_net(local_rank, rt_cnn):
    """one weight upload (1.03 GB) per process"""
    from vanishing_points_2017_amd import cnn
    key = (local_rank, id(rt_cnn))
    if key not in _NETS:
        _NETS[key] = (cnn.Net(cnn.synthetic_weights(0), cnn.synthetic_mean(0), device=local_rank, runtime=rt_cnn),
                      cnn.synthetic_weights(0), cnn.synthetic_mean(0))
    return _NETS[key]


def run_workload(args, dist, rank, local_rank, world):
    """K timed steps of one workload; returns the JSON line as a dict (rank 0) or None."""
    import torch
    from vanishing_points_2017_amd import cnn, em as gem, sharding
    from vanishing_points_2017_amd.runtime import get_runtime
    count = args.images or (102 if args.workload == "yud" else 512)
    sliced = args.em_mode in ("slice", "serial") and args.workload == "yud"
    if sliced:
        # Time-sliced EM launches: an image that is still iterating when its launch's budget is spent is parked in
        # HBM and resumed by the next launch, so no launch holds CUs for a 99-iteration straggler.
        #   slice : CNN stream + ONE EM stream; an EM launch holds at most em_wgs CUs for at most the budget, its
        #           workgroups leave as soon as no image is waiting, and the CNN's kernels take every other CU
        #   serial: one stream, CNN(k) on all CUs, then an EM launch on all CUs
        n_lanes = 1
        serial = args.em_mode == "serial"
        # (round 5, CNN at 3.9 ms alone: 96 workgroups x 4 ms 16.4-16.5 k images/s, 128 x 3 16.6 k, 112 x 4 16.2 k, 72 x 5 15.1 k;
        #  whole-batch launches on three lanes 15.6 k -- each lane is held for its slowest image, ~17 ms)
        em_wgs = 0 if serial else (args.em_wgs if args.em_wgs > 0 else max(8, (count * 27) // 17))   # 160 of 102 (scripts/knobs_r5c.sh)
        slice_ms = args.em_slice_ms if args.em_slice_ms > 0 else (1.6 if serial else 2.5)
        rt = get_runtime(local_rank, "main" if serial else "em")
        rt_cnn = rt if serial else get_runtime(local_rank, "cnn")
        rt.handle.em_set_workgroups(em_wgs)
        lanes = [rt]
    else:
        # lanes (library handle + HIP stream each) on the same GPU: one for the CNN, --em-lanes for the EM,
        # used round-robin.  Steps are software-pipelined: the EM of step k (one persistent workgroup per
        # image: 102 of 256 CUs, and a tail of a few images still iterating) overlaps the CNN of step k+1 and
        # the EM of steps k+1, k+2.  All K steps complete inside the timed region.
        n_lanes = max(1, args.em_lanes)
        lanes = [get_runtime(local_rank, "em%d" % i) for i in range(n_lanes)]
        rt = lanes[0]
        rt_cnn = get_runtime(local_rank, "cnn", priority=args.cnn_priority)
        em_wgs = args.em_wgs if args.em_wgs >= 0 else (max(8, (count * 5) // 17) if args.workload == "yud" else 0)   # 30 of 102: measured optimum (26: -1.5 %, 34: -1.5 %)
        for r in lanes:
            r.handle.em_set_workgroups(em_wgs)
    scenes, kw = make_workload(args.workload, rank, count)
    net, weights, mean = get_net(local_rank, rt_cnn)
    net.set_profiling(True)
    net.set_fusion(args.cnn_fusion)
    net.set_precision(args.cnn_precision)
    net.set_algorithm(args.cnn_algorithm)
    params = gem._params(kw)
    d = gem.upload_batch(rt, scenes)                     # inputs resident in HBM before the timed region
    l_pristine = d["l"].clone()
    l_lane = [d["l"]] + [d["l"].clone() for _ in range(n_lanes - 1)]   # EM normalises l in place
    rt.synchronize()
    n_lines = np.diff(d["offsets"])
    max_vp = 64
    if sliced:
        rt.handle.em_set_time_slice(slice_ms, int(n_lines.max()))

    sphere_cnn = d["sphere"]
    image_ids = torch.arange(rank * count, (rank + 1) * count, dtype=torch.int64, device=rt.tdev)
    alive = []                                           # sliced mode: a step's buffers live until the flush

    def step_sliced(k, prior=None, sphere=None):
        """prior: response maps the EM uses instead of this step's CNN output (the CNN runs all the same); sphere: the step's own
        raster buffer (from_lines leg) instead of the resident rasters"""
        sph = sphere_cnn if sphere is None else sphere
        with rt_cnn.on_stream():
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            resp = net.forward_device(sph)               # B x 20 x 20 fp32
            e[1].record()
        with rt.on_stream():
            if rt is not rt_cnn:
                rt.stream.wait_event(e[1])               # EM(k) needs CNN(k)
            l_buf = l_pristine.clone()                   # parked images keep reading their step's inputs
            e[2].record()
            out = gem.em_batch_device(rt, d["offsets"], l_buf, d["lp"], (resp if prior is None else prior).reshape(-1, 400), sph,
                                      d["init_vp"], params, max_vp=max_vp)
            e[3].record()
        alive.append((resp, l_buf, out, sph))
        return e, out

    def finish_sliced():
        """End of a run of steps: finish every parked image; then the one collective of the path."""
        with rt.on_stream():
            rt.handle.em_flush()
            if dist is not None:
                rec = torch.cat([sharding.device_records(torch, image_ids, a[2]) for a in alive], 0)
                alive[-1][2]["records"] = sharding.gather_device(dist, rec)

    # lanes mode: a step is ONE call into the library (vpk_pipeline_step: CNN, stream dependency, copy of the lines, EM
    # launch, gather records -- all enqueued from C++).  A ring of 2 x lanes buffer sets; each set's reuse is ordered on
    # the device (its guard event), never by the host.
    from vanishing_points_2017_amd import pipeline
    ring = []
    if not sliced:
        ring = [pipeline.Step(rt_cnn, lanes[j % n_lanes], d, params, l_in=l_pristine, max_vp=max_vp,
                              records=dist is not None, image_ids=image_ids, timing=False) for j in range(2 * n_lanes)]
    quads = {}
    active = {"ring": ring}                              # which ring of buffer sets step_lanes enqueues (see the fixture-prior leg)

    def step_lanes(k):
        st = active["ring"][k % len(ring)]
        if k not in quads:                               # (steps outside the timed loop; the timed ones are made ahead)
            quads[k] = pipeline.event_quad(rt_cnn, st.rt_em)
        e, handles = quads[k]
        out = st.enqueue(handles)
        if dist is not None:                             # the one collective: gather the result records
            with st.rt_em.on_stream():
                out = dict(out, records=sharding.gather_device(dist, st.records))
        return e, out

    step = step_sliced if sliced else step_lanes

    def sync_all():
        if sliced and alive:
            finish_sliced()
        if dist is not None:
            with rt.on_stream():
                dist.barrier()
        rt_cnn.synchronize()
        for r in lanes:
            r.synchronize()
        torch.cuda.synchronize()
        del alive[:-1]                                   # the last step's outputs feed the statistics below

    for k in range(n_lanes):                             # setup: every lane allocates its workspace once
        step(k)
    sync_all()
    for k in range(args.warmup):
        step(k)
    sync_all()
    if dist is not None:                                 # RCCL writes its banner through C stdio: push it out now,
        import ctypes                                    # so that the JSON line below is the last line on stdout
        ctypes.CDLL(None).fflush(None)
    if not sliced:
        for k in range(args.steps):
            quads[args.warmup + k] = pipeline.event_quad(rt_cnn, ring[(args.warmup + k) % len(ring)].rt_em)
        sync_all()
    net.set_profiling(True)                              # (restarts the per-layer averages: the timed passes only)
    t0 = time.perf_counter()
    evs = []
    call_s = []
    for k in range(args.steps):
        tk = time.perf_counter()
        e, out = step(args.warmup + k)
        call_s.append(time.perf_counter() - tk)
        evs.append(e)
    submit_s = time.perf_counter() - t0
    sync_all()
    elapsed = time.perf_counter() - t0
    layer_ms, layer_passes = net.mean_layer_ms()         # averaged over the timed steps' passes (HIP events on the CNN stream)
    # the default arithmetic's value range (include/vpk.h: vpk_cnn_range_flags): no activation of any forward so far was clamped, and
    # the last step's response maps are finite -- a run that left the calibrated range is not a measurement
    cnn_range = net.range_flags()
    resp_last = alive[-1][0] if sliced else ring[(args.warmup + args.steps - 1) % len(ring)].resp
    cnn_finite = bool(torch.isfinite(resp_last).all().item())
    assert cnn_range == 0 and cnn_finite, "CNN left the fp16-pair range (flags %#x) or produced non-finite maps" % cnn_range
    out = {q: (v.clone() if torch.is_tensor(v) else v) for q, v in out.items()}   # the ring's buffers are reused by the legs below
    rank_elapsed = [elapsed]
    if dist is not None:
        tall = torch.zeros(dist.get_world_size(), dtype=torch.float64, device=rt.tdev)
        dist.all_gather_into_tensor(tall, torch.tensor([elapsed], dtype=torch.float64, device=rt.tdev))
        rank_elapsed = [float(x) for x in tall.cpu()]
        elapsed = max(rank_elapsed)                      # the job's time is its slowest rank's

    cnn_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    em_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
    # The same K steps once more with the EM fed the FIXTURE response maps (the generator's: what the reference was given and
    # what the parity object runs) instead of the random-weight CNN's output.  The CNN is executed all the same (its output is
    # written and left unused): the work of a deployment whose net produces the priors the fixtures stand for.
    fixture = None
    if args.workload == "yud" and not args.no_alt:
        if sliced:
            step_fix = lambda k: step_sliced(k, prior=d["cnn"])
        else:
            ring_fix = [pipeline.Step(rt_cnn, lanes[j % n_lanes], d, params, l_in=l_pristine, max_vp=max_vp, records=dist is not None,
                                      image_ids=image_ids, timing=False, em_prior=d["cnn"]) for j in range(2 * n_lanes)]
            active["ring"] = ring_fix
            step_fix = step
        for k in range(max(args.warmup, n_lanes)):
            step_fix(k)
        sync_all()
        t1 = time.perf_counter()
        for k in range(args.steps):
            e_fix, out_fix = step_fix(args.warmup + k)
        sync_all()
        fix_elapsed = time.perf_counter() - t1
        if dist is not None:
            tmax = torch.tensor([fix_elapsed], dtype=torch.float64, device=rt.tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            fix_elapsed = float(tmax.item())
        it_fix = out_fix["iterations"].cpu().numpy()
        assert not (out_fix["status"] == 3).any().item()
        fixture = {"value": count * world * args.steps / fix_elapsed, "ms_per_step": fix_elapsed / args.steps * 1e3,
                   "iterations_mean": float(it_fix.mean()), "iterations_max": int(it_fix.max())}
        active["ring"] = ring
    # The same K steps once more with EXACT-OPERAND arithmetic only (vpk_cnn_set_algorithm(2), round 5's first default: conv1, conv2
    # and fc6 on exact bf16 triples -- six products per f32 product --, conv3..5 Winograd on the f32 matrix cores, fc7 / fc8 f32).
    # Reported beside the headline number, never as it: `value` is the default run above (fp16 pairs, see `dtype`); a reader who only
    # accepts operands that are represented exactly finds that number here.
    alt = None
    if args.workload == "yud" and args.cnn_precision == 0 and args.cnn_algorithm == 4 and not args.no_alt:
        alt_wgs = em_wgs if sliced else (args.em_wgs if args.em_wgs >= 0 else max(8, (count * 47) // 100))
        net.set_algorithm(2)
        for r in lanes:
            r.handle.em_set_workgroups(alt_wgs)
        for k in range(max(args.warmup, n_lanes)):
            step(k)
        sync_all()
        t1 = time.perf_counter()
        evs_alt = [step(args.warmup + k)[0] for k in range(args.steps)]
        alt_submit = time.perf_counter() - t1
        sync_all()
        alt_elapsed = time.perf_counter() - t1
        if dist is not None:
            tmax = torch.tensor([alt_elapsed], dtype=torch.float64, device=rt.tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            alt_elapsed = float(tmax.item())
        alt = {"cnn": "vpk_cnn_set_algorithm(2): conv1, conv2, fc6 multiply EXACT bf16 pieces (three per f32 operand, six products per f32 "
                      "product), conv3..5 Winograd F(2x2,3x3) on v_mfma_f32, fc7 / fc8 v_mfma_f32 -- every operand represented exactly",
               "value": count * world * args.steps / alt_elapsed, "unit": "images/s",
               "ms_per_step": alt_elapsed / args.steps * 1e3, "steps": args.steps,
               "host_submit_ms_per_step": alt_submit / args.steps * 1e3,
               "stage_ms": {"cnn": float(np.mean([e[0].elapsed_time(e[1]) for e in evs_alt])),
                            "em": float(np.mean([e[2].elapsed_time(e[3]) for e in evs_alt])), "em_workgroups": alt_wgs},
               "cnn_layer_ms": {k: round(v, 4) for k, v in net.last_layer_ms().items()}}
        net.set_algorithm(args.cnn_algorithm)
        for r in lanes:
            r.handle.em_set_workgroups(em_wgs)
    # The same K steps once more STARTING FROM THE LINES: every step first rasterises its batch (vpk_sphere_raster on a stream
    # of its own; the CNN of the step waits for it on the device), i.e. sphere raster -> CNN -> EM, SURVEY 8d's full metric.
    # Reported beside the headline number (whose inputs include the rasters, as the contract's "resident in HBM" says).
    from_lines = None
    if args.workload == "yud" and not args.no_alt and not args.no_from_lines:
        from vanishing_points_2017_amd import sphere_mapping, _lib
        rt_r = get_runtime(local_rank, "raster")
        offs = _lib.host_i64(d["offsets"])
        size_px = int(d["sphere"].shape[-1])
        n_ring = max(len(ring), 2)
        with rt_r.on_stream():
            spheres = [torch.empty_like(d["sphere"]) for _ in range(n_ring)]
            ev_r = [torch.cuda.Event(enable_timing=False) for _ in range(n_ring)]
            ev_t = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ring_r = [] if sliced else [pipeline.Step(rt_cnn, lanes[j % n_lanes], dict(d, sphere=spheres[j]), params, l_in=l_pristine,
                                                  max_vp=max_vp, records=dist is not None, image_ids=image_ids, timing=False)
                                    for j in range(len(ring))]
        rt_r.synchronize()
        ev_t[0].record(rt_r.stream)                      # the raster alone, once (also its workspace allocation and table)
        sphere_mapping.raster_batch_device(rt_r, l_pristine, offs, size_px, 0.1, out=spheres[0])
        rt_r.synchronize()
        ev_t[0].record(rt_r.stream)
        sphere_mapping.raster_batch_device(rt_r, l_pristine, offs, size_px, 0.1, out=spheres[0])
        ev_t[1].record(rt_r.stream)
        rt_r.synchronize()
        raster_alone_ms = ev_t[0].elapsed_time(ev_t[1])
        # the same batch once without any overlap (raster, then CNN, then EM, each waited for): what the pipelined steps must equal
        with rt_cnn.on_stream():
            resp0 = net.forward_device(spheres[0])
        rt_cnn.synchronize()
        with rt.on_stream():
            plain = gem.em_batch_device(rt, d["offsets"], l_pristine.clone(), d["lp"], resp0.reshape(-1, 400), spheres[0],
                                        d["init_vp"], params, max_vp=max_vp)
            rt.handle.em_flush()                         # (time-sliced mode: finish what the launch parked)
        rt.synchronize()
        plain = {q: plain[q].clone() for q in ("iterations", "status", "num_vp", "vp_assoc", "vp")}

        def step_from_lines(k):
            if sliced:                                   # a raster buffer of its own per step: parked images read theirs until the flush
                with rt_r.on_stream():
                    sph = torch.empty_like(d["sphere"])
                    ev = torch.cuda.Event(enable_timing=False)
                sphere_mapping.raster_batch_device(rt_r, l_pristine, offs, size_px, 0.1, out=sph)
                ev.record(rt_r.stream)
                rt_cnn.stream.wait_event(ev)
                return step_sliced(k, sphere=sph)[1]
            j = k % len(ring_r)
            st = ring_r[j]
            rt_r.stream.wait_event(st.guard)             # the EM that read this raster buffer last has finished
            sphere_mapping.raster_batch_device(rt_r, l_pristine, offs, size_px, 0.1, out=spheres[j])
            ev_r[j].record(rt_r.stream)
            rt_cnn.stream.wait_event(ev_r[j])
            o = st.enqueue()
            if dist is not None:
                with st.rt_em.on_stream():
                    o = dict(o, records=sharding.gather_device(dist, st.records))
            return o

        def sync_lines():
            rt_r.synchronize()
            sync_all()

        for k in range(max(args.warmup, len(ring_r), 2)):
            step_from_lines(k)
        sync_lines()
        t1 = time.perf_counter()
        for k in range(args.steps):
            o_lines = step_from_lines(args.warmup + k)
        sync_lines()
        fl_elapsed = time.perf_counter() - t1
        if dist is not None:
            tmax = torch.tensor([fl_elapsed], dtype=torch.float64, device=rt.tdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            fl_elapsed = float(tmax.item())
        from_lines = {"path": "vpk_sphere_raster (own stream) -> CNN -> EM per step; the step's inputs are the line sets only",
                      "value": count * world * args.steps / fl_elapsed, "unit": "images/s",
                      "ms_per_step": fl_elapsed / args.steps * 1e3, "steps": args.steps,
                      "raster_alone_ms": raster_alone_ms, "raster_alone_images_per_s": count / raster_alone_ms * 1e3,
                      "note": "same inputs as the headline run: there the rasters of these line sets were made by the same "
                              "vpk_sphere_raster before the clock started, here every step makes them again",
                      "results_equal_unpipelined_pass": bool(all(torch.equal(o_lines[q], plain[q]) for q in plain)),
                      "results_equal_headline_run": bool(all(torch.equal(o_lines[q], out[q]) for q in plain))}
    assert not (out["status"] == 3).any().item(), "an image found no working-set slot (VPK_EM_NO_SLOT): results incomplete"
    iters = out["iterations"].cpu().numpy()
    status = out["status"].cpu().numpy()
    nvp = out["num_vp"].cpu().numpy()

    if rank == 0:
        total_images = count * world * args.steps
        value = total_images / elapsed
        # ---- roofline of the dominant kernel (live HIP-event timings from this run) ----
        def newest_profile(suffix):
            """profiles/rNN_<suffix> of the latest round that has one (measured with rocprofv3, scripts/profile_round.sh)."""
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
            if not files:
                return {}, None
            try:
                with open(files[-1]) as fh:
                    return json.load(fh), os.path.relpath(files[-1], ROOT)
            except (OSError, ValueError):
                return {}, None
        traffic, traffic_src = newest_profile("pmc_traffic.json")   # HBM bytes per launch: --pmc FETCH_SIZE / WRITE_SIZE passes
        mfma, mfma_src = newest_profile("pmc_mfma.json")            # SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE passes
        m_avg = np.maximum(nvp, 1)
        evals = iters + 1 + 4                            # loop E-steps + initial + finalisation (:344,:398,:415 and the merge/prune ones: >= 4)
        # EM batch kernel (one launch per step): algorithmic bytes B_EM of SURVEY 8d,
        # 8 N^2 (I+1) + I (64 N + 16 M N) per image with I = E-step evaluations
        b_em = float(np.sum(8.0 * n_lines ** 2 * (evals + 1) + evals * (64.0 * n_lines + 16.0 * m_avg * n_lines)))
        key = "yud_102" if (args.workload == "yud" and count == 102) else (
            "stress_512x1000x8x50" if (args.workload == "stress" and count == 512) else None)
        t = traffic.get(key) if key else None
        roof_em = {"kernel": "em_batch_kernel", "bound": "hbm", "achieved": b_em / (em_ms * 1e-3) / 1e9,
                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "traffic": (t["hbm_read_bytes"] + t["hbm_write_bytes"]) if t else None, "traffic_source": traffic_src,
                   "evaluations_I": float(evals.mean()),
                   "note": "achieved = modelled algorithmic bytes (SURVEY 8d B_EM with I = E-step evaluations per image = loop "
                           "iterations + 1 initial + 4 in the finalisation; evaluations_I = its mean; SURVEY's figure for the stress "
                           "unit takes I = 50, this run's is 54) / mean kernel time of the timed steps, measured on overlapped streams; "
                           "traffic = HBM bytes per launch from the PMC passes"}
        # ---- the CNN stream's dominant kernel: conv2's (the layer with the most arithmetic and the longest kernel).  `achieved` and
        #      `frac` count what the matrix pipes EXECUTE (tile padding, Winograd's product count, the six bf16 products per f32
        #      product included) against the peak of the instruction type the kernel issues -- a distance to that kernel's own bound,
        #      never above 1; the layer's algorithmic rate (2 x MACs of the direct convolution) is `direct_equivalent_tflops`
        name = max(cnn.Net.LAYER_FLOP, key=lambda k: cnn.Net.LAYER_FLOP[k])
        setting = dict(fusion=args.cnn_fusion, precision=args.cnn_precision, algorithm=args.cnn_algorithm, batch=count)
        peak_of = {"f32": MFMA_F32_PEAK_TFLOPS, "bf16": MFMA_BF16_PEAK_TFLOPS, "f16": MFMA_BF16_PEAK_TFLOPS}
        ex_flop, pipe = cnn.Net.executed_flop(name, **setting)
        if args.cnn_precision == 1:
            kname, klabel = "conv_gemm_split_kernel", "conv_gemm_split_kernel(conv2)"
        elif args.cnn_algorithm >= 2:
            kname, klabel = "conv_pieces_kernelILi5E", "conv_pieces_kernel<5>(conv2)"
        elif args.cnn_algorithm == 1:
            kname, klabel = "conv5x5_winograd_kernel", "conv5x5_winograd_kernel(conv2)"
        else:
            kname, klabel = "conv_gemm_dma_kernelILi2ELi2ELi2ELi2ELb0", "conv_gemm_dma_kernel<2,2,2,2>(conv2)"
        roof_cnn = {"kernel": klabel, "bound": "mfma",
                    "matrix_instruction": {"bf16": "v_mfma_f32_32x32x16_bf16", "f16": "v_mfma_f32_32x32x16_f16", "f32": "v_mfma_f32_32x32x2_f32"}[pipe],
                    "sustained_peak_with_changing_operands": MFMA_SUSTAINED_TFLOPS.get(pipe),
                    "achieved": ex_flop * count / (layer_ms[name] * 1e-3) / 1e12, "peak": peak_of[pipe], "unit": "TFLOP/s", "traffic": None,
                    "direct_equivalent_tflops": cnn.Net.LAYER_FLOP[name] * count / (layer_ms[name] * 1e-3) / 1e12,
                    "executed_gflop_per_launch": ex_flop * count / 1e9,
                    "note": "achieved = matrix-core flops the kernel executes per launch (cnn.Net.executed_flop: %s) / the layer's mean duration "
                            "over the timed steps (HIP events on the CNN stream, %d passes; includes the layer's input conversion kernel), "
                            "i.e. with whatever ran beside it; peak = dense peak of that instruction type; direct_equivalent_tflops = the "
                            "layer's algorithmic flops (2 x MACs of the direct f32 convolution, SURVEY 8d) over the same time"
                            "; sustained_peak_with_changing_operands = what the matrix cores hold with operands that change between "
                            "instructions (scripts/ubench/mfma_f16_pairs.hip, profiles/r05_mfma_sustained.txt: the chip drops to ~1.6-1.7 GHz under "
                            "that load) -- the kernel's practical ceiling, reported beside the nominal peak that `frac` is priced against"
                            % ({"bf16": "6 bf16 products per f32 product on 4 x 32-pixel tiles", "f16": "3 fp16 products per f32 product on 4 x 32-pixel tiles",
                                "f32": "f32 products"}[pipe], layer_passes)}
        roof_cnn["frac"] = roof_cnn["achieved"] / roof_cnn["peak"]
        tc = traffic.get(key + "_conv2") if key else None        # HBM bytes per launch of THIS kernel (--pmc FETCH_SIZE / WRITE_SIZE passes)
        if tc and tc.get("kernel", "") in klabel:
            roof_cnn["traffic"] = tc["hbm_read_bytes"] + tc["hbm_write_bytes"]
        roof_cnn["traffic_source"] = traffic_src
        # the same kernel without the EM beside it: kernel trace of the CNN alone (scripts/time_cnn.py; profiles/rNN_cnn_kernel_stats.csv)
        import csv
        import glob
        traces = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_cnn_kernel_stats.csv")))
        if traces and count == 102 and args.cnn_precision == 0:
            try:
                rows = list(csv.DictReader(open(traces[-1])))
                avg_us = [float(r["AverageUs"]) for r in rows if kname in r["Name"]]
                if avg_us:
                    a_tf = ex_flop * count / (avg_us[0] * 1e-6) / 1e12
                    roof_cnn["alone"] = {"achieved": a_tf, "unit": "TFLOP/s", "frac": a_tf / peak_of[pipe], "kernel_us": avg_us[0],
                                         "source": os.path.relpath(traces[-1], ROOT),
                                         "note": "average launch of the kernel in the kernel trace of the CNN alone (executed flops, as above)"}
            except (OSError, ValueError, KeyError):
                pass
        pm = None
        if args.workload == "yud" and args.cnn_precision == 0:
            for k_, v_ in (mfma.get("cnn_alone_B102") or {}).items():
                if kname.split("IL")[0] in k_ and "conv2" in k_:
                    pm = v_
        if pm:
            roof_cnn["pmc_serialised_pass"] = {"mfma_util": pm["mfma_util"], "shader_clock_ghz": pm["shader_clock_ghz"], "source": mfma_src,
                                               "note": "SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles and GRBM_GUI_ACTIVE / duration from a rocprofv3 --pmc pass of "
                                                       "the CNN alone (dispatches serialised)"}
        # the whole CNN stream: time the matrix pipes would need at their peaks for what the net executes, against the stream's period
        pipe_ms = 0.0
        ex_tf = {"f32": 0.0, "bf16": 0.0, "f16": 0.0}
        for lname in cnn.Net.LAYER_FLOP:
            f_, p_ = cnn.Net.executed_flop(lname, **setting)
            ex_tf[p_] += f_ * count / 1e12
            pipe_ms += f_ * count / (peak_of[p_] * 1e12) * 1e3
        cnn_stream = {"period_ms": cnn_ms, "executed_tflop_f32_mfma": ex_tf["f32"], "executed_tflop_bf16_mfma": ex_tf["bf16"],
                      "executed_tflop_f16_mfma": ex_tf["f16"], "matrix_pipe_ms_at_peak": pipe_ms, "frac": pipe_ms / cnn_ms,
                      "note": "frac = (executed f32-MFMA flops / 157.3 TF + executed bf16- and fp16-MFMA flops / 2500 TF) / the CNN stream's period per step "
                              "(HIP events around the forward on its stream, beside the EM): the share of the period the matrix pipes would be "
                              "busy at their peak rates"}
        # dominant = the kernel on the stream that bounds the step: the CNN stream runs one forward per step, each EM
        # stream one batch every n_lanes steps; whichever takes longer per step
        roof = roof_em if (em_ms / n_lanes >= cnn_ms or args.workload == "stress") else roof_cnn
        roof["frac"] = roof["achieved"] / roof["peak"]
        roof["dominant_rule"] = ("EM launch %.2f ms / %d lanes = %.2f ms per step vs CNN stream %.2f ms per step"
                                 % (em_ms, n_lanes, em_ms / n_lanes, cnn_ms))
        line = {
            "metric": "images/sec (LSD->CNN->EM) + horizon-AUC parity, YUD-shape, 1/2/4/8 GPU" if args.workload == "yud"
                      else "images/sec (CNN->EM), synthetic stress, 1->8 GPU roofline scan",
            "metric_note": "BASELINE.json's metric name; the timed region is CNN forward -> EM refinement with inputs resident "
                           "in HBM (no LSD, no rasteriser); horizon-AUC parity is the 'parity' object, outside the timed region",
            "value": value, "unit": "images/s", "n_gpus": world, "ranks_seen": world if dist is None else dist.get_world_size(),
            "ms_per_step_per_rank": {"min": min(rank_elapsed) / args.steps * 1e3, "max": max(rank_elapsed) / args.steps * 1e3,
                                     "all": [round(x / args.steps * 1e3, 4) for x in rank_elapsed],
                                     "note": "every rank's own wall time of the K timed steps / K: a skewed rank shows here; "
                                             "ms_per_step and value use the max"},
            "rank_elapsed_ms": {"min": min(rank_elapsed) * 1e3, "max": max(rank_elapsed) * 1e3,
                                "note": "wall time of the K timed steps per rank (barrier + synchronize on both sides); "
                                        "value uses the max"},
            "steps": args.steps, "warmup": args.warmup, "host_threads_per_rank": torch.get_num_threads(),
            "ms_per_step": elapsed / args.steps * 1e3, "host_submit_ms_per_step": submit_s / args.steps * 1e3,
            "host_enqueue_ms_per_step": float(np.median(call_s)) * 1e3,
            "host_enqueue_unblocked_ms": float(np.min(call_s)) * 1e3,
            "host_note": "host_enqueue = median duration of a step's enqueue call (one vpk_pipeline_step in lanes mode), "
                         "host_enqueue_unblocked = the shortest one (a call that did not have to wait for a free header "
                         "buffer: the host's own work per step); "
                         "host_submit = wall time of the whole submit loop / steps, which includes flow control: "
                         "vpk_em_batch keeps four pinned header buffers per handle and waits for the oldest launch's "
                         "header copy once five launches are queued on a handle",
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "cnn_algorithm": cnn_algorithm_text(args),
            "dtype": dtype_text(args),
            "data": "synthetic (seeded line sets of the config's shape; their rasters made by vpk_sphere_raster = the reference's "
                    "own, hash-checked in 'parity'; random-init AlexNet-500 weights)",
            "config": {"workload": "configs[1] YUD-shape: %d images/GPU, N~U{100..400} lines, 3 VPs, CNN+EM"
                                   % count if args.workload == "yud" else
                                   "configs[4] stress: %d of 10 000 images per launch and GPU (%d distinct scenes) x 1000 lines x 8 VP "
                                   "candidates x 50 EM iterations" % (count, min(count, 16)),
                       "images_per_gpu": count, "parallelism": "image-sharded x%d" % world},
            "stage_ms": ({"cnn": cnn_ms, "em_slice": em_ms, "em_mode": args.em_mode, "em_slice_budget_ms": slice_ms,
                          "em_workgroups": em_wgs,
                          "note": "time-sliced EM launches: images unfinished at the end of a launch are parked and resumed by "
                                  "the next one, the last step is followed by a flush; slice = CNN stream + one EM stream, "
                                  "serial = one stream"}
                         if sliced else
                         {"cnn": cnn_ms, "em": em_ms, "em_mode": "lanes", "em_lanes": n_lanes, "em_workgroups": em_wgs,
                          "note": "stages of consecutive steps overlap (1 CNN stream + em_lanes EM streams)"}),
            "cnn_range_check": {"range_flags": cnn_range, "response_maps_finite": cnn_finite,
                                "note": "vpk_cnn_range_flags over every forward up to the end of the timed steps (0 = no scaled fp16-pair "
                                        "activation was clamped); asserted"},
            "cnn_layer_ms": {k: round(v, 4) for k, v in layer_ms.items()},
            "cnn_layer_ms_note": "mean over the %d timed steps' passes (HIP events between the layers on the CNN stream)" % layer_passes,
            "em_stats": {"iterations_mean": float(iters.mean()), "iterations_max": int(iters.max()),
                         "ok_images": int((status == 0).sum()), "lines_mean": float(n_lines.mean())},
            "roofline": roof,
            "roofline_secondary": roof_cnn if roof is roof_em else roof_em,
            "cnn_stream": cnn_stream,
        }
        if fixture:
            line["value_fixture_prior"] = fixture["value"]
            line["fixture_prior"] = dict(fixture, unit="images/s", steps=args.steps,
                                         note="the same K steps with the CNN executed and the EM fed the generator's response maps "
                                              "(the priors the 'parity' object and the reference's stored results use) instead of "
                                              "the random-weight CNN's output, which 'value' uses (em_stats: its iteration counts)")
        if dist is not None:            # the one collective of the path, as this run executed it
            rec = out.get("records")
            g_info = {"backend": dist.get_backend(), "records": None, "width": None, "image_ids_complete": None}
            if rec is not None:
                # lanes mode: one all_gather per step (world x B rows); slice mode: one at the flush that ends a run of steps, with the
                # records of every step still alive (world x steps x B rows, rank-major)
                per_rank = int(rec.shape[0]) // world
                g_steps = per_rank // count
                ids = rec[:, 0].to(torch.int64).cpu().reshape(world, g_steps, count)
                want = (torch.arange(world, dtype=torch.int64)[:, None, None] * count + torch.arange(count, dtype=torch.int64)[None, None, :])
                g_info.update(records=int(rec.shape[0]), width=int(rec.shape[1]), steps_in_gather=g_steps, records_per_step=count * world,
                              image_ids_complete=bool(torch.equal(ids, want.expand(world, g_steps, count))))
            g_info["note"] = ("slice mode: ONE all_gather of the ranks' fixed-size result records at the flush that ends the run (the records of "
                              "every step whose buffers are alive); lanes mode: one per step on the EM lane's stream (sharding.gather_device)")
            line["gather"] = g_info
        if alt:
            line["alt_exact_operands"] = alt
        if from_lines is not None:
            line["from_lines"] = from_lines
        if args.workload == "yud":
            line["parity"] = reference_parity(rt, gem, scenes, d, l_pristine, params, max_vp, rank * count)
        if not args.no_cpu_baseline and world == 1:          # the CPU baseline is timed on rank 0 at N = 1 only
            sample = args.cpu_sample or (6 if args.workload == "yud" else 1)
            line["cpu_baseline"] = cpu_baseline(scenes, kw, weights, mean, sample)
            if line.get("parity") and line["parity"].get("reference_seconds_per_image"):
                # the reference's own loop code, timed where it can run (build container, 8 cores, joblib pools)
                line["cpu_baseline"]["reference_images_per_s"] = 1.0 / line["parity"]["reference_seconds_per_image"]
                line["cpu_baseline"]["reference_note"] = ("the reference's EM itself (Python loops + joblib, 8-core build "
                                                          "container, CNN excluded) on the same 102 scenes; the 'port' is "
                                                          "this repo's vectorised oracle on this host")
        return line
    return None


if __name__ == "__main__":
    main()
