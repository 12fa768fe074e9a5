"""Time-sliced EM launches (vpk_em_set_time_slice / vpk_em_flush): images suspended at a launch's deadline
and resumed by later launches must give bit-identical results to uninterrupted runs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host(out):
    return {k: v.cpu().numpy() for k, v in out.items() if v is not None}


@pytest.mark.parametrize("slice_ms,wgs", [(0.3, 0), (1.5, 0), (0.2, 24)])
def test_sliced_batches_equal_uninterrupted_batches(slice_ms, wgs):
    import torch
    from vanishing_points_2017_amd import em as gem, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0, "slice%d" % wgs)
    rt.handle.em_set_workgroups(wgs)
    scenes = list(synth.config_scenes(2, count=60, start=40)) + [next(synth.config_scenes(2, count=1, start=12))]
    p = gem._params({})
    d = gem.upload_batch(rt, scenes)
    l0 = d["l"].clone()
    rt.handle.em_set_time_slice(0.0)
    ref = gem.em_batch_device(rt, d["offsets"], l0.clone(), d["lp"], d["cnn"], d["sphere"], None, p)
    rt.synchronize()
    ref = _host(ref)
    assert ref["iterations"].max() == 99                  # the straggler is in the batch
    rt.handle.em_set_time_slice(slice_ms, int(np.diff(d["offsets"]).max()))
    keep = []
    for step in range(6):                                 # six calls in flight before the flush (24 workgroups: overload,
                                                          # images wait unstarted across launches)
        lb = l0.clone()
        keep.append((lb, gem.em_batch_device(rt, d["offsets"], lb, d["lp"], d["cnn"], d["sphere"], None, p)))
    with rt.on_stream():
        rt.handle.em_flush()
    rt.synchronize()
    for lb, out in keep:
        got = _host(out)
        for k in ("status", "iterations", "num_vp", "vp_assoc", "flags"):
            assert np.array_equal(got[k], ref[k]), k
        for b in range(len(scenes)):
            m = int(ref["num_vp"][b])
            for k in ("vp", "sigma", "counts", "counts_weighted"):
                assert np.array_equal(got[k][b, :m], ref[k][b, :m]), (k, b)
        assert torch.equal(lb, keep[0][0])
    rt.handle.em_set_time_slice(0.0)
    again = gem.em_batch_device(rt, d["offsets"], l0.clone(), d["lp"], d["cnn"], d["sphere"], None, p)
    rt.synchronize()
    again = _host(again)
    assert np.array_equal(again["vp_assoc"], ref["vp_assoc"])


def test_layout_change_while_images_are_parked_is_refused():
    from vanishing_points_2017_amd import _lib, em as gem, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    rt = get_runtime(0, "slice_guard")
    small = [next(synth.config_scenes(2, count=1, start=12))]
    big = [synth.make_scene(9, 700, 3)]
    p = gem._params({})
    rt.handle.em_set_time_slice(0.05, 0)
    ds, db = gem.upload_batch(rt, small), gem.upload_batch(rt, big)
    out = gem.em_batch_device(rt, ds["offsets"], ds["l"], ds["lp"], ds["cnn"], ds["sphere"], None, p)
    with pytest.raises(_lib.VpkError):
        gem.em_batch_device(rt, db["offsets"], db["l"], db["lp"], db["cnn"], db["sphere"], None, p)
    with rt.on_stream():
        rt.handle.em_flush()
    rt.synchronize()
    assert int(out["iterations"].cpu()[0]) == 99
    rt.handle.em_set_time_slice(0.0)


def test_full_parked_lists_make_images_run_on_instead_of_overflowing(monkeypatch):
    """The device-side lists of parked images have a capacity (8192 not-yet-started, 1024 suspended images).  With the
    capacities shrunk to 3 and 2 (VPK_EM_WAIT_CAP / VPK_EM_STARTED_CAP, read at vpk_create) a 40-image batch on 6
    workgroups with a 0.2 ms budget overflows both: images that find the waiting list full start anyway and run to
    completion (no deadline: they must not take a slot and suspend at once), images that find the suspended list full
    are resumed on the spot.  Nothing is lost, nothing is written past a list, the launch ends, and every result
    equals the uninterrupted one."""
    from vanishing_points_2017_amd import em as gem, synth
    from vanishing_points_2017_amd.runtime import get_runtime
    monkeypatch.setenv("VPK_EM_WAIT_CAP", "3")
    monkeypatch.setenv("VPK_EM_STARTED_CAP", "2")
    rt = get_runtime(0, "slice_caps")
    monkeypatch.delenv("VPK_EM_WAIT_CAP")
    monkeypatch.delenv("VPK_EM_STARTED_CAP")
    rt.handle.em_set_workgroups(6)
    scenes = list(synth.config_scenes(2, count=39, start=10)) + [next(synth.config_scenes(2, count=1, start=12))]
    p = gem._params({})
    d = gem.upload_batch(rt, scenes)
    l0 = d["l"].clone()
    rt.handle.em_set_time_slice(0.0)
    ref = gem.em_batch_device(rt, d["offsets"], l0.clone(), d["lp"], d["cnn"], d["sphere"], None, p)
    rt.synchronize()
    ref = _host(ref)
    assert ref["iterations"].max() == 99
    rt.handle.em_set_time_slice(0.2, int(np.diff(d["offsets"]).max()))
    keep = []
    for step in range(3):
        lb = l0.clone()
        keep.append((lb, gem.em_batch_device(rt, d["offsets"], lb, d["lp"], d["cnn"], d["sphere"], None, p)))
    with rt.on_stream():
        rt.handle.em_flush()
    rt.synchronize()
    rt.handle.em_set_time_slice(0.0)
    for lb, out in keep:
        got = _host(out)
        assert (got["status"] == ref["status"]).all() and (got["status"] != 3).all()
        for k in ("iterations", "num_vp", "vp_assoc", "flags"):
            assert np.array_equal(got[k], ref[k]), k
        for b in range(len(scenes)):
            m = int(ref["num_vp"][b])
            assert np.array_equal(got["vp"][b, :m], ref["vp"][b, :m])
