"""One pipeline step -- CNN forward -> EM refinement of one batch -- enqueued by ONE call into the library
(C-ABI: vpk_pipeline_step), for pipelines that overlap consecutive batches on two streams.

The reference runs run_cnn over the whole data set and then run_em (evaluation.py:254-329), image by image.  Here a
`Step` owns the device buffers of one batch in flight (response maps, working copy of the lines, every EM output, the
gather records); `Step.enqueue()` costs the host one ctypes call: the CNN, the stream dependency, the copy of the
lines, the EM launch and the record build are all enqueued from C++."""
import ctypes

import numpy as np

from . import _lib

MAX_VP = 64


class Step(object):
    """Buffers of one batch in flight on (rt_cnn, rt_em).  ``d``: em.upload_batch() of the batch (resident inputs);
    ``l_in``: the pristine lines (device, sum N x 3).  ``records=True`` also builds sharding-layout records.
    ``em_prior``: B x 400 response maps (device) the EM uses instead of this step's CNN output (the CNN still runs)."""

    def __init__(self, rt_cnn, rt_em, d, params, l_in=None, max_vp=MAX_VP, records=False, image_ids=None, timing=True,
                 em_prior=None):
        torch = rt_em.torch
        self.rt_cnn, self.rt_em = rt_cnn, rt_em
        self.offsets = _lib.host_i64(d["offsets"])
        batch = self.offsets.shape[0] - 1
        total = int(self.offsets[-1])
        dev = rt_em.tdev
        self.params = params
        self.l_in = d["l"] if l_in is None else l_in
        with torch.cuda.device(dev):
            self.resp = torch.empty((batch, 20, 20), dtype=torch.float32, device=dev)
            self.l_work = torch.empty_like(self.l_in)
            self.out = {
                "vp": torch.empty((batch, max_vp, 3), dtype=torch.float64, device=dev),
                "sigma": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
                "counts": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
                "counts_weighted": torch.empty((batch, max_vp), dtype=torch.float64, device=dev),
                "num_vp": torch.empty((batch,), dtype=torch.int32, device=dev),
                "vp_assoc": torch.empty((max(total, 1),), dtype=torch.int64, device=dev),
                "iterations": torch.empty((batch,), dtype=torch.int32, device=dev),
                "status": torch.empty((batch,), dtype=torch.int32, device=dev),
                "flags": torch.empty((batch,), dtype=torch.int32, device=dev),
            }
            self.records = None
            if records:
                self.records = torch.empty((batch, int(rt_em.lib.vpk_record_width())), dtype=torch.float64, device=dev)
                self.image_ids = image_ids if image_ids is not None else torch.arange(batch, dtype=torch.int64, device=dev)
            # guards the buffers: recorded behind every step's EM, waited for by the next step's CNN (on the device)
            self.guard = torch.cuda.Event(enable_timing=False)
            self.guard.record(rt_em.stream)
            self.events = None
            self._ev = None
            if timing:
                self.events = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                for k, e in enumerate(self.events):                      # materialise the hipEvent_t handles
                    e.record(rt_cnn.stream if k < 2 else rt_em.stream)
                self._ev = (ctypes.c_void_p * 4)(*[e.cuda_event for e in self.events])
        ptr = rt_em.ptr
        o = self.out
        init = d.get("init_vp")
        self._args = _lib.StepArgs(
            sphere=ptr(d["sphere"]).value, batch=batch, sphere_size=int(d["sphere"].shape[-1]), cnn_out=ptr(self.resp).value,
            offsets=self.offsets.ctypes.data, l_in=ptr(self.l_in).value, l_work=ptr(self.l_work).value, lp=ptr(d["lp"]).value,
            init_vp=None if init is None else ptr(init).value, n_init=0 if init is None else int(init.shape[-2]),
            max_vp=max_vp, params=ctypes.addressof(params), vp_out=ptr(o["vp"]).value, sigma_out=ptr(o["sigma"]).value,
            counts_out=ptr(o["counts"]).value, counts_w_out=ptr(o["counts_weighted"]).value,
            num_vp_out=ptr(o["num_vp"]).value, assoc_out=ptr(o["vp_assoc"]).value, iterations_out=ptr(o["iterations"]).value,
            status_out=ptr(o["status"]).value, flags_out=ptr(o["flags"]).value,
            records=None if self.records is None else ptr(self.records).value,
            image_ids=None if self.records is None else ptr(self.image_ids).value,
            events=None if self._ev is None else ctypes.addressof(self._ev), reuse_event=self.guard.cuda_event,
            em_prior=None if em_prior is None else ptr(em_prior).value)
        self._keep = (d, init, em_prior)

    def enqueue(self, events=None):
        """CNN(batch) on rt_cnn's stream, EM(batch) on rt_em's stream behind it.  Asynchronous.  ``events``: a
        (c_void_p * 4) of hipEvent_t handles to record instead of the step's own (see event_quad)."""
        if events is not None:
            self._args.events = ctypes.addressof(events)
        self.rt_em.check(self.rt_em.lib.vpk_pipeline_step(self.rt_cnn.h, self.rt_em.h, ctypes.byref(self._args)))
        return self.out

    def check_cnn_range(self):
        """The steps enqueued so far ran the CNN asynchronously: nothing has looked at its value range yet (include/vpk.h:
        vpk_cnn_range_flags).  Waits for the CNN stream and raises VpkRangeError if a scaled fp16-pair activation of ANY forward on
        that handle since the last check was clamped -- the response maps (and the EM results refined on them) of those steps are then
        not the net's.  A pipeline calls this once per run, before it trusts the run's results."""
        w = ctypes.c_uint32(0)
        rc = self.rt_cnn.lib.vpk_cnn_range_flags(self.rt_cnn.h, ctypes.byref(w))
        if rc == -6:
            raise _lib.VpkRangeError("libvpk error -6: %s" % self.rt_cnn.lib.vpk_last_error(self.rt_cnn.h).decode(), int(w.value))
        self.rt_cnn.check(rc)

    def stage_ms(self):
        """(CNN ms, EM ms) of the last enqueue(), once both streams have passed it."""
        e = self.events
        return e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])


def event_quad(rt_cnn, rt_em):
    """Four timing events (before / after the CNN, before / after the EM) with their handles materialised:
    (list of torch events, the (c_void_p * 4) to pass to Step.enqueue)."""
    torch = rt_em.torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    for k, e in enumerate(ev):
        e.record(rt_cnn.stream if k < 2 else rt_em.stream)
    return ev, (ctypes.c_void_p * 4)(*[e.cuda_event for e in ev])
