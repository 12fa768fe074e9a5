"""Per-device runtime: one libvpk handle + one torch stream that both torch plumbing and the
library's kernels are enqueued on (so they are ordered without extra synchronisation)."""
import ctypes

from . import _lib

_runtimes = {}


class Runtime(object):
    def __init__(self, device=0, priority=0):
        import torch
        if not torch.cuda.is_available():
            raise _lib.VpkError("no GPU visible: the vanishing-point hot path runs on MI355X only "
                                "(there is no CPU fallback)")
        self.torch = torch
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.Stream(device=self.tdev, priority=int(priority))   # (-1: high, as torch counts)
        self.handle = _lib.Handle(self.device, stream=self.stream.cuda_stream)
        self.lib = self.handle.lib
        self.h = self.handle.h

    def on_stream(self):
        return self.torch.cuda.stream(self.stream)

    def check(self, rc):
        self.handle.check(rc)

    def synchronize(self):
        self.stream.synchronize()

    @staticmethod
    def ptr(t):
        return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def get_runtime(device=0, lane="main", priority=0):
    """Process-wide runtime of ``device``.  ``lane`` names an independent (handle, stream) pair on the
    same GPU, e.g. one for the CNN and one for the EM so that consecutive batches overlap.  ``priority``
    (used when the lane is created): HIP stream priority, -1 = high."""
    key = (int(device), lane)
    if key not in _runtimes:
        _runtimes[key] = Runtime(int(device), priority)
    return _runtimes[key]
