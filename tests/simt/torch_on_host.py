"""install(): make the REAL torch hand out host tensors where a script asks for device ones — for rehearsing bench.py against the SIMT
emulator (SL_BENCH_DRY_RUN=1), whose "device memory" is host memory: `device=cuda` is dropped from tensor factories, torch.cuda's few
entry points bench.py uses become no-ops, everything else (tensor arithmetic, torch.distributed over gloo) is torch itself.  Test
infrastructure; nothing in the product imports it, and what a rehearsal prints is never a measurement."""
import time


def install():
    import torch

    def host(fn):
        def wrapped(*a, **k):
            k.pop("device", None)
            return fn(*a, **k)
        return wrapped

    for name in ("empty", "zeros", "ones", "full", "arange", "tensor", "empty_like", "zeros_like", "rand", "randn"):
        setattr(torch, name, host(getattr(torch, name)))
    torch.Tensor.cuda = lambda self, *a, **k: self

    class _Stream:
        cuda_stream = 0

    class _Event:
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self, stream=None):
            self.t = time.perf_counter()

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    c = torch.cuda
    c.is_available = lambda: True
    import os
    c.device_count = lambda: max(1, int(os.environ.get("SIMT_DEVICES", "1") or 1))
    c.set_device = lambda d: None
    c.synchronize = lambda d=None: None
    c.empty_cache = lambda: None
    c.current_stream = lambda d=None: _Stream()
    c.Event = _Event
    return torch
