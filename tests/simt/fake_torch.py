"""A stand-in for the few things bench.py asks of torch — device buffers whose address it hands to the C ABI, and device synchronisation —
for ONE purpose: tests/test_simt_emulated.py runs `bench.py` end to end against the SIMT emulator (whose "device memory" is host memory)
with SL_BENCH_DRY_RUN=1, so that the whole line — synthesis through the ABI, layouts, the parity gate and its CPU child, warm-up and
timed steps, the column-structure sweep, the cpu_baseline child, the JSON — executes in the CPU suite.  The line such a run prints
carries `"dry_run"` and `"value": null`: it is a rehearsal of the code path, never a measurement.  Nothing in the product imports this."""
import types

import numpy as np

int32, int64, float64 = np.int32, np.int64, np.float64


class Tensor(np.ndarray):
    def data_ptr(self):
        return self.ctypes.data

    def abs(self):
        return np.abs(self).view(Tensor)

    def item(self):
        return np.asarray(self).item()

    def numel(self):
        return int(self.size)

    def element_size(self):
        return int(self.itemsize)

    def clone(self):
        return self.copy().view(Tensor)

    def cpu(self):
        return self

    def numpy(self):
        return np.asarray(self)

    def cuda(self, *a, **k):
        return self


def _t(a):
    return np.ascontiguousarray(a).view(Tensor)


def device(kind, index=0):
    return ("cuda", int(index))


def empty(*shape, dtype=float64, device=None):
    return _t(np.empty(shape[0] if len(shape) == 1 else shape, dtype=dtype))


def zeros(*shape, dtype=float64, device=None):
    return _t(np.zeros(shape[0] if len(shape) == 1 else shape, dtype=dtype))


def ones(*shape, dtype=float64, device=None):
    return _t(np.ones(shape[0] if len(shape) == 1 else shape, dtype=dtype))


def full(shape, value, dtype=float64, device=None):
    return _t(np.full(shape, value, dtype=dtype))


def empty_like(a):
    return _t(np.empty_like(np.asarray(a)))


def arange(n, device=None, dtype=float64):
    return _t(np.arange(n, dtype=dtype))


def remainder(a, b):
    return _t(np.remainder(np.asarray(a), b))


def tensor(data, dtype=float64, device=None):
    return _t(np.array(data, dtype=dtype))


def from_numpy(a):
    return _t(a)


class _Stream:
    cuda_stream = 0


class _Event:
    def __init__(self, enable_timing=False):
        self.t = 0.0

    def record(self, stream=None):
        import time
        self.t = time.perf_counter()

    def synchronize(self):
        pass

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


cuda = types.SimpleNamespace(Event=_Event, is_available=lambda: True, device_count=lambda: 1, set_device=lambda d: None, synchronize=lambda d=None: None,
                             empty_cache=lambda: None, current_stream=lambda d=None: _Stream())
