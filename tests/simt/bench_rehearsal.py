"""Rehearsal plumbing of bench.py (TEST INFRASTRUCTURE; loaded by bench.py only under SL_BENCH_DRY_RUN=1 with the emulator library).

bench.py is edited between GPU accesses and run unattended by the driver; the CPU suite executes the whole file against the SIMT emulator so
that a typo does not cost a GPU call.  Two things make that possible and neither belongs in a measuring run: torch hands out HOST tensors
where bench.py asks for device ones (the emulator's "device memory" is host memory: torch_on_host.py), and the line that comes out keeps its
structure but loses every figure that would be a measurement."""
import importlib.util
from pathlib import Path

HERE = Path(__file__).resolve().parent


def install():
    spec = importlib.util.spec_from_file_location("torch_on_host", str(HERE / "torch_on_host.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.install()


def dry_run_line(out):
    """what a rehearsal under the emulator may print: the structure of the line, every number that would be a measurement removed"""
    def scrub(o):
        if isinstance(o, dict):
            return {k: (None if k in TIMED_KEYS and not isinstance(v, (dict, list)) else scrub(v)) for k, v in o.items()}
        if isinstance(o, list):
            return [scrub(v) for v in o]
        return o
    out = scrub(out)
    out["dry_run"] = "rehearsal of bench.py's code path under the SIMT emulator (tests/simt): host fibers, not an MI355X — no figure of this line is a measurement"
    return out



TIMED_KEYS = {"value", "ms_per_step", "achieved", "frac", "launch_ms", "timed_region_device_ms_per_step", "algorithmic_over_copy_ceiling_6290", "rows_iter_per_s",
              "achieved_GBps", "roofline_frac", "nnz_iter_per_s", "floor_ms", "single_thread_simd4", "all_threads_rowchunk", "n1_ms_per_step", "slice_ms_per_step",
              "device_ms_per_step_slowest_rank", "roofline_frac_per_gpu", "spmv_s_per_step", "vector_passes_s_per_step"}      # (floor_ms and the ceiling derived from it are constants of the model, not measurements — floor_ms is scrubbed with the rest for simplicity)
