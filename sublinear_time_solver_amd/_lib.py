"""ctypes binding of libsublinear_hip.so (the C ABI of include/sublinear_hip.h).

There is no CPU fallback: if the shared library is missing this module raises at
import-of-symbols time, and every compute call returns SL_DEVICE_ERROR without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libsublinear_hip.so"

SL_OK = 0
STATUS_NAMES = {
    0: "OK", 1: "MatrixNotDiagonallyDominant", 2: "NumericalInstability", 3: "ConvergenceFailure",
    4: "InvalidInput", 5: "DimensionMismatch", 6: "UnsupportedMatrixFormat", 7: "MemoryAllocationError",
    8: "IndexOutOfBounds", 9: "InvalidSparseMatrix", 10: "AlgorithmError", 11: "DeviceError",
}
SL_MEM_HOST, SL_MEM_DEVICE = 0, 1
SL_ORDER_CSR_SEQUENTIAL, SL_ORDER_SIMD4, SL_ORDER_ANY = 0, 1, 2
SL_START_ZERO, SL_START_REFERENCE_DEFAULT, SL_START_INITIAL_GUESS = 0, 1, 2
SL_RESIDUAL_TRUE, SL_RESIDUAL_REFERENCE_SCALED = 0, 1
SL_SYSTEM_FORWARD, SL_SYSTEM_BACKWARD, SL_SYSTEM_DANGLING_IDENTITY = 0, 1, 2
SL_MATRIX_WITH_TRANSPOSE, SL_MATRIX_KEEP_CSR, SL_MATRIX_COLUMN_PANELS, SL_MATRIX_NO_COLUMN_PANELS, SL_MATRIX_ORDER_ANY = 1, 2, 4, 8, 16
SL_MATRIX_ROW_SLICE = 32
SL_WALK_STREAM_BLOCKS, SL_WALK_STREAM_SERIAL = 0, 1
SL_NORM_L1, SL_NORM_L2, SL_NORM_LINF, SL_NORM_WEIGHTED = 0, 1, 2, 3
SL_CONV_RESIDUAL_NORM, SL_CONV_RELATIVE_RESIDUAL, SL_CONV_SOLUTION_CHANGE, SL_CONV_RELATIVE_SOLUTION_CHANGE, SL_CONV_COMBINED = 0, 1, 2, 3, 4

u64, u32, i32, f64 = C.c_uint64, C.c_uint32, C.c_int32, C.c_double
vp = C.c_void_p


class MatrixInfo(C.Structure):
    _fields_ = [("n_rows", u64), ("n_cols", u64), ("nnz", u64), ("row_offset", u64), ("padded_nnz", u64),
                ("n_slices", u64), ("device_bytes", u64), ("bandwidth", u64), ("max_row_nnz", u32), ("min_row_nnz", u32),
                ("uniform_width", u32), ("has_transpose", u32), ("long_row_threshold", u32), ("n_long_rows", u32),
                ("column_panels", u32), ("reserved", u32)]


class SparsityInfo(C.Structure):      # sl_sparsity_info (SparsityInfo, types.rs:114-129)
    _fields_ = [("nnz", u64), ("rows", u64), ("cols", u64), ("sparsity_ratio", f64), ("avg_nnz_per_row", f64),
                ("max_nnz_per_row", u64), ("bandwidth", u64), ("is_banded", C.c_int32), ("reserved", C.c_int32)]


class NeumannOptions(C.Structure):
    _fields_ = [("tolerance", f64), ("max_iterations", u64), ("max_terms", u64), ("series_tolerance", f64),
                ("order", i32), ("start", i32), ("residual", i32), ("mem", i32), ("collect_stats", i32),
                ("compute_error_bounds", i32)]


class NeumannResult(C.Structure):
    _fields_ = [("iterations", u64), ("terms_computed", u64), ("matvec_count", u64), ("residual_norm", f64),
                ("last_term_norm", f64), ("error_bound", f64), ("total_time_ms", f64), ("device_time_ms", f64),
                ("bytes_moved", u64), ("converged", i32), ("series_converged", i32)]


class PushOptions(C.Structure):
    _fields_ = [("theta", f64), ("max_rounds", u64), ("order", i32), ("mem", i32), ("dense_switch", f64),
                ("theta_rows", C.c_void_p)]


class PushResult(C.Structure):
    _fields_ = [("rounds", u64), ("pushes", u64), ("rows_touched", u64), ("dense_rounds", u64),
                ("residual_norm", f64), ("device_time_ms", f64), ("converged", i32), ("reserved", i32)]


class SouthwellOptions(C.Structure):
    _fields_ = [("epsilon", f64), ("max_iterations", u64), ("mem", i32), ("reserved", i32)]


class SouthwellResult(C.Structure):
    _fields_ = [("iterations", u64), ("residual_norm", f64), ("device_time_ms", f64), ("converged", i32), ("reserved", i32)]


class EstimateResult(C.Structure):
    _fields_ = [("estimate", f64), ("residual_l1", f64), ("rounds", u64), ("pushes", u64), ("rows_touched", u64),
                ("device_time_ms", f64), ("converged", i32), ("reserved", i32)]


class WalkResult(C.Structure):
    _fields_ = [("estimate", f64), ("variance", f64), ("num_samples", u64), ("device_time_ms", f64)]


class RandomWalkResult(C.Structure):     # sl_random_walk_result
    _fields_ = [("iterations", u64), ("num_walks", u64), ("residual", f64), ("total_variance", f64), ("device_time_ms", f64),
                ("converged", i32), ("reserved", i32)]


class CommInfo(C.Structure):
    _fields_ = [("rank", i32), ("world", i32), ("device", i32), ("transport", i32), ("halo_allreduce", i32), ("ranks_joined", i32),
                ("failed", i32), ("reserved", i32)]


class AclOptions(C.Structure):
    _fields_ = [("alpha", f64), ("epsilon", f64), ("queue_threshold", f64), ("max_pushes", u64), ("adaptive_threshold", i32), ("mem", i32)]


class AclResult(C.Structure):
    _fields_ = [("push_count", u64), ("nodes_visited", u64), ("residual_norm", f64), ("device_time_ms", f64), ("stopped_by", i32), ("reserved", i32)]


class CgOptions(C.Structure):
    _fields_ = [("tolerance", f64), ("max_iterations", u64), ("order", i32), ("mem", i32)]


class CgResult(C.Structure):
    _fields_ = [("iterations", u64), ("matvec_count", u64), ("residual_norm", f64), ("total_time_ms", f64),
                ("device_time_ms", f64), ("converged", i32), ("reserved", i32)]


ABI_VERSION = 5      # SL_ABI_VERSION of include/sublinear_hip.h

# every symbol include/sublinear_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "sl_abi_version": (C.c_int, []),
    "sl_last_error_message": (C.c_char_p, []),
    "sl_status_string": (C.c_char_p, [C.c_int]),
    "sl_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "sl_device_name": (C.c_int, [C.c_int, C.c_char_p, u64]),
    "sl_set_device": (C.c_int, [C.c_int]),
    "sl_set_stream": (C.c_int, [vp]),
    "sl_synchronize": (C.c_int, []),
    "sl_release_workspace": (None, []),
    "sl_matrix_create_from_triplets": (C.c_int, [u64, vp, vp, vp, u64, u64, u32, C.POINTER(vp)]),
    "sl_matrix_create_csr": (C.c_int, [u64, u64, u64, vp, vp, vp, C.c_int, u64, u32, C.POINTER(vp)]),
    "sl_matrix_destroy": (None, [vp]),
    "sl_matrix_get_info": (C.c_int, [vp, C.POINTER(MatrixInfo)]),
    "sl_matrix_scale": (C.c_int, [vp, f64]),
    "sl_matrix_add_diagonal": (C.c_int, [vp, f64]),
    "sl_matrix_download_csr": (C.c_int, [vp, vp, vp, vp]),
    "sl_matrix_is_diagonally_dominant": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "sl_matrix_diagonal_inverse": (C.c_int, [vp, vp, C.c_int]),
    "sl_matrix_diagonal_dominance_factor": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(f64)]),
    "sl_matrix_spectral_radius_estimate": (C.c_int, [vp, C.POINTER(f64)]),
    "sl_matrix_get": (C.c_int, [vp, u64, u64, C.POINTER(C.c_int), C.POINTER(f64)]),
    "sl_matrix_row": (C.c_int, [vp, u64, u64, vp, vp, C.POINTER(u64)]),
    "sl_matrix_col": (C.c_int, [vp, u64, u64, vp, vp, C.POINTER(u64)]),
    "sl_matrix_frobenius_norm": (C.c_int, [vp, C.POINTER(f64)]),
    "sl_matrix_sparsity_info": (C.c_int, [vp, C.POINTER(SparsityInfo)]),
    "sl_spmv": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "sl_spmv_add": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "sl_dot": (C.c_int, [u64, vp, vp, C.POINTER(f64), C.c_int]),
    "sl_axpy": (C.c_int, [u64, f64, vp, vp, C.c_int]),
    "sl_l2_norm": (C.c_int, [u64, vp, C.POINTER(f64), C.c_int]),
    "sl_l1_norm": (C.c_int, [u64, vp, C.POINTER(f64), C.c_int]),
    "sl_linf_norm": (C.c_int, [u64, vp, C.POINTER(f64), C.c_int]),
    "sl_compute_norm": (C.c_int, [u64, vp, C.c_int, C.POINTER(f64), C.c_int]),
    "sl_compute_residual": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int]),
    "sl_check_convergence": (C.c_int, [f64, f64, C.c_int, f64, u64, vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "sl_neumann_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int]),
    "sl_matrix_partials_capacity": (C.c_int, [vp, C.POINTER(u64)]),
    "sl_neumann_step_partials": (C.c_int, [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_uint32), C.c_int]),
    "sl_reduce_partials": (C.c_int, [vp, C.c_uint32, vp]),
    "sl_residual_norm2": (C.c_int, [vp, vp, vp, vp, vp, C.c_int]),
    "sl_neumann_run_steps": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, u64, C.POINTER(C.c_float)]),
    "sl_neumann_options_default": (None, [C.POINTER(NeumannOptions)]),
    "sl_neumann_options_streaming": (None, [C.POINTER(NeumannOptions)]),
    "sl_neumann_result_meets_quality_criteria": (C.c_int, [C.POINTER(NeumannResult), f64]),
    "sl_neumann_solve": (C.c_int, [vp, vp, vp, C.POINTER(NeumannOptions), vp, vp, C.POINTER(NeumannResult)]),
    "sl_neumann_state_create": (C.c_int, [vp, vp, vp, C.POINTER(NeumannOptions), C.POINTER(vp)]),
    "sl_neumann_state_destroy": (None, [vp]),
    "sl_neumann_state_update_rhs": (C.c_int, [vp, u64, vp, vp]),
    "sl_neumann_state_run": (C.c_int, [vp, vp, C.POINTER(NeumannResult)]),
    "sl_neumann_state_solution": (C.c_int, [vp, vp, C.c_int]),
    "sl_neumann_state_current_term": (C.c_int, [vp, u64, u64, vp, C.c_int]),
    "sl_neumann_state_solution_rows": (C.c_int, [vp, u64, u64, vp, C.c_int]),
    "sl_neumann_state_reset": (C.c_int, [vp]),
    "sl_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)]),
    "sl_comm_destroy": (None, [vp]),
    "sl_comm_rank": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sl_comm_barrier": (C.c_int, [vp]),
    "sl_comm_allgather_u64": (C.c_int, [vp, u64, vp]),
    "sl_comm_info": (C.c_int, [vp, C.POINTER(CommInfo)]),
    "sl_neumann_state_verify_exchange": (C.c_int, [vp, C.POINTER(u64)]),
    "sl_balanced_row_bounds": (C.c_int, [u64, vp, C.c_int, vp]),
    "sl_neumann_state_create_partitioned": (C.c_int, [vp, vp, vp, vp, C.POINTER(NeumannOptions), C.POINTER(vp)]),
    "sl_neumann_state_run_steps": (C.c_int, [vp, u64, C.POINTER(f64), C.POINTER(C.c_float)]),
    "sl_push_options_default": (None, [C.POINTER(PushOptions)]),
    "sl_push_solve": (C.c_int, [vp, vp, C.POINTER(PushOptions), vp, vp, vp, u64, C.POINTER(u64),
                                C.POINTER(PushResult)]),
    "sl_push_graph_create": (C.c_int, [u64, vp, vp, vp, C.c_int, C.POINTER(vp)]),
    "sl_push_graph_destroy": (None, [vp]),
    "sl_push_graph_size": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64)]),
    "sl_push_graph_degrees": (C.c_int, [vp, vp, vp, C.c_int]),
    "sl_push_graph_system": (C.c_int, [vp, f64, u32, u32, C.POINTER(vp)]),
    "sl_acl_options_default": (None, [C.POINTER(AclOptions)]),
    "sl_forward_push_acl": (C.c_int, [vp, u64, vp, C.POINTER(AclOptions), vp, vp, vp, u64, C.POINTER(AclResult)]),
    "sl_backward_push_acl": (C.c_int, [vp, u64, vp, C.POINTER(AclOptions), vp, vp, vp, u64, C.POINTER(AclResult)]),
    "sl_forward_push_acl_with_target": (C.c_int, [vp, u64, u64, f64, C.POINTER(AclOptions), vp, vp, vp, u64, C.POINTER(AclResult)]),
    "sl_backward_push_acl_with_source": (C.c_int, [vp, u64, u64, f64, C.POINTER(AclOptions), vp, vp, vp, u64, C.POINTER(AclResult)]),
    "sl_acl_extrapolated_solution": (C.c_int, [u64, f64, vp, vp, vp, C.c_int]),
    "sl_backward_push_acl_reachability": (C.c_int, [vp, u64, C.POINTER(AclOptions), vp, C.POINTER(AclResult)]),
    "sl_southwell_options_default": (None, [C.POINTER(SouthwellOptions)]),
    "sl_forward_push_southwell": (C.c_int, [vp, vp, C.POINTER(SouthwellOptions), vp, vp, vp, u64, C.POINTER(SouthwellResult)]),
    "sl_estimate_entry": (C.c_int, [vp, vp, C.c_int, u64, f64, u64, C.POINTER(EstimateResult)]),
    "sl_synth_sdd_device": (C.c_int, [u64, u32, u64, u64, u64, u64, vp, vp, vp, vp]),
    "sl_estimate_entry_random_walk": (C.c_int, [vp, vp, C.c_int, u64, f64, u32, C.c_int, u64, vp, C.POINTER(WalkResult)]),
    "sl_solve_random_walk": (C.c_int, [vp, vp, C.c_int, f64, u32, C.c_int, u64, vp, vp, C.POINTER(RandomWalkResult)]),
    "sl_cg_options_default": (None, [C.POINTER(CgOptions)]),
    "sl_cg_solve": (C.c_int, [vp, vp, C.POINTER(CgOptions), vp, C.POINTER(CgResult)]),
    "sl_estimate_entry_transposed": (C.c_int, [vp, vp, C.c_int, u64, f64, u64, C.POINTER(EstimateResult)]),
    "sl_query_session_create": (C.c_int, [vp, C.c_int, vp, C.c_int, C.POINTER(vp)]),
    "sl_query_session_estimate": (C.c_int, [vp, u64, f64, u64, C.POINTER(EstimateResult)]),
    "sl_query_session_estimate_batch": (C.c_int, [vp, u64, vp, f64, u64, u32, vp]),
    "sl_query_session_destroy": (None, [vp]),
    "sl_matrix_transpose": (C.c_int, [vp, u32, C.POINTER(vp)]),
    "sl_synth_pagerank_device": (C.c_int, [u64, u64, f64, u32, u32, vp, vp, vp, C.POINTER(u64)]),
}

_lib = None


class SolverError(RuntimeError):
    """Mirror of the reference's SolverError enum (src/error.rs:16-140): `.kind` is the variant name."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.kind = STATUS_NAMES.get(status, "Unknown")
        super().__init__(f"{self.kind}: {message}")


def _share_hip_runtime_with_torch() -> None:
    """A process must hold ONE HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (same SONAME as /opt/rocm's); if this library pulled in /opt/rocm's copy first, a later
    `import torch` would load a second runtime and find no GPU.  When torch is installed but not
    yet imported, map its copy first so both sides resolve to the same runtime (torch itself is NOT
    imported here: it is plumbing for callers that want it, not a dependency)."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Load the HIP library; raise loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SUBLINEAR_HIP_LIB", str(LIB_PATH))
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: build it with `make -C sublinear_time_solver_amd/csrc` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.sl_abi_version() != ABI_VERSION:
        raise ImportError(f"libsublinear_hip ABI version {lib.sl_abi_version()}, this package binds {ABI_VERSION}: rebuild csrc/")
    _lib = lib
    return lib


def check(status: int) -> None:
    if status != SL_OK:
        msg = load().sl_last_error_message()
        raise SolverError(status, msg.decode() if msg else "")


def ptr(a) -> int:
    """Raw address of a numpy array (host) or torch tensor (device)."""
    if a is None:
        return None
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data
