"""Host-side mirror of the reference's solver interfaces over the C ABI.

Names, argument meaning and error behaviour follow the reference:
  * Rust crate surface: SparseMatrix::from_triplets (src/matrix/mod.rs:160-199), Matrix trait
    methods (matrix/mod.rs:25-104), SolverOptions / SolverResult (src/solver/mod.rs:20-195),
    NeumannSolver (src/solver/neumann.rs:24-92, solve :469-555), ForwardPushSolver /
    query_single_entry (src/solver/forward_push.rs:67-231); errors are SolverError variants
    (src/error.rs:16-140) carried by `SolverError.kind`.
  * shipped TypeScript surface: SublinearSolver(config).solve / .estimateEntry
    (src/core/solver.ts:36-111, 550-659) with the same result field names.

All compute happens in libsublinear_hip.so; this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib as L
from ._lib import SolverError


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


@dataclass
class SolverOptions:
    """src/solver/mod.rs:20-62 (defaults :47-62)."""
    tolerance: float = 1e-6
    max_iterations: int = 1000
    initial_guess: Optional[Sequence[float]] = None
    collect_stats: bool = False
    compute_error_bounds: bool = False

    @staticmethod
    def high_precision() -> "SolverOptions":   # solver/mod.rs:66-76
        return SolverOptions(tolerance=1e-12, max_iterations=5000, collect_stats=True, compute_error_bounds=True)

    @staticmethod
    def fast() -> "SolverOptions":             # solver/mod.rs:79-89
        return SolverOptions(tolerance=1e-3, max_iterations=100)

    @staticmethod
    def streaming(interval: int) -> "SolverOptions":   # solver/mod.rs:101-116 (sl_neumann_options_streaming)
        o = SolverOptions(tolerance=1e-4, max_iterations=1000, collect_stats=True, compute_error_bounds=False)
        o.streaming_interval = interval          # paces the caller's loop over NeumannState.run_steps (PartialSolution, solver/mod.rs:197-214)
        return o


@dataclass
class SolverResult:
    """src/solver/mod.rs:118-195."""
    solution: np.ndarray
    residual_norm: float
    iterations: int
    converged: bool
    error_bounds: Optional[float] = None
    stats: Optional[dict] = None
    term_norms: Optional[np.ndarray] = None

    def meets_quality_criteria(self, tolerance: float) -> bool:
        """SolverResult::meets_quality_criteria, solver/mod.rs:192-195"""
        return bool(self.converged and self.residual_norm <= tolerance)


class SparseMatrix:
    """Device-resident sparse matrix (CSR in, row-slice layout in HBM)."""

    def __init__(self, handle: int, rows: int, cols: int):
        self._h = handle
        self._rows, self._cols = rows, cols

    # -- constructors -------------------------------------------------------------------
    @classmethod
    def from_triplets(cls, triplets, rows: int, cols: int, with_transpose: bool = False, keep_csr: bool = False):
        """SparseMatrix::from_triplets, matrix/mod.rs:160-199 (validation, zero dropping, stable sort)."""
        lib = L.load()
        t = list(triplets)
        r = np.asarray([x[0] for x in t], dtype=np.int64)
        c = np.asarray([x[1] for x in t], dtype=np.int64)
        v = _f64([x[2] for x in t])
        if (r < 0).any() or (c < 0).any():
            raise SolverError(8, "negative index in triplet")
        r = r.astype(np.uint64)
        c = c.astype(np.uint64)
        h = L.vp()
        flags = (L.SL_MATRIX_WITH_TRANSPOSE if with_transpose else 0) | (L.SL_MATRIX_KEEP_CSR if keep_csr else 0)
        L.check(lib.sl_matrix_create_from_triplets(len(t), L.ptr(r), L.ptr(c), L.ptr(v), rows, cols, flags, C.byref(h)))
        return cls(h.value, rows, cols)

    @classmethod
    def from_csr(cls, row_ptr, col_idx, values, rows: int, cols: int, row_offset: int = 0,
                 with_transpose: bool = False, keep_csr: bool = False, device: bool = False, column_panels=None, order_any: bool = False,
                 row_slice: bool = False):
        """Adopt CSRStorage arrays (matrix/sparse.rs:16-23); device=True: torch CUDA tensors.
        column_panels: None = the library decides (large systems with columns all over the vector), True / False = force / forbid
        the second, panel-ordered copy of the entries."""
        lib = L.load()
        if not device:
            row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
            col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
            values = _f64(values)
            nnz = int(values.size)
        else:
            nnz = int(values.numel())
        h = L.vp()
        flags = (L.SL_MATRIX_WITH_TRANSPOSE if with_transpose else 0) | (L.SL_MATRIX_KEEP_CSR if keep_csr else 0)
        if column_panels is not None:
            flags |= L.SL_MATRIX_COLUMN_PANELS if column_panels else L.SL_MATRIX_NO_COLUMN_PANELS
        if row_slice:      # rows [row_offset, row_offset + rows) of a SQUARE system of `cols` rows — said explicitly for the range that starts at row 0 (add_diagonal)
            flags |= L.SL_MATRIX_ROW_SLICE
        if order_any:      # the order-free column stream: what SolverOptions(order=SL_ORDER_ANY) solves run on (results to rounding, not bit for bit)
            flags |= L.SL_MATRIX_ORDER_ANY
        L.check(lib.sl_matrix_create_csr(rows, cols, nnz, L.ptr(row_ptr), L.ptr(col_idx), L.ptr(values),
                                         L.SL_MEM_DEVICE if device else L.SL_MEM_HOST, row_offset, flags, C.byref(h)))
        return cls(h.value, rows, cols)

    @classmethod
    def from_dense(cls, data, **kw):
        """SparseMatrix::from_dense, matrix/mod.rs:204-223: zeros filtered out."""
        a = np.asarray(data, dtype=np.float64)
        rows, cols = a.shape
        rr, cc = np.nonzero(a)
        return cls.from_triplets(zip(rr.tolist(), cc.tolist(), a[rr, cc].tolist()), rows, cols, **kw)

    @classmethod
    def identity(cls, size: int, **kw):
        """SparseMatrix::identity, matrix/mod.rs:226-229"""
        return cls.from_triplets(((i, i, 1.0) for i in range(size)), size, size, **kw)

    @classmethod
    def diagonal(cls, diag, **kw):
        """SparseMatrix::diagonal, matrix/mod.rs:232-239: zero entries of `diag` are not stored"""
        d = _f64(diag)
        return cls.from_triplets(((i, i, float(v)) for i, v in enumerate(d) if v != 0.0), d.size, d.size, **kw)

    @classmethod
    def from_scipy(cls, A, **kw):
        A = A.tocsr()
        A.sort_indices()
        return cls.from_csr(A.indptr, A.indices, A.data, A.shape[0], A.shape[1], **kw)

    # -- Matrix trait (matrix/mod.rs:25-104) ----------------------------------------------
    def rows(self) -> int:
        return self._rows

    def cols(self) -> int:
        return self._cols

    def is_square(self) -> bool:
        return self._rows == self._cols

    def info(self) -> L.MatrixInfo:
        i = L.MatrixInfo()
        L.check(L.load().sl_matrix_get_info(self._h, C.byref(i)))
        return i

    def nnz(self) -> int:
        return int(self.info().nnz)

    def is_diagonally_dominant(self) -> bool:
        f = C.c_int(0)
        L.check(L.load().sl_matrix_is_diagonally_dominant(self._h, C.byref(f)))
        return bool(f.value)

    def diagonal_inverse(self) -> np.ndarray:
        d = np.empty(self._rows, dtype=np.float64)
        L.check(L.load().sl_matrix_diagonal_inverse(self._h, L.ptr(d), L.SL_MEM_HOST))
        return d

    def multiply_vector(self, x, order: int = L.SL_ORDER_CSR_SEQUENTIAL) -> np.ndarray:
        """Matrix::multiply_vector, matrix/mod.rs:415-439 (DimensionMismatch on bad lengths)."""
        x = _f64(x)
        if x.size != self._cols:
            raise SolverError(5, f"expected {self._cols}, actual {x.size} in matrix_vector_multiply")
        y = np.empty(self._rows, dtype=np.float64)
        L.check(L.load().sl_spmv(self._h, L.ptr(x), L.ptr(y), order, L.SL_MEM_HOST))
        return y

    def multiply_vector_add(self, x, result) -> np.ndarray:
        """Matrix::multiply_vector_add, matrix/mod.rs:441-465: result += A x IN PLACE (a float64 numpy array), the running sum of row i
        seeded with result[i] (sparse.rs:192-203); DimensionMismatch on bad lengths, as the reference checks x first, then result."""
        x = _f64(x)
        if x.size != self._cols:
            raise SolverError(5, f"expected {self._cols}, actual {x.size} in matrix_vector_multiply_add")
        if not (isinstance(result, np.ndarray) and result.dtype == np.float64 and result.flags.c_contiguous):
            raise SolverError(4, "result must be a contiguous float64 numpy array (it is updated in place)")
        if result.size != self._rows:
            raise SolverError(5, f"expected {self._rows}, actual {result.size} in matrix_vector_multiply_add")
        L.check(L.load().sl_spmv_add(self._h, L.ptr(x), L.ptr(result), L.SL_ORDER_CSR_SEQUENTIAL, L.SL_MEM_HOST))
        return result

    def diagonal_dominance_factor(self) -> Optional[float]:
        """Matrix::diagonal_dominance_factor, matrix/mod.rs:487-514: min |a_ii| / sum |a_ij| over rows with off-diagonal weight, or None"""
        has, f = C.c_int(0), C.c_double(0.0)
        L.check(L.load().sl_matrix_diagonal_dominance_factor(self._h, C.byref(has), C.byref(f)))
        return f.value if has.value else None

    def spectral_radius_estimate(self) -> float:
        """Matrix::spectral_radius_estimate, matrix/mod.rs:83-100 (Gershgorin)"""
        r = C.c_double(0.0)
        L.check(L.load().sl_matrix_spectral_radius_estimate(self._h, C.byref(r)))
        return r.value

    def conditioning_info(self) -> dict:
        """Matrix::conditioning_info, matrix/mod.rs:548-556 (condition_number / is_positive_definite stay None there too)"""
        return {"condition_number": None, "is_diagonally_dominant": self.is_diagonally_dominant(),
                "diagonal_dominance_factor": self.diagonal_dominance_factor(), "spectral_radius": self.spectral_radius_estimate(),
                "is_positive_definite": None}

    def get(self, row: int, col: int) -> Optional[float]:
        """Matrix::get, matrix/mod.rs:33 / :383-395 -> CSRStorage::get sparse.rs:142-155: None out of bounds or where nothing is stored;
        a row holding the column twice answers with the entry the reference's binary search lands on"""
        if row < 0 or col < 0:
            return None
        found, v = C.c_int(0), C.c_double(0.0)
        L.check(L.load().sl_matrix_get(self._h, row, col, C.byref(found), C.byref(v)))
        return v.value if found.value else None

    def row_iter(self, row: int):
        """Matrix::row_iter, matrix/mod.rs:37 (CSRStorage::row_iter sparse.rs:158-176): (column, value) pairs in stored order; a row out
        of bounds is empty"""
        if row < 0:
            return iter(())
        lib, n = L.load(), C.c_uint64(0)
        cap = max(1, int(self.info().max_row_nnz))
        co, va = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.float64)
        L.check(lib.sl_matrix_row(self._h, row, cap, L.ptr(co), L.ptr(va), C.byref(n)))
        return iter(list(zip(co[: n.value].tolist(), va[: n.value].tolist())))

    def col_iter(self, col: int):
        """Matrix::col_iter, matrix/mod.rs:41 (CSRColIter sparse.rs:273-298): (row, value) pairs, rows ascending, one pair per row"""
        if col < 0:
            return iter(())
        lib, n = L.load(), C.c_uint64(0)
        L.check(lib.sl_matrix_col(self._h, col, 0, None, None, C.byref(n)))
        cap = max(1, n.value)
        ro, va = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.float64)
        L.check(lib.sl_matrix_col(self._h, col, cap, L.ptr(ro), L.ptr(va), C.byref(n)))
        return iter(list(zip(ro[: n.value].tolist(), va[: n.value].tolist())))

    def scale(self, factor: float) -> None:
        """SparseMatrix::scale, matrix/mod.rs:346-354 over CSRStorage::scale, sparse.rs:229-233: every stored value *= factor, in place on
        the device (every layout copy).  A `&mut self` method: no solve may be running on the matrix; states / sessions created before
        keep their old D^-1 and must be re-created."""
        L.check(L.load().sl_matrix_scale(self._h, float(factor)))

    def add_diagonal(self, alpha: float) -> None:
        """SparseMatrix::add_diagonal, matrix/mod.rs:356-372 over CSRStorage::add_diagonal, sparse.rs:236-248: A += alpha I in place; a row
        without a stored diagonal entry is silently skipped as in the reference; non-square: InvalidInput."""
        L.check(L.load().sl_matrix_add_diagonal(self._h, float(alpha)))

    def frobenius_norm(self) -> float:
        """Matrix::frobenius_norm, matrix/mod.rs:74-82 (tree-reduced on the device: equal to the reference's sequential sum to rounding)"""
        r = C.c_double(0.0)
        L.check(L.load().sl_matrix_frobenius_norm(self._h, C.byref(r)))
        return r.value

    def sparsity_info(self) -> dict:
        """Matrix::sparsity_info, matrix/mod.rs:523-545: the fields of SparsityInfo (types.rs:114-129)"""
        i = L.SparsityInfo()
        L.check(L.load().sl_matrix_sparsity_info(self._h, C.byref(i)))
        return {"nnz": int(i.nnz), "dimensions": (int(i.rows), int(i.cols)), "sparsity_ratio": i.sparsity_ratio, "avg_nnz_per_row": i.avg_nnz_per_row,
                "max_nnz_per_row": int(i.max_nnz_per_row), "bandwidth": int(i.bandwidth), "is_banded": bool(i.is_banded)}

    FORMATS = ("CSR", "CSC", "COO", "GraphAdjacency")        # SparseFormat, matrix/mod.rs:106-116

    def format_name(self) -> str:
        """Matrix::format_name, matrix/mod.rs:557-564"""
        return getattr(self, "_format", "CSR")

    def convert_to_format(self, new_format: str) -> None:
        """SparseMatrix::convert_to_format, matrix/mod.rs:244-296.  No device work: every storage the reference can convert a SparseMatrix
        into is filled from to_triplets() of the one before, so its multiply loop adds a row's products in the CSR loop's own sequence —
        the same bits (tests/test_storage_formats_host.py shows it on restatements of all four loops, duplicates and every conversion path
        included).  The device copy keeps its layouts; only the name changes.  GraphAdjacency of a non-square matrix is refused: the
        reference's GraphStorage::from_triplets silently DROPS entries whose column is >= rows (sparse.rs:655-690)."""
        if new_format not in self.FORMATS:
            raise SolverError(6, f"Unsupported matrix format: {new_format}")
        if new_format == "GraphAdjacency" and self._rows != self._cols:
            raise SolverError(6, "GraphAdjacency of a non-square matrix: the reference drops the entries beyond column `rows`; not mirrored")
        self._format = new_format

    def to_triplets(self):
        """SparseMatrix::to_triplets, matrix/mod.rs:298-305 (CSRStorage::to_triplets, sparse.rs:211-226): (row, col, value) in stored order"""
        rp, ci, va = self.to_csr()
        rows = np.repeat(np.arange(rp.size - 1), np.diff(rp.astype(np.int64)))
        return list(zip(rows.tolist(), ci.tolist(), va.tolist()))

    def to_csr(self):
        """SparseMatrix::as_csr, matrix/mod.rs:315-321: the raw arrays where the matrix keeps them, else written back from the row slices"""
        i = self.info()
        rp = np.empty(i.n_rows + 1, dtype=np.uint32)
        ci = np.empty(i.nnz, dtype=np.uint32)
        va = np.empty(i.nnz, dtype=np.float64)
        L.check(L.load().sl_matrix_download_csr(self._h, L.ptr(rp), L.ptr(ci), L.ptr(va)))
        return rp, ci, va

    def transpose(self, with_transpose: bool = False, keep_csr: bool = False) -> "SparseMatrix":
        """A^T as its own matrix (CompressedSparseRow::transpose, src/graph/mod.rs:92-130); needs with_transpose at create."""
        h = L.vp()
        flags = (L.SL_MATRIX_WITH_TRANSPOSE if with_transpose else 0) | (L.SL_MATRIX_KEEP_CSR if keep_csr else 0)
        L.check(L.load().sl_matrix_transpose(self._h, flags, C.byref(h)))
        return SparseMatrix(h.value, self._cols, self._rows)

    def close(self):
        if self._h:
            L.load().sl_matrix_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class NeumannSolver:
    """src/solver/neumann.rs:24-92.  `start` / `residual` select the reference-compat quirks
    (SURVEY.md §0.3): defaults are the mathematically exact series."""
    max_terms: int = 50
    series_tolerance: float = 1e-8
    adaptive_truncation: bool = True
    order: int = L.SL_ORDER_CSR_SEQUENTIAL
    start: int = L.SL_START_ZERO
    residual: int = L.SL_RESIDUAL_TRUE

    @staticmethod
    def high_precision() -> "NeumannSolver":   # neumann.rs:63-65
        return NeumannSolver(100, 1e-12)

    @staticmethod
    def fast() -> "NeumannSolver":             # neumann.rs:68-70
        return NeumannSolver(20, 1e-6)

    def algorithm_name(self) -> str:
        return "neumann"

    def _options(self, matrix, b, options):
        lib = L.load()
        if matrix.is_square() and b.size != matrix.rows():           # neumann.rs:154-160
            raise SolverError(5, f"expected {matrix.rows()}, actual {b.size} in neumann_initialization")
        o = L.NeumannOptions()
        lib.sl_neumann_options_default(C.byref(o))
        o.tolerance, o.max_iterations = options.tolerance, options.max_iterations
        o.max_terms, o.series_tolerance = self.max_terms, self.series_tolerance
        o.order, o.start, o.residual, o.mem = self.order, self.start, self.residual, L.SL_MEM_HOST
        o.collect_stats = int(options.collect_stats)
        o.compute_error_bounds = int(options.compute_error_bounds and self.adaptive_truncation)
        guess = None
        if options.initial_guess is not None:
            guess = _f64(options.initial_guess)
            if guess.size != matrix.rows():                          # neumann.rs:198-204
                raise SolverError(5, f"expected {matrix.rows()}, actual {guess.size} in initial_guess")
            o.start = L.SL_START_INITIAL_GUESS
        return o, guess

    # SolverAlgorithm::{initialize, update_rhs, extract_solution} (solver/mod.rs:223-333) over the state object of the ABI
    def initialize(self, matrix: SparseMatrix, b, options: Optional[SolverOptions] = None) -> "NeumannState":
        options = options or SolverOptions()
        b = _f64(b)
        o, guess = self._options(matrix, b, options)
        h = C.c_void_p()
        L.check(L.load().sl_neumann_state_create(matrix._h, L.ptr(b), L.ptr(guess), C.byref(o), C.byref(h)))
        return NeumannState(h, matrix, self.max_terms, options)

    def initialize_partitioned(self, comm: "Communicator", local_rows: SparseMatrix, b_local, options: Optional[SolverOptions] = None) -> "NeumannState":
        """NeumannState::new over a row partition (sl_neumann_state_create_partitioned): `local_rows` = this rank's rows with global
        column ids (SparseMatrix.from_csr(..., row_offset=lo)), `b_local` its part of the right-hand side.  Collective."""
        options = options or SolverOptions()
        b = _f64(b_local)
        if b.size != local_rows.rows():
            raise SolverError(5, f"expected {local_rows.rows()}, actual {b.size} in neumann_initialization")
        o, guess = self._options(local_rows, b, options)
        h = C.c_void_p()
        L.check(L.load().sl_neumann_state_create_partitioned(comm._h, local_rows._h, L.ptr(b), L.ptr(guess), C.byref(o), C.byref(h)))
        st = NeumannState(h, local_rows, self.max_terms, options)
        st._comm = comm
        return st

    def update_rhs(self, state: "NeumannState", delta_b) -> None:
        state.update_rhs(delta_b)

    def extract_solution(self, state: "NeumannState") -> np.ndarray:
        return state.solution()

    def is_converged(self, state: "NeumannState") -> bool:
        return state.converged

    def solve(self, matrix: SparseMatrix, b, options: Optional[SolverOptions] = None) -> SolverResult:
        options = options or SolverOptions()
        lib = L.load()
        b = _f64(b)
        if matrix.is_square() and b.size != matrix.rows():           # neumann.rs:154-160
            raise SolverError(5, f"expected {matrix.rows()}, actual {b.size} in neumann_initialization")
        o = L.NeumannOptions()
        lib.sl_neumann_options_default(C.byref(o))
        o.tolerance, o.max_iterations = options.tolerance, options.max_iterations
        o.max_terms, o.series_tolerance = self.max_terms, self.series_tolerance
        o.order, o.start, o.residual, o.mem = self.order, self.start, self.residual, L.SL_MEM_HOST
        o.collect_stats = int(options.collect_stats)
        o.compute_error_bounds = int(options.compute_error_bounds and self.adaptive_truncation)
        guess = None
        if options.initial_guess is not None:
            guess = _f64(options.initial_guess)
            if guess.size != matrix.rows():                          # neumann.rs:198-204
                raise SolverError(5, f"expected {matrix.rows()}, actual {guess.size} in initial_guess")
            o.start = L.SL_START_INITIAL_GUESS
        x = np.empty(matrix.rows(), dtype=np.float64)
        tn = np.zeros(max(self.max_terms, 1), dtype=np.float64)
        r = L.NeumannResult()
        st = lib.sl_neumann_solve(matrix._h, L.ptr(b), L.ptr(guess), C.byref(o), L.ptr(x), L.ptr(tn), C.byref(r))
        if st != L.SL_OK:
            msg = lib.sl_last_error_message().decode()
            err = SolverError(st, msg)
            if st == 3:  # ConvergenceFailure still carries the partial result (the reference drops it)
                err.result = SolverResult(x, r.residual_norm, int(r.iterations), False)
            raise err
        stats = None
        if options.collect_stats:
            stats = {"total_time_ms": r.total_time_ms, "matvec_count": int(r.matvec_count),
                     "device_time_ms": r.device_time_ms, "bytes_moved": int(r.bytes_moved)}
        return SolverResult(x, r.residual_norm, int(r.iterations), bool(r.converged),
                            r.error_bound if r.error_bound >= 0 else None, stats, tn[: int(r.terms_computed)].copy())


class Communicator:
    """sl_comm: one process per GPU of one node; every rank of the job passes the same `name` (the rendezvous is a shared-memory
    block that disappears once all ranks have joined).  No torch / MPI / RCCL involved (include/sublinear_hip.h, multi-GPU)."""

    def __init__(self, rank: int, world: int, name: str):
        self._h = C.c_void_p()
        L.check(L.load().sl_comm_create(int(rank), int(world), name.encode(), C.byref(self._h)))
        self.rank, self.world = int(rank), int(world)

    def barrier(self) -> None:
        L.check(L.load().sl_comm_barrier(self._h))

    def allgather_u64(self, value: int) -> list:
        out = (C.c_uint64 * self.world)()
        L.check(L.load().sl_comm_allgather_u64(self._h, int(value), out))
        return [int(v) for v in out]

    def allgather_f64(self, value: float) -> list:
        bits = int(np.array([value], dtype=np.float64).view(np.uint64)[0])
        return [float(np.array([b], dtype=np.uint64).view(np.float64)[0]) for b in self.allgather_u64(bits)]

    def close(self) -> None:
        if self._h:
            L.load().sl_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NeumannState:
    """NeumannState (src/solver/neumann.rs:95-137) living on the device behind sl_neumann_state_*: the matrix layout, D^-1 and the
    vectors stay resident across update_rhs / run / reset — what an incremental caller of the reference's SolverAlgorithm holds."""

    def __init__(self, handle, matrix: SparseMatrix, max_terms: int, options: SolverOptions):
        self._h, self._matrix, self._max_terms, self._options = handle, matrix, max_terms, options   # the matrix must outlive the state
        self.converged = False
        self.last = None

    def update_rhs(self, delta_b) -> None:
        """NeumannSolver::update_rhs(state, &[(index, delta)]) — neumann.rs:436-462, in list order"""
        pairs = list(delta_b)
        idx = np.ascontiguousarray([int(i) for i, _ in pairs], dtype=np.uint64)
        val = np.ascontiguousarray([float(v) for _, v in pairs], dtype=np.float64)
        L.check(L.load().sl_neumann_state_update_rhs(self._h, len(pairs), L.ptr(idx), L.ptr(val)))

    def run(self) -> SolverResult:
        """the loop of NeumannSolver::solve (neumann.rs:477-555) from the state's current position"""
        tn = np.zeros(max(self._max_terms, 1), dtype=np.float64)
        r = L.NeumannResult()
        st = L.load().sl_neumann_state_run(self._h, L.ptr(tn), C.byref(r))
        self.converged = bool(r.converged)
        res = SolverResult(self.solution(), r.residual_norm, int(r.iterations), bool(r.converged),
                           r.error_bound if r.error_bound >= 0 else None,
                           {"matvec_count": int(r.matvec_count), "device_time_ms": r.device_time_ms, "terms_computed": int(r.terms_computed)},
                           tn[: min(int(r.terms_computed), tn.size)].copy())
        self.last = res
        if st != L.SL_OK:
            err = SolverError(st, L.load().sl_last_error_message().decode())
            err.result = res
            raise err
        return res

    def solution(self) -> np.ndarray:
        x = np.empty(self._matrix.rows(), dtype=np.float64)
        L.check(L.load().sl_neumann_state_solution(self._h, L.ptr(x), L.SL_MEM_HOST))
        return x

    def run_steps(self, steps: int):
        """`steps` fused steps without the stop rule (the measurement loop): returns (||t||^2 of the last step, device ms)"""
        nrm, ms = C.c_double(0.0), C.c_float(0.0)
        L.check(L.load().sl_neumann_state_run_steps(self._h, int(steps), C.byref(nrm), C.byref(ms)))
        return nrm.value, ms.value

    def reset(self) -> None:
        """SolverState::reset (neumann.rs:367-378): solution = 0, current term = the (updated) scaled rhs, counters cleared"""
        L.check(L.load().sl_neumann_state_reset(self._h))
        self.converged = False

    def close(self) -> None:
        if self._h:
            L.load().sl_neumann_state_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class ConjugateGradientSolver:
    """OptimizedConjugateGradientSolver (src/optimized_solver.rs:167-295; config defaults :119-127) and
    FastConjugateGradient (src/fast_solver.rs:110-178) over sl_cg_solve."""
    max_iterations: int = 1000
    tolerance: float = 1e-6
    order: int = L.SL_ORDER_CSR_SEQUENTIAL

    def solve(self, matrix: SparseMatrix, b) -> SolverResult:
        lib = L.load()
        b = _f64(b)
        if not matrix.is_square():
            raise SolverError(4, "Matrix must be square")                                    # optimized_solver.rs:187-190
        if b.size != matrix.rows():
            raise SolverError(5, "Right-hand side vector length must match matrix size")     # :191-193
        o = L.CgOptions()
        lib.sl_cg_options_default(C.byref(o))
        o.tolerance, o.max_iterations, o.order, o.mem = self.tolerance, self.max_iterations, self.order, L.SL_MEM_HOST
        x = np.empty(matrix.rows(), dtype=np.float64)
        r = L.CgResult()
        L.check(lib.sl_cg_solve(matrix._h, L.ptr(b), C.byref(o), L.ptr(x), C.byref(r)))
        return SolverResult(x, r.residual_norm, int(r.iterations), bool(r.converged), None,
                            {"matvec_count": int(r.matvec_count), "total_time_ms": r.total_time_ms, "device_time_ms": r.device_time_ms})


@dataclass
class PushSolver:
    """Synchronous thresholded residual push: the data-parallel member of
    ForwardPushSolver::push_node (forward_push.rs:179-216) / TS solveForwardPush (solver.ts:437-522)."""
    theta: float = 1e-6
    max_rounds: int = 10_000
    order: int = L.SL_ORDER_CSR_SEQUENTIAL
    dense_switch: float = 1.0 / 16.0
    theta_rows: object = None      # optional per-row thresholds (used instead of theta): the degree-scaled rule of forward_push.rs:93-99

    def solve(self, matrix: SparseMatrix, b, x0=None, log_frontier: int = 0):
        lib = L.load()
        b = _f64(b)
        n = matrix.rows()
        if b.size != n:
            raise SolverError(5, f"expected {n}, actual {b.size}")
        o = L.PushOptions()
        lib.sl_push_options_default(C.byref(o))
        o.theta, o.max_rounds, o.order, o.mem, o.dense_switch = self.theta, self.max_rounds, self.order, L.SL_MEM_HOST, self.dense_switch
        th = None
        if self.theta_rows is not None:
            th = _f64(self.theta_rows)
            if th.size != n:
                raise SolverError(5, f"theta_rows: expected {n}, actual {th.size}")
            o.theta_rows = th.ctypes.data
        x = np.zeros(n, dtype=np.float64) if x0 is None else _f64(x0).copy()
        r = np.empty(n, dtype=np.float64)
        log = np.zeros(max(log_frontier, 1), dtype=np.uint32)
        words = L.u64(0)
        res = L.PushResult()
        L.check(lib.sl_push_solve(matrix._h, L.ptr(b), C.byref(o), L.ptr(x), L.ptr(r),
                                  L.ptr(log) if log_frontier else None, log_frontier, C.byref(words), C.byref(res)))
        out = {"solution": x, "residual": r, "rounds": int(res.rounds), "pushes": int(res.pushes),
               "rows_touched": int(res.rows_touched), "dense_rounds": int(res.dense_rounds),
               "residual_norm": res.residual_norm, "converged": bool(res.converged),
               "device_time_ms": res.device_time_ms}
        if log_frontier:
            out["frontier_log"] = log[: int(words.value)].copy()
        return out


@dataclass
class GaussSouthwellSolver:
    """TS solveForwardPush (src/core/solver.ts:437-522) in the reference's own visiting order: one push per step at the first
    index of largest |r_i|.  Sequential by definition — for order-exact parity (push sequence, iteration count, solution bits);
    PushSolver is the throughput path.  Raises SolverError(ConvergenceFailure) after max_iterations pushes like the reference;
    `on_failure="return"` hands back the partial state instead."""
    epsilon: float = 1e-6
    max_iterations: int = 1000

    def solve(self, matrix: SparseMatrix, b, log_pushes: bool = False, on_failure: str = "raise"):
        lib = L.load()
        b = _f64(b)
        n = matrix.rows()
        if b.size != n:
            raise SolverError(5, f"expected {n}, actual {b.size}")
        o = L.SouthwellOptions()
        lib.sl_southwell_options_default(C.byref(o))
        o.epsilon, o.max_iterations, o.mem = self.epsilon, self.max_iterations, L.SL_MEM_HOST
        x, r = np.zeros(n), np.zeros(n)
        cap = min(self.max_iterations, 1 << 26) if log_pushes else 0
        log = np.zeros(max(cap, 1), dtype=np.uint32)
        res = L.SouthwellResult()
        st = lib.sl_forward_push_southwell(matrix._h, L.ptr(b), C.byref(o), L.ptr(x), L.ptr(r), L.ptr(log) if cap else None, cap, C.byref(res))
        if st != 0 and not (st == 3 and on_failure == "return"):
            L.check(st)
        out = {"solution": x, "residual_vector": r, "iterations": int(res.iterations), "residual": res.residual_norm,
               "converged": bool(res.converged), "device_time_ms": res.device_time_ms}
        if log_pushes:
            out["push_log"] = log[: min(cap, int(res.iterations))].copy()
        return out


def estimate_entry(matrix: SparseMatrix, b, row: int, theta: float = 1e-8, max_rounds: int = 100_000,
                   matrix_is_transpose: bool = False, device: bool = False):
    """x_row = e_row^T A^-1 b by local push on A^T (sl_estimate_entry); matrix_is_transpose: `matrix` already
    holds A^T (sl_estimate_entry_transposed).  device=True: b is a torch CUDA tensor."""
    lib = L.load()
    if not device:
        b = _f64(b)
    if not (0 <= row < matrix.rows()):
        raise SolverError(4, f"Row index {row} out of bounds. Matrix has {matrix.rows()} rows")
    res = L.EstimateResult()
    fn = lib.sl_estimate_entry_transposed if matrix_is_transpose else lib.sl_estimate_entry
    L.check(fn(matrix._h, L.ptr(b), L.SL_MEM_DEVICE if device else L.SL_MEM_HOST, row, theta, max_rounds, C.byref(res)))
    return res


# ---- solver::utils (solver/mod.rs:363-461) on device-reduced vectors -----------------------------------------------------------------
NORM_TYPES = {"l1": L.SL_NORM_L1, "l2": L.SL_NORM_L2, "linf": L.SL_NORM_LINF, "weighted": L.SL_NORM_WEIGHTED}
CONVERGENCE_MODES = {"residual_norm": L.SL_CONV_RESIDUAL_NORM, "relative_residual": L.SL_CONV_RELATIVE_RESIDUAL,
                     "solution_change": L.SL_CONV_SOLUTION_CHANGE, "relative_solution_change": L.SL_CONV_RELATIVE_SOLUTION_CHANGE,
                     "combined": L.SL_CONV_COMBINED}


def _norm(fn_name: str, v) -> float:
    v = _f64(v)
    out = C.c_double(0.0)
    L.check(getattr(L.load(), fn_name)(v.size, L.ptr(v), C.byref(out), L.SL_MEM_HOST))
    return out.value


def l2_norm(v) -> float:
    """utils::l2_norm, solver/mod.rs:369-371"""
    return _norm("sl_l2_norm", v)


def l1_norm(v) -> float:
    """utils::l1_norm, solver/mod.rs:374-376"""
    return _norm("sl_l1_norm", v)


def linf_norm(v) -> float:
    """utils::linf_norm, solver/mod.rs:379-381 (NaN entries are ignored as f64::max ignores them)"""
    return _norm("sl_linf_norm", v)


def compute_norm(v, norm_type: str = "l2") -> float:
    """utils::compute_norm, solver/mod.rs:384-391 (weighted falls back to l2 there)"""
    if norm_type not in NORM_TYPES:
        raise SolverError(4, f"Unknown norm type: {norm_type}")
    v = _f64(v)
    out = C.c_double(0.0)
    L.check(L.load().sl_compute_norm(v.size, L.ptr(v), NORM_TYPES[norm_type], C.byref(out), L.SL_MEM_HOST))
    return out.value


def compute_residual(matrix: "SparseMatrix", x, b, order: int = L.SL_ORDER_CSR_SEQUENTIAL) -> np.ndarray:
    """utils::compute_residual, solver/mod.rs:394-405: r = A x - b"""
    x, b = _f64(x), _f64(b)
    if x.size != matrix.cols() or b.size != matrix.rows():
        raise SolverError(5, f"Dimension mismatch: x {x.size} / b {b.size} against a {matrix.rows()} x {matrix.cols()} matrix")
    r = np.empty(matrix.rows())
    L.check(L.load().sl_compute_residual(matrix._h, L.ptr(x), L.ptr(b), L.ptr(r), order, L.SL_MEM_HOST))
    return r


def check_convergence(residual_norm: float, tolerance: float, mode: str, b_norm: float, prev_solution, current_solution) -> bool:
    """utils::check_convergence, solver/mod.rs:408-461"""
    if mode not in CONVERGENCE_MODES:
        raise SolverError(4, f"Unknown convergence mode: {mode}")
    cur = _f64(current_solution)
    prev = None if prev_solution is None else _f64(prev_solution)
    out = C.c_int(0)
    L.check(L.load().sl_check_convergence(float(residual_norm), float(tolerance), CONVERGENCE_MODES[mode], float(b_norm), cur.size,
                                          L.ptr(prev), L.ptr(cur), L.SL_MEM_HOST, C.byref(out)))
    return bool(out.value)


WALK_STREAMS = {"blocks": L.SL_WALK_STREAM_BLOCKS, "reference": L.SL_WALK_STREAM_SERIAL, "serial": L.SL_WALK_STREAM_SERIAL}


def _walk_stream(stream: str) -> int:
    if stream not in WALK_STREAMS:
        raise SolverError(4, f"Unknown random-walk stream: {stream} (blocks | reference)")
    return WALK_STREAMS[stream]


def random_walk_solve(matrix: SparseMatrix, b, epsilon: float, seed: int, num_walks: int = 0, stream: str = "blocks") -> dict:
    """solveRandomWalk, core/solver.ts:278-357 (the `random-walk` method of SublinearSolver.solve): every coordinate from
    max(100, ceil(1 / eps^2)) absorbing walks, mean and sample variance per coordinate, residual ||A x - b||_2, converged = residual < eps.
    stream = "blocks": one lane per walk, each reading its own block of the seed's stream (throughput); "reference": the ONE stream walked
    serially as the reference does — every x_i and variance bit-identical to solver.ts for the same seed (include/sublinear_hip.h, sl_walk_stream).
    Never raises on a missed epsilon (the TS-shaped caller does); the matrix needs its raw CSR (keep_csr / with_transpose)."""
    b = _f64(b)
    n = matrix.rows()
    if b.size != n:
        raise SolverError(5, f"Vector length {b.size} does not match matrix rows {n}")
    x, var, res = np.empty(n), np.empty(n), L.RandomWalkResult()
    st = L.load().sl_solve_random_walk(matrix._h, L.ptr(b), L.SL_MEM_HOST, float(epsilon), seed & 0xFFFFFFFF, _walk_stream(stream), int(num_walks),
                                       L.ptr(x), L.ptr(var), C.byref(res))
    if st not in (0, 3):
        L.check(st)
    return {"solution": x, "variances": var, "iterations": int(res.iterations), "num_walks": int(res.num_walks), "residual": res.residual,
            "total_variance": res.total_variance, "converged": bool(res.converged), "device_ms": res.device_time_ms}


class QuerySession:
    """Many single-entry queries against one system (ForwardPushSolver::new + query_single_entry,
    forward_push.rs:52-66, 224-231): setup once, then every query costs only the rows its push touches
    (sl_query_session_*).  matrix_is_transpose: `matrix` already holds A^T."""

    def __init__(self, matrix: SparseMatrix, b, matrix_is_transpose: bool = False, device: bool = False):
        lib = L.load()
        self._matrix, self._b = matrix, (b if device else _f64(b))          # keep both alive for the session's lifetime
        size = int(self._b.numel()) if device else self._b.size
        if size != matrix.rows():
            raise SolverError(5, f"Vector length {size} does not match matrix rows {matrix.rows()}")
        h = L.vp()
        L.check(lib.sl_query_session_create(matrix._h, int(matrix_is_transpose), L.ptr(self._b),
                                            L.SL_MEM_DEVICE if device else L.SL_MEM_HOST, C.byref(h)))
        self._h = h.value

    def estimate(self, row: int, theta: float = 1e-8, max_rounds: int = 100_000):
        if not (0 <= row < self._matrix.rows()):
            raise SolverError(4, f"Row index {row} out of bounds. Matrix has {self._matrix.rows()} rows")
        res = L.EstimateResult()
        L.check(L.load().sl_query_session_estimate(self._h, row, theta, max_rounds, C.byref(res)))
        return res

    def estimate_batch(self, rows, theta: float = 1e-8, max_rounds: int = 100_000, lanes: int = 0):
        """many independent queries at once (sl_query_session_estimate_batch): lanes with a state, a stream and a host thread each;
        every result equals the one-at-a-time answer bit for bit"""
        rows = np.ascontiguousarray([int(r) for r in rows], dtype=np.uint64)
        n = self._matrix.rows()
        if rows.size and int(rows.max()) >= n:
            raise SolverError(4, f"Row index {int(rows.max())} out of bounds. Matrix has {n} rows")
        res = (L.EstimateResult * max(1, rows.size))()
        L.check(L.load().sl_query_session_estimate_batch(self._h, int(rows.size), L.ptr(rows), theta, max_rounds, int(lanes), res))
        return list(res)[: rows.size]

    def close(self):
        if getattr(self, "_h", None):
            L.load().sl_query_session_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# Shipped TypeScript surface (src/core/solver.ts)
# ------------------------------------------------------------------------------------------------
def _matrix_from_json(matrix, **kw) -> SparseMatrix:
    """core/types.ts:6-22: {rows, cols, format:'dense', data:number[][]} or
    {rows, cols, format:'coo', values, rowIndices, colIndices} (CJS CLI nests COO under `data`,
    bin/cli.js:484-490); also accepts scipy sparse matrices and SparseMatrix."""
    if isinstance(matrix, SparseMatrix):
        return matrix
    if hasattr(matrix, "tocsr"):
        return SparseMatrix.from_scipy(matrix, **kw)
    fmt = matrix.get("format")
    rows, cols = int(matrix["rows"]), int(matrix["cols"])
    if fmt == "dense":
        data = np.asarray(matrix["data"], dtype=np.float64)
        if data.shape != (rows, cols):
            raise SolverError(5, f"dense data shape {data.shape} != ({rows}, {cols})")
        return SparseMatrix.from_dense(data, **kw)
    if fmt == "coo":
        src = matrix["data"] if isinstance(matrix.get("data"), dict) else matrix
        v, r, c = src["values"], src["rowIndices"], src["colIndices"]
        if not (len(v) == len(r) == len(c)):
            raise SolverError(4, "COO arrays differ in length")
        return SparseMatrix.from_triplets(zip(r, c, v), rows, cols, **kw)
    raise SolverError(6, f"Unsupported matrix format: {fmt}")


class SublinearSolver:
    """new SublinearSolver({method, epsilon, maxIterations, timeout?, seed?}) — core/solver.ts:36-58.

    `stream` (random-walk method and the random-walk branch of estimate_entry): "blocks" = one lane per walk, every walk its own block of
    the seed's stream; "reference" = the reference's ONE serial stream — results bit-identical to solver.ts for the same seed.

    `neumann` runs the exact series (the TS sign bug of solver.ts:157-163 is NOT reproduced,
    SURVEY.md §0.3); `forward-push` / `backward-push` / `bidirectional` (aliases in the reference,
    solver.ts:527-545) run the push: `push_order="reference"` = solveForwardPush's own order, one Gauss-Southwell push per
    iteration (GaussSouthwellSolver; `iterations` = pushes, as the reference reports them), `"synchronous"` = the data-parallel
    thresholded push with theta = epsilon (`iterations` = rounds), `"auto"` (default) = reference order up to 4096 rows."""

    def __init__(self, method: str = "neumann", epsilon: float = 1e-6, max_iterations: int = 1000,
                 timeout: Optional[float] = None, seed: Optional[int] = None, push_order: str = "auto", stream: str = "blocks"):
        if push_order not in ("auto", "reference", "synchronous"):
            raise SolverError(4, f"Unknown push_order: {push_order}")
        self.push_order = push_order
        _walk_stream(stream)
        self.stream = stream
        if method not in ("neumann", "random-walk", "forward-push", "backward-push", "bidirectional"):
            raise SolverError(4, f"Unknown method: {method}")
        if not (epsilon > 0):
            raise SolverError(4, "epsilon must be positive")
        self.method, self.epsilon, self.max_iterations, self.timeout, self.seed = method, epsilon, max_iterations, timeout, seed

    def solve(self, matrix, vector, progress_callback=None) -> dict:
        """solve(matrix, vector, progressCallback?) — core/solver.ts:58-111; the callback gets one report {iteration, residual, elapsed} when the
        device loop has ended (the loop runs on the device: there is no per-iteration host turn to report from)"""
        import time
        t0 = time.perf_counter()
        push = self.method in ("forward-push", "backward-push", "bidirectional")
        if isinstance(matrix, dict):
            # the reference's gate, in its order (solver.ts:59-78): validateMatrix, vector length against the COLUMNS, then analyzeMatrix —
            # "Matrix is not diagonally dominant" (row or column dominance, read the way its getEntry / getRowSum read the matrix: golden G14)
            from . import io
            analysis = io.analyze_matrix(matrix)                           # (validates first)
            if len(vector) != int(matrix["cols"]):
                raise SolverError(5, f"Vector length {len(vector)} does not match matrix columns {int(matrix['cols'])}")
            if not analysis["isDiagonallyDominant"]:
                raise SolverError(1, "Matrix is not diagonally dominant")
            if not push and self.method == "neumann" and analysis["dominanceType"] != "row":
                push = True      # Neumann needs ROW dominance (neumann.rs:139-170); a column-dominant system goes through the push, which does not
        m = _matrix_from_json(matrix, with_transpose=push, keep_csr=self.method == "random-walk")
        b = _f64(vector)
        if b.size != m.rows():
            raise SolverError(5, f"Vector length {b.size} does not match matrix rows {m.rows()}")
        if self.method == "random-walk":
            # solveRandomWalk (solver.ts:278-357): max(100, ceil(1 / eps^2)) walks per coordinate; a stream per walk from the seed (the
            # reference seeds ONE stream with `seed || Date.now()`); a residual that misses epsilon raises as the reference throws (:335-341)
            import time as _t
            seed = (self.seed if self.seed is not None else int(_t.time() * 1e3)) & 0xFFFFFFFF
            rw = random_walk_solve(m, b, self.epsilon, seed, stream=self.stream)
            if not rw["converged"]:
                raise SolverError(3, "Random walk sampling failed to achieve desired accuracy")
            sol, it, res, conv = rw["solution"], rw["iterations"], rw["residual"], True
        elif not push:
            ns = NeumannSolver(max_terms=self.max_iterations, series_tolerance=self.epsilon)
            r = ns.solve(m, b, SolverOptions(tolerance=self.epsilon, max_iterations=self.max_iterations))
            sol, it, res, conv = r.solution, r.iterations, r.residual_norm, r.converged
        elif self.push_order == "reference" or (self.push_order == "auto" and m.rows() <= 4096):
            gs = GaussSouthwellSolver(epsilon=self.epsilon, max_iterations=self.max_iterations).solve(m, b)      # raises ConvergenceFailure like solver.ts:509-515
            sol, it, res, conv = gs["solution"], gs["iterations"], gs["residual"], True
        else:
            pr = PushSolver(theta=self.epsilon, max_rounds=self.max_iterations).solve(m, b)
            if not pr["converged"]:
                raise SolverError(3, f"Forward push failed to converge after {self.max_iterations} iterations")
            sol, it, res, conv = pr["solution"], pr["rounds"], pr["residual_norm"], True
        elapsed_ms = (time.perf_counter() - t0) * 1e3
        if self.timeout and elapsed_ms > self.timeout:               # TimeoutController.checkTimeout, core/utils.ts:319-325 (measured per solve)
            raise SolverError(3, f"Operation timed out after {self.timeout}ms")
        if progress_callback is not None:
            progress_callback({"iteration": it, "residual": res, "elapsed": elapsed_ms})
        return {"solution": sol, "iterations": it, "residual": res, "converged": conv, "method": self.method,
                "computeTime": elapsed_ms, "memoryUsed": int(m.info().device_bytes)}

    def compute_pagerank(self, adjacency, damping: float = 0.85, epsilon: float = 1e-6, max_iterations: int = 1000, personalized=None) -> np.ndarray:
        """computePageRank(adjacency, {damping, epsilon, maxIterations, personalized?}) — core/solver.ts:664-722: the system
        (I - d P^T) x = (1 - d) / n (or `personalized`), P_ji = adj[j][i] / out_j for out_j > 0 (a dangling node's column stays zero, as there),
        solved by THIS solver's method with the call's epsilon / maxIterations; returns the solution vector.  The system is assembled in
        CSR (the reference builds a dense n x n table)."""
        from . import io, generators as G
        if not (0.0 <= damping <= 1.0):                                  # ValidationUtils.validateRange, core/utils.ts
            raise SolverError(4, "damping must be between 0 and 1")
        if not (epsilon > 0):
            raise SolverError(4, "epsilon must be a positive number")
        r, c, v, rows, cols = io.matrix_to_triplets(adjacency)
        if rows != cols:
            raise SolverError(5, "Adjacency matrix must be square")
        # the reference's own arithmetic and its own reading of the adjacency (first stored match of a duplicated entry): the system carries
        # its bits (golden G13, tests/golden/make_golden_ts_pagerank.py)
        arp, aci, ava = G.adjacency_csr_first_match(r, c, v, rows)
        rp, ci, va, b = G.pagerank_system(rows, arp, aci, ava, damping)
        if personalized is not None:
            b = _f64(personalized)
            if b.size != rows:
                raise SolverError(5, f"Vector length {b.size} does not match matrix rows {rows}")
        m = SparseMatrix.from_csr(rp, ci, va, rows, rows, with_transpose=True, keep_csr=True)
        inner = SublinearSolver(method=self.method, epsilon=epsilon, max_iterations=max_iterations, timeout=self.timeout, seed=self.seed, push_order=self.push_order,
                                stream=self.stream)
        return inner.solve(m, b)["solution"]

    def estimate_entry(self, matrix, vector, row: int, column: int = 0, epsilon: Optional[float] = None,
                       confidence: float = 0.95, method: str = "neumann", entry_of: str = "solution") -> dict:
        """estimateEntry(matrix, vector, {row, column, epsilon, confidence, method}) -> {estimate,
        variance, confidence} (solver.ts:550-554).  entry_of = "solution" (default): x_row = (A^-1 vector)_row — what the reference's
        random-walk branch estimates and what the name says.  entry_of = "inverse": (A^-1)[row][column], `vector` IGNORED — what the
        reference's non-random-walk branch computes (it solves A x = e_column and reads x[row], solver.ts:603-620; listed as a defect in
        SURVEY §8, offered for callers that relied on it; by the same local push: e_row^T A^-1 e_column)."""
        if entry_of not in ("solution", "inverse"):
            raise SolverError(4, f"Unknown entry_of: {entry_of}")
        m = _matrix_from_json(matrix, with_transpose=True)
        if not (0 <= row < m.rows()):
            raise SolverError(4, f"Row index {row} out of bounds. Matrix has {m.rows()} rows (valid range: 0-{m.rows() - 1})")
        if not (0 <= column < m.cols()):
            raise SolverError(4, f"Column index {column} out of bounds. Matrix has {m.cols()} columns")
        b = _f64(vector)
        if b.size != m.rows():
            raise SolverError(5, f"Vector length {b.size} does not match matrix rows {m.rows()}")
        eps = epsilon if epsilon is not None else self.epsilon
        if entry_of == "inverse" and method != "random-walk":
            b = np.zeros(m.rows())
            b[column] = 1.0
        if method == "random-walk":                                      # solver.ts:585-601, 630-648
            res = L.WalkResult()
            seed = (self.seed if self.seed is not None else 0) & 0xFFFFFFFF
            L.check(L.load().sl_estimate_entry_random_walk(m._h, L.ptr(b), L.SL_MEM_HOST, row, eps, seed, _walk_stream(self.stream), 0, None, C.byref(res)))
            return {"estimate": res.estimate, "variance": res.variance, "confidence": confidence, "numSamples": int(res.num_samples)}
        r = estimate_entry(m, b, row, theta=eps * 1e-2, max_rounds=self.max_iterations * 100)
        return {"estimate": r.estimate, "variance": 0.0, "confidence": 1.0 if r.converged else 0.5,
                "residual_l1": r.residual_l1}
