/*
 * sublinear_hip.h — C ABI of the MI355X-native push / Neumann-series solver.
 *
 * This is the drop-in boundary for ONE hot path of ruvnet/sublinear-time-solver:
 * the CSR residual-push SpMV + per-row diagonal normalisation + frontier
 * compaction that sits under the crate's `solve()` / `estimateEntry()` surface.
 * A Rust (cgo / ctypes / N-API ...) binding needs only this header; there are no
 * C++ or torch types in any signature.  INTEGRATION.md shows the Rust shim.
 *
 * Every entry point names the reference interface it replaces (paths relative
 * to the reference root, @ 2025-09-19).
 *
 * Conventions
 *   - every function returns sl_status (0 = OK); nothing throws or aborts across
 *     the ABI; sl_last_error_message() gives the detail text of the calling
 *     thread's last failure.
 *   - pointers are HOST pointers unless the parameter is documented as device
 *     memory or the call takes an `sl_mem` selector.
 *   - sl_matrix owns its device copies (uploaded/converted once at create) and is
 *     immutable afterwards: it may be shared by concurrent solves.
 *   - all arithmetic is IEEE binary64, indices are uint32 (types.rs:19-22:
 *     Precision = f64, IndexType = u32).
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     returns SL_DEVICE_ERROR.
 */
#ifndef SUBLINEAR_HIP_H
#define SUBLINEAR_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3 (round 5): + sl_spmv_add, sl_matrix_diagonal_dominance_factor, sl_matrix_spectral_radius_estimate, and the round-4 additions
 * (sl_neumann_state_current_term / _solution_rows, sl_backward_push_acl_with_source / _reachability, sl_acl_extrapolated_solution):
 * a library that lacks any of them answers 2 and is refused by the bindings before a symbol lookup can fail. */
/* 4 (round 5): + the element / iterator / norm side of trait Matrix — sl_matrix_get, sl_matrix_row, sl_matrix_col,
 * sl_matrix_frobenius_norm, sl_matrix_sparsity_info (with the calls of version 3, `impl Matrix for HipMatrix` is complete) — and
 * sl_solve_random_walk, the `random-walk` method of the TS solve(). */
/* 5 (round 6): + sl_device_name; the in-place mutators sl_matrix_scale / sl_matrix_add_diagonal; sl_l1_norm / sl_linf_norm /
 * sl_compute_norm / sl_compute_residual / sl_check_convergence (solver::utils); sl_neumann_options_streaming; and a `stream` argument
 * on sl_estimate_entry_random_walk / sl_solve_random_walk (SL_WALK_STREAM_SERIAL = the reference's one serial stream, bit for bit). */
#define SL_ABI_VERSION 5

/* ---- status codes: 1:1 with SolverError variants (src/error.rs:16-140) ---------- */
typedef enum {
    SL_OK = 0,
    SL_NOT_DIAGONALLY_DOMINANT = 1, /* MatrixNotDiagonallyDominant  error.rs:18 */
    SL_NUMERICAL_INSTABILITY = 2,   /* NumericalInstability         error.rs:28 */
    SL_CONVERGENCE_FAILURE = 3,     /* ConvergenceFailure           error.rs:38 */
    SL_INVALID_INPUT = 4,           /* InvalidInput                 error.rs:50 */
    SL_DIMENSION_MISMATCH = 5,      /* DimensionMismatch            error.rs:58 */
    SL_UNSUPPORTED_FORMAT = 6,      /* UnsupportedMatrixFormat      error.rs:68 */
    SL_ALLOCATION = 7,              /* MemoryAllocationError        error.rs:78 */
    SL_INDEX_OUT_OF_BOUNDS = 8,     /* IndexOutOfBounds             error.rs:86 */
    SL_INVALID_SPARSE_MATRIX = 9,   /* InvalidSparseMatrix          error.rs:96 */
    SL_ALGORITHM_ERROR = 10,        /* AlgorithmError               error.rs:104 */
    SL_DEVICE_ERROR = 11            /* new: HIP / RCCL runtime failure */
} sl_status;

typedef enum { SL_MEM_HOST = 0, SL_MEM_DEVICE = 1 } sl_mem;

/* summation order of one row's dot product */
typedef enum {
    SL_ORDER_CSR_SEQUENTIAL = 0, /* CSRStorage::multiply_vector, matrix/sparse.rs:187-203 */
    SL_ORDER_SIMD4 = 1,          /* simd_ops::matrix_vector_multiply_simd, simd_ops.rs:20-88 */
    /* Any order (opt-in): the caller accepts row sums added in whatever order the device finds fastest — results equal the
     * reference's to rounding (<= 1e-10 relative on x, BASELINE north_star's tolerance) instead of bit for bit, and differ by rounding
     * from run to run.  Takes the order-free column stream of a matrix created with SL_MATRIX_ORDER_ANY (uniformly random columns:
     * accumulation by LDS atomics, a CU's entries sorted by column); on any other matrix it runs the CSR order.  Neumann / SpMV /
     * residual only: the thresholded push (frontier lists bit-exact by contract) always runs an exact order. */
    SL_ORDER_ANY = 2
} sl_order;

typedef enum {
    SL_START_ZERO = 0,              /* SolverOptions.initial_guess = Some(zeros): exact series  */
    SL_START_REFERENCE_DEFAULT = 1, /* initial_guess = None: x0 = D^-1 b (neumann.rs:197-208)   */
    SL_START_INITIAL_GUESS = 2      /* caller-provided warm start                               */
} sl_start;

typedef enum {
    SL_RESIDUAL_TRUE = 0,            /* ||A x - b||_2                                            */
    SL_RESIDUAL_REFERENCE_SCALED = 1 /* ||A x - D^-1 b||_2, the quirk of neumann.rs:302-318      */
} sl_residual;

typedef struct sl_matrix sl_matrix; /* opaque, device resident */

/* matrix create flags */
#define SL_MATRIX_DEFAULT 0u
#define SL_MATRIX_WITH_TRANSPOSE 1u /* also build the column-structure needed by push / estimateEntry */
#define SL_MATRIX_KEEP_CSR 2u       /* keep the raw CSR arrays on the device next to the slice layout */
#define SL_MATRIX_COLUMN_PANELS 4u  /* build the column-panel layout whatever the size (by default: only where it pays) */
#define SL_MATRIX_NO_COLUMN_PANELS 8u /* never build it (saves 14 B per entry of HBM) */
#define SL_MATRIX_ROW_SLICE 32u     /* the rows are a row range of a SQUARE system of n_cols rows (what row_offset > 0 implies; this flag says it
                                       for the range that starts at row 0): sl_matrix_add_diagonal accepts such a matrix */
#define SL_MATRIX_ORDER_ANY 16u     /* where column panels pay, build the ORDER-FREE column stream (12 B per entry) instead of the ordered
                                       one: what SL_ORDER_ANY solves run on; exact orders on such a matrix take the row-slice kernels */

/* ---- library ------------------------------------------------------------------- */
int sl_abi_version(void);
const char *sl_last_error_message(void);
const char *sl_status_string(sl_status s);
sl_status sl_device_count(int *count);
/* hipDeviceProp_t::gcnArchName of `device` ("gfx950:sramecc+:xnack-" on an MI355X), NUL-terminated and cut to `capacity` bytes:
 * what a caller (and the test suite's first line) prints to show WHICH device the library runs on. */
sl_status sl_device_name(int device, char *name, uint64_t capacity);
sl_status sl_set_device(int device);
/* all launches of the calling thread go to `hip_stream` (a hipStream_t); NULL = default stream */
sl_status sl_set_stream(void *hip_stream);
sl_status sl_synchronize(void);
/* Solve calls take their device workspace (term vectors, frontier lists) from a per-thread cache instead of
 * allocating per call; this returns the cached buffers of the calling thread to the driver.  Cache size limit:
 * env SL_WORKSPACE_CACHE_MB (default 16384). */
void sl_release_workspace(void);

/* ---- a1 / a3: matrices ---------------------------------------------------------
 * replaces SparseMatrix::from_triplets (matrix/mod.rs:160-199) + COOStorage::from_triplets
 * (sparse.rs:530-548) + CSRStorage::from_coo (sparse.rs:80-132): bounds / finiteness
 * validation in input order, exact zeros dropped, STABLE sort by (row, col),
 * duplicates kept as separate entries. */
sl_status sl_matrix_create_from_triplets(uint64_t n_triplets, const uint64_t *rows, const uint64_t *cols,
                                         const double *values, uint64_t n_rows, uint64_t n_cols,
                                         uint32_t flags, sl_matrix **out);
/* adopts an existing CSRStorage {values, col_indices, row_ptr} (sparse.rs:16-23) —
 * what `SparseMatrix::as_csr` hands out.  `row_offset` > 0 makes this the row slice
 * [row_offset, row_offset + n_rows) of a larger square system (multi-GPU row-range
 * partition; column ids stay global, n_cols = global dimension). */
sl_status sl_matrix_create_csr(uint64_t n_rows, uint64_t n_cols, uint64_t nnz, const uint32_t *row_ptr,
                               const uint32_t *col_idx, const double *values, sl_mem where,
                               uint64_t row_offset, uint32_t flags, sl_matrix **out);
void sl_matrix_destroy(sl_matrix *m);

typedef struct {
    uint64_t n_rows, n_cols, nnz, row_offset;
    uint64_t padded_nnz;      /* entries stored in the slice layout (>= nnz) */
    uint64_t n_slices;        /* 64-row slices */
    uint64_t device_bytes;    /* HBM held by this matrix */
    uint64_t bandwidth;       /* max |col - row| over stored entries: selects the LDS band kernel */
    uint32_t max_row_nnz, min_row_nnz;
    uint32_t uniform_width;   /* != 0: every row has exactly this many entries */
    uint32_t has_transpose;
    uint32_t long_row_threshold; /* rows with more entries than this are served by the long-row kernel (2.5 x mean length, in [24, 256]) */
    uint32_t n_long_rows;
    uint32_t column_panels;   /* != 0: the matrix also carries a column-panel layout (columns spread beyond the LDS window and the L2):
                               * 1 = dynamic tiles (ragged matrices), 2 = paced persistent blocks (balanced matrices), 3 = the same with
                               * block-local rows and narrow panels (bands too wide for the LDS window: the gathers hit the CU's L1) */
    uint32_t reserved;
} sl_matrix_info;
sl_status sl_matrix_get_info(const sl_matrix *m, sl_matrix_info *info);
/* In-place mutators — SparseMatrix::scale / add_diagonal (matrix/mod.rs:346-372) over CSRStorage::scale / add_diagonal
 * (matrix/sparse.rs:229-248).  They are the two `&mut self` methods of the matrix: the caller must hold the matrix exclusively (no
 * solve running on it); Neumann states / query sessions created before the call keep the D^-1 they computed and must be re-created.
 *   sl_matrix_scale         every stored value *= factor (one rounded product per entry, in every layout copy the matrix carries).
 *   sl_matrix_add_diagonal  A += alpha I: in every row the entry `binary_search` finds at the row's own column gets += alpha; a row
 *                           WITHOUT a stored diagonal entry is silently skipped (sparse.rs:236-248: "we'd need to restructure the
 *                           matrix"), a row that stores its diagonal twice changes only the entry the halving search lands on.
 *                           Non-square matrix: SL_INVALID_INPUT (matrix/mod.rs:356-361) — except a row slice of a square system
 *                           (row_offset > 0 or SL_MATRIX_ROW_SLICE), whose "own column" is row_offset + row.
 * Both keep every device layout consistent (row slices, raw CSR, transpose in place; the sorted column streams of a matrix that
 * carries column panels are rebuilt from the updated rows). */
sl_status sl_matrix_scale(sl_matrix *m, double factor);
sl_status sl_matrix_add_diagonal(sl_matrix *m, double alpha);
/* SparseMatrix::as_csr / to_triplets (matrix/mod.rs:298-321): the CSR arrays back on the host — row_ptr (n_rows + 1), col_idx / values (nnz);
 * the raw copy where the matrix keeps one (SL_MATRIX_KEEP_CSR), otherwise written back from the row-slice layout (same arrays, same order) */
sl_status sl_matrix_download_csr(const sl_matrix *m, uint32_t *row_ptr, uint32_t *col_idx, double *values);

/* a6: SparseMatrix::is_diagonally_dominant (matrix/mod.rs:467-485), weak ROW dominance */
sl_status sl_matrix_is_diagonally_dominant(const sl_matrix *m, int *is_dd);
/* Matrix::diagonal_dominance_factor (matrix/mod.rs:487-514): min over the rows that have off-diagonal weight of |a_ii| / sum_j |a_ij|
 * (per row exactly as a6: last diagonal entry seen, |.| summed left to right).  *has_factor = 0 is the reference's None (no row has
 * off-diagonal entries, or the minimum is not finite), then *factor = 0.  The minimum of exactly computed row ratios: bit-exact. */
sl_status sl_matrix_diagonal_dominance_factor(const sl_matrix *m, int *has_factor, double *factor);
/* Matrix::spectral_radius_estimate (matrix/mod.rs:83-100): Gershgorin, max over rows of |a_ii| + sum_j |a_ij|; bit-exact.
 * With the two calls above it fills ConditioningInfo (matrix/mod.rs:548-556). */
sl_status sl_matrix_spectral_radius_estimate(const sl_matrix *m, double *radius);
/* Matrix::get (matrix/mod.rs:33, SparseMatrix::get :383-395 -> CSRStorage::get sparse.rs:142-155): *found = 0 is the reference's None
 * — row or column out of bounds, or no stored entry (an exact zero is never stored: from_triplets drops it).  A row that holds the
 * column more than once (duplicates are kept as separate entries, sparse.rs:80-132): Rust's slice::binary_search promises only that
 * "any one of the matches could be returned", and WHICH one depends on the standard library the reference is built with.  This
 * library (and its CPU checker) mirrors the classical halving search — lo = 0, hi = len; mid = lo + (hi - lo) / 2; return at the
 * first mid whose key is equal, else lo = mid + 1 / hi = mid — which is what std's binary_search_by did up to Rust 1.81; std >= 1.82
 * runs a branchless search without the early return and may land on another of the equal keys.  So: without duplicates in a row
 * (every fixture and generator of the reference) the answer is THE entry; with duplicates it is one of the stored values, the one
 * named above — bit-exact against a reference built with Rust <= 1.81, "a" match against newer ones (the same holds for the
 * diagonal lookups of NeumannState::new, sl_matrix_col and sl_matrix_add_diagonal, which use the same search).
 * `row` counts from the first row of this matrix (a row slice: local row), `col` is the global column. */
sl_status sl_matrix_get(const sl_matrix *m, uint64_t row, uint64_t col, int *found, double *value);
/* Matrix::row_iter (matrix/mod.rs:37, CSRStorage::row_iter sparse.rs:158-176): the (column, value) pairs of one row in stored order
 * = ascending column, duplicates in input order.  *count = the row's length (0 for a row out of bounds: the reference's empty
 * iterator); the first min(*count, capacity) pairs are written (cols / values may be null when capacity is 0). */
sl_status sl_matrix_row(const sl_matrix *m, uint64_t row, uint64_t capacity, uint32_t *cols, double *values, uint64_t *count);
/* Matrix::col_iter (matrix/mod.rs:41, CSRColIter sparse.rs:273-298): row after row ascending, the pair (row, get(row, col)) of every
 * row that holds the column — ONE pair per row even where the row holds it twice (the reference searches each row once).
 * *count = the number of such rows; the first min(*count, capacity) pairs are written. */
sl_status sl_matrix_col(const sl_matrix *m, uint64_t col, uint64_t capacity, uint32_t *rows, double *values, uint64_t *count);
/* Matrix::frobenius_norm (matrix/mod.rs:74-82): sqrt of the sum of value^2 over the stored entries.  The reference adds the squares
 * one after the other in row-major order; the device sums every row in stored order and the rows in a fixed tree — deterministic, and
 * equal to the reference's value to rounding (tests: 1e-12 relative, like the device's other tree-reduced norms). */
sl_status sl_matrix_frobenius_norm(const sl_matrix *m, double *norm);
/* Matrix::sparsity_info (matrix/mod.rs:523-545; SparsityInfo, types.rs:114-129, ::new :344-369) */
typedef struct {
    uint64_t nnz, rows, cols;         /* dimensions = (rows, cols) */
    double sparsity_ratio;            /* nnz / (rows * cols) as f64 / f64; 0 for an empty shape */
    double avg_nnz_per_row;           /* nnz / rows; 0 without rows */
    uint64_t max_nnz_per_row;
    uint64_t bandwidth;               /* max |row - col| over the stored entries: always Some(..) in the reference (0 without entries) */
    int32_t is_banded;                /* bandwidth < rows / 4 (integer division, matrix/mod.rs:542) */
    int32_t reserved;
} sl_sparsity_info;
sl_status sl_matrix_sparsity_info(const sl_matrix *m, sl_sparsity_info *info);
/* a7 (first half): D^-1 with the reference's rejection rules (neumann.rs:172-188):
 * missing diagonal or |d| < 1e-14 -> SL_INVALID_SPARSE_MATRIX.  dinv: n_rows doubles. */
sl_status sl_matrix_diagonal_inverse(const sl_matrix *m, double *dinv, sl_mem where);

/* ---- a2 / a3 / a5: primitives ----------------------------------------------------
 * y = A x — Matrix::multiply_vector (matrix/mod.rs:415-439) in either summation order. */
sl_status sl_spmv(const sl_matrix *m, const double *x, double *y, sl_order order, sl_mem where);
/* y += A x — Matrix::multiply_vector_add (matrix/mod.rs:47, 441-465) over CSRStorage::multiply_vector_add (sparse.rs:192-203): the
 * running sum of row i starts from y_i, (y_i + a_0 x_0) + a_1 x_1 + ..., so the bits differ from sl_spmv followed by an add.
 * SL_ORDER_CSR_SEQUENTIAL (SL_ORDER_ANY runs the same order); SL_ORDER_SIMD4: SL_INVALID_INPUT — simd_ops.rs has no accumulating
 * form.  x (n_cols) and y (n_rows) must not alias. */
sl_status sl_spmv_add(const sl_matrix *m, const double *x, double *y, sl_order order, sl_mem where);
/* simd_ops::dot_product_simd / axpy_simd (simd_ops.rs:116-189), solver::utils::l2_norm
 * (solver/mod.rs:369-371).  Device reductions use a fixed tree: run-to-run
 * deterministic, equal to the sequential CPU sum to rounding (not bitwise). */
sl_status sl_dot(uint64_t n, const double *x, const double *y, double *out, sl_mem where);
sl_status sl_axpy(uint64_t n, double alpha, const double *x, double *y, sl_mem where);
sl_status sl_l2_norm(uint64_t n, const double *x, double *out, sl_mem where);
/* solver::utils::l1_norm / linf_norm (solver/mod.rs:374-381): sum of |x_i| (fixed tree, equal to the sequential sum to rounding);
 * max of |x_i| folded from 0.0 with f64::max — exact in any order; a NaN entry is ignored exactly as f64::max ignores it. */
sl_status sl_l1_norm(uint64_t n, const double *x, double *out, sl_mem where);
sl_status sl_linf_norm(uint64_t n, const double *x, double *out, sl_mem where);
/* solver::utils::compute_norm (solver/mod.rs:384-391) over NormType (types.rs): Weighted falls back to L2 as in the reference. */
typedef enum { SL_NORM_L1 = 0, SL_NORM_L2 = 1, SL_NORM_LINF = 2, SL_NORM_WEIGHTED = 3 } sl_norm_type;
sl_status sl_compute_norm(uint64_t n, const double *x, sl_norm_type norm_type, double *out, sl_mem where);
/* solver::utils::compute_residual (solver/mod.rs:394-405): residual = A x, then residual_i -= b_i (per row bit-identical to the
 * reference: the row sum in `order`, then one subtraction).  x: n_cols, b / residual: n_rows. */
sl_status sl_compute_residual(const sl_matrix *m, const double *x, const double *b, double *residual, sl_order order, sl_mem where);
/* solver::utils::check_convergence (solver/mod.rs:408-461) over ConvergenceMode (types.rs:30-41): ResidualNorm: residual_norm <=
 * tolerance; RelativeResidual: residual_norm / b_norm <= tolerance (b_norm > 0, else the absolute test); SolutionChange:
 * ||current - prev||_2 <= tolerance; RelativeSolutionChange: ||current - prev||_2 / ||prev||_2 <= tolerance (||prev|| > 0, else the
 * absolute test) — both false without a previous solution (prev_solution == NULL); Combined: residual_norm <= tolerance and
 * (b_norm == 0 or residual_norm / b_norm <= tolerance).  The two vector modes reduce their sums of squares on the device (fixed tree:
 * the reference's sequential sums to rounding; n entries each in `where`); the other modes are host arithmetic on the caller's numbers. */
typedef enum { SL_CONV_RESIDUAL_NORM = 0, SL_CONV_RELATIVE_RESIDUAL = 1, SL_CONV_SOLUTION_CHANGE = 2, SL_CONV_RELATIVE_SOLUTION_CHANGE = 3,
               SL_CONV_COMBINED = 4 } sl_convergence_mode;
sl_status sl_check_convergence(double residual_norm, double tolerance, sl_convergence_mode mode, double b_norm, uint64_t n,
                               const double *prev_solution, const double *current_solution, sl_mem where, int *converged);

/* ---- a8 / a9: the fused Neumann step (device pointers only) -------------------------
 * One pass of NeumannState::apply_iteration_matrix (neumann.rs:280-299) +
 * `solution += term` and the term norm of compute_next_term (:264-274):
 *     t_out_i = t_in[i0+i] - dinv_i * (A t_in)_i ;  x_i += t_out_i ;  *norm2 = sum t_out_i^2
 * t_in has n_cols entries (global), t_out / x / dinv have n_rows entries (local slice;
 * i0 = row_offset).  Per-row arithmetic is bit-identical to the reference's scalar
 * loops (product rounded, then added, column order).  norm2 is a device double. */
sl_status sl_neumann_step(const sl_matrix *m, const double *dinv, const double *t_in, double *t_out,
                          double *x, double *norm2, sl_order order);
/* a10 on device pointers (row slices included): r = A x - rhs over the local rows, *norm2 = sum r_i^2
 * (NeumannState::update_residual, neumann.rs:302-318; rhs = b for the true residual, D^-1 b for the
 * reference's scaled one).  x_full has n_cols entries, rhs / r_out n_rows; r_out may be NULL. */
sl_status sl_residual_norm2(const sl_matrix *m, const double *x_full, const double *rhs, double *r_out, double *norm2,
                            sl_order order);
/* The fused step for callers that cut one rank's rows into several matrices (multi-GPU: the rows within w of a slice
 * boundary first, so that their halo travels while the interior rows compute).  Each piece leaves its per-block partial
 * sums of ||t_out||^2 in `partials` (device; capacity from sl_matrix_partials_capacity) and reports how many it wrote;
 * one fixed-order sl_reduce_partials over the concatenated pieces closes the step. */
sl_status sl_matrix_partials_capacity(const sl_matrix *m, uint64_t *count);
sl_status sl_neumann_step_partials(const sl_matrix *m, const double *dinv, const double *t_in, double *t_out, double *x,
                                   double *partials, uint32_t *n_partials, sl_order order);
sl_status sl_reduce_partials(const double *partials, uint32_t n, double *norm2);

/* same launch sequence repeated `steps` times with ping-pong buffers t_a -> t_b -> t_a ...
 * bracketed by HIP events on the launch stream; *elapsed_ms is the device time.
 * After the call the newest term is in t_a when `steps` is even, t_b when odd. */
sl_status sl_neumann_run_steps(const sl_matrix *m, const double *dinv, double *t_a, double *t_b, double *x,
                               double *norm2, sl_order order, uint64_t steps, float *elapsed_ms);

/* ---- a7..a12: NeumannSolver::solve (neumann.rs:469-555) --------------------------- */
typedef struct {
    double tolerance;        /* SolverOptions.tolerance        solver/mod.rs:47-62  (1e-6)  */
    uint64_t max_iterations; /* SolverOptions.max_iterations                         (1000) */
    uint64_t max_terms;      /* NeumannSolver.max_terms        neumann.rs:58-60      (50)   */
    double series_tolerance; /* NeumannSolver.series_tolerance                       (1e-8) */
    int32_t order;           /* sl_order    */
    int32_t start;           /* sl_start    */
    int32_t residual;        /* sl_residual */
    int32_t mem;             /* sl_mem of b / initial_guess / x_out */
    int32_t collect_stats;   /* SolverOptions.collect_stats */
    int32_t compute_error_bounds; /* SolverOptions.compute_error_bounds (neumann.rs:321-347) */
} sl_neumann_options;
void sl_neumann_options_default(sl_neumann_options *o);
/* SolverOptions::streaming(interval) (solver/mod.rs:101-116): tolerance 1e-4, max_iterations 1000, collect_stats on, no error
 * bounds; everything else as sl_neumann_options_default.  (streaming_interval itself paces the reference's PartialSolution
 * callbacks, solver/mod.rs:197-214 — a host-side concern of the caller's loop over sl_neumann_state_run_steps; it is not a field
 * of the device options.) */
void sl_neumann_options_streaming(sl_neumann_options *o);

typedef struct {
    uint64_t iterations;     /* SolverResult.iterations */
    uint64_t terms_computed;
    uint64_t matvec_count;   /* SolverStats.matvec_count */
    double residual_norm;    /* SolverResult.residual_norm */
    double last_term_norm;
    double error_bound;      /* ErrorBounds::upper_bound_only of estimate_error_bounds (neumann.rs:321-347): needs compute_error_bounds and a
                                converged series; < 0 = the reference's None (also when the norm estimate is >= 1 or NaN).  One term
                                computed gives 0.0 as in the reference; est^terms is f64::powi's square-and-multiply. */
    double total_time_ms;    /* SolverStats.total_time_ms (host wall) */
    double device_time_ms;   /* HIP-event time of the iteration loop */
    uint64_t bytes_moved;    /* algorithmic HBM bytes of the loop */
    int32_t converged;       /* SolverResult.converged */
    int32_t series_converged;
} sl_neumann_result;

/* SolverResult::meets_quality_criteria (solver/mod.rs:192-195): converged && residual_norm <= tolerance; 1 / 0 */
int sl_neumann_result_meets_quality_criteria(const sl_neumann_result *r, double tolerance);

/* x_out (n_rows) is always written, also on SL_CONVERGENCE_FAILURE (the reference drops
 * it, neumann.rs:523-530).  term_norms may be NULL, else receives one l2 norm per term
 * (capacity max_terms).  initial_guess is read only for SL_START_INITIAL_GUESS. */
sl_status sl_neumann_solve(const sl_matrix *m, const double *b, const double *initial_guess,
                           const sl_neumann_options *opts, double *x_out, double *term_norms,
                           sl_neumann_result *result);

/* ---- NeumannState as an object: SolverAlgorithm::initialize / update_rhs / extract_solution (solver/mod.rs:223-333) --------
 * sl_neumann_state_create   = NeumannState::new (neumann.rs:139-249): same checks and errors as sl_neumann_solve; the state owns
 *                             device copies of b, D^-1, the scaled rhs, the solution and the current term.
 * sl_neumann_state_run      = the loop of NeumannSolver::solve (neumann.rs:477-555) from the state's current position (iteration
 *                             count from 0, terms / matvec counters carried by the state); create + run + solution == sl_neumann_solve.
 * sl_neumann_state_update_rhs = NeumannSolver::update_rhs (neumann.rs:436-462), statement for statement: for every (index, delta)
 *                             IN LIST ORDER rhs[i] += delta * dinv[i] and solution[i] += the same; then current_term = rhs,
 *                             terms_computed = 0, series_converged = false.  An index >= n ends the call with
 *                             SL_INDEX_OUT_OF_BOUNDS: the pairs before it stay applied and the series state is not reset, as in
 *                             the reference.  (b itself is updated too, so the TRUE residual follows the new right-hand side.)
 * sl_neumann_state_reset    = SolverState::reset (neumann.rs:367-378).   indices / deltas are HOST arrays. */
typedef struct sl_neumann_state sl_neumann_state;
sl_status sl_neumann_state_create(const sl_matrix *m, const double *b, const double *initial_guess, const sl_neumann_options *opts,
                                  sl_neumann_state **out);
void sl_neumann_state_destroy(sl_neumann_state *st);
sl_status sl_neumann_state_update_rhs(sl_neumann_state *st, uint64_t count, const uint64_t *indices, const double *deltas);
sl_status sl_neumann_state_run(sl_neumann_state *st, double *term_norms, sl_neumann_result *result);
sl_status sl_neumann_state_solution(const sl_neumann_state *st, double *x_out, sl_mem where);
sl_status sl_neumann_state_reset(sl_neumann_state *st);
/* Rows [first_row, first_row + count) — LOCAL row numbers of the state (a partitioned state: of the rank's range) — of
 * NeumannState::current_term / NeumannState::solution (neumann.rs:104-107), without moving the whole vector: what a caller samples
 * to check an iteration against its own CPU arithmetic (bench.py's parity gate).  first_row + count > rows: SL_INDEX_OUT_OF_BOUNDS. */
sl_status sl_neumann_state_current_term(const sl_neumann_state *st, uint64_t first_row, uint64_t count, double *t_out, sl_mem where);
sl_status sl_neumann_state_solution_rows(const sl_neumann_state *st, uint64_t first_row, uint64_t count, double *x_out, sl_mem where);

/* ---- (a-P) / a13 / a14: synchronous thresholded residual push ------------------------
 * The data-parallel member of the reference's push family — ForwardPushSolver::push_node
 * (solver/forward_push.rs:179-216) and TS solveForwardPush (src/core/solver.ts:437-522):
 * invariant r = b - A x; every round pushes all rows with |r_i * dinv_i| >= theta
 * (frontier in ascending index order, built by wavefront ballot / prefix scan):
 *     delta_i = r_i * dinv_i ;  x_i += delta_i ;  r -= A delta  (row-wise, column order)
 * Needs SL_MATRIX_WITH_TRANSPOSE (candidate-row expansion walks columns of A). */
typedef struct {
    double theta;            /* frontier threshold                                  */
    uint64_t max_rounds;     /* ForwardPushConfig.max_pushes analogue, per round    */
    int32_t order;           /* sl_order */
    int32_t mem;             /* sl_mem of b / x / r */
    double dense_switch;     /* frontier fraction above which a round runs the dense kernel (default 1/16) */
    const double *theta_rows;/* NULL, or n per-row thresholds (memory space `mem`) used INSTEAD of theta: row i enters a frontier when
                              * |r_i dinv_i| >= theta_rows[i] — the degree-scaled admission / skip rule of the ACL push,
                              * residual[u] >= epsilon * max(out_degree(u), 1) (forward_push.rs:93-99, graph/mod.rs:171-212) */
} sl_push_options;
void sl_push_options_default(sl_push_options *o);

typedef struct {
    uint64_t rounds;
    uint64_t pushes;         /* sum of |F| */
    uint64_t rows_touched;   /* sum of |candidate rows| */
    uint64_t dense_rounds;
    double residual_norm;    /* l2(r) at exit */
    double device_time_ms;
    int32_t converged;       /* frontier became empty */
    int32_t reserved;
} sl_push_result;

/* ---- multi-GPU: one process per GPU of ONE node, row-range partition ----------------------------------
 * Precedent in the reference: simd_ops::parallel_matrix_vector_multiply (src/simd_ops.rs:201-239) hides row chunks behind one
 * call.  Here a rank holds a contiguous row range on its own GPU (the process's current device):
 *   sl_comm_create(rank, world, name)      every rank of the job calls it with the same `name` (no '/'); the rendezvous is a POSIX
 *                                          shared-memory block /dev/shm/slcomm_<name> (removed again once all ranks have joined).
 *                                          No MPI / RCCL / launcher dependence: any host language that can start N processes can
 *                                          drive N GPUs.  world <= 16.
 *   sl_matrix_create_csr(rows, n_global, nnz, ..., row_offset = first row of the rank, ...)   the rank's rows, GLOBAL column ids
 *   sl_neumann_state_create_partitioned    NeumannState::new over the partition (collective; b / initial guess = the rank's rows)
 *   sl_neumann_state_run / _run_steps / _update_rhs / _reset / _solution / _destroy            as for one GPU; collective calls —
 *                                          every rank makes the same calls in the same order; _solution returns the rank's rows;
 *                                          _update_rhs takes GLOBAL row indices (the same list on every rank).
 * Per iteration a rank (i) runs the fused step on its rows, (ii) publishes its share of ||t||^2 and waits for all ranks' shares
 * (summed in rank order: the same bits everywhere, so every rank takes the same stop decision), (iii) pulls the pieces of the new
 * term its columns reach (the measured bandwidth of its rows) out of its peers' vectors over xGMI (IPC-mapped device memory);
 * every 5th iteration the solution travels the same way for the residual (neumann.rs:489-491).  Where every rank has interior
 * rows beyond the largest reach (banded systems), the step runs BOUNDARY FIRST: the blocks at both ends of the rank's range, a
 * "halo ready" handshake and the pulls on a second stream beside the interior blocks, joined before the sum (SL_DIST_OVERLAP=0
 * keeps the plain order).  Per-row results equal the
 * one-GPU solve bit for bit; norms are sums of per-rank sums (equal to ~1e-16 relative).  A peer that never arrives turns
 * into SL_DEVICE_ERROR after SL_COMM_TIMEOUT_MS (20 s), never into a hung queue; a rank that fails locally between two collective
 * points marks the communicator, its peers' waits end at once, and no further collective runs on it.
 * TRANSPORTS (environment, the same on every rank): SL_COMM_TRANSPORT=ipc (default; ranks may share a GPU) as described above;
 * SL_COMM_TRANSPORT=rccl — the collectives SURVEY §8(e) names, inside the library (librccl resolved at run time, one rank per GPU):
 * ncclAllGather of the term when every rank needs every row (uniform columns), one group of ncclSend / ncclRecv for halo strips
 * (SL_COMM_HALO=allreduce: ONE ncclAllReduce over a compact buffer of all ranks' strips, -0.0 elsewhere — BASELINE north_star's
 * wording), ncclAllGather of the ranks' partial sums, added in rank order on the device.  Both transports are copies: same bits. */
typedef struct sl_comm sl_comm;
sl_status sl_comm_create(int rank, int world, const char *rendezvous_name, sl_comm **out);
void sl_comm_destroy(sl_comm *c);
sl_status sl_comm_rank(const sl_comm *c, int *rank, int *world);
sl_status sl_comm_barrier(sl_comm *c);                                      /* drains the calling thread's stream, then all ranks meet */
typedef struct sl_comm_info_t {
    int32_t rank, world, device;
    int32_t transport;        /* 0 = ipc, 1 = rccl */
    int32_t halo_allreduce;   /* rccl: halo strips as one all-reduce over the compact buffer */
    int32_t ranks_joined;     /* ranks that took part in the rendezvous (= world on a live communicator) */
    int32_t failed;           /* a rank failed or did not arrive: the communicator refuses further collectives */
    int32_t reserved;
} sl_comm_info_t;
sl_status sl_comm_info(const sl_comm *c, sl_comm_info_t *info);
sl_status sl_comm_allgather_u64(sl_comm *c, uint64_t mine, uint64_t *all); /* e.g. row counts -> row ranges; all[world] */
/* row ranges with equal shares of STORED ENTRIES from a host row_ptr (rank r starts at the first row whose prefix
 * reaches r * nnz / world); bounds[world + 1].  Pure host arithmetic: needs neither a device nor a communicator. */
sl_status sl_balanced_row_bounds(uint64_t n_rows, const uint32_t *row_ptr, int world, uint64_t *bounds);
sl_status sl_neumann_state_create_partitioned(sl_comm *c, const sl_matrix *local_rows, const double *b_local, const double *initial_guess_local,
                                              const sl_neumann_options *opts, sl_neumann_state **out);
/* `steps` fused steps (a8 + a9) from the state's current term without the stop rule — the measurement loop; *last_norm2 = ||t||^2
 * of the last step (over all ranks), *elapsed_ms = device time of the loop on this rank.  One GPU or partitioned. */
sl_status sl_neumann_state_run_steps(sl_neumann_state *st, uint64_t steps, double *last_norm2, float *elapsed_ms);
/* Collective check of the last exchange of the current term: every piece a rank holds of its peers' rows against the owner's own
 * copy (position-weighted checksums of the bit patterns); *pieces_bad = 0 on every rank when the transport moved what the owners
 * wrote.  What bench.py asks before it reports a multi-GPU figure.  One GPU: nothing to compare, 0. */
sl_status sl_neumann_state_verify_exchange(sl_neumann_state *st, uint64_t *pieces_bad);

/* ---- a13, graph side: PushGraph, the PageRank / PPR systems over it, and the ACL push in the spec's own visiting order ----------
 * sl_push_graph = PushGraph (src/graph/adjacency.rs:199-277): the weighted adjacency in CSR (u32 indices; the entries of a row are
 * walked in the order given, as forward_neighbors does), its transpose (graph/mod.rs:92-130), degrees = row sums and reverse degrees
 * = column sums, each added left to right (graph/mod.rs:81-89) — all built and kept on the device.
 * sl_push_graph_system: I - (1 - alpha) P^T (SL_SYSTEM_FORWARD) or I - (1 - alpha) P (SL_SYSTEM_BACKWARD), P_uv = w_uv / deg_u, a node
 * without out-weight keeps its mass (P_uu = 1, forward_push.rs:210-215; SL_SYSTEM_DANGLING_IDENTITY: it contributes nothing), assembled on the device and returned as an sl_matrix (row diagonally
 * dominant in the backward form, column dominant in the forward one): what sl_push_solve / sl_estimate_entry* / sl_query_session_* and
 * the TS computePageRank path (core/solver.ts:664-722, d = 1 - alpha) run on.
 * sl_forward_push_acl / _with_target / sl_backward_push_acl / _with_source: ForwardPushSolver::solve_single_source / solve_multi_source
 * (forward_push.rs:67-177), solve_with_target (:233-290) and BackwardPushSolver::solve_single_target / solve_multi_target /
 * solve_with_source (backward_push.rs:67-176, 238-293) with the
 * WorkQueue's order (graph/mod.rs:132-213: largest priority first; equal priorities: larger node id — the reference leaves it open):
 * push_count, nodes_visited and every bit of estimate / residual equal the CPU restatement's.  Sequential across pushes by
 * definition: for parity and small graphs; the throughput path is sl_push_solve on the system matrix with theta_rows. */
typedef struct sl_push_graph sl_push_graph;
sl_status sl_push_graph_create(uint64_t num_nodes, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, sl_mem where,
                               sl_push_graph **out);
void sl_push_graph_destroy(sl_push_graph *g);
sl_status sl_push_graph_size(const sl_push_graph *g, uint64_t *num_nodes, uint64_t *num_edges);
sl_status sl_push_graph_degrees(const sl_push_graph *g, double *out_degrees, double *in_degrees, sl_mem where);   /* either may be NULL */
#define SL_SYSTEM_FORWARD 0u              /* I - (1 - alpha) P^T */
#define SL_SYSTEM_BACKWARD 1u             /* I - (1 - alpha) P   */
#define SL_SYSTEM_DANGLING_IDENTITY 2u    /* a node without out-weight contributes nothing (its column / row stays the identity's): TS
                                             computePageRank's rule (solver.ts:690-700, mass leaks) instead of the push spec's self loop */
sl_status sl_push_graph_system(const sl_push_graph *g, double alpha, uint32_t system_flags, uint32_t matrix_flags, sl_matrix **out);
typedef struct sl_acl_options {   /* ForwardPushConfig / BackwardPushConfig, forward_push.rs:24-49 */
    double alpha;             /* 0.15 */
    double epsilon;           /* 1e-6 */
    double queue_threshold;   /* 1e-8 */
    uint64_t max_pushes;      /* 1 000 000 */
    int32_t adaptive_threshold; /* 1 */
    int32_t mem;              /* sl_mem of estimate / residual (sources are always host indices) */
} sl_acl_options;
void sl_acl_options_default(sl_acl_options *o);
typedef struct sl_acl_result {    /* ForwardPushResult scalars, forward_push.rs:10-22 */
    uint64_t push_count, nodes_visited;
    double residual_norm, device_time_ms;
    int32_t stopped_by;       /* 0 nothing ran (source / target out of range), 1 queue empty, 2 max_pushes, 3 target precision reached */
    int32_t reserved;
} sl_acl_result;
/* estimate / residual: num_nodes doubles each; push_log (may be NULL): the pushed nodes in order, up to log_cap */
sl_status sl_forward_push_acl(const sl_push_graph *g, uint64_t n_sources, const uint64_t *sources, const sl_acl_options *o, double *estimate,
                              double *residual, uint32_t *push_log, uint64_t log_cap, sl_acl_result *res);
sl_status sl_backward_push_acl(const sl_push_graph *g, uint64_t n_targets, const uint64_t *targets, const sl_acl_options *o, double *estimate,
                               double *residual, uint32_t *push_log, uint64_t log_cap, sl_acl_result *res);
sl_status sl_forward_push_acl_with_target(const sl_push_graph *g, uint64_t source, uint64_t target, double target_precision,
                                          const sl_acl_options *o, double *estimate, double *residual, uint32_t *push_log, uint64_t log_cap,
                                          sl_acl_result *res);
/* BackwardPushSolver::solve_with_source (backward_push.rs:238-293): unit mass at `target`, pushed over the reverse adjacency in the
 * WorkQueue's order; ends as soon as estimate[source] > source_precision and residual[source] < 0.1 source_precision (:262-264, checked
 * before every pop: stopped_by = 3); source or target out of range: the empty result (:243-251), not an error. */
sl_status sl_backward_push_acl_with_source(const sl_push_graph *g, uint64_t source, uint64_t target, double source_precision,
                                           const sl_acl_options *o, double *estimate, double *residual, uint32_t *push_log, uint64_t log_cap,
                                           sl_acl_result *res);
/* {Forward,Backward}PushSolver::extrapolated_solution (forward_push.rs:292-301, backward_push.rs:302-311): solution = estimate, then
 * solution[i] += alpha * residual[i] (product rounded, then added).  count doubles each, all three in memory space `where`. */
sl_status sl_acl_extrapolated_solution(uint64_t count, double alpha, const double *estimate, const double *residual, double *solution, sl_mem where);
/* BackwardPushSolver::reachability_probabilities (backward_push.rs:296-299): solve_single_target(target), then extrapolated_solution
 * of its result; solution: num_nodes doubles in o->mem; *res = the scalars of the push. */
sl_status sl_backward_push_acl_reachability(const sl_push_graph *g, uint64_t target, const sl_acl_options *o, double *solution, sl_acl_result *res);

/* ---- a14 in the reference's own visiting order: TS solveForwardPush (src/core/solver.ts:437-522) ----
 * Gauss-Southwell: every step pushes the FIRST index of largest |r_i| (r = b - A x, x0 = 0), p = r_i / a_ii, x_i += p, r_i = 0,
 * r_j -= a_ji p over column i; stops when max |r_i| < epsilon; `iterations` = pushes.  Sequential across pushes by definition
 * (the |F| = 1 member of the push family): this entry point is for order-exact parity with the reference — push sequence,
 * iteration count, solution bits — the throughput path is sl_push_solve.  Needs SL_MATRIX_WITH_TRANSPOSE.
 * Status: SL_CONVERGENCE_FAILURE after max_iterations pushes (solver.ts:509-515; x_out / r_out / result still filled),
 * SL_NUMERICAL_INSTABILITY for |a_ii| < 1e-15 (:471-473).  push_log (may be NULL): the pushed index of every step, up to log_cap.
 * Matrices that store the same (i, j) twice: the reference reads entries through MatrixOperations.getEntry / getDiagonal, which return
 * the FIRST stored match of a COO input, so later duplicates never enter its arithmetic; this entry point works on the CSR the Rust
 * side defines (duplicates are separate entries that a product sums, sparse.rs:80-132) — every stored duplicate of a column is
 * subtracted and the diagonal is the entry the midpoint search of CSRStorage::get lands on.  Parity with solver.ts is therefore
 * stated for matrices without duplicates (every fixture and golden of the reference); coalesce duplicates first where they occur. */
typedef struct {
    double epsilon;          /* SolverConfig.epsilon        src/core/types.ts:28-46 */
    uint64_t max_iterations; /* SolverConfig.maxIterations  */
    int32_t mem;             /* sl_mem of b / x_out / r_out */
    int32_t reserved;
} sl_southwell_options;
void sl_southwell_options_default(sl_southwell_options *o);
typedef struct {
    uint64_t iterations;     /* pushes = SolverResult.iterations of the reference */
    double residual_norm;    /* l2(r) after the last push; +inf before the first (solver.ts:446) */
    double device_time_ms;
    int32_t converged;
    int32_t reserved;
} sl_southwell_result;
sl_status sl_forward_push_southwell(const sl_matrix *m, const double *b, const sl_southwell_options *opts, double *x_out, double *r_out,
                                    uint32_t *push_log, uint64_t log_cap, sl_southwell_result *result);

/* x: in = x0 / out = solution; r_out: residual at exit (may be NULL).
 * frontier_log (may be NULL): per round, |F| followed by the ascending indices, up to
 * frontier_cap words; *frontier_words = words written. */
sl_status sl_push_solve(const sl_matrix *m, const double *b, const sl_push_options *opts, double *x,
                        double *r_out, uint32_t *frontier_log, uint64_t frontier_cap,
                        uint64_t *frontier_words, sl_push_result *result);

/* ---- estimateEntry (src/core/solver.ts:550-659; Rust analogue
 * ForwardPushSolver::query_single_entry, forward_push.rs:224-231) -------------------------
 * x_row = e_row^T A^-1 b by local push on A^T from e_row: y ~ A^-T e_row, estimate = y.b,
 * |x_row - estimate| <= ||r_y||_1 * ||x||_inf.  `m` must be created WITH_TRANSPOSE.
 * Work is proportional to the rows the push touches, not to n. */
typedef struct {
    double estimate;
    double residual_l1;      /* ||r_y||_1 : multiply by a bound on ||x||_inf for the error bound */
    uint64_t rounds, pushes, rows_touched;
    double device_time_ms;
    int32_t converged;
    int32_t reserved;
} sl_estimate_result;
sl_status sl_estimate_entry(const sl_matrix *m, const double *b, sl_mem where, uint64_t row,
                            double theta, uint64_t max_rounds, sl_estimate_result *result);
/* same query when the caller already holds A^T as `mt` (e.g. the row-stochastic side I - alpha P of a
 * PageRank system whose solve matrix is I - alpha P^T, src/core/solver.ts:664-722; PushGraph keeps both
 * orientations, src/graph/adjacency.rs:199-224): the push runs on mt's own rows and may use its dense kernel. */
sl_status sl_estimate_entry_transposed(const sl_matrix *mt, const double *b, sl_mem where, uint64_t row,
                                       double theta, uint64_t max_rounds, sl_estimate_result *result);
/* Query sessions — the object form of the same query (ForwardPushSolver::new + query after query,
 * forward_push.rs:52-66, 224-231): D^-1, the state vectors and the device copy of b are set up once; each
 * query then costs only the rows its push touches (seed one row, device-driven sparse rounds, sums and reset over
 * the touched rows).  matrix_is_transpose = 1: `m` already holds A^T (as for sl_estimate_entry_transposed).
 * `m` (and b, when where = SL_MEM_DEVICE) must outlive the session.  One session serves one thread at a time. */
typedef struct sl_query_session sl_query_session;
sl_status sl_query_session_create(const sl_matrix *m, int matrix_is_transpose, const double *b, sl_mem where,
                                  sl_query_session **out);
sl_status sl_query_session_estimate(sl_query_session *q, uint64_t row, double theta, uint64_t max_rounds,
                                    sl_estimate_result *result);
/* `count` independent queries of one session at once (query_single_entry for many pairs, forward_push.rs:224-231): they run on up to
 * `lanes` lanes (0 = 8) — per lane a state of its own (the session's vectors cloned on first use, ~1.2 GB at n = 10^7; matrix, D^-1
 * source and b shared), a HIP stream and a host thread — so that the launch trains of different queries overlap on the device.
 * results[i] equals what sl_query_session_estimate(rows[i]) returns, bit for bit, whatever the lane.  The first failing query's
 * status is returned (the other results are still filled). */
sl_status sl_query_session_estimate_batch(sl_query_session *q, uint64_t count, const uint64_t *rows, double theta, uint64_t max_rounds,
                                          uint32_t lanes, sl_estimate_result *results);
void sl_query_session_destroy(sl_query_session *q);

/* A^T as a matrix of its own (CompressedSparseRow::transpose, src/graph/mod.rs:92-130; PushGraph::from_matrix
 * adjacency.rs:212-224).  `m` must have been created WITH_TRANSPOSE. */
sl_status sl_matrix_transpose(const sl_matrix *m, uint32_t flags, sl_matrix **out);

/* ---- Monte-Carlo branch of estimateEntry ---------------------------------------------
 * TS estimateEntry with method 'random-walk' (src/core/solver.ts:585-601,630-648; walk rule :390-432):
 * numSamples = max(100, ceil(1/epsilon^2)) absorbing walks from `row`.  All walks draw from the reference's ONE stream
 * createSeededRandom(seed) (core/utils.ts:161-168), in the order `stream` selects (sl_walk_stream below).  num_samples = 0 derives the
 * count from epsilon.  walk_values (may be NULL) receives the per-walk estimates.  Needs the raw CSR
 * (SL_MATRIX_KEEP_CSR or SL_MATRIX_WITH_TRANSPOSE). */
/* Which draws a walk reads.  Both forms draw from the reference's ONE generator createSeededRandom(seed) (core/utils.ts:161-168).
 *   SL_WALK_STREAM_BLOCKS (default, the throughput form): the stream is cut into blocks, walk number s reads it from position
 *     s * stride, one lane per walk.  stride = 2048 draws (a walk uses at most 2000) while all walks of the call fit the generator's
 *     period, total_walks * 2048 <= 2^32; beyond that the largest power of two <= 2^32 / total_walks (so that no two walks of a call
 *     START at the same position; walks longer than the stride read into their successor's block), and more than 2^28 walks in one
 *     call are refused (SL_INVALID_INPUT: the stride would fall below 16 draws).  Walk 0 equals the reference's first walk draw for
 *     draw; the estimate agrees with the reference's statistically, not bit for bit.
 *   SL_WALK_STREAM_SERIAL (the reference as written): ONE stream walked serially — walk s starts where walk s - 1 stopped
 *     (solver.ts:585-601, 300-326), the sums of mean and variance are added in walk order.  Every per-walk value, `estimate` and
 *     `variance` (solve: every x_i, variance_i, total_variance) are bit-identical to the reference's for the same
 *     (matrix, b, row, epsilon, seed).  The serial dependency is only WHERE in the stream a walk starts; what a walk does from a
 *     given position is independent of the others.  So the device simulates a walk from EVERY position of a window of the stream
 *     (one lane per position, the state there by jump-ahead), records value and draws used, and then follows the reference's chain
 *     position -> position + draws used through the window (per-4096-position tables in LDS, one short serial pass over the chunks):
 *     the reference's walks, in its order, at a cost of (mean draws per walk) simulated walks per walk kept.
 *     SL_WALK_SERIAL_PLAIN=1 (environment) runs the reference as written on ONE lane instead — the cross-check of the tests. */
typedef enum { SL_WALK_STREAM_BLOCKS = 0, SL_WALK_STREAM_SERIAL = 1 } sl_walk_stream;
typedef struct {
    double estimate;         /* mean of the walk values            (solver.ts:630)  */
    double variance;         /* sample variance, N - 1 denominator (solver.ts:631-633) */
    uint64_t num_samples;
    double device_time_ms;
} sl_walk_result;
sl_status sl_estimate_entry_random_walk(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double epsilon,
                                        uint32_t seed, sl_walk_stream stream, uint64_t num_samples, double *walk_values,
                                        sl_walk_result *result);

/* The `random-walk` METHOD of SublinearSolver.solve — solveRandomWalk, src/core/solver.ts:278-357: every coordinate i estimated by
 * numWalks = max(100, ceil(1/epsilon^2)) absorbing walks from i (num_walks = 0 derives it; the walk rule of performRandomWalk
 * :390-432 as above), x[i] = their mean, total_variance = the sum over coordinates of the sample variances (N - 1); then
 * residual = ||A x - b||_2 and converged = residual < epsilon.  The reference THROWS when it is not (CONVERGENCE_FAILED,
 * :335-341): SL_CONVERGENCE_FAILURE, with x, variances and *res filled all the same.  Streams: walk w of coordinate i is walk number
 * i * numWalks + w of the solve (total_walks = n * numWalks); SL_WALK_STREAM_BLOCKS reads createSeededRandom(seed) from that number's
 * block, SL_WALK_STREAM_SERIAL walks the one stream serially as the reference does (sl_walk_stream above).  variances (n doubles)
 * may be NULL.  Needs the raw CSR. */
typedef struct {
    uint64_t iterations;      /* coordinates estimated = n (what the reference reports as `iterations`, solver.ts:321, 349) */
    uint64_t num_walks;       /* per coordinate */
    double residual;
    double total_variance;
    double device_time_ms;
    int32_t converged;
    int32_t reserved;
} sl_random_walk_result;
sl_status sl_solve_random_walk(const sl_matrix *m, const double *b, sl_mem where, double epsilon, uint32_t seed, sl_walk_stream stream,
                               uint64_t num_walks, double *x, double *variances, sl_random_walk_result *res);

/* ---- conjugate gradient behind the same SpMV ------------------------------------------
 * OptimizedConjugateGradientSolver::solve (src/optimized_solver.rs:182-295) == FastConjugateGradient::solve
 * (src/fast_solver.rs:126-178): x0 = 0, r = p = b, stop at r.r <= tolerance^2, break when |p.Ap| < 1e-16. */
typedef struct {
    double tolerance;        /* OptimizedSolverConfig.tolerance (1e-6), optimized_solver.rs:119-127 */
    uint64_t max_iterations; /* OptimizedSolverConfig.max_iterations (1000) */
    int32_t order;           /* sl_order of the SpMV */
    int32_t mem;             /* sl_mem of b / x_out */
} sl_cg_options;
void sl_cg_options_default(sl_cg_options *o);
typedef struct {
    uint64_t iterations;     /* OptimizedSolverResult.iterations */
    uint64_t matvec_count;   /* OptimizedSolverStats.matvec_count */
    double residual_norm;    /* sqrt(r.r) */
    double total_time_ms, device_time_ms;
    int32_t converged;
    int32_t reserved;
} sl_cg_result;
/* Deviation from the reference, on purpose: when r.r turns non-finite the reference loop (optimized_solver.rs:221-263) keeps
 * iterating on NaNs until max_iterations and returns Ok(converged = false); this entry point stops at once with
 * SL_NUMERICAL_INSTABILITY (x_out = the last iterate, result filled) — same "not converged" outcome, no wasted launches. */
sl_status sl_cg_solve(const sl_matrix *m, const double *b, const sl_cg_options *opts, double *x_out, sl_cg_result *result);

/* ---- synthetic inputs, generated in HBM (bench / tests) -------------------
 * S-DD(n, k, seed, w): rows [row_lo, row_hi) of the seeded diagonally dominant system;
 * writes device arrays: row_ptr (rows+1), col_idx / values (rows*k), b (rows). */
sl_status sl_synth_sdd_device(uint64_t n, uint32_t k, uint64_t seed, uint64_t half_bandwidth,
                              uint64_t row_lo, uint64_t row_hi, uint32_t *row_ptr, uint32_t *col_idx,
                              double *values, double *b);
/* S-PR(n, seed, alpha): power-law digraph (out-degree dmin * 2^g, g geometric so that P(d >= x) ~ x^-1.1,
 * capped at dmax; targets floor(n u^2): preferential to low ids) as the ROW-dominant matrix
 * M = I - alpha P (P row-stochastic), whose transpose is the PageRank system of core/solver.ts:664-722.
 * Two calls: with col_idx == NULL only row_ptr (n+1) is written and *nnz returned; then with arrays of *nnz. */
sl_status sl_synth_pagerank_device(uint64_t n, uint64_t seed, double alpha, uint32_t dmin, uint32_t dmax,
                                   uint32_t *row_ptr, uint32_t *col_idx, double *values, uint64_t *nnz);

#ifdef __cplusplus
}
#endif
#endif /* SUBLINEAR_HIP_H */
