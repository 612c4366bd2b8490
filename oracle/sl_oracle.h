/*
 * sl_oracle.h — CPU restatement of the reference's push / Neumann hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the C-ABI library
 * under sublinear_time_solver_amd/csrc or the host classes above it) may
 * include, link or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * The reference (ruvnet/sublinear-time-solver @ 2025-09-19) is a Rust crate
 * that cannot be built in this image (no cargo/rustc, deps not vendored), so
 * this is a restatement in plain C, every function citing the reference
 * file:line it follows.  It is pinned (tests/test_oracle_golden.py) against
 *   - the reference's own unit-test known answers,
 *   - outputs of the reference's runnable Python and JS Jacobi solvers,
 *     captured in this container by tests/golden/make_golden.py (G1/G2),
 *   - the push property tests of tests/rust/push_tests.rs (G5) and the
 *     TS forward-push case of tests/mcp/mcp-tool-tests.js (G6), LCG (G7).
 *
 * All arithmetic is IEEE binary64, compiled with -ffp-contract=off so that a
 * product is rounded before it is added, exactly as rustc emits for the
 * reference's scalar loops.
 */
#ifndef SL_ORACLE_H
#define SL_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: 1:1 with SolverError variants, src/error.rs:16-140 */
enum {
    ORC_OK = 0,
    ORC_NOT_DIAGONALLY_DOMINANT = 1,
    ORC_NUMERICAL_INSTABILITY = 2,
    ORC_CONVERGENCE_FAILURE = 3,
    ORC_INVALID_INPUT = 4,
    ORC_DIMENSION_MISMATCH = 5,
    ORC_UNSUPPORTED_FORMAT = 6,
    ORC_ALLOCATION = 7,
    ORC_INDEX_OUT_OF_BOUNDS = 8,
    ORC_INVALID_SPARSE_MATRIX = 9,
    ORC_ALGORITHM_ERROR = 10
};

/* summation order of the row dot product */
enum { ORC_ORDER_CSR_SEQUENTIAL = 0, ORC_ORDER_SIMD4 = 1 };
/* x0 of the Neumann state */
enum { ORC_START_ZERO = 0, ORC_START_REFERENCE_DEFAULT = 1, ORC_START_INITIAL_GUESS = 2 };
/* which rhs update_residual subtracts */
enum { ORC_RESIDUAL_TRUE = 0, ORC_RESIDUAL_REFERENCE_SCALED = 1 };

typedef struct {
    double tolerance;          /* SolverOptions.tolerance, solver/mod.rs:47-62 (1e-6) */
    uint64_t max_iterations;   /* SolverOptions.max_iterations (1000) */
    uint64_t max_terms;        /* NeumannSolver.max_terms, neumann.rs:58-60 (50) */
    double series_tolerance;   /* NeumannSolver.series_tolerance (1e-8) */
    int32_t order;             /* ORC_ORDER_* */
    int32_t start;             /* ORC_START_* */
    int32_t residual;          /* ORC_RESIDUAL_* */
    int32_t threads;           /* >1: row-chunk threaded SpMV (simd_ops.rs:201-239) */
} orc_neumann_opts;

typedef struct {
    uint64_t iterations;
    uint64_t terms_computed;
    uint64_t matvec_count;
    double residual_norm;
    int32_t converged;
    int32_t series_converged;
} orc_neumann_result;

/* ---- a1: triplets -> CSR (matrix/mod.rs:160-199, sparse.rs:530-548, 80-132) ---- */
int orc_csr_from_triplets(uint64_t ntrip, const uint64_t *tr, const uint64_t *tc, const double *tv,
                          uint64_t rows, uint64_t cols,
                          uint32_t *row_ptr, uint32_t *col_idx, double *values, uint64_t *nnz_out);
/* CSRStorage::get, sparse.rs:142-155.  returns 1 and *out if present, else 0 */
int orc_csr_get(const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                uint64_t rows, uint64_t r, uint64_t c, double *out);

/* ---- a2/a3/a4: SpMV ---- */
void orc_spmv_csr_sequential(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                             const double *values, const double *x, double *y);
void orc_spmv_simd4(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                    const double *values, const double *x, double *y);
void orc_spmv_parallel(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                       const double *values, const double *x, double *y, int threads);

/* CSRStorage::multiply_vector_add, sparse.rs:192-203 (y += A x, the running sum seeded with y_i) */
void orc_spmv_add_csr_sequential(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                                 const double *values, const double *x, double *y);

/* ---- a5: vector primitives ---- */
double orc_dot_simd4(uint64_t n, const double *x, const double *y);
double orc_dot_sequential(uint64_t n, const double *x, const double *y);
void orc_axpy(uint64_t n, double alpha, const double *x, double *y);
double orc_l2_norm(uint64_t n, const double *v);
double orc_l1_norm(uint64_t n, const double *v);
double orc_linf_norm(uint64_t n, const double *v);

/* ---- a6: diagonal dominance (matrix/mod.rs:467-485) ---- */
int orc_is_diagonally_dominant(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                               const double *values);

/* Matrix::diagonal_dominance_factor (matrix/mod.rs:487-514): 1 + *factor for Some, 0 for None */
int orc_diagonal_dominance_factor(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                                  const double *values, double *factor);
/* Matrix::spectral_radius_estimate (matrix/mod.rs:83-100) */
double orc_spectral_radius_estimate(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values);
/* the element / iterator / norm side of trait Matrix: get (matrix/mod.rs:383-395), row_iter (sparse.rs:158-176), col_iter
 * (CSRColIter sparse.rs:273-298), frobenius_norm (matrix/mod.rs:74-82), sparsity_info (matrix/mod.rs:523-545, types.rs:344-369) */
int orc_matrix_get(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                   uint64_t r, uint64_t c, double *out);
uint64_t orc_csr_row(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, uint64_t r,
                     uint64_t cap, uint32_t *cols_out, double *vals_out);
uint64_t orc_csr_col(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, uint64_t c,
                     uint64_t cap, uint32_t *rows_out, double *vals_out);
double orc_frobenius_norm(uint64_t rows, const uint32_t *row_ptr, const double *values);
void orc_sparsity_info(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx, uint64_t out_u[3], double out_f[2]);
/* f64::powi as rustc emits it (compiler-rt __powidf2: square and multiply) */
double orc_powi(double a, int b);
/* NeumannState::estimate_error_bounds (neumann.rs:321-347): 1 + *bound for Some, 0 for None */
int orc_neumann_error_bound(uint64_t n, const double *current_term, const double *rhs, uint64_t terms_computed,
                            int series_converged, double *bound);

/* ---- a7..a11: NeumannState::new + NeumannSolver::solve ---- */
int orc_neumann_init(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx,
                     const double *values, uint64_t b_len, const double *b,
                     double *dinv, double *rhs);
int orc_neumann_solve(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx,
                      const double *values, uint64_t b_len, const double *b,
                      const double *initial_guess, const orc_neumann_opts *opts,
                      double *x_out, double *term_out, double *term_norms /* [max_terms] or NULL */,
                      orc_neumann_result *res);

/* bench.py cpu_baseline: `steps` passes of a8 + a9 on rows [0, rows) with a gathered vector of any length */
double orc_neumann_steps(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                         const double *dinv, double *t, double *x, double *tmp, uint64_t steps, int order, int threads);
/* the same with its time split (seconds in the SpMV / in the vector passes + norm); parallel_passes = 1: the passes row-chunk threaded
 * too — beyond the reference, a labelled second figure of the CPU baseline */
double orc_neumann_steps_split(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *dinv, double *t,
                               double *x, double *tmp, uint64_t steps, int order, int threads, int parallel_passes, double *sec_spmv, double *sec_vec);

/* ---- (a-P) synchronous thresholded push, SURVEY.md §8 (a-P) ---- */
typedef struct {
    double theta;            /* frontier threshold on |r_i * dinv_i| */
    uint64_t max_rounds;
    int32_t order;
    int32_t pad;
    const double *theta_rows; /* NULL, or one threshold per row used instead of theta (degree-scaled rule, forward_push.rs:93-99) */
} orc_push_opts;
typedef struct {
    uint64_t rounds;
    uint64_t pushes;          /* sum of |F| over rounds */
    uint64_t rows_touched;    /* sum of |candidate rows| over rounds */
    double residual_norm;     /* l2(r) at exit */
    int32_t converged;        /* frontier became empty */
    int32_t pad;
} orc_push_result;
/* frontier_log: if non-NULL receives, per round, |F| then the ascending indices
 * (capacity frontier_cap uint32 words; logging stops silently when full);
 * *frontier_words = words written. */
int orc_push_sync_solve(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                        const double *values, const double *b, const orc_push_opts *opts,
                        double *x /* in: x0, out */, double *r /* out */,
                        uint32_t *frontier_log, uint64_t frontier_cap, uint64_t *frontier_words,
                        orc_push_result *res);

/* ---- a13: ACL forward / backward push (forward_push.rs:67-216, backward_push.rs:67-220) ---- */
typedef struct {
    double alpha;            /* 0.15 */
    double epsilon;          /* 1e-6 */
    uint64_t max_pushes;     /* 1_000_000 */
    double queue_threshold;  /* 1e-8 */
    int32_t adaptive_threshold;
    int32_t pad;
} orc_acl_opts;
typedef struct {
    uint64_t push_count;
    uint64_t nodes_visited;
    double residual_norm;
} orc_acl_result;
/* graph = weighted adjacency in CSR with u32 indices (the spec uses usize; widths
 * do not change arithmetic).  sources: unit mass split uniformly. */
int orc_acl_forward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                         const double *weights, uint64_t nsrc, const uint64_t *sources,
                         const orc_acl_opts *opts, double *estimate, double *residual,
                         orc_acl_result *res);
int orc_acl_backward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                          const double *weights, uint64_t ntgt, const uint64_t *targets,
                          const orc_acl_opts *opts, double *estimate, double *residual,
                          orc_acl_result *res);
/* the same with the sequence of pushed nodes logged (backward = 1: the backward push) */
int orc_acl_forward_push_logged(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t nsrc,
                                const uint64_t *sources, const orc_acl_opts *opts, double *estimate, double *residual, orc_acl_result *res,
                                int backward, uint32_t *push_log, uint64_t log_cap);
/* ForwardPushSolver::solve_with_target, forward_push.rs:233-290 */
int orc_acl_forward_push_with_target(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t source,
                                     uint64_t target, double target_precision, const orc_acl_opts *opts, double *estimate, double *residual,
                                     orc_acl_result *res, uint32_t *push_log, uint64_t log_cap);
/* BackwardPushSolver::solve_with_source, backward_push.rs:238-293 */
int orc_acl_backward_push_with_source(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t source,
                                      uint64_t target, double source_precision, const orc_acl_opts *opts, double *estimate, double *residual,
                                      orc_acl_result *res, uint32_t *push_log, uint64_t log_cap);
/* {Forward,Backward}PushSolver::extrapolated_solution, forward_push.rs:292-301 / backward_push.rs:302-311 */
void orc_acl_extrapolated_solution(uint64_t n, double alpha, const double *estimate, const double *residual, double *solution);
/* CompressedSparseRow::transpose, graph/mod.rs:92-130 */
void orc_csr_transpose(uint64_t nrows, uint64_t ncols, const uint32_t *row_ptr, const uint32_t *col_idx,
                       const double *values, uint32_t *t_row_ptr, uint32_t *t_col_idx, double *t_values);

/* ---- a14: TS solveForwardPush, Gauss-Southwell (src/core/solver.ts:437-522) ---- */
typedef struct {
    uint64_t iterations;
    double residual;         /* norm2(r) after the last push */
    int32_t converged;
    int32_t pad;
} orc_ts_push_result;
int orc_ts_forward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                        const double *values, const double *b, double epsilon, uint64_t max_iterations,
                        double *x, double *r, orc_ts_push_result *res);

/* BackwardPushSolver::combine_with_forward (backward_push.rs:314-333) */
double orc_acl_combine_with_forward(uint64_t n_backward, uint64_t n_forward, double alpha, const double *b_est, const double *b_res,
                                    const double *f_est, const double *f_res);

/* ---- a15: TS LCG + random-walk estimateEntry (core/utils.ts:161-168, solver.ts:359-432,585-648) ---- */
void orc_ts_lcg(uint32_t seed, uint64_t count, double *out);
uint64_t orc_walk_stride(uint64_t total_walks);            /* draws between the starting points of consecutive walks of a call */
int orc_ts_random_walk_serial(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values_a, const double *b,
                              uint64_t start_row, uint64_t num_samples, uint32_t seed, double *values, double *mean, double *variance);
/* the multiply loops of the storages convert_to_format() can produce (sparse.rs:409-430, 584-597, 763-773), entries in THEIR order */
void orc_spmv_coo(uint64_t rows, uint64_t nnz, const uint32_t *row_idx, const uint32_t *col_idx, const double *values, const double *x, double *y);
void orc_spmv_csc(uint64_t rows, uint64_t cols, const uint32_t *col_ptr, const uint32_t *row_idx, const double *values, const double *x, double *y);
void orc_coo_to_csc(uint64_t cols, uint64_t nnz, const uint32_t *row_in, const uint32_t *col_in, const double *val_in, uint32_t *col_ptr, uint32_t *row_out, double *val_out);
void orc_spmv_graph(uint64_t nodes, uint64_t nnz, const uint32_t *row_idx, const uint32_t *col_idx, const double *values, uint64_t x_len, const double *x, double *y);
void orc_csr_scale(uint64_t nnz, double *values, double factor);                                   /* sparse.rs:229-233 */
uint64_t orc_csr_add_diagonal(uint64_t rows, uint64_t row_offset, const uint32_t *row_ptr, const uint32_t *col_idx, double *values,
                              double alpha);                                                         /* sparse.rs:236-248 */
uint32_t orc_ts_lcg_jump(uint32_t state, uint64_t k);      /* the state k draws further on (the per-walk blocks of the one stream) */
int orc_ts_random_walk_estimate(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                                const double *values, const double *b, uint64_t start_row,
                                double epsilon, uint32_t seed, double *mean, double *variance,
                                uint64_t *num_samples);

/* solveRandomWalk (solver.ts:278-357): per_walk_streams = 0 the reference's ONE shared stream, 1 = a stream per walk (the device's form) */
int orc_ts_random_walk_solve(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *b,
                             double epsilon, uint32_t seed, uint64_t num_walks, int per_walk_streams,
                             double *x, double *variances, double *residual, double *total_variance);

#ifdef __cplusplus
}
#endif
#endif
