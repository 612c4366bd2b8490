// sublinear_solver.hpp — header-only C++ host mirror of the reference crate's solver interface over
// the C ABI (sublinear_hip.h).  The reference is compiled code (Rust) and no Rust toolchain exists in
// the build image, so the host side above the ABI is C++ with the crate's names, argument meaning and
// error behaviour (Result<T, SolverError> becomes a thrown SolverError carrying the variant).
//
//   reference                                         here
//   SparseMatrix::from_triplets  matrix/mod.rs:160    sublinear::SparseMatrix::from_triplets
//   Matrix::{rows,cols,nnz,is_diagonally_dominant,
//            multiply_vector,multiply_vector_add,
//            diagonal_dominance_factor,
//            spectral_radius_estimate} matrix/mod.rs:25-104 same names
//   SolverOptions (+presets)     solver/mod.rs:20-116 sublinear::SolverOptions
//   SolverResult / SolverStats   solver/mod.rs:118-195, types.rs:88-109
//   NeumannSolver::{new,default,high_precision,fast,solve}  solver/neumann.rs:24-92,469-555
//   ForwardPushSolver::query_single_entry analogue    solver/forward_push.rs:224-231 -> estimate_entry
//   SolverError variants         error.rs:16-140      sublinear::SolverError::kind
#pragma once
#include <cstdint>
#include <cstring>
#include <limits>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <utility>
#include <algorithm>
#include <vector>

#include "sublinear_hip.h"

namespace sublinear {

using Precision = double;      // types.rs:19
using IndexType = uint32_t;    // types.rs:22

class SolverError : public std::runtime_error {
public:
    SolverError(sl_status k, const std::string &msg) : std::runtime_error(std::string(sl_status_string(k)) + ": " + msg), kind(k) {}
    sl_status kind;
    // error.rs:147-160: which failures a caller may retry with different settings
    bool is_recoverable() const
    {
        return kind == SL_CONVERGENCE_FAILURE || kind == SL_NUMERICAL_INSTABILITY || kind == SL_NOT_DIAGONALLY_DOMINANT;
    }
};

inline void check(sl_status s)
{
    if (s != SL_OK) throw SolverError(s, sl_last_error_message());
}

struct SolverOptions {                       // solver/mod.rs:20-62
    Precision tolerance = 1e-6;
    size_t max_iterations = 1000;
    std::optional<std::vector<Precision>> initial_guess;
    bool collect_stats = false;
    bool compute_error_bounds = false;
    static SolverOptions high_precision() { SolverOptions o; o.tolerance = 1e-12; o.max_iterations = 5000; o.collect_stats = true; o.compute_error_bounds = true; return o; }
    static SolverOptions fast() { SolverOptions o; o.tolerance = 1e-3; o.max_iterations = 100; return o; }
    // SolverOptions::streaming (solver/mod.rs:101-116); the interval paces the caller's loop over NeumannState::run_steps
    size_t streaming_interval = 0;
    static SolverOptions streaming(size_t interval) { SolverOptions o; o.tolerance = 1e-4; o.max_iterations = 1000; o.collect_stats = true; o.streaming_interval = interval; return o; }
};

struct SolverStats {                         // types.rs:88-109 (fields the path fills) + device additions
    double total_time_ms = 0, device_time_ms = 0;
    size_t matvec_count = 0;
    uint64_t bytes_moved = 0;
};

struct SolverResult {                        // solver/mod.rs:118-195
    std::vector<Precision> solution;
    Precision residual_norm = 0;
    size_t iterations = 0;
    bool converged = false;
    std::optional<Precision> error_bound;
    std::optional<SolverStats> stats;
    // SolverResult::meets_quality_criteria (solver/mod.rs:192-195)
    bool meets_quality_criteria(Precision tolerance) const { return converged && residual_norm <= tolerance; }
};

class SparseMatrix {
public:
    using Triplet = std::tuple<size_t, size_t, Precision>;
    SparseMatrix(const SparseMatrix &) = delete;
    SparseMatrix &operator=(const SparseMatrix &) = delete;
    SparseMatrix(SparseMatrix &&o) noexcept : h_(o.h_), rows_(o.rows_), cols_(o.cols_), format_(o.format_) { o.h_ = nullptr; }
    ~SparseMatrix() { if (h_) sl_matrix_destroy(h_); }

    static SparseMatrix from_triplets(const std::vector<Triplet> &t, size_t rows, size_t cols, bool with_transpose = false)
    {
        std::vector<uint64_t> r(t.size()), c(t.size());
        std::vector<double> v(t.size());
        for (size_t k = 0; k < t.size(); ++k) { r[k] = std::get<0>(t[k]); c[k] = std::get<1>(t[k]); v[k] = std::get<2>(t[k]); }
        sl_matrix *h = nullptr;
        check(sl_matrix_create_from_triplets(t.size(), r.data(), c.data(), v.data(), rows, cols,
                                             with_transpose ? SL_MATRIX_WITH_TRANSPOSE : SL_MATRIX_DEFAULT, &h));
        return SparseMatrix(h, rows, cols);
    }
    // SparseMatrix::from_dense (matrix/mod.rs:204-223: row-major data, zeros not stored), identity (:226-229), diagonal (:232-239)
    static SparseMatrix from_dense(const std::vector<Precision> &data, size_t rows, size_t cols)
    {
        if (data.size() != rows * cols) throw SolverError(SL_DIMENSION_MISMATCH, "dense_to_sparse_conversion");
        std::vector<Triplet> t;
        for (size_t i = 0; i < data.size(); ++i) if (data[i] != 0.0) t.emplace_back(i / cols, i % cols, data[i]);
        return from_triplets(t, rows, cols);
    }
    static SparseMatrix identity(size_t size)
    {
        std::vector<Triplet> t;
        for (size_t i = 0; i < size; ++i) t.emplace_back(i, i, 1.0);
        return from_triplets(t, size, size);
    }
    static SparseMatrix diagonal(const std::vector<Precision> &diag)
    {
        std::vector<Triplet> t;
        for (size_t i = 0; i < diag.size(); ++i) if (diag[i] != 0.0) t.emplace_back(i, i, diag[i]);
        return from_triplets(t, diag.size(), diag.size());
    }
    // adopt CSRStorage arrays (sparse.rs:16-23)
    static SparseMatrix from_csr(const std::vector<IndexType> &row_ptr, const std::vector<IndexType> &col_indices,
                                 const std::vector<Precision> &values, size_t rows, size_t cols, bool with_transpose = false)
    {
        sl_matrix *h = nullptr;
        check(sl_matrix_create_csr(rows, cols, values.size(), row_ptr.data(), col_indices.data(), values.data(), SL_MEM_HOST, 0,
                                   with_transpose ? SL_MATRIX_WITH_TRANSPOSE : SL_MATRIX_DEFAULT, &h));
        return SparseMatrix(h, rows, cols);
    }
    size_t rows() const { return rows_; }
    size_t cols() const { return cols_; }
    bool is_square() const { return rows_ == cols_; }
    size_t nnz() const { sl_matrix_info i; check(sl_matrix_get_info(h_, &i)); return i.nnz; }
    bool is_diagonally_dominant() const { int f = 0; check(sl_matrix_is_diagonally_dominant(h_, &f)); return f != 0; }
    // Matrix::multiply_vector (matrix/mod.rs:415-439): DimensionMismatch on wrong lengths
    void multiply_vector(const std::vector<Precision> &x, std::vector<Precision> &result, sl_order order = SL_ORDER_CSR_SEQUENTIAL) const
    {
        if (x.size() != cols_) throw SolverError(SL_DIMENSION_MISMATCH, "matrix_vector_multiply: x");
        if (result.size() != rows_) throw SolverError(SL_DIMENSION_MISMATCH, "matrix_vector_multiply: result");
        check(sl_spmv(h_, x.data(), result.data(), order, SL_MEM_HOST));
    }
    // Matrix::multiply_vector_add (matrix/mod.rs:441-465): result += A x, the running sum of row i seeded with result[i] (sparse.rs:192-203)
    void multiply_vector_add(const std::vector<Precision> &x, std::vector<Precision> &result) const
    {
        if (x.size() != cols_) throw SolverError(SL_DIMENSION_MISMATCH, "matrix_vector_multiply_add: x");
        if (result.size() != rows_) throw SolverError(SL_DIMENSION_MISMATCH, "matrix_vector_multiply_add: result");
        check(sl_spmv_add(h_, x.data(), result.data(), SL_ORDER_CSR_SEQUENTIAL, SL_MEM_HOST));
    }
    // Matrix::diagonal_dominance_factor (matrix/mod.rs:487-514) / spectral_radius_estimate (:83-100)
    std::optional<Precision> diagonal_dominance_factor() const
    {
        int has = 0; double f = 0.0;
        check(sl_matrix_diagonal_dominance_factor(h_, &has, &f));
        return has ? std::optional<Precision>(f) : std::nullopt;
    }
    Precision spectral_radius_estimate() const { double r = 0.0; check(sl_matrix_spectral_radius_estimate(h_, &r)); return r; }
    // Matrix::get (matrix/mod.rs:383-395 over CSRStorage::get, sparse.rs:142-155): nullopt out of bounds or where nothing is stored
    std::optional<Precision> get(size_t row, size_t col) const
    {
        int found = 0; double v = 0.0;
        check(sl_matrix_get(h_, row, col, &found, &v));
        return found ? std::optional<Precision>(v) : std::nullopt;
    }
    // Matrix::row_iter (sparse.rs:158-176) / col_iter (CSRColIter sparse.rs:273-298: one pair per row): (index, value) pairs
    std::vector<std::pair<IndexType, Precision>> row_iter(size_t row) const { return pairs(row, false); }
    std::vector<std::pair<IndexType, Precision>> col_iter(size_t col) const { return pairs(col, true); }
    // Matrix::frobenius_norm (matrix/mod.rs:74-82; tree-reduced on the device: the reference's value to rounding)
    Precision frobenius_norm() const { double r = 0.0; check(sl_matrix_frobenius_norm(h_, &r)); return r; }
    // Matrix::sparsity_info (matrix/mod.rs:523-545; the fields of SparsityInfo, types.rs:114-129)
    sl_sparsity_info sparsity_info() const { sl_sparsity_info i; check(sl_matrix_sparsity_info(h_, &i)); return i; }
    // SparseMatrix::scale / add_diagonal (matrix/mod.rs:346-372 over sparse.rs:229-248): in place on the device, every layout copy;
    // non-const like the reference's `&mut self` (no solve may be running on the matrix; re-create states / sessions afterwards).
    // add_diagonal skips rows without a stored diagonal entry; SL_INVALID_INPUT for a non-square matrix
    // SparseMatrix::to_triplets (matrix/mod.rs:298-305): (row, col, value) in stored order; the raw CSR where the matrix keeps one, else
    // written back from the row slices
    std::vector<Triplet> to_triplets() const
    {
        sl_matrix_info i;
        check(sl_matrix_get_info(h_, &i));
        std::vector<uint32_t> rp(i.n_rows + 1), ci(i.nnz ? i.nnz : 1);
        std::vector<double> va(i.nnz ? i.nnz : 1);
        check(sl_matrix_download_csr(h_, rp.data(), ci.data(), va.data()));
        std::vector<Triplet> out;
        out.reserve(i.nnz);
        for (uint64_t r = 0; r < i.n_rows; ++r) for (uint32_t k = rp[r]; k < rp[r + 1]; ++k) out.emplace_back(r, ci[k], va[k]);
        return out;
    }
    void scale(Precision factor) { check(sl_matrix_scale(h_, factor)); }
    void add_diagonal(Precision alpha) { check(sl_matrix_add_diagonal(h_, alpha)); }
    // Matrix::format_name (matrix/mod.rs:557-564) / SparseMatrix::convert_to_format (:244-296).  No device work: every storage the
    // reference converts a SparseMatrix into is filled from to_triplets() of the one before, so its multiply loop adds a row's products
    // in the CSR loop's own sequence — the same bits (tests/test_storage_formats_host.py); only the name changes.  GraphAdjacency of a
    // non-square matrix is refused (GraphStorage::from_triplets drops entries whose column is >= rows, sparse.rs:655-690)
    const char *format_name() const { return format_; }
    void convert_to_format(const std::string &new_format)
    {
        for (const char *f : {"CSR", "CSC", "COO", "GraphAdjacency"})
            if (new_format == f) {
                if (new_format == "GraphAdjacency" && rows_ != cols_) throw SolverError(SL_UNSUPPORTED_FORMAT, "GraphAdjacency of a non-square matrix");
                format_ = f;
                return;
            }
        throw SolverError(SL_UNSUPPORTED_FORMAT, "Unsupported matrix format: " + new_format);
    }
    const sl_matrix *handle() const { return h_; }

private:
    std::vector<std::pair<IndexType, Precision>> pairs(size_t index, bool by_column) const
    {
        uint64_t n = 0;
        check(by_column ? sl_matrix_col(h_, index, 0, nullptr, nullptr, &n) : sl_matrix_row(h_, index, 0, nullptr, nullptr, &n));
        std::vector<uint32_t> idx(n ? n : 1);
        std::vector<double> val(n ? n : 1);
        check(by_column ? sl_matrix_col(h_, index, n, idx.data(), val.data(), &n) : sl_matrix_row(h_, index, n, idx.data(), val.data(), &n));
        std::vector<std::pair<IndexType, Precision>> out;
        for (uint64_t k = 0; k < n; ++k) out.emplace_back(idx[k], val[k]);
        return out;
    }
    SparseMatrix(sl_matrix *h, size_t r, size_t c) : h_(h), rows_(r), cols_(c) {}
    sl_matrix *h_;
    size_t rows_, cols_;
    const char *format_ = "CSR";
};

enum class StepResult { Continue, Converged, Failed };        // solver/mod.rs:197-221

// solver::utils (solver/mod.rs:363-461): vectors reduced on the device (sums: fixed tree, the reference's sequential sum to rounding)
namespace utils {
enum class NormType { L1 = SL_NORM_L1, L2 = SL_NORM_L2, LInfinity = SL_NORM_LINF, Weighted = SL_NORM_WEIGHTED };                  // types.rs:46-55
enum class ConvergenceMode { ResidualNorm = SL_CONV_RESIDUAL_NORM, RelativeResidual = SL_CONV_RELATIVE_RESIDUAL, SolutionChange = SL_CONV_SOLUTION_CHANGE,
                             RelativeSolutionChange = SL_CONV_RELATIVE_SOLUTION_CHANGE, Combined = SL_CONV_COMBINED };             // types.rs:30-41
inline Precision l2_norm(const std::vector<Precision> &v) { double r = 0; check(sl_l2_norm(v.size(), v.data(), &r, SL_MEM_HOST)); return r; }
inline Precision l1_norm(const std::vector<Precision> &v) { double r = 0; check(sl_l1_norm(v.size(), v.data(), &r, SL_MEM_HOST)); return r; }
inline Precision linf_norm(const std::vector<Precision> &v) { double r = 0; check(sl_linf_norm(v.size(), v.data(), &r, SL_MEM_HOST)); return r; }
inline Precision compute_norm(const std::vector<Precision> &v, NormType t)
{
    double r = 0;
    check(sl_compute_norm(v.size(), v.data(), static_cast<sl_norm_type>(t), &r, SL_MEM_HOST));
    return r;
}
inline void compute_residual(const SparseMatrix &m, const std::vector<Precision> &x, const std::vector<Precision> &b, std::vector<Precision> &residual)
{
    residual.resize(m.rows());
    check(sl_compute_residual(m.handle(), x.data(), b.data(), residual.data(), SL_ORDER_CSR_SEQUENTIAL, SL_MEM_HOST));
}
inline bool check_convergence(Precision residual_norm, Precision tolerance, ConvergenceMode mode, Precision b_norm,
                              const std::vector<Precision> *prev_solution, const std::vector<Precision> &current_solution)
{
    int out = 0;
    check(sl_check_convergence(residual_norm, tolerance, static_cast<sl_convergence_mode>(mode), b_norm, current_solution.size(),
                               prev_solution ? prev_solution->data() : nullptr, current_solution.data(), SL_MEM_HOST, &out));
    return out != 0;
}
} // namespace utils

// sl_comm: one process per GPU of one node (include/sublinear_hip.h, multi-GPU); every rank passes the same name
class Communicator {
public:
    Communicator(int rank, int world, const std::string &name) { check(sl_comm_create(rank, world, name.c_str(), &c_)); }
    Communicator(const Communicator &) = delete;
    Communicator &operator=(const Communicator &) = delete;
    ~Communicator() { if (c_) sl_comm_destroy(c_); }
    void barrier() const { check(sl_comm_barrier(c_)); }
    sl_comm *handle() const { return c_; }

private:
    sl_comm *c_ = nullptr;
};

// NeumannState (neumann.rs:95-137) + trait SolverState (solver/mod.rs:336-351) on the device
class NeumannState {
public:
    NeumannState(NeumannState &&o) noexcept : h_(o.h_), n_(o.n_), last_(o.last_), ran_(o.ran_) { o.h_ = nullptr; }
    NeumannState(const NeumannState &) = delete;
    NeumannState &operator=(const NeumannState &) = delete;
    ~NeumannState() { if (h_) sl_neumann_state_destroy(h_); }
    Precision residual_norm() const { return ran_ ? last_.residual_norm : std::numeric_limits<Precision>::infinity(); }
    size_t matvec_count() const { return (size_t)last_.matvec_count; }
    size_t iterations() const { return (size_t)last_.iterations; }
    void reset() { check(sl_neumann_state_reset(h_)); ran_ = false; }                     // neumann.rs:367-378

private:
    friend class NeumannSolver;
    NeumannState(sl_neumann_state *h, size_t n) : h_(h), n_(n) { std::memset(&last_, 0, sizeof(last_)); }
    sl_neumann_state *h_;
    size_t n_;
    sl_neumann_result last_;
    bool ran_ = false;
};

class NeumannSolver {                        // neumann.rs:24-92
public:
    NeumannSolver(size_t max_terms, Precision series_tolerance) : max_terms_(max_terms), series_tolerance_(series_tolerance) {}
    NeumannSolver() : NeumannSolver(50, 1e-8) {}                                         // Default, :58-60
    static NeumannSolver high_precision() { return NeumannSolver(100, 1e-12); }          // :63-65
    static NeumannSolver fast() { return NeumannSolver(20, 1e-6); }                      // :68-70
    NeumannSolver &with_order(sl_order o) { order_ = o; return *this; }
    // reference-compat quirks; the defaults are the exact series
    NeumannSolver &with_reference_quirks(bool on) { start_ = on ? SL_START_REFERENCE_DEFAULT : SL_START_ZERO; residual_ = on ? SL_RESIDUAL_REFERENCE_SCALED : SL_RESIDUAL_TRUE; return *this; }
    const char *algorithm_name() const { return "neumann"; }

    SolverResult solve(const SparseMatrix &matrix, const std::vector<Precision> &b, const SolverOptions &options = SolverOptions()) const
    {
        if (matrix.is_square() && b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "neumann_initialization");   // :154-160
        sl_neumann_options o;
        sl_neumann_options_default(&o);
        o.tolerance = options.tolerance; o.max_iterations = options.max_iterations;
        o.max_terms = max_terms_; o.series_tolerance = series_tolerance_;
        o.order = order_; o.start = start_; o.residual = residual_; o.mem = SL_MEM_HOST;
        o.collect_stats = options.collect_stats; o.compute_error_bounds = options.compute_error_bounds;
        const double *guess = nullptr;
        if (options.initial_guess) {
            if (options.initial_guess->size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "initial_guess");        // :198-204
            guess = options.initial_guess->data();
            o.start = SL_START_INITIAL_GUESS;
        }
        SolverResult out;
        out.solution.resize(matrix.rows());
        sl_neumann_result r;
        check(sl_neumann_solve(matrix.handle(), b.data(), guess, &o, out.solution.data(), nullptr, &r));
        out.residual_norm = r.residual_norm; out.iterations = r.iterations; out.converged = r.converged != 0;
        if (r.error_bound >= 0) out.error_bound = r.error_bound;
        if (options.collect_stats) { SolverStats s; s.total_time_ms = r.total_time_ms; s.device_time_ms = r.device_time_ms; s.matvec_count = r.matvec_count; s.bytes_moved = r.bytes_moved; out.stats = s; }
        return out;
    }

    // ---- trait SolverAlgorithm (solver/mod.rs:223-333): initialize / step / is_converged / extract_solution / update_rhs --------
    // The state lives on the device (sl_neumann_state_*); `comm` != nullptr: NeumannState::new over a row partition (`matrix` = this
    // rank's rows with global column ids, `b` = its part of the right-hand side; every call on the state is then collective).
    NeumannState initialize(const SparseMatrix &matrix, const std::vector<Precision> &b, const SolverOptions &options = SolverOptions(),
                            const Communicator *comm = nullptr) const;
    // One call runs the whole loop of neumann.rs:477-555 on the device (the reference's own `step` cannot iterate: :393-419)
    StepResult step(NeumannState &state) const;
    bool is_converged(const NeumannState &state) const;
    std::vector<Precision> extract_solution(const NeumannState &state) const;
    void update_rhs(NeumannState &state, const std::vector<std::pair<size_t, Precision>> &delta_b) const;     // neumann.rs:436-462

private:
    sl_neumann_options abi_options(const SparseMatrix &matrix, const SolverOptions &options, const double **guess) const
    {
        sl_neumann_options o;
        sl_neumann_options_default(&o);
        o.tolerance = options.tolerance; o.max_iterations = options.max_iterations;
        o.max_terms = max_terms_; o.series_tolerance = series_tolerance_;
        o.order = order_; o.start = start_; o.residual = residual_; o.mem = SL_MEM_HOST;
        o.collect_stats = options.collect_stats; o.compute_error_bounds = options.compute_error_bounds;
        *guess = nullptr;
        if (options.initial_guess) {
            if (options.initial_guess->size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "initial_guess");        // :198-204
            *guess = options.initial_guess->data();
            o.start = SL_START_INITIAL_GUESS;
        }
        return o;
    }
    size_t max_terms_;
    Precision series_tolerance_;
    sl_order order_ = SL_ORDER_CSR_SEQUENTIAL;
    sl_start start_ = SL_START_ZERO;
    sl_residual residual_ = SL_RESIDUAL_TRUE;
};

inline NeumannState NeumannSolver::initialize(const SparseMatrix &matrix, const std::vector<Precision> &b, const SolverOptions &options,
                                              const Communicator *comm) const
{
    if (b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "neumann_initialization");     // :154-160
    const double *guess = nullptr;
    const sl_neumann_options o = abi_options(matrix, options, &guess);
    sl_neumann_state *st = nullptr;
    if (comm) check(sl_neumann_state_create_partitioned(comm->handle(), matrix.handle(), b.data(), guess, &o, &st));
    else check(sl_neumann_state_create(matrix.handle(), b.data(), guess, &o, &st));
    return NeumannState(st, matrix.rows());
}
inline StepResult NeumannSolver::step(NeumannState &state) const
{
    const sl_status st = sl_neumann_state_run(state.h_, nullptr, &state.last_);
    state.ran_ = true;
    if (st == SL_OK) return StepResult::Converged;
    if (st == SL_CONVERGENCE_FAILURE) return StepResult::Failed;
    check(st);
    return StepResult::Failed;
}
inline bool NeumannSolver::is_converged(const NeumannState &state) const { return state.ran_ && state.last_.converged != 0; }
inline std::vector<Precision> NeumannSolver::extract_solution(const NeumannState &state) const
{
    std::vector<Precision> x(state.n_);
    check(sl_neumann_state_solution(state.h_, x.data(), SL_MEM_HOST));
    return x;
}
inline void NeumannSolver::update_rhs(NeumannState &state, const std::vector<std::pair<size_t, Precision>> &delta_b) const
{
    std::vector<uint64_t> idx; std::vector<double> val;
    for (const auto &p : delta_b) { idx.push_back(p.first); val.push_back(p.second); }
    check(sl_neumann_state_update_rhs(state.h_, idx.size(), idx.data(), val.data()));
    state.last_.converged = 0;
}

// TS solveForwardPush (src/core/solver.ts:437-522) in the reference's own visiting order: one Gauss-Southwell push per iteration
class GaussSouthwellSolver {
public:
    explicit GaussSouthwellSolver(Precision epsilon = 1e-6, size_t max_iterations = 1000) : epsilon_(epsilon), max_iterations_(max_iterations) {}
    SolverResult solve(const SparseMatrix &matrix, const std::vector<Precision> &b, std::vector<uint32_t> *push_log = nullptr) const
    {
        if (b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "forward push");
        sl_southwell_options o;
        sl_southwell_options_default(&o);
        o.epsilon = epsilon_; o.max_iterations = max_iterations_;
        SolverResult out;
        out.solution.resize(matrix.rows());
        sl_southwell_result r;
        if (push_log) push_log->assign(max_iterations_, 0u);
        check(sl_forward_push_southwell(matrix.handle(), b.data(), &o, out.solution.data(), nullptr, push_log ? push_log->data() : nullptr,
                                        push_log ? push_log->size() : 0, &r));
        if (push_log) push_log->resize(r.iterations);
        out.residual_norm = r.residual_norm; out.iterations = r.iterations; out.converged = r.converged != 0;
        return out;
    }

private:
    Precision epsilon_;
    size_t max_iterations_;
};

struct PushResult {
    std::vector<Precision> solution, residual;
    size_t rounds = 0, pushes = 0;
    Precision residual_norm = 0;
    bool converged = false;
};

// thresholded residual push (ForwardPushConfig.epsilon -> theta, max_pushes -> max_rounds; forward_push.rs:26-49)
class PushSolver {
public:
    explicit PushSolver(Precision theta = 1e-6, size_t max_rounds = 1000000) : theta_(theta), max_rounds_(max_rounds) {}
    PushResult solve(const SparseMatrix &matrix, const std::vector<Precision> &b) const
    {
        if (b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "push");
        sl_push_options o;
        sl_push_options_default(&o);
        o.theta = theta_; o.max_rounds = max_rounds_;
        PushResult out;
        out.solution.assign(matrix.rows(), 0.0);
        out.residual.resize(matrix.rows());
        sl_push_result r;
        check(sl_push_solve(matrix.handle(), b.data(), &o, out.solution.data(), out.residual.data(), nullptr, 0, nullptr, &r));
        out.rounds = r.rounds; out.pushes = r.pushes; out.residual_norm = r.residual_norm; out.converged = r.converged != 0;
        return out;
    }
    // ForwardPushSolver::query_single_entry (forward_push.rs:224-231) / TS estimateEntry (solver.ts:550-659)
    Precision query_single_entry(const SparseMatrix &matrix, const std::vector<Precision> &b, size_t row, Precision *error_l1 = nullptr) const
    {
        sl_estimate_result r;
        check(sl_estimate_entry(matrix.handle(), b.data(), SL_MEM_HOST, row, theta_, max_rounds_, &r));
        if (error_l1) *error_l1 = r.residual_l1;
        return r.estimate;
    }

private:
    Precision theta_;
    size_t max_rounds_;
};


// ---- the graph side of the push spec: PushGraph, ForwardPushSolver, BackwardPushSolver (reference names) -----------------------------
// PushGraph (src/graph/adjacency.rs:199-277) on the device behind sl_push_graph_*; ForwardPushConfig (forward_push.rs:24-49);
// ForwardPushResult (:10-22).  solve_single_source / solve_multi_source / solve_with_target / solve_single_target run the spec's OWN
// visiting order (WorkQueue pops, graph/mod.rs:132-213): push_count, nodes_visited and every bit as the reference's loop.
struct ForwardPushConfig {
    Precision alpha = 0.15, epsilon = 1e-6, queue_threshold = 1e-8;
    size_t max_pushes = 1000000;
    bool adaptive_threshold = true;
};
using BackwardPushConfig = ForwardPushConfig;
struct ForwardPushResult {
    std::vector<Precision> estimate, residual;
    size_t push_count = 0, nodes_visited = 0;
    Precision residual_norm = 0;
};
using BackwardPushResult = ForwardPushResult;

class PushGraph {
public:
    // PushGraph::from_matrix: the weighted adjacency in CSR (u32 indices)
    PushGraph(size_t num_nodes, const std::vector<uint32_t> &row_ptr, const std::vector<uint32_t> &col_idx, const std::vector<Precision> &weights) : n_(num_nodes)
    {
        if (row_ptr.size() != num_nodes + 1 || col_idx.size() != weights.size()) throw SolverError(SL_DIMENSION_MISMATCH, "push graph arrays");
        check(sl_push_graph_create(num_nodes, row_ptr.data(), col_idx.data(), weights.data(), SL_MEM_HOST, &g_));
    }
    // PushGraph::from_edges (adjacency.rs:226-238): endpoints out of range are skipped
    static PushGraph from_edges(size_t num_nodes, std::vector<std::tuple<size_t, size_t, Precision>> edges)
    {
        edges.erase(std::remove_if(edges.begin(), edges.end(), [&](const auto &e) { return std::get<0>(e) >= num_nodes || std::get<1>(e) >= num_nodes; }), edges.end());
        std::stable_sort(edges.begin(), edges.end(), [](const auto &a, const auto &b) { return std::get<0>(a) != std::get<0>(b) ? std::get<0>(a) < std::get<0>(b) : std::get<1>(a) < std::get<1>(b); });
        std::vector<uint32_t> rp(num_nodes + 1, 0), ci;
        std::vector<Precision> w;
        for (const auto &e : edges) { ++rp[std::get<0>(e) + 1]; ci.push_back((uint32_t)std::get<1>(e)); w.push_back(std::get<2>(e)); }
        for (size_t i = 0; i < num_nodes; ++i) rp[i + 1] += rp[i];
        return PushGraph(num_nodes, rp, ci, w);
    }
    PushGraph(PushGraph &&o) noexcept : n_(o.n_), g_(o.g_) { o.g_ = nullptr; }
    PushGraph(const PushGraph &) = delete;
    PushGraph &operator=(const PushGraph &) = delete;
    ~PushGraph() { if (g_) sl_push_graph_destroy(g_); }
    size_t num_nodes() const { return n_; }
    size_t num_edges() const { uint64_t e = 0; check(sl_push_graph_size(g_, nullptr, &e)); return (size_t)e; }
    std::vector<Precision> degrees() const { std::vector<Precision> d(n_); check(sl_push_graph_degrees(g_, d.data(), nullptr, SL_MEM_HOST)); return d; }
    std::vector<Precision> reverse_degrees() const { std::vector<Precision> d(n_); check(sl_push_graph_degrees(g_, nullptr, d.data(), SL_MEM_HOST)); return d; }
    Precision out_degree(size_t node) const { return node < n_ ? degrees()[node] : 0.0; }
    Precision in_degree(size_t node) const { return node < n_ ? reverse_degrees()[node] : 0.0; }
    const sl_push_graph *handle() const { return g_; }

private:
    size_t n_ = 0;
    sl_push_graph *g_ = nullptr;
};

class ForwardPushSolver {
public:
    ForwardPushSolver(const PushGraph &graph, ForwardPushConfig config = {}) : graph_(graph), config_(config) {}
    ForwardPushResult solve_single_source(size_t source) const { const uint64_t s = source; return run(false, 1, &s, nullptr, 0.0); }        // forward_push.rs:67-122
    ForwardPushResult solve_multi_source(const std::vector<size_t> &sources) const                                                           // :125-177
    {
        std::vector<uint64_t> s(sources.begin(), sources.end());
        return run(false, s.size(), s.data(), nullptr, 0.0);
    }
    Precision query_single_entry(size_t source, size_t target) const                                                                        // :224-231
    {
        const ForwardPushResult r = solve_single_source(source);
        return target < r.estimate.size() ? r.estimate[target] : 0.0;
    }
    ForwardPushResult solve_with_target(size_t source, size_t target, Precision target_precision) const                                      // :233-290
    {
        const uint64_t s = source, t = target;
        return run(false, 1, &s, &t, target_precision);
    }
    std::vector<Precision> extrapolated_solution(const ForwardPushResult &r) const                                                          // :292-301
    {
        std::vector<Precision> x = r.estimate;
        for (size_t i = 0; i < x.size(); ++i) x[i] += config_.alpha * r.residual[i];
        return x;
    }

protected:
    ForwardPushResult run(bool backward, uint64_t count, const uint64_t *nodes, const uint64_t *target, Precision target_precision) const
    {
        sl_acl_options o;
        sl_acl_options_default(&o);
        o.alpha = config_.alpha; o.epsilon = config_.epsilon; o.queue_threshold = config_.queue_threshold; o.max_pushes = config_.max_pushes;
        o.adaptive_threshold = config_.adaptive_threshold ? 1 : 0; o.mem = SL_MEM_HOST;
        ForwardPushResult out;
        const size_t n = graph_.num_nodes();
        out.estimate.assign(n ? n : 1, 0.0); out.residual.assign(n ? n : 1, 0.0);
        sl_acl_result r;
        // with a stop node: forward = solve_with_target(source = nodes[0], target = *target); backward = solve_with_source(source = *target, target = nodes[0])
        if (target && backward) check(sl_backward_push_acl_with_source(graph_.handle(), *target, nodes[0], target_precision, &o, out.estimate.data(), out.residual.data(), nullptr, 0, &r));
        else if (target) check(sl_forward_push_acl_with_target(graph_.handle(), nodes[0], *target, target_precision, &o, out.estimate.data(), out.residual.data(), nullptr, 0, &r));
        else if (backward) check(sl_backward_push_acl(graph_.handle(), count, nodes, &o, out.estimate.data(), out.residual.data(), nullptr, 0, &r));
        else check(sl_forward_push_acl(graph_.handle(), count, nodes, &o, out.estimate.data(), out.residual.data(), nullptr, 0, &r));
        out.estimate.resize(n); out.residual.resize(n);
        out.push_count = r.push_count; out.nodes_visited = r.nodes_visited; out.residual_norm = r.residual_norm;
        return out;
    }
    const PushGraph &graph_;
    ForwardPushConfig config_;
};

class BackwardPushSolver : private ForwardPushSolver {      // backward_push.rs:67-334
public:
    BackwardPushSolver(const PushGraph &graph, BackwardPushConfig config = {}) : ForwardPushSolver(graph, config) {}
    BackwardPushResult solve_single_target(size_t target) const { const uint64_t t = target; return run(true, 1, &t, nullptr, 0.0); }             // :67-122
    BackwardPushResult solve_multi_target(const std::vector<size_t> &targets) const                                                         // :125-176
    {
        std::vector<uint64_t> t(targets.begin(), targets.end());
        return run(true, t.size(), t.data(), nullptr, 0.0);
    }
    Precision query_transition_probability(size_t source, size_t target) const                                                              // :228-235
    {
        const BackwardPushResult r = solve_single_target(target);
        return source < r.estimate.size() ? r.estimate[source] : 0.0;
    }
    BackwardPushResult solve_with_source(size_t source, size_t target, Precision source_precision) const                                    // :238-293
    {
        const uint64_t s = source, t = target;
        return run(true, 1, &t, &s, source_precision);
    }
    std::vector<Precision> reachability_probabilities(size_t target) const                                                                  // :296-299
    {
        sl_acl_options o;
        sl_acl_options_default(&o);
        o.alpha = config_.alpha; o.epsilon = config_.epsilon; o.queue_threshold = config_.queue_threshold; o.max_pushes = config_.max_pushes;
        o.adaptive_threshold = config_.adaptive_threshold ? 1 : 0; o.mem = SL_MEM_HOST;
        const size_t n = graph_.num_nodes();
        std::vector<Precision> x(n ? n : 1, 0.0);
        sl_acl_result r;
        check(sl_backward_push_acl_reachability(graph_.handle(), target, &o, x.data(), &r));
        x.resize(n);
        return x;
    }
    std::vector<Precision> extrapolated_solution(const BackwardPushResult &r) const                                                         // :302-311
    {
        std::vector<Precision> x(r.estimate.size() ? r.estimate.size() : 1, 0.0);
        check(sl_acl_extrapolated_solution(r.estimate.size(), config_.alpha, r.estimate.data(), r.residual.data(), x.data(), SL_MEM_HOST));
        x.resize(r.estimate.size());
        return x;
    }
    // :314-333 — three additions per node, one after the other, in the reference's order (a left-to-right fold over host vectors)
    Precision combine_with_forward(const BackwardPushResult &backward_result, const std::vector<Precision> &forward_estimate,
                                   const std::vector<Precision> &forward_residual) const
    {
        Precision total = 0.0;
        const size_t k = std::min(backward_result.estimate.size(), forward_estimate.size());
        for (size_t i = 0; i < k; ++i) {
            total += backward_result.estimate[i] * forward_estimate[i];
            total += backward_result.residual[i] * forward_estimate[i] * config_.alpha;
            total += backward_result.estimate[i] * forward_residual[i] * config_.alpha;
        }
        return total;
    }
};

// BidirectionalPushSolver (backward_push.rs:337-410): forward from the source, backward from the target, combined — or the single
// direction that starts from the node of much higher degree
class BidirectionalPushSolver {
public:
    BidirectionalPushSolver(const PushGraph &graph, ForwardPushConfig forward_config = {}, BackwardPushConfig backward_config = {})
        : graph_(graph), forward_(graph, forward_config), backward_(graph, backward_config) {}
    Precision solve_bidirectional(size_t source, size_t target) const                                                                       // :359-377
    {
        const ForwardPushResult f = forward_.solve_single_source(source);
        const BackwardPushResult b = backward_.solve_single_target(target);
        return backward_.combine_with_forward(b, f.estimate, f.residual);
    }
    Precision adaptive_solve(size_t source, size_t target) const                                                                            // :380-410
    {
        const size_t n = graph_.num_nodes();
        if (source >= n || target >= n) return 0.0;
        const Precision out_s = graph_.out_degree(source), in_t = graph_.in_degree(target);
        if (out_s > in_t * 2.0) return backward_.query_transition_probability(source, target);
        if (in_t > out_s * 2.0) return forward_.query_single_entry(source, target);
        return solve_bidirectional(source, target);
    }

private:
    const PushGraph &graph_;
    ForwardPushSolver forward_;
    BackwardPushSolver backward_;
};

// Many single-entry queries against one system: ForwardPushSolver::new(graph, config) once, query_single_entry per
// query (forward_push.rs:52-66, 224-231).  Setup is paid once; a query costs the rows its push touches.  `matrix` and
// `b` are captured: the matrix must outlive the session.
class QuerySession {
public:
    QuerySession(const SparseMatrix &matrix, const std::vector<Precision> &b, bool matrix_is_transpose = false)
    {
        if (b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "query session");
        check(sl_query_session_create(matrix.handle(), matrix_is_transpose ? 1 : 0, b.data(), SL_MEM_HOST, &q_));
    }
    QuerySession(const QuerySession &) = delete;
    QuerySession &operator=(const QuerySession &) = delete;
    ~QuerySession() { if (q_) sl_query_session_destroy(q_); }
    // x_row = (A^-1 b)_row; *error_l1 * max|x| bounds the error
    Precision query_single_entry(size_t row, Precision theta = 1e-8, size_t max_rounds = 100000, Precision *error_l1 = nullptr,
                                 size_t *rows_touched = nullptr) const
    {
        sl_estimate_result r;
        check(sl_query_session_estimate(q_, row, theta, max_rounds, &r));
        if (error_l1) *error_l1 = r.residual_l1;
        if (rows_touched) *rows_touched = r.rows_touched;
        return r.estimate;
    }

    // many independent queries at once, on lanes (sl_query_session_estimate_batch): results as the one-at-a-time answers, bit for bit
    std::vector<Precision> query_batch(const std::vector<size_t> &rows, Precision theta = 1e-8, size_t max_rounds = 100000, unsigned lanes = 0) const
    {
        std::vector<uint64_t> r64(rows.begin(), rows.end());
        std::vector<sl_estimate_result> res(rows.size() ? rows.size() : 1);
        check(sl_query_session_estimate_batch(q_, r64.size(), r64.data(), theta, max_rounds, lanes, res.data()));
        std::vector<Precision> out(rows.size());
        for (size_t i = 0; i < rows.size(); ++i) out[i] = res[i].estimate;
        return out;
    }

private:
    sl_query_session *q_ = nullptr;
};

// The crate's public free functions of src/simd_ops.rs (re-exported from lib.rs:83-87), the reference's argument order, on the device.
// A one-shot product pays the upload and the layout build; multiply by the same matrix again through SparseMatrix.
inline void spmv_once(const std::vector<Precision> &values, const std::vector<IndexType> &col_indices, const std::vector<IndexType> &row_ptr,
                      const std::vector<Precision> &x, std::vector<Precision> &y, sl_order order)
{
    const size_t rows = row_ptr.empty() ? 0 : row_ptr.size() - 1;
    if (y.size() != rows) throw SolverError(SL_DIMENSION_MISMATCH, "matrix_vector_multiply: y");
    SparseMatrix::from_csr(row_ptr, col_indices, values, rows, x.size()).multiply_vector(x, y, order);
}
// simd_ops::matrix_vector_multiply_simd (:20-88): rows of >= 8 entries in the 4-lane order, shorter ones sequentially
inline void matrix_vector_multiply_simd(const std::vector<Precision> &values, const std::vector<IndexType> &col_indices, const std::vector<IndexType> &row_ptr,
                                        const std::vector<Precision> &x, std::vector<Precision> &y) { spmv_once(values, col_indices, row_ptr, x, y, SL_ORDER_SIMD4); }
// simd_ops::parallel_matrix_vector_multiply (:201-239): row chunks, every row summed sequentially; the thread count has no meaning here
inline void parallel_matrix_vector_multiply(const std::vector<Precision> &values, const std::vector<IndexType> &col_indices, const std::vector<IndexType> &row_ptr,
                                            const std::vector<Precision> &x, std::vector<Precision> &y, std::optional<size_t> = std::nullopt)
{ spmv_once(values, col_indices, row_ptr, x, y, SL_ORDER_CSR_SEQUENTIAL); }
// simd_ops::dot_product_simd (:116-147; the device's sum is a fixed tree: the 4-lane sum to rounding) / axpy_simd (:158-189)
inline Precision dot_product_simd(const std::vector<Precision> &x, const std::vector<Precision> &y)
{
    if (x.size() != y.size()) throw SolverError(SL_DIMENSION_MISMATCH, "dot_product: lengths differ");
    double out = 0.0;
    check(sl_dot(x.size(), x.data(), y.data(), &out, SL_MEM_HOST));
    return out;
}
inline void axpy_simd(Precision alpha, const std::vector<Precision> &x, std::vector<Precision> &y)
{
    if (x.size() != y.size()) throw SolverError(SL_DIMENSION_MISMATCH, "axpy: lengths differ");
    check(sl_axpy(x.size(), alpha, x.data(), y.data(), SL_MEM_HOST));
}

// OptimizedConjugateGradientSolver (optimized_solver.rs:167-295; config defaults :119-127) / FastConjugateGradient
// (fast_solver.rs:110-178) over sl_cg_solve
class ConjugateGradientSolver {
public:
    explicit ConjugateGradientSolver(size_t max_iterations = 1000, Precision tolerance = 1e-6) : max_iterations_(max_iterations), tolerance_(tolerance) {}
    SolverResult solve(const SparseMatrix &matrix, const std::vector<Precision> &b) const
    {
        if (b.size() != matrix.rows()) throw SolverError(SL_DIMENSION_MISMATCH, "Right-hand side vector length must match matrix size");
        sl_cg_options o;
        sl_cg_options_default(&o);
        o.tolerance = tolerance_; o.max_iterations = max_iterations_;
        SolverResult out;
        out.solution.resize(matrix.rows());
        sl_cg_result r;
        check(sl_cg_solve(matrix.handle(), b.data(), &o, out.solution.data(), &r));
        out.residual_norm = r.residual_norm; out.iterations = r.iterations; out.converged = r.converged != 0;
        return out;
    }

private:
    size_t max_iterations_;
    Precision tolerance_;
};

} // namespace sublinear
