//! rust/sublinear_hip.rs — the reference-side binding of libsublinear_hip.so (MI355X): what a maintainer of ruvnet/sublinear-time-solver
//! adds as `src/solver/hip.rs` behind `feature = "hip"` (declared in src/solver/mod.rs next to `pub mod neumann;`, mod.rs:14, and
//! re-exported from lib.rs like `NeumannSolver`, lib.rs:70-97).  SOURCE ONLY: the build image has no rustc / cargo, so this file is
//! not compiled here; what IS checked here (tests/test_rust_shim.py, no GPU needed) is that every `extern "C"` declaration and every
//! `#[repr(C)]` struct below agrees with include/sublinear_hip.h — names, argument counts, pointer / scalar kinds and widths, field
//! order — so the shim cannot drift from the ABI it binds.
//!
//! Interfaces it implements: `trait Matrix` (src/matrix/mod.rs:25-104) for the device-resident `HipMatrix`; `trait SolverAlgorithm` /
//! `trait SolverState` (src/solver/mod.rs:223-351) over `&dyn Matrix`; `OptimizedConjugateGradientSolver` (src/optimized_solver.rs:167-350); `ForwardPushSolver` / `BackwardPushSolver` over `PushGraph` (src/solver/forward_push.rs:52-301,
//! src/solver/backward_push.rs:60-311, src/graph/adjacency.rs:199-277).
//! build.rs: `println!("cargo:rustc-link-lib=dylib=sublinear_hip")` + a `rustc-link-search` for the directory holding the `.so`.

use crate::error::{Result, SolverError};
use crate::matrix::Matrix;
use crate::solver::{SolverAlgorithm, SolverOptions, SolverResult, SolverState, StepResult};
use crate::types::{ErrorBounds, MemoryInfo, NodeId, Precision};
use crate::graph::PushGraph;
use crate::solver::forward_push::{ForwardPushConfig, ForwardPushResult};
use crate::solver::backward_push::{BackwardPushConfig, BackwardPushResult};
use core::ffi::{c_char, c_int, c_void};

#[repr(C)] pub struct SlMatrix { _private: [u8; 0] }

#[repr(C)] #[derive(Default)]
pub struct SlNeumannOptions {
    tolerance: f64, max_iterations: u64, max_terms: u64, series_tolerance: f64,
    order: i32, start: i32, residual: i32, mem: i32, collect_stats: i32, compute_error_bounds: i32,
}
#[repr(C)] #[derive(Default)]
pub struct SlNeumannResult {
    iterations: u64, terms_computed: u64, matvec_count: u64, residual_norm: f64, last_term_norm: f64,
    error_bound: f64, total_time_ms: f64, device_time_ms: f64, bytes_moved: u64,
    converged: i32, series_converged: i32,
}

#[repr(C)] pub struct SlNeumannState { _private: [u8; 0] }
#[repr(C)] #[derive(Default)]
pub struct SlMatrixInfo { n_rows: u64, n_cols: u64, nnz: u64, row_offset: u64, padded_nnz: u64, n_slices: u64, device_bytes: u64, bandwidth: u64,
                          max_row_nnz: u32, min_row_nnz: u32, uniform_width: u32, has_transpose: u32, long_row_threshold: u32, n_long_rows: u32,
                          column_panels: u32, reserved: u32 }
#[repr(C)] #[derive(Default)]
pub struct SlSparsityInfo { nnz: u64, rows: u64, cols: u64, sparsity_ratio: f64, avg_nnz_per_row: f64, max_nnz_per_row: u64, bandwidth: u64,
                            is_banded: i32, reserved: i32 }

#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_last_error_message() -> *const c_char;
    fn sl_matrix_create_from_triplets(n: u64, rows: *const u64, cols: *const u64, vals: *const f64,
                                      n_rows: u64, n_cols: u64, flags: u32, out: *mut *mut SlMatrix) -> c_int;
    fn sl_matrix_destroy(m: *mut SlMatrix);
    // the device-served half of `trait Matrix` (matrix/mod.rs:25-104)
    fn sl_spmv(m: *const SlMatrix, x: *const f64, y: *mut f64, order: c_int, mem: c_int) -> c_int;
    fn sl_spmv_add(m: *const SlMatrix, x: *const f64, y: *mut f64, order: c_int, mem: c_int) -> c_int;
    fn sl_matrix_is_diagonally_dominant(m: *const SlMatrix, is_dd: *mut c_int) -> c_int;
    fn sl_matrix_diagonal_dominance_factor(m: *const SlMatrix, has_factor: *mut c_int, factor: *mut f64) -> c_int;
    fn sl_matrix_spectral_radius_estimate(m: *const SlMatrix, radius: *mut f64) -> c_int;
    fn sl_matrix_get_info(m: *const SlMatrix, info: *mut SlMatrixInfo) -> c_int;
    fn sl_matrix_get(m: *const SlMatrix, row: u64, col: u64, found: *mut c_int, value: *mut f64) -> c_int;
    fn sl_matrix_row(m: *const SlMatrix, row: u64, capacity: u64, cols: *mut u32, values: *mut f64, count: *mut u64) -> c_int;
    fn sl_matrix_col(m: *const SlMatrix, col: u64, capacity: u64, rows: *mut u32, values: *mut f64, count: *mut u64) -> c_int;
    fn sl_matrix_frobenius_norm(m: *const SlMatrix, norm: *mut f64) -> c_int;
    fn sl_matrix_sparsity_info(m: *const SlMatrix, info: *mut SlSparsityInfo) -> c_int;
    // the two `&mut self` methods of SparseMatrix (matrix/mod.rs:346-372)
    fn sl_matrix_scale(m: *mut SlMatrix, factor: f64) -> c_int;
    fn sl_matrix_add_diagonal(m: *mut SlMatrix, alpha: f64) -> c_int;
    // solver::utils (solver/mod.rs:363-461)
    fn sl_l2_norm(n: u64, x: *const f64, out: *mut f64, mem: c_int) -> c_int;
    fn sl_l1_norm(n: u64, x: *const f64, out: *mut f64, mem: c_int) -> c_int;
    fn sl_linf_norm(n: u64, x: *const f64, out: *mut f64, mem: c_int) -> c_int;
    fn sl_compute_norm(n: u64, x: *const f64, norm_type: c_int, out: *mut f64, mem: c_int) -> c_int;
    fn sl_compute_residual(m: *const SlMatrix, x: *const f64, b: *const f64, residual: *mut f64, order: c_int, mem: c_int) -> c_int;
    fn sl_check_convergence(residual_norm: f64, tolerance: f64, mode: c_int, b_norm: f64, n: u64, prev_solution: *const f64,
                            current_solution: *const f64, mem: c_int, converged: *mut c_int) -> c_int;
    fn sl_neumann_options_streaming(o: *mut SlNeumannOptions);
    fn sl_neumann_result_meets_quality_criteria(r: *const SlNeumannResult, tolerance: f64) -> c_int;
    fn sl_neumann_options_default(o: *mut SlNeumannOptions);
    fn sl_neumann_solve(m: *const SlMatrix, b: *const f64, initial_guess: *const f64,
                        opts: *const SlNeumannOptions, x_out: *mut f64, term_norms: *mut f64,
                        result: *mut SlNeumannResult) -> c_int;
    // NeumannState as an object (initialize / update_rhs / extract_solution / reset)
    fn sl_neumann_state_create(m: *const SlMatrix, b: *const f64, initial_guess: *const f64, opts: *const SlNeumannOptions,
                               out: *mut *mut SlNeumannState) -> c_int;
    fn sl_neumann_state_destroy(s: *mut SlNeumannState);
    fn sl_neumann_state_update_rhs(s: *mut SlNeumannState, count: u64, indices: *const u64, deltas: *const f64) -> c_int;
    fn sl_neumann_state_run(s: *mut SlNeumannState, term_norms: *mut f64, result: *mut SlNeumannResult) -> c_int;
    fn sl_neumann_state_solution(s: *const SlNeumannState, x_out: *mut f64, mem: c_int) -> c_int;
    fn sl_neumann_state_reset(s: *mut SlNeumannState) -> c_int;
}

fn to_error(status: c_int, iterations: usize, residual: f64, tol: f64) -> SolverError {
    let msg = unsafe { std::ffi::CStr::from_ptr(sl_last_error_message()) }.to_string_lossy().into_owned();
    match status {                                            // 1:1 with error.rs:16-140
        1 => SolverError::MatrixNotDiagonallyDominant { row: 0, diagonal: 0.0, off_diagonal_sum: 0.0 },
        2 => SolverError::NumericalInstability { reason: msg, iteration: iterations, residual_norm: residual },
        3 => SolverError::ConvergenceFailure { iterations, residual_norm: residual, tolerance: tol, algorithm: "neumann".into() },
        4 => SolverError::InvalidInput { message: msg, parameter: None },
        5 => SolverError::DimensionMismatch { expected: 0, actual: 0, operation: msg },
        8 => SolverError::IndexOutOfBounds { index: 0, max_index: 0, context: msg },
        9 => SolverError::InvalidSparseMatrix { reason: msg, position: None },
        _ => SolverError::AlgorithmError { algorithm: "neumann-hip".into(), message: msg, context: vec![] },
    }
}

/// Device copy of a `&dyn Matrix`, built ONCE (upload + row-slice layout) and kept for every later solve on the same matrix.
/// Key = the address and shape of the borrowed matrix plus its nnz: the crate's matrices are immutable behind `&dyn Matrix`.
pub struct HipMatrix { handle: *mut SlMatrix, key: (usize, usize, usize, usize) }
unsafe impl Send for HipMatrix {}            // sl_matrix is immutable after create and shareable across threads (header, "Threading")
unsafe impl Sync for HipMatrix {}
impl HipMatrix {
    fn key_of(m: &dyn Matrix) -> (usize, usize, usize, usize) { (m as *const dyn Matrix as *const () as usize, m.rows(), m.cols(), m.nnz()) }
    pub fn upload(matrix: &dyn Matrix) -> Result<Self> {
        // CSR arrays via the trait: to_triplets (matrix/mod.rs:298-305) keeps the builder rules in ONE place
        let t = matrix.to_triplets()?;                                  // Vec<(usize, usize, Precision)>
        let (r, c, v): (Vec<u64>, Vec<u64>, Vec<f64>) =
            t.iter().fold((vec![], vec![], vec![]), |mut a, &(i, j, x)| { a.0.push(i as u64); a.1.push(j as u64); a.2.push(x); a });
        let mut h: *mut SlMatrix = core::ptr::null_mut();
        let st = unsafe { sl_matrix_create_from_triplets(v.len() as u64, r.as_ptr(), c.as_ptr(), v.as_ptr(),
                                                         matrix.rows() as u64, matrix.cols() as u64, 0, &mut h) };
        if st != 0 { return Err(to_error(st, 0, f64::INFINITY, 0.0)); }
        Ok(Self { handle: h, key: Self::key_of(matrix) })
    }
}
/// `trait Matrix` (matrix/mod.rs:25-104), every method served from the device copy with the reference's rules: multiply_vector
/// (:415-439), multiply_vector_add (:441-465, running sums seeded with `result`, sparse.rs:192-203), is_diagonally_dominant (:467-485),
/// diagonal_dominance_factor (:487-514), spectral_radius_estimate (:83-100), conditioning_info (:548-556); get (:383-395 — the entry
/// CSRStorage::get's binary search lands on, sparse.rs:142-155), row_iter (sparse.rs:158-176), col_iter (CSRColIter sparse.rs:273-298: one
/// pair per row), frobenius_norm (:74-82, tree-reduced: equal to rounding), sparsity_info (:523-545).  A `HipMatrix` can therefore be
/// handed to anything that takes `&dyn Matrix` — `NeumannSolver::solve` included — without a host copy of the entries beside it.
impl HipMatrix {
    fn dims(&self) -> (usize, usize) { (self.key.1, self.key.2) }
    fn pairs(&self, index: usize, by_column: bool) -> Vec<(crate::types::IndexType, Precision)> {
        let mut n: u64 = 0;
        let call = |cap: u64, i: *mut u32, v: *mut f64, n: &mut u64| unsafe {
            if by_column { sl_matrix_col(self.handle, index as u64, cap, i, v, n) } else { sl_matrix_row(self.handle, index as u64, cap, i, v, n) } };
        if call(0, core::ptr::null_mut(), core::ptr::null_mut(), &mut n) != 0 || n == 0 { return Vec::new(); }
        let (mut idx, mut val) = (vec![0u32; n as usize], vec![0.0f64; n as usize]);
        if call(n, idx.as_mut_ptr(), val.as_mut_ptr(), &mut n) != 0 { return Vec::new(); }
        idx.into_iter().zip(val).collect()
    }
}
impl Matrix for HipMatrix {
    fn rows(&self) -> usize { self.key.1 }
    fn cols(&self) -> usize { self.key.2 }
    fn nnz(&self) -> usize { self.key.3 }
    /// The device copy is built from `to_triplets()` of whatever storage the borrowed matrix holds; every storage the crate can convert a
    /// SparseMatrix into (matrix/mod.rs:244-296) multiplies to the CSR loop's bits (tests/test_storage_formats_host.py), so one layout serves all
    fn format_name(&self) -> &'static str { "CSR" }
    fn get(&self, row: usize, col: usize) -> Option<Precision> {
        let (mut found, mut v): (c_int, f64) = (0, 0.0);
        if unsafe { sl_matrix_get(self.handle, row as u64, col as u64, &mut found, &mut v) } == 0 && found != 0 { Some(v) } else { None }
    }
    fn row_iter(&self, row: usize) -> Box<dyn Iterator<Item = (crate::types::IndexType, Precision)> + '_> { Box::new(self.pairs(row, false).into_iter()) }
    fn col_iter(&self, col: usize) -> Box<dyn Iterator<Item = (crate::types::IndexType, Precision)> + '_> { Box::new(self.pairs(col, true).into_iter()) }
    fn multiply_vector(&self, x: &[Precision], result: &mut [Precision]) -> Result<()> {
        let (rows, cols) = self.dims();
        if x.len() != cols { return Err(SolverError::DimensionMismatch { expected: cols, actual: x.len(), operation: "matrix_vector_multiply".into() }); }
        if result.len() != rows { return Err(SolverError::DimensionMismatch { expected: rows, actual: result.len(), operation: "matrix_vector_multiply".into() }); }
        let st = unsafe { sl_spmv(self.handle, x.as_ptr(), result.as_mut_ptr(), 0 /* SL_ORDER_CSR_SEQUENTIAL */, 0 /* SL_MEM_HOST */) };
        if st != 0 { Err(to_error(st, 0, f64::INFINITY, 0.0)) } else { Ok(()) }
    }
    fn multiply_vector_add(&self, x: &[Precision], result: &mut [Precision]) -> Result<()> {
        let (rows, cols) = self.dims();
        if x.len() != cols { return Err(SolverError::DimensionMismatch { expected: cols, actual: x.len(), operation: "matrix_vector_multiply_add".into() }); }
        if result.len() != rows { return Err(SolverError::DimensionMismatch { expected: rows, actual: result.len(), operation: "matrix_vector_multiply_add".into() }); }
        let st = unsafe { sl_spmv_add(self.handle, x.as_ptr(), result.as_mut_ptr(), 0 /* SL_ORDER_CSR_SEQUENTIAL */, 0 /* SL_MEM_HOST */) };
        if st != 0 { Err(to_error(st, 0, f64::INFINITY, 0.0)) } else { Ok(()) }
    }
    fn is_diagonally_dominant(&self) -> bool {
        let mut f: c_int = 0;
        unsafe { sl_matrix_is_diagonally_dominant(self.handle, &mut f) == 0 && f != 0 }
    }
    fn diagonal_dominance_factor(&self) -> Option<Precision> {
        let (mut has, mut f): (c_int, f64) = (0, 0.0);
        if unsafe { sl_matrix_diagonal_dominance_factor(self.handle, &mut has, &mut f) } == 0 && has != 0 { Some(f) } else { None }
    }
    fn spectral_radius_estimate(&self) -> Precision {
        let mut r = 0.0;
        unsafe { sl_matrix_spectral_radius_estimate(self.handle, &mut r) };
        r
    }
    fn frobenius_norm(&self) -> Precision {
        let mut r = 0.0;
        unsafe { sl_matrix_frobenius_norm(self.handle, &mut r) };
        r
    }
    fn sparsity_info(&self) -> crate::types::SparsityInfo {
        let mut i = SlSparsityInfo::default();
        unsafe { sl_matrix_sparsity_info(self.handle, &mut i) };
        crate::types::SparsityInfo { nnz: i.nnz as usize, dimensions: (i.rows as usize, i.cols as usize), sparsity_ratio: i.sparsity_ratio,
                                     avg_nnz_per_row: i.avg_nnz_per_row, max_nnz_per_row: i.max_nnz_per_row as usize,
                                     bandwidth: Some(i.bandwidth as usize), is_banded: i.is_banded != 0 }
    }
    fn conditioning_info(&self) -> crate::types::ConditioningInfo {
        crate::types::ConditioningInfo { condition_number: None, is_diagonally_dominant: self.is_diagonally_dominant(),
                                         diagonal_dominance_factor: self.diagonal_dominance_factor(),
                                         spectral_radius: Some(self.spectral_radius_estimate()), is_positive_definite: None }
    }
}
impl Drop for HipMatrix { fn drop(&mut self) { unsafe { sl_matrix_destroy(self.handle) } } }
/// SparseMatrix::scale / add_diagonal (matrix/mod.rs:346-372 over CSRStorage::scale / add_diagonal, sparse.rs:229-248), in place on the
/// device: `&mut self` is exactly the exclusivity the ABI asks for (no solve running on the matrix).  add_diagonal skips rows without a
/// stored diagonal entry as the reference does; InvalidInput for a non-square matrix.
impl HipMatrix {
    pub fn scale(&mut self, factor: Precision) { unsafe { sl_matrix_scale(self.handle, factor) }; }
    pub fn add_diagonal(&mut self, alpha: Precision) -> Result<()> {
        let st = unsafe { sl_matrix_add_diagonal(self.handle, alpha) };
        if st != 0 { Err(to_error(st, 0, f64::INFINITY, 0.0)) } else { Ok(()) }
    }
}

/// solver::utils (solver/mod.rs:363-461) with the vectors reduced on the device: same names, same argument meaning.  Sums are tree
/// reductions (equal to the reference's sequential sums to rounding); linf_norm and every comparison are exact.
pub mod hip_utils {
    use super::*;
    use crate::types::{ConvergenceMode, NormType};
    fn norm(f: unsafe extern "C" fn(u64, *const f64, *mut f64, c_int) -> c_int, v: &[Precision]) -> Precision {
        let mut out = 0.0;
        unsafe { f(v.len() as u64, v.as_ptr(), &mut out, 0 /* SL_MEM_HOST */) };
        out
    }
    pub fn l2_norm(v: &[Precision]) -> Precision { norm(sl_l2_norm, v) }
    pub fn l1_norm(v: &[Precision]) -> Precision { norm(sl_l1_norm, v) }
    pub fn linf_norm(v: &[Precision]) -> Precision { norm(sl_linf_norm, v) }
    pub fn compute_norm(v: &[Precision], norm_type: NormType) -> Precision {
        let t = match norm_type { NormType::L1 => 0, NormType::L2 => 1, NormType::LInfinity => 2, NormType::Weighted => 3 };
        let mut out = 0.0;
        unsafe { sl_compute_norm(v.len() as u64, v.as_ptr(), t, &mut out, 0) };
        out
    }
    pub fn compute_residual(matrix: &HipMatrix, x: &[Precision], b: &[Precision], residual: &mut [Precision]) -> Result<()> {
        if x.len() != matrix.cols() { return Err(SolverError::DimensionMismatch { expected: matrix.cols(), actual: x.len(), operation: "matrix_vector_multiply".into() }); }
        if residual.len() != matrix.rows() || b.len() != matrix.rows() {
            return Err(SolverError::DimensionMismatch { expected: matrix.rows(), actual: residual.len().min(b.len()), operation: "matrix_vector_multiply".into() });
        }
        let st = unsafe { sl_compute_residual(matrix.handle, x.as_ptr(), b.as_ptr(), residual.as_mut_ptr(), 0, 0) };
        if st != 0 { Err(to_error(st, 0, f64::INFINITY, 0.0)) } else { Ok(()) }
    }
    pub fn check_convergence(residual_norm: Precision, tolerance: Precision, mode: ConvergenceMode, b_norm: Precision,
                             prev_solution: Option<&[Precision]>, current_solution: &[Precision]) -> bool {
        let m = match mode { ConvergenceMode::ResidualNorm => 0, ConvergenceMode::RelativeResidual => 1, ConvergenceMode::SolutionChange => 2,
                             ConvergenceMode::RelativeSolutionChange => 3, ConvergenceMode::Combined => 4 };
        let mut out: c_int = 0;
        let prev = prev_solution.map(|p| p.as_ptr()).unwrap_or(core::ptr::null());
        unsafe { sl_check_convergence(residual_norm, tolerance, m, b_norm, current_solution.len() as u64, prev, current_solution.as_ptr(), 0, &mut out) };
        out != 0
    }
    /// SolverOptions::streaming (solver/mod.rs:101-116) as the device options; SolverResult::meets_quality_criteria (:192-195)
    pub fn streaming_options() -> SlNeumannOptions { let mut o = SlNeumannOptions::default(); unsafe { sl_neumann_options_streaming(&mut o) }; o }
    pub fn meets_quality_criteria(r: &SlNeumannResult, tolerance: Precision) -> bool { unsafe { sl_neumann_result_meets_quality_criteria(r, tolerance) != 0 } }
}

/// Same constructor surface as NeumannSolver (neumann.rs:48-92).  The solver caches the device matrix of the last system it saw.
pub struct HipNeumannSolver { pub max_terms: usize, pub series_tolerance: Precision, cache: std::sync::Mutex<Option<std::sync::Arc<HipMatrix>>> }
impl Default for HipNeumannSolver { fn default() -> Self { Self { max_terms: 50, series_tolerance: 1e-8, cache: Default::default() } } }
impl HipNeumannSolver {
    fn device_matrix(&self, matrix: &dyn Matrix) -> Result<std::sync::Arc<HipMatrix>> {
        let mut slot = self.cache.lock().unwrap();
        if let Some(m) = slot.as_ref() { if m.key == HipMatrix::key_of(matrix) { return Ok(m.clone()); } }
        let m = std::sync::Arc::new(HipMatrix::upload(matrix)?);
        *slot = Some(m.clone());
        Ok(m)
    }
    fn options(&self, options: &SolverOptions) -> (SlNeumannOptions, *const f64) {
        // SolverOptions (mod.rs:22-45) + solver fields (neumann.rs:24-33)
        let mut o = SlNeumannOptions::default();
        unsafe { sl_neumann_options_default(&mut o) };
        o.tolerance = options.tolerance; o.max_iterations = options.max_iterations as u64;
        o.max_terms = self.max_terms as u64; o.series_tolerance = self.series_tolerance;
        o.collect_stats = options.collect_stats as i32; o.compute_error_bounds = options.compute_error_bounds as i32;
        let guess = options.initial_guess.as_ref().map(|g| { o.start = 2; g.as_ptr() }).unwrap_or(core::ptr::null());
        // o.start = 1; o.residual = 1;   // opt in to the reference's x0 = D^-1 b / scaled-residual behaviour
        (o, guess)
    }
}

/// NeumannState (neumann.rs:95-137) living on the device: the ABI's state object + the matrix it borrows.
pub struct HipState { raw: *mut SlNeumannState, _matrix: std::sync::Arc<HipMatrix>, n: usize, last: SlNeumannResult, tolerance: Precision }
impl Drop for HipState { fn drop(&mut self) { unsafe { sl_neumann_state_destroy(self.raw) } } }
impl SolverState for HipState {
    fn residual_norm(&self) -> Precision { if self.last.iterations == 0 && self.last.matvec_count == 0 { Precision::INFINITY } else { self.last.residual_norm } }
    fn matvec_count(&self) -> usize { self.last.matvec_count as usize }
    /// estimate_error_bounds (neumann.rs:321-347) of the state the last run ended in; the ABI reports None as a negative bound
    fn error_bounds(&self) -> Option<ErrorBounds> {
        if self.last.error_bound >= 0.0 { Some(ErrorBounds::upper_bound_only(self.last.error_bound, crate::types::ErrorBoundMethod::NeumannTruncation)) } else { None }
    }
    /// neumann.rs:219-227 counts its five n-vectors; the device state holds six (b, dinv, rhs, x, two term buffers) and the matrix's
    /// layouts in HBM are known exactly (`sl_matrix_info.device_bytes`), where the reference leaves `matrix_memory_bytes` a TODO
    fn memory_usage(&self) -> MemoryInfo {
        let mut i = SlMatrixInfo::default();
        unsafe { sl_matrix_get_info(self._matrix.handle, &mut i) };
        let vectors = self.n * 8 * 6;
        MemoryInfo { current_usage_bytes: vectors + i.device_bytes as usize, peak_usage_bytes: vectors + i.device_bytes as usize,
                     matrix_memory_bytes: i.device_bytes as usize, vector_memory_bytes: vectors, workspace_memory_bytes: 0,
                     allocation_count: 6, deallocation_count: 0 }
    }
    fn reset(&mut self) { unsafe { sl_neumann_state_reset(self.raw) }; self.last = SlNeumannResult::default(); }     // neumann.rs:367-378
}

impl SolverAlgorithm for HipNeumannSolver {
    type State = HipState;
    fn initialize(&self, matrix: &dyn Matrix, b: &[Precision], options: &SolverOptions) -> Result<HipState> {          // NeumannState::new
        if b.len() != matrix.rows() {                                   // neumann.rs:154-160
            return Err(SolverError::DimensionMismatch { expected: matrix.rows(), actual: b.len(), operation: "neumann_initialization".into() });
        }
        let dm = self.device_matrix(matrix)?;
        let (o, guess) = self.options(options);
        let mut raw = core::ptr::null_mut();
        let st = unsafe { sl_neumann_state_create(dm.handle, b.as_ptr(), guess, &o, &mut raw) };
        if st != 0 { return Err(to_error(st, 0, f64::INFINITY, options.tolerance)); }
        Ok(HipState { raw, _matrix: dm, n: matrix.rows(), last: SlNeumannResult::default(), tolerance: options.tolerance })
    }
    /// One call runs the whole loop of neumann.rs:477-555 on the device (the reference's own `step` cannot iterate at all,
    /// neumann.rs:393-419); it reports Converged / Failed like a last step would.
    fn step(&self, s: &mut HipState) -> Result<StepResult> {
        let st = unsafe { sl_neumann_state_run(s.raw, core::ptr::null_mut(), &mut s.last) };
        match st { 0 => Ok(StepResult::Converged),
                   3 => Ok(StepResult::Failed("convergence failure".into())),
                   _ => Err(to_error(st, s.last.iterations as usize, s.last.residual_norm, s.tolerance)) }
    }
    fn is_converged(&self, s: &HipState) -> bool { s.last.converged != 0 }
    fn extract_solution(&self, s: &HipState) -> Vec<Precision> {
        let mut x = vec![0.0; s.n];
        unsafe { sl_neumann_state_solution(s.raw, x.as_mut_ptr(), 0 /* SL_MEM_HOST */) };
        x
    }
    /// neumann.rs:436-462 on the device state, statement for statement (rhs and solution += delta * dinv, series state reset;
    /// IndexOutOfBounds for an index >= n).  An update is never dropped: any other failure comes back as an error.
    fn update_rhs(&self, s: &mut HipState, d: &[(usize, Precision)]) -> Result<()> {
        let (i, v): (Vec<u64>, Vec<f64>) = d.iter().map(|&(i, v)| (i as u64, v)).unzip();
        let st = unsafe { sl_neumann_state_update_rhs(s.raw, d.len() as u64, i.as_ptr(), v.as_ptr()) };
        if st != 0 { return Err(to_error(st, 0, s.last.residual_norm, s.tolerance)); }
        s.last.converged = 0;
        Ok(())
    }
    fn algorithm_name(&self) -> &'static str { "neumann-hip" }

    fn solve(&self, matrix: &dyn Matrix, b: &[Precision], options: &SolverOptions) -> Result<SolverResult> {
        if b.len() != matrix.rows() {                                   // neumann.rs:154-160
            return Err(SolverError::DimensionMismatch { expected: matrix.rows(), actual: b.len(), operation: "neumann_initialization".into() });
        }
        let dm = self.device_matrix(matrix)?;                           // (i) uploaded and laid out once per matrix, reused by later solves
        let (o, guess) = self.options(options);                         // (ii)
        let mut x = vec![0.0; matrix.rows()];
        let mut res = SlNeumannResult::default();
        let st = unsafe { sl_neumann_solve(dm.handle, b.as_ptr(), guess, &o, x.as_mut_ptr(), core::ptr::null_mut(), &mut res) };
        // (iii) status -> SolverError / SolverResult (mod.rs:118-195)
        if st != 0 { return Err(to_error(st, res.iterations as usize, res.residual_norm, options.tolerance)); }
        let mut out = if res.converged != 0 { SolverResult::success(x, res.residual_norm, res.iterations as usize) }
                      else { SolverResult::failure(x, res.residual_norm, res.iterations as usize) };
        if options.collect_stats {
            let mut s = crate::types::SolverStats::new();
            s.total_time_ms = res.total_time_ms; s.matvec_count = res.matvec_count as usize;
            out.stats = Some(s);
        }
        if options.compute_error_bounds && res.error_bound >= 0.0 {      // neumann.rs:549-551
            out.error_bounds = Some(ErrorBounds::upper_bound_only(res.error_bound, crate::types::ErrorBoundMethod::NeumannTruncation));
        }
        Ok(out)
    }
}

// ---- single-entry queries: ForwardPushSolver over a query session ---------------------------------------------------------------
#[repr(C)] pub struct SlQuerySession { _private: [u8; 0] }
#[repr(C)] #[derive(Default)]
pub struct SlEstimateResult { estimate: f64, residual_l1: f64, rounds: u64, pushes: u64, rows_touched: u64,
                              device_time_ms: f64, converged: i32, reserved: i32 }
#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_query_session_create(m: *const SlMatrix, matrix_is_transpose: c_int, b: *const f64, mem: c_int,
                               out: *mut *mut SlQuerySession) -> c_int;
    fn sl_query_session_estimate(q: *mut SlQuerySession, row: u64, theta: f64, max_rounds: u64,
                                 res: *mut SlEstimateResult) -> c_int;
    fn sl_query_session_destroy(q: *mut SlQuerySession);
}

/// ForwardPushSolver::new + query_single_entry (src/solver/forward_push.rs:52-66, 224-231): the session is created once per
/// (graph, rhs); every query then costs the rows its push touches.  `matrix` is a handle made with flag 1
/// (SL_MATRIX_WITH_TRANSPOSE) by sl_matrix_create_from_triplets, as in `solve` above; it is owned here.
pub struct HipForwardPush { matrix: *mut SlMatrix, session: *mut SlQuerySession, epsilon: Precision, max_pushes: u64 }

impl HipForwardPush {
    pub fn new(matrix: *mut SlMatrix, b: &[Precision], config: &ForwardPushConfig) -> Result<Self> {
        let mut session = core::ptr::null_mut();
        let st = unsafe { sl_query_session_create(matrix, 0, b.as_ptr(), 0 /* SL_MEM_HOST */, &mut session) };
        if st != 0 { return Err(to_error(st, 0, f64::INFINITY, config.epsilon)); }
        Ok(Self { matrix, session, epsilon: config.epsilon, max_pushes: config.max_pushes as u64 })
    }
    pub fn query_single_entry(&mut self, row: NodeId) -> Result<(Precision, Precision)> {      // (estimate, ||r||_1)
        let mut r = SlEstimateResult::default();
        let st = unsafe { sl_query_session_estimate(self.session, row as u64, self.epsilon, self.max_pushes, &mut r) };
        if st != 0 { return Err(to_error(st, r.rounds as usize, r.residual_l1, self.epsilon)); }
        Ok((r.estimate, r.residual_l1))
    }
}
impl Drop for HipForwardPush { fn drop(&mut self) { unsafe { sl_query_session_destroy(self.session); sl_matrix_destroy(self.matrix) } } }

// ---- the graph side: ForwardPushSolver / BackwardPushSolver::new(graph, config) served from the header alone -------------------------
#[repr(C)] pub struct SlPushGraph { _private: [u8; 0] }
#[repr(C)] pub struct SlAclOptions { alpha: f64, epsilon: f64, queue_threshold: f64, max_pushes: u64, adaptive_threshold: i32, mem: i32 }
#[repr(C)] #[derive(Default)]
pub struct SlAclResult { push_count: u64, nodes_visited: u64, residual_norm: f64, device_time_ms: f64, stopped_by: i32, reserved: i32 }
#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_push_graph_create(n: u64, row_ptr: *const u32, col_idx: *const u32, weights: *const f64, mem: c_int, out: *mut *mut SlPushGraph) -> c_int;
    fn sl_push_graph_destroy(g: *mut SlPushGraph);
    fn sl_push_graph_degrees(g: *const SlPushGraph, out_deg: *mut f64, in_deg: *mut f64, mem: c_int) -> c_int;
    fn sl_push_graph_system(g: *const SlPushGraph, alpha: f64, system_flags: u32, matrix_flags: u32, out: *mut *mut SlMatrix) -> c_int;
    fn sl_forward_push_acl(g: *const SlPushGraph, n_sources: u64, sources: *const u64, o: *const SlAclOptions, estimate: *mut f64,
                           residual: *mut f64, push_log: *mut u32, log_cap: u64, res: *mut SlAclResult) -> c_int;
    fn sl_forward_push_acl_with_target(g: *const SlPushGraph, source: u64, target: u64, target_precision: f64, o: *const SlAclOptions,
                                       estimate: *mut f64, residual: *mut f64, push_log: *mut u32, log_cap: u64, res: *mut SlAclResult) -> c_int;
    fn sl_backward_push_acl(g: *const SlPushGraph, n_targets: u64, targets: *const u64, o: *const SlAclOptions, estimate: *mut f64,
                            residual: *mut f64, push_log: *mut u32, log_cap: u64, res: *mut SlAclResult) -> c_int;
    fn sl_backward_push_acl_with_source(g: *const SlPushGraph, source: u64, target: u64, source_precision: f64, o: *const SlAclOptions,
                                        estimate: *mut f64, residual: *mut f64, push_log: *mut u32, log_cap: u64, res: *mut SlAclResult) -> c_int;
    fn sl_acl_extrapolated_solution(count: u64, alpha: f64, estimate: *const f64, residual: *const f64, solution: *mut f64, mem: c_int) -> c_int;
    fn sl_backward_push_acl_reachability(g: *const SlPushGraph, target: u64, o: *const SlAclOptions, solution: *mut f64, res: *mut SlAclResult) -> c_int;
    fn sl_acl_options_default(o: *mut SlAclOptions);
}

fn acl_options(alpha: f64, epsilon: f64, queue_threshold: f64, max_pushes: usize, adaptive: bool) -> SlAclOptions {
    SlAclOptions { alpha, epsilon, queue_threshold, max_pushes: max_pushes as u64, adaptive_threshold: adaptive as i32, mem: 0 /* SL_MEM_HOST */ }
}

/// ForwardPushSolver over PushGraph (src/solver/forward_push.rs:52-301, src/graph/adjacency.rs:199-277).  The graph lives on the device;
/// `solve_single_source` / `solve_multi_source` / `solve_with_target` / `query_single_entry` run the spec's OWN visiting order (WorkQueue
/// pops) — push_count / nodes_visited / bits as the reference's loop; the system sl_push_graph_system assembles (SL_SYSTEM_FORWARD,
/// WITH_TRANSPOSE) is what the data-parallel push and the local single-entry query (`HipForwardPush` above) run on.
pub struct HipForwardPushSolver { graph: *mut SlPushGraph, system: *mut SlMatrix, config: ForwardPushConfig, n: usize }
impl HipForwardPushSolver {
    pub fn new(graph: &PushGraph, config: ForwardPushConfig) -> Result<Self> {
        let a = &graph.adjacency;                                  // CompressedSparseRow { row_ptr, col_indices, values }
        let (rp, ci): (Vec<u32>, Vec<u32>) = (a.row_ptr.iter().map(|&v| v as u32).collect(), a.col_indices.iter().map(|&v| v as u32).collect());
        let (mut g, mut m) = (core::ptr::null_mut(), core::ptr::null_mut());
        let st = unsafe { sl_push_graph_create(a.nrows as u64, rp.as_ptr(), ci.as_ptr(), a.values.as_ptr(), 0, &mut g) };
        if st != 0 { return Err(to_error(st, 0, f64::INFINITY, config.epsilon)); }
        let st = unsafe { sl_push_graph_system(g, config.alpha, 0 /* SL_SYSTEM_FORWARD */, 1 /* WITH_TRANSPOSE */, &mut m) };
        if st != 0 { unsafe { sl_push_graph_destroy(g) }; return Err(to_error(st, 0, f64::INFINITY, config.epsilon)); }
        Ok(Self { graph: g, system: m, config, n: a.nrows })
    }
    pub fn solve_with_target(&self, source: usize, target: usize, target_precision: f64) -> ForwardPushResult {    // forward_push.rs:233-290
        let o = acl_options(self.config.alpha, self.config.epsilon, self.config.queue_threshold, self.config.max_pushes, self.config.adaptive_threshold);
        let (mut est, mut res, mut r) = (vec![0.0; self.n], vec![0.0; self.n], SlAclResult::default());
        unsafe { sl_forward_push_acl_with_target(self.graph, source as u64, target as u64, target_precision, &o, est.as_mut_ptr(),
                                                 res.as_mut_ptr(), core::ptr::null_mut(), 0, &mut r) };
        ForwardPushResult { estimate: est, residual: res, push_count: r.push_count as usize, nodes_visited: r.nodes_visited as usize, residual_norm: r.residual_norm }
    }
}
impl HipForwardPushSolver {
    /// solve_single_source / solve_multi_source in the spec's own order (forward_push.rs:67-177)
    pub fn solve_multi_source_exact(&self, sources: &[usize]) -> ForwardPushResult {
        let s: Vec<u64> = sources.iter().map(|&v| v as u64).collect();
        let o = acl_options(self.config.alpha, self.config.epsilon, self.config.queue_threshold, self.config.max_pushes, self.config.adaptive_threshold);
        let (mut est, mut res, mut r) = (vec![0.0; self.n], vec![0.0; self.n], SlAclResult::default());
        unsafe { sl_forward_push_acl(self.graph, s.len() as u64, s.as_ptr(), &o, est.as_mut_ptr(), res.as_mut_ptr(), core::ptr::null_mut(), 0, &mut r) };
        ForwardPushResult { estimate: est, residual: res, push_count: r.push_count as usize, nodes_visited: r.nodes_visited as usize, residual_norm: r.residual_norm }
    }
    pub fn solve_single_source_exact(&self, source: usize) -> ForwardPushResult { self.solve_multi_source_exact(&[source]) }
    /// the reference's own names (forward_push.rs:67, :125, :224): the spec's visiting order, i.e. its bits
    pub fn solve_single_source(&self, source: usize) -> ForwardPushResult { self.solve_multi_source_exact(&[source]) }
    pub fn solve_multi_source(&self, sources: &[usize]) -> ForwardPushResult { self.solve_multi_source_exact(sources) }
    pub fn query_single_entry(&self, source: usize, target: usize) -> f64 {                                            // :224-231
        let r = self.solve_single_source(source);
        if target < r.estimate.len() { r.estimate[target] } else { 0.0 }
    }
    pub fn extrapolated_solution(&self, result: &ForwardPushResult) -> Vec<f64> {                                        // forward_push.rs:292-301
        let mut x = vec![0.0; result.estimate.len()];
        unsafe { sl_acl_extrapolated_solution(x.len() as u64, self.config.alpha, result.estimate.as_ptr(), result.residual.as_ptr(), x.as_mut_ptr(), 0) };
        x
    }
}
impl Drop for HipForwardPushSolver { fn drop(&mut self) { unsafe { sl_matrix_destroy(self.system); sl_push_graph_destroy(self.graph) } } }

/// BackwardPushSolver over PushGraph (src/solver/backward_push.rs:60-311) in the spec's OWN visiting order: the WorkQueue's pops over the
/// reverse adjacency, in-degrees in the admission rule — push_count / nodes_visited / every bit of estimate and residual as the
/// reference's loop.  The reverse adjacency and both degree vectors live on the device (built once by sl_push_graph_create).
pub struct HipBackwardPushSolver { graph: *mut SlPushGraph, config: BackwardPushConfig, n: usize }
impl HipBackwardPushSolver {
    pub fn new(graph: &PushGraph, config: BackwardPushConfig) -> Result<Self> {                                          // backward_push.rs:62-64
        let a = &graph.adjacency;
        let (rp, ci): (Vec<u32>, Vec<u32>) = (a.row_ptr.iter().map(|&v| v as u32).collect(), a.col_indices.iter().map(|&v| v as u32).collect());
        let mut g = core::ptr::null_mut();
        let st = unsafe { sl_push_graph_create(a.nrows as u64, rp.as_ptr(), ci.as_ptr(), a.values.as_ptr(), 0, &mut g) };
        if st != 0 { return Err(to_error(st, 0, f64::INFINITY, config.epsilon)); }
        Ok(Self { graph: g, config, n: a.nrows })
    }
    fn options(&self) -> SlAclOptions {
        acl_options(self.config.alpha, self.config.epsilon, self.config.queue_threshold, self.config.max_pushes, self.config.adaptive_threshold)
    }
    fn result(est: Vec<f64>, res: Vec<f64>, r: SlAclResult) -> BackwardPushResult {
        BackwardPushResult { estimate: est, residual: res, push_count: r.push_count as usize, nodes_visited: r.nodes_visited as usize, residual_norm: r.residual_norm }
    }
    pub fn solve_single_target(&self, target: usize) -> BackwardPushResult { self.solve_multi_target(&[target]) }         // :67-122 (one target: unit mass)
    pub fn solve_multi_target(&self, targets: &[usize]) -> BackwardPushResult {                                          // :125-176
        let t: Vec<u64> = targets.iter().map(|&v| v as u64).collect();
        let (o, mut est, mut res, mut r) = (self.options(), vec![0.0; self.n], vec![0.0; self.n], SlAclResult::default());
        unsafe { sl_backward_push_acl(self.graph, t.len() as u64, t.as_ptr(), &o, est.as_mut_ptr(), res.as_mut_ptr(), core::ptr::null_mut(), 0, &mut r) };
        Self::result(est, res, r)
    }
    pub fn query_transition_probability(&self, source: usize, target: usize) -> f64 {                                    // :228-235
        let r = self.solve_single_target(target);
        if source < r.estimate.len() { r.estimate[source] } else { 0.0 }
    }
    /// :238-293 — ends once estimate[source] > source_precision && residual[source] < 0.1 source_precision (:262-264); source or target
    /// out of range: the empty result (:243-251)
    pub fn solve_with_source(&self, source: usize, target: usize, source_precision: f64) -> BackwardPushResult {
        let (o, mut est, mut res, mut r) = (self.options(), vec![0.0; self.n], vec![0.0; self.n], SlAclResult::default());
        unsafe { sl_backward_push_acl_with_source(self.graph, source as u64, target as u64, source_precision, &o, est.as_mut_ptr(),
                                                  res.as_mut_ptr(), core::ptr::null_mut(), 0, &mut r) };
        Self::result(est, res, r)
    }
    pub fn reachability_probabilities(&self, target: usize) -> Vec<f64> {                                                // :296-299, one call on the device
        let (o, mut x, mut r) = (self.options(), vec![0.0; self.n], SlAclResult::default());
        unsafe { sl_backward_push_acl_reachability(self.graph, target as u64, &o, x.as_mut_ptr(), &mut r) };
        x
    }
    pub fn extrapolated_solution(&self, result: &BackwardPushResult) -> Vec<f64> {                                       // :302-311
        let mut x = vec![0.0; result.estimate.len()];
        unsafe { sl_acl_extrapolated_solution(x.len() as u64, self.config.alpha, result.estimate.as_ptr(), result.residual.as_ptr(), x.as_mut_ptr(), 0) };
        x
    }
}
impl HipBackwardPushSolver {
    /// :314-333 — three adds per node in the reference's order; a left-to-right fold over the callers' host slices stays on the host
    pub fn combine_with_forward(&self, backward_result: &BackwardPushResult, forward_estimate: &[f64], forward_residual: &[f64]) -> f64 {
        let alpha = self.config.alpha;
        (0..backward_result.estimate.len().min(forward_estimate.len())).fold(0.0, |t, i| {
            let t = t + backward_result.estimate[i] * forward_estimate[i];
            let t = t + backward_result.residual[i] * forward_estimate[i] * alpha;
            t + backward_result.estimate[i] * forward_residual[i] * alpha })
    }
}
impl Drop for HipBackwardPushSolver { fn drop(&mut self) { unsafe { sl_push_graph_destroy(self.graph) } } }

/// BidirectionalPushSolver (src/solver/backward_push.rs:337-410) over the two device-resident solvers above: built once, not per query
/// as the reference rebuilds them (`self.graph.clone()` per call, :360-367); the degrees of the heuristic come from the host PushGraph.
pub struct HipBidirectionalPushSolver { graph: PushGraph, forward: HipForwardPushSolver, backward: HipBackwardPushSolver }
impl HipBidirectionalPushSolver {
    pub fn new(graph: PushGraph, forward_config: ForwardPushConfig, backward_config: BackwardPushConfig) -> Result<Self> {   // :346-357
        let (forward, backward) = (HipForwardPushSolver::new(&graph, forward_config)?, HipBackwardPushSolver::new(&graph, backward_config)?);
        Ok(Self { graph, forward, backward })
    }
    pub fn solve_bidirectional(&self, source: usize, target: usize) -> f64 {                                           // :359-377
        let (f, b) = (self.forward.solve_single_source(source), self.backward.solve_single_target(target));
        self.backward.combine_with_forward(&b, &f.estimate, &f.residual)
    }
    pub fn adaptive_solve(&self, source: usize, target: usize) -> f64 {                                                // :380-410
        let n = self.graph.num_nodes();
        if source >= n || target >= n { return 0.0; }
        let (out_s, in_t) = (self.graph.out_degree(source), self.graph.in_degree(target));
        if out_s > in_t * 2.0 { self.backward.query_transition_probability(source, target) }
        else if in_t > out_s * 2.0 { self.forward.query_single_entry(source, target) }
        else { self.solve_bidirectional(source, target) }
    }
}

// ---- conjugate gradient behind the same SpMV: OptimizedConjugateGradientSolver (src/optimized_solver.rs:167-350) ----------------------
#[repr(C)] #[derive(Default)]
pub struct SlCgOptions { tolerance: f64, max_iterations: u64, order: i32, mem: i32 }
#[repr(C)] #[derive(Default)]
pub struct SlCgResult { iterations: u64, matvec_count: u64, residual_norm: f64, total_time_ms: f64, device_time_ms: f64, converged: i32, reserved: i32 }
#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_matrix_create_csr(n_rows: u64, n_cols: u64, nnz: u64, row_ptr: *const u32, col_idx: *const u32, values: *const f64, mem: c_int,
                            row_offset: u64, flags: u32, out: *mut *mut SlMatrix) -> c_int;
    fn sl_cg_options_default(o: *mut SlCgOptions);
    fn sl_cg_solve(m: *const SlMatrix, b: *const f64, opts: *const SlCgOptions, x_out: *mut f64, result: *mut SlCgResult) -> c_int;
}
use crate::matrix::sparse::CSRStorage;
use crate::optimized_solver::{OptimizedSolverConfig, OptimizedSolverResult, OptimizedSolverStats, OptimizedSparseMatrix};

/// Same constructor, `solve`, `get_last_iteration_count` and `solve_with_callback` as the reference's solver (:174-350); the CSR arrays of
/// the matrix are adopted as they are (`sl_matrix_create_csr`), the loop — x0 = 0, r = p = b, stop at r.r <= tol^2, break at |p.Ap| < 1e-16
/// (:209-263) — runs on the device with the SpMV in the reference's summation order.  `OptimizedSparseMatrix` keeps its `CSRStorage`
/// private (:17-21): the one line the crate gains is `pub(crate) fn storage(&self) -> &CSRStorage { &self.storage }` next to `nnz()` (:64).
pub struct HipConjugateGradientSolver { config: OptimizedSolverConfig, stats: OptimizedSolverStats }
impl HipConjugateGradientSolver {
    pub fn new(config: OptimizedSolverConfig) -> Self { Self { config, stats: OptimizedSolverStats::default() } }                 // :174-179
    pub fn solve(&mut self, matrix: &OptimizedSparseMatrix, b: &[Precision]) -> core::result::Result<OptimizedSolverResult, String> {   // :182-295
        let (rows, cols) = matrix.dimensions();
        self.solve_csr(matrix.storage(), rows, cols, b)
    }
    pub fn solve_csr(&mut self, a: &CSRStorage, rows: usize, cols: usize, b: &[Precision]) -> core::result::Result<OptimizedSolverResult, String> {
        if rows != cols { return Err("Matrix must be square".to_string()); }                                                        // :187-190
        if b.len() != rows { return Err("Right-hand side vector length must match matrix size".to_string()); }                      // :191-193
        self.stats = OptimizedSolverStats::default();
        let mut h: *mut SlMatrix = core::ptr::null_mut();
        let st = unsafe { sl_matrix_create_csr(rows as u64, cols as u64, a.values.len() as u64, a.row_ptr.as_ptr(), a.col_indices.as_ptr(),
                                               a.values.as_ptr(), 0 /* SL_MEM_HOST */, 0, 0, &mut h) };
        if st != 0 { return Err(unsafe { std::ffi::CStr::from_ptr(sl_last_error_message()) }.to_string_lossy().into_owned()); }
        let mut o = SlCgOptions::default();
        unsafe { sl_cg_options_default(&mut o) };
        o.tolerance = self.config.tolerance; o.max_iterations = self.config.max_iterations as u64;
        let (mut x, mut r) = (vec![0.0; rows], SlCgResult::default());
        let st = unsafe { sl_cg_solve(h, b.as_ptr(), &o, x.as_mut_ptr(), &mut r) };
        unsafe { sl_matrix_destroy(h) };
        if st != 0 && st != 3 /* CONVERGENCE_FAILURE: the reference returns Ok(converged = false), :286-293 */ {
            return Err(unsafe { std::ffi::CStr::from_ptr(sl_last_error_message()) }.to_string_lossy().into_owned());
        }
        self.stats.matvec_count = r.matvec_count as usize;                     // dot_product_count / axpy_count stay 0 there too: the loop inlines them
        self.stats.total_flops = self.stats.matvec_count * a.values.len() * 2 + r.iterations as usize * rows * 6;                  // :276-277
        if r.total_time_ms > 0.0 {                                                                                                  // :279-283
            self.stats.average_bandwidth_gbs = (self.stats.total_flops * 8) as f64 / 1e9 / (r.total_time_ms / 1000.0);
            self.stats.average_gflops = self.stats.total_flops as f64 / (r.total_time_ms * 1e6);
        }
        Ok(OptimizedSolverResult { solution: x, residual_norm: r.residual_norm, iterations: r.iterations as usize, converged: r.converged != 0,
                                   computation_time_ms: r.total_time_ms, performance_stats: self.stats.clone() })
    }
    pub fn get_last_iteration_count(&self) -> usize { self.stats.matvec_count }                                                    // :331-333
    pub fn solve_with_callback<F: FnMut(&OptimizedSolverStats)>(&mut self, matrix: &OptimizedSparseMatrix, b: &[Precision], _chunk_size: usize,
                                                                 mut callback: F) -> core::result::Result<OptimizedSolverResult, String> {   // :336-350
        let r = self.solve(matrix, b)?;
        callback(&r.performance_stats);                                        // (the reference never calls it; once, at the end, here)
        Ok(r)
    }
}

// ---- the crate's public free functions of src/simd_ops.rs (re-exported from lib.rs:83-87), same signatures, on the device ----------
#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_dot(n: u64, x: *const f64, y: *const f64, out: *mut f64, mem: c_int) -> c_int;
    fn sl_axpy(n: u64, alpha: f64, x: *const f64, y: *mut f64, mem: c_int) -> c_int;
}
/// simd_ops::matrix_vector_multiply_simd (:20-88): rows of >= 8 entries in the 4-lane order, shorter ones sequentially (SL_ORDER_SIMD4 = 1).
/// A one-shot call pays the upload and the layout build; callers that multiply by the same matrix again keep a `HipMatrix`.
pub fn matrix_vector_multiply_simd(values: &[Precision], col_indices: &[u32], row_ptr: &[u32], x: &[Precision], y: &mut [Precision]) {
    spmv_once(values, col_indices, row_ptr, x, y, 1 /* SL_ORDER_SIMD4 */)
}
/// simd_ops::parallel_matrix_vector_multiply (:201-239): row chunks over threads, each row summed sequentially (SL_ORDER_CSR_SEQUENTIAL = 0);
/// `num_threads` has no meaning on the device
pub fn parallel_matrix_vector_multiply(values: &[Precision], col_indices: &[u32], row_ptr: &[u32], x: &[Precision], y: &mut [Precision], _num_threads: Option<usize>) {
    spmv_once(values, col_indices, row_ptr, x, y, 0 /* SL_ORDER_CSR_SEQUENTIAL */)
}
fn spmv_once(values: &[Precision], col_indices: &[u32], row_ptr: &[u32], x: &[Precision], y: &mut [Precision], order: c_int) {
    let rows = row_ptr.len().saturating_sub(1);
    assert_eq!(y.len(), rows);
    let mut h: *mut SlMatrix = core::ptr::null_mut();
    let st = unsafe { sl_matrix_create_csr(rows as u64, x.len() as u64, values.len() as u64, row_ptr.as_ptr(), col_indices.as_ptr(), values.as_ptr(),
                                           0 /* SL_MEM_HOST */, 0, 0, &mut h) };
    assert_eq!(st, 0, "sl_matrix_create_csr failed");                    // the reference indexes out of bounds (a panic) on the same inputs
    let st = unsafe { sl_spmv(h, x.as_ptr(), y.as_mut_ptr(), order, 0) };
    unsafe { sl_matrix_destroy(h) };
    assert_eq!(st, 0, "sl_spmv failed");
}
/// simd_ops::dot_product_simd (:116-147) / axpy_simd (:158-189); the device's dot is a fixed tree (the reference's 4-lane sum to rounding)
pub fn dot_product_simd(x: &[Precision], y: &[Precision]) -> Precision {
    assert_eq!(x.len(), y.len());
    let mut out = 0.0;
    let st = unsafe { sl_dot(x.len() as u64, x.as_ptr(), y.as_ptr(), &mut out, 0) };
    assert_eq!(st, 0, "sl_dot failed");
    out
}
pub fn axpy_simd(alpha: Precision, x: &[Precision], y: &mut [Precision]) {
    assert_eq!(x.len(), y.len());
    let st = unsafe { sl_axpy(x.len() as u64, alpha, x.as_ptr(), y.as_mut_ptr(), 0) };
    assert_eq!(st, 0, "sl_axpy failed");
}

// ---- one process per GPU: the communicator and the partitioned NeumannState (src/simd_ops.rs:201-239 is the crate's precedent: row
// chunks behind one call).  A host that starts N processes drives N GPUs through these alone. -------------------------------------------
#[repr(C)] pub struct SlComm { _private: [u8; 0] }
#[repr(C)] #[derive(Default)]
pub struct SlCommInfoT { rank: i32, world: i32, device: i32, transport: i32, halo_allreduce: i32, ranks_joined: i32, failed: i32, reserved: i32 }
#[link(name = "sublinear_hip")]
extern "C" {
    fn sl_comm_create(rank: c_int, world: c_int, rendezvous_name: *const c_char, out: *mut *mut SlComm) -> c_int;
    fn sl_comm_destroy(c: *mut SlComm);
    fn sl_comm_barrier(c: *mut SlComm) -> c_int;
    fn sl_comm_info(c: *const SlComm, info: *mut SlCommInfoT) -> c_int;
    fn sl_matrix_create_csr(n_rows: u64, n_cols: u64, nnz: u64, row_ptr: *const u32, col_idx: *const u32, values: *const f64,
                            mem: c_int, row_offset: u64, flags: u32, out: *mut *mut SlMatrix) -> c_int;
    fn sl_neumann_state_create_partitioned(c: *mut SlComm, local_rows: *const SlMatrix, b_local: *const f64, initial_guess_local: *const f64,
                                           opts: *const SlNeumannOptions, out: *mut *mut SlNeumannState) -> c_int;
    fn sl_neumann_state_run_steps(s: *mut SlNeumannState, steps: u64, last_norm2: *mut f64, elapsed_ms: *mut f32) -> c_int;
    fn sl_neumann_state_verify_exchange(s: *mut SlNeumannState, pieces_bad: *mut u64) -> c_int;
    fn sl_neumann_state_current_term(s: *const SlNeumannState, first_row: u64, count: u64, t_out: *mut f64, mem: c_int) -> c_int;
    fn sl_neumann_state_solution_rows(s: *const SlNeumannState, first_row: u64, count: u64, x_out: *mut f64, mem: c_int) -> c_int;
}

/// This rank's rows [lo, hi) of a system of n_global rows (CSR with GLOBAL column ids) as a NeumannState over the communicator: the
/// SolverAlgorithm calls above then act collectively (every rank makes the same calls in the same order); `extract_solution` returns
/// the rank's rows.
pub fn initialize_partitioned(comm: *mut SlComm, n_global: usize, lo: usize, row_ptr: &[u32], col_idx: &[u32], values: &[f64],
                              b_local: &[Precision], o: &SlNeumannOptions) -> Result<(*mut SlMatrix, *mut SlNeumannState)> {
    let (mut m, mut s) = (core::ptr::null_mut(), core::ptr::null_mut());
    let rows = row_ptr.len() - 1;
    let st = unsafe { sl_matrix_create_csr(rows as u64, n_global as u64, values.len() as u64, row_ptr.as_ptr(), col_idx.as_ptr(), values.as_ptr(),
                                           0 /* SL_MEM_HOST */, lo as u64, 0, &mut m) };
    if st != 0 { return Err(to_error(st, 0, f64::INFINITY, o.tolerance)); }
    let st = unsafe { sl_neumann_state_create_partitioned(comm, m, b_local.as_ptr(), core::ptr::null(), o, &mut s) };
    if st != 0 { unsafe { sl_matrix_destroy(m) }; return Err(to_error(st, 0, f64::INFINITY, o.tolerance)); }
    Ok((m, s))
}
