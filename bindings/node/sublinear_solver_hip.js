'use strict';
/**
 * Host side of the reference's SHIPPED surface, in its own language: `SublinearSolver` of src/core/solver.ts
 * (npm `sublinear-time-solver`, dist/core/solver.js) over libsublinear_hip.so (MI355X) through the N-API addon in this
 * directory.  Same constructor config, method names, argument meaning, result field names, error class and error codes
 * (src/core/types.ts:28-46, 79-108), so `require('./bindings/node')` replaces `require('sublinear-time-solver')` for
 * this path:
 *
 *     const { SublinearSolver } = require('./bindings/node');
 *     const s = new SublinearSolver({ method: 'neumann', epsilon: 1e-8, maxIterations: 1000 });
 *     const { solution, iterations, residual, converged } = await s.solve(matrix, vector);
 *     const { estimate } = await s.estimateEntry(matrix, vector, { row: 3, column: 0, epsilon: 1e-6, confidence: 0.95, method: 'neumann' });
 *
 * What runs where: validation and the JSON matrix model (core/matrix.ts:11-55) stay in JS; every loop over matrix
 * entries of solve / estimateEntry / computePageRank runs on the GPU.  Deliberate differences from the reference
 * (SURVEY.md §0.3): `neumann` sums the exact series for A = D - R (the shipped JS flips the sign of R,
 * solver.ts:157-163); the push methods run the synchronous thresholded push with theta = epsilon; there is no CPU path —
 * without a GPU every call rejects with a DeviceError.
 */
const path = require('path');
const native = require(path.join(__dirname, 'sublinear_hip.node'));

const ErrorCodes = Object.freeze({                       // core/types.ts:99-108
  NOT_DIAGONALLY_DOMINANT: 'E001', CONVERGENCE_FAILED: 'E002', INVALID_MATRIX: 'E003', TIMEOUT: 'E004',
  INVALID_DIMENSIONS: 'E005', NUMERICAL_INSTABILITY: 'E006', MEMORY_LIMIT_EXCEEDED: 'E007', INVALID_PARAMETERS: 'E008'
});

class SolverError extends Error {                        // core/types.ts:88-97
  constructor(message, code, details) {
    super(message);
    this.name = 'SolverError';
    this.code = code;
    this.details = details;
  }
}

// sl_status (include/sublinear_hip.h) -> reference error code
const STATUS_TO_CODE = { 1: 'E001', 2: 'E006', 3: 'E002', 4: 'E008', 5: 'E005', 6: 'E003', 7: 'E007', 8: 'E003', 9: 'E003', 10: 'E002' };

function rethrow(e) {
  if (e instanceof SolverError) throw e;
  if (e && typeof e.status === 'number') throw new SolverError(e.message, STATUS_TO_CODE[e.status] || 'E002', { status: e.status, kind: e.kind });
  throw e;
}

// ---- the JSON matrix model (core/types.ts:6-22; the CJS CLI nests COO arrays under `data`, bin/cli.js:484-490) ----
function cooArrays(matrix) {
  const src = (matrix.data && !Array.isArray(matrix.data) && typeof matrix.data === 'object') ? matrix.data : matrix;
  return { values: src.values, rowIndices: src.rowIndices, colIndices: src.colIndices };
}

const MatrixOperations = {
  /** core/matrix.ts:11-55 */
  validateMatrix(matrix) {
    if (!matrix) throw new SolverError('Matrix is required', ErrorCodes.INVALID_MATRIX);
    if (!(matrix.rows > 0) || !(matrix.cols > 0)) throw new SolverError('Matrix dimensions must be positive', ErrorCodes.INVALID_DIMENSIONS);
    if (matrix.format === 'dense') {
      if (!Array.isArray(matrix.data) || matrix.data.length !== matrix.rows) throw new SolverError('Dense matrix data must be array of rows', ErrorCodes.INVALID_MATRIX);
      for (let i = 0; i < matrix.rows; i++) {
        if (!matrix.data[i] || matrix.data[i].length !== matrix.cols) throw new SolverError(`Row ${i} has invalid length`, ErrorCodes.INVALID_MATRIX);
      }
    } else if (matrix.format === 'coo') {
      const { values, rowIndices, colIndices } = cooArrays(matrix);
      if (!values || !rowIndices || !colIndices) throw new SolverError('COO matrix must have values, rowIndices, and colIndices arrays', ErrorCodes.INVALID_MATRIX);
      if (values.length !== rowIndices.length || values.length !== colIndices.length) throw new SolverError('COO matrix arrays must have same length', ErrorCodes.INVALID_MATRIX);
      for (let k = 0; k < values.length; k++) {
        if (!(rowIndices[k] >= 0 && rowIndices[k] < matrix.rows)) throw new SolverError(`Invalid row index ${rowIndices[k]}`, ErrorCodes.INVALID_MATRIX);
        if (!(colIndices[k] >= 0 && colIndices[k] < matrix.cols)) throw new SolverError(`Invalid column index ${colIndices[k]}`, ErrorCodes.INVALID_MATRIX);
      }
    } else {
      throw new SolverError(`Unsupported matrix format: ${matrix.format}`, ErrorCodes.INVALID_MATRIX);
    }
  },

  /** (row, col, value) typed arrays; dense zeros are not emitted (SparseMatrix::from_dense, matrix/mod.rs:204-223) */
  toTriplets(matrix) {
    if (matrix.format === 'dense') {
      let nnz = 0;
      for (let i = 0; i < matrix.rows; i++) for (let j = 0; j < matrix.cols; j++) if (matrix.data[i][j] !== 0) nnz++;
      const r = new Float64Array(nnz), c = new Float64Array(nnz), v = new Float64Array(nnz);
      let k = 0;
      for (let i = 0; i < matrix.rows; i++) {
        const row = matrix.data[i];
        for (let j = 0; j < matrix.cols; j++) if (row[j] !== 0) { r[k] = i; c[k] = j; v[k] = row[j]; k++; }
      }
      return { r, c, v };
    }
    const { values, rowIndices, colIndices } = cooArrays(matrix);
    return { r: Float64Array.from(rowIndices), c: Float64Array.from(colIndices), v: Float64Array.from(values) };
  },

  /** core/matrix.ts:327-351 with checkDiagonalDominance (:211-258), isSymmetric (:263-296), calculateSparsity (:301-322) — with the
   *  reference's own reading of the matrix, so that every field (the bits of dominanceStrength included) is what its TypeScript returns
   *  (golden G14, tests/golden/reference_ts_analyze.json): the diagonal and the symmetry test read entries through getEntry, i.e. the FIRST
   *  stored match of a duplicated COO entry (:105-112); the off-diagonal row / column sums add |value| over ALL stored entries in storage
   *  order (getRowSum / getColumnSum :143-206); the loop over the rows stops at the first zero diagonal (:228-234).  O(entries), where the
   *  reference walks the whole array per row. */
  analyzeMatrix(matrix) {
    MatrixOperations.validateMatrix(matrix);
    const n = matrix.rows, square = matrix.rows === matrix.cols, dense = matrix.format === 'dense';
    let isRow = false, isCol = false, strength = 0, symmetric = false, stored = 0;
    const first = new Map();                                   // (row, col) -> the value getEntry returns
    const rowOff = new Float64Array(n), colOff = new Float64Array(square ? n : 0);
    const visit = (i, j, val) => {
      if (!square) return;
      const key = i * n + j;
      if (!first.has(key)) first.set(key, val);
      if (i !== j) { rowOff[i] += Math.abs(val); colOff[j] += Math.abs(val); }
    };
    if (dense) {
      for (let i = 0; i < matrix.rows; i++) for (let j = 0; j < matrix.cols; j++) { const val = matrix.data[i][j]; if (Math.abs(val) > 1e-15) stored++; if (val !== 0) visit(i, j, val); }
    } else {
      const { values, rowIndices, colIndices } = cooArrays(matrix);
      stored = values.length;
      for (let k = 0; k < values.length; k++) visit(rowIndices[k], colIndices[k], values[k]);
    }
    if (square) {
      symmetric = true;
      for (const [key, val] of first) {
        const i = Math.floor(key / n), j = key - i * n;
        if (i !== j && Math.abs(val - (first.has(j * n + i) ? first.get(j * n + i) : 0)) > 1e-10) { symmetric = false; break; }
      }
      let zeroDiag = false, minR = Infinity, minC = Infinity;
      isRow = true; isCol = true;
      for (let i = 0; i < n; i++) {
        const a = Math.abs(first.has(i * n + i) ? first.get(i * n + i) : 0);
        if (a === 0) { zeroDiag = true; break; }
        if (a - rowOff[i] < 0) isRow = false; else minR = Math.min(minR, (a - rowOff[i]) / a);
        if (a - colOff[i] < 0) isCol = false; else minC = Math.min(minC, (a - colOff[i]) / a);
      }
      if (zeroDiag) { isRow = false; isCol = false; minR = 0; minC = 0; }
      strength = Math.max(isRow ? minR : 0, isCol ? minC : 0);
    }
    return {
      isDiagonallyDominant: isRow || isCol, dominanceType: isRow ? 'row' : (isCol ? 'column' : 'none'), dominanceStrength: strength,
      isSymmetric: symmetric, sparsity: 1 - stored / (matrix.rows * matrix.cols), size: { rows: matrix.rows, cols: matrix.cols }
    };
  }
};

function validatePositiveNumber(x, name) {
  if (typeof x !== 'number' || !(x > 0) || !isFinite(x)) throw new SolverError(`${name} must be a positive number`, ErrorCodes.INVALID_PARAMETERS);
}

/** device matrix for the duration of fn */
function withDeviceMatrix(matrix, withTranspose, fn) {
  const { r, c, v } = MatrixOperations.toTriplets(matrix);
  let h;
  try { h = native.createMatrix(matrix.rows, matrix.cols, r, c, v, withTranspose); } catch (e) { rethrow(e); }
  try { return fn(h); } catch (e) { return rethrow(e); } finally { native.destroyMatrix(h); }
}

/** The system computePageRank assembles (core/solver.ts:679-698), sparse, with the reference's own arithmetic so that it carries its bits
 *  (golden G13): out_j = the row's entries added left to right in column order; S[i][j] = [i == j] - damping * (adj[j][i] / out_j) for
 *  out_j > 0 (a dangling node's column stays the identity's); MatrixOperations.getEntry reads the FIRST stored match of a duplicated COO
 *  entry (matrix.ts:105-112), so later duplicates are dropped, a stored 0 included.  Returns a COO matrix, entries by column then diagonal. */
function pageRankSystem(adjacency, damping) {
  const config = { damping };
  const n = adjacency.rows;
  const { r, c, v } = MatrixOperations.toTriplets(adjacency);
  const order = Array.from(v.keys()).sort((a, b) => (r[a] - r[b]) || (c[a] - c[b]) || (a - b));
  const er = [], ec = [], ev = [];
  for (let q = 0; q < order.length; q++) {
    const k = order[q];
    if (q > 0 && r[order[q - 1]] === r[k] && c[order[q - 1]] === c[k]) continue;     // a later duplicate
    if (v[k] !== 0) { er.push(r[k]); ec.push(c[k]); ev.push(v[k]); }
  }
  const out = new Float64Array(n);
  for (let k = 0; k < ev.length; k++) out[er[k]] += ev[k];
  const diag = new Float64Array(n).fill(1);
  const sys = { rows: n, cols: n, format: 'coo', values: [], rowIndices: [], colIndices: [] };
  for (let k = 0; k < ev.length; k++) {
    const j = er[k], i = ec[k];
    if (!(out[j] > 0)) continue;                                                     // a dangling node's column stays the identity's
    const prob = ev[k] / out[j];
    if (i === j) diag[i] = 1 - config.damping * prob;
    else { sys.rowIndices.push(i); sys.colIndices.push(j); sys.values.push(-(config.damping * prob)); }
  }
  for (let i = 0; i < n; i++) if (diag[i] !== 0) { sys.rowIndices.push(i); sys.colIndices.push(i); sys.values.push(diag[i]); }
  return sys;
}

const WALK_STREAMS = { blocks: 0, reference: 1, serial: 1 };
const METHODS = ['neumann', 'random-walk', 'forward-push', 'backward-push', 'bidirectional'];

class SublinearSolver {
  /** core/solver.ts:36-56: { method, epsilon, maxIterations, timeout?, enableProgress?, seed? }
   *  + stream?: 'blocks' (default: one lane per walk, every walk its own block of the seed's stream) | 'reference' (the reference's ONE
   *  serial stream: random-walk results bit-identical to solver.ts for the same seed; include/sublinear_hip.h, sl_walk_stream) */
  constructor(config) {
    if (!config || METHODS.indexOf(config.method) < 0) throw new SolverError(`Unknown method: ${config && config.method}`, ErrorCodes.INVALID_PARAMETERS);
    validatePositiveNumber(config.epsilon, 'epsilon');
    if (!Number.isInteger(config.maxIterations) || config.maxIterations < 1 || config.maxIterations > 1e6) {
      throw new SolverError('maxIterations must be an integer between 1 and 1000000', ErrorCodes.INVALID_PARAMETERS);
    }
    if (config.timeout) validatePositiveNumber(config.timeout, 'timeout');
    if (config.stream !== undefined && WALK_STREAMS[config.stream] === undefined) {
      throw new SolverError(`Unknown random-walk stream: ${config.stream}`, ErrorCodes.INVALID_PARAMETERS);
    }
    this.config = Object.assign({}, config);
    this.walkStream = WALK_STREAMS[config.stream === undefined ? 'blocks' : config.stream];
  }

  /** core/solver.ts:58-111 -> { solution, iterations, residual, converged, method, computeTime, memoryUsed } */
  async solve(matrix, vector, progressCallback) {
    MatrixOperations.validateMatrix(matrix);
    if (vector.length !== matrix.cols) {
      throw new SolverError(`Vector length ${vector.length} does not match matrix columns ${matrix.cols}`, ErrorCodes.INVALID_DIMENSIONS);
    }
    const analysis = MatrixOperations.analyzeMatrix(matrix);
    if (!analysis.isDiagonallyDominant) throw new SolverError('Matrix is not diagonally dominant', ErrorCodes.NOT_DIAGONALLY_DOMINANT, { analysis });
    const t0 = process.hrtime.bigint();
    const method = this.config.method;
    // Neumann needs ROW dominance (neumann.rs:139-170); a column-dominant system goes through the push, which does not
    const push = method !== 'neumann' || analysis.dominanceType !== 'row';
    const b = Float64Array.from(vector);
    const out = withDeviceMatrix(matrix, push || method === 'random-walk', (h) => {
      if (method === 'random-walk') {
        // solveRandomWalk (solver.ts:278-357): max(100, ceil(1 / eps^2)) walks per coordinate, a stream per walk from config.seed
        // (the reference seeds ONE stream with `seed || Date.now()`); a residual that misses epsilon throws, as there (:335-341)
        const seed = (this.config.seed !== undefined ? this.config.seed : Date.now()) >>> 0;
        const w = native.randomWalkSolve(h, b, this.config.epsilon, seed, this.walkStream);
        if (!w.converged) {
          throw new SolverError('Random walk sampling failed to achieve desired accuracy', ErrorCodes.CONVERGENCE_FAILED,
                                { finalResidual: w.residualNorm, variance: Math.sqrt(w.totalVariance) });
        }
        return { solution: Array.from(w.solution), iterations: w.iterations, residual: w.residualNorm, converged: true, memoryUsed: w.deviceBytes };
      }
      if (!push) {
        const r = native.neumannSolve(h, b, { tolerance: this.config.epsilon, maxIterations: this.config.maxIterations,
                                              maxTerms: this.config.maxIterations, seriesTolerance: this.config.epsilon });
        return { solution: Array.from(r.solution), iterations: r.iterations, residual: r.residualNorm, converged: r.converged, memoryUsed: r.deviceBytes };
      }
      // solveForwardPush (solver.ts:437-522): up to 4096 rows in the reference's own order — one Gauss-Southwell push per iteration,
      // `iterations` = pushes as the reference reports them (ConvergenceFailure after maxIterations pushes comes back as an
      // exception of the native call); larger systems through the data-parallel thresholded push (`iterations` = rounds)
      if (matrix.rows <= 4096 && this.config.pushOrder !== 'synchronous') {
        const g = native.forwardPushSouthwell(h, b, { epsilon: this.config.epsilon, maxIterations: this.config.maxIterations });
        return { solution: Array.from(g.solution), iterations: g.iterations, residual: g.residualNorm, converged: true, memoryUsed: g.deviceBytes };
      }
      const r = native.pushSolve(h, b, { theta: this.config.epsilon, maxRounds: this.config.maxIterations });
      if (!r.converged) throw new SolverError(`Forward push failed to converge after ${this.config.maxIterations} iterations`, ErrorCodes.CONVERGENCE_FAILED);
      return { solution: Array.from(r.solution), iterations: r.rounds, residual: r.residualNorm, converged: true, memoryUsed: r.deviceBytes };
    });
    out.method = method;
    out.computeTime = Number(process.hrtime.bigint() - t0) / 1e6;
    if (this.config.timeout && out.computeTime > this.config.timeout) {      // TimeoutController.checkTimeout, core/utils.ts:319-325 (measured per solve)
      throw new SolverError(`Operation timed out after ${this.config.timeout}ms`, ErrorCodes.TIMEOUT);
    }
    if (progressCallback) progressCallback({ iteration: out.iterations, residual: out.residual, elapsed: out.computeTime });
    return out;
  }

  /** core/solver.ts:550-659: x_row = (A^-1 vector)_row -> { estimate, variance, confidence } */
  async estimateEntry(matrix, vector, config) {
    MatrixOperations.validateMatrix(matrix);
    if (!(config.row >= 0 && config.row < matrix.rows)) {
      throw new SolverError(`Row index ${config.row} out of bounds. Matrix has ${matrix.rows} rows (valid range: 0-${matrix.rows - 1})`, ErrorCodes.INVALID_PARAMETERS);
    }
    if (!(config.column >= 0 && config.column < matrix.cols)) {
      throw new SolverError(`Column index ${config.column} out of bounds. Matrix has ${matrix.cols} columns`, ErrorCodes.INVALID_PARAMETERS);
    }
    if (vector.length !== matrix.rows) throw new SolverError(`Vector length ${vector.length} does not match matrix rows ${matrix.rows}`, ErrorCodes.INVALID_DIMENSIONS);
    const eps = config.epsilon !== undefined ? config.epsilon : this.config.epsilon;
    // config.entryOf = 'solution' (default): x_row = (A^-1 vector)_row, what the reference's random-walk branch estimates; 'inverse':
    // (A^-1)[row][column] with `vector` ignored — what the reference's non-random-walk branch computes (A x = e_column, x[row]: solver.ts:603-620)
    if (config.entryOf !== undefined && config.entryOf !== 'solution' && config.entryOf !== 'inverse') {
      throw new SolverError(`Unknown entryOf: ${config.entryOf}`, ErrorCodes.INVALID_PARAMETERS);
    }
    const walk = config.method === 'random-walk' || config.method === 'monte-carlo';
    const b = Float64Array.from(vector);
    if (config.entryOf === 'inverse' && !walk) { b.fill(0); b[config.column] = 1; }
    return withDeviceMatrix(matrix, true, (h) => {
      if (walk) {        // solver.ts:585-601, 630-648
        const seed = (this.config.seed !== undefined ? this.config.seed : 0) >>> 0;
        const r = native.estimateEntryRandomWalk(h, b, config.row, eps, seed, this.walkStream);
        return { estimate: r.estimate, variance: r.variance, confidence: config.confidence, numSamples: r.numSamples };
      }
      const r = native.estimateEntry(h, b, config.row, eps * 1e-2, this.config.maxIterations * 100);
      return { estimate: r.estimate, variance: 0, confidence: r.converged ? 1 : 0.5, residualL1: r.residualL1 };
    });
  }

  /** core/solver.ts:664-722: (I - damping P^T) x = (1 - damping)/n (or `personalized`), assembled sparse instead of dense */
  async computePageRank(adjacency, config) {
    MatrixOperations.validateMatrix(adjacency);
    if (!(config.damping >= 0 && config.damping <= 1)) throw new SolverError('damping must be between 0 and 1', ErrorCodes.INVALID_PARAMETERS);
    validatePositiveNumber(config.epsilon, 'epsilon');
    if (adjacency.rows !== adjacency.cols) throw new SolverError('Adjacency matrix must be square', ErrorCodes.INVALID_DIMENSIONS);
    const n = adjacency.rows;
    const sys = pageRankSystem(adjacency, config.damping);
    const rhs = config.personalized || new Array(n).fill(1 * ((1 - config.damping) / n));
    const solver = new SublinearSolver({ method: this.config.method, epsilon: config.epsilon, maxIterations: config.maxIterations, timeout: this.config.timeout });
    return (await solver.solve(sys, rhs)).solution;
  }
}

module.exports = { SublinearSolver, SolverError, ErrorCodes, MatrixOperations, pageRankSystem, native };
