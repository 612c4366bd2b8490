'use strict';
/* The reference's shipped surface (src/core/solver.ts) in its own language, over the GPU library: node tests/js/surface_test.js [gpu]
 * Without `gpu`: host logic only (config / matrix validation, analyzeMatrix, the loud DeviceError).  With `gpu`: solves. */
const assert = require('assert');
const path = require('path');
const { SublinearSolver, SolverError, ErrorCodes, MatrixOperations, pageRankSystem, native } = require(path.join(__dirname, '..', '..', 'bindings', 'node'));

const onGpu = process.argv[2] === 'gpu';

function gauss(A, b) {                                   // dense reference solution (partial pivoting)
  const n = b.length, M = A.map((r, i) => r.concat([b[i]]));
  for (let c = 0; c < n; c++) {
    let p = c;
    for (let r = c + 1; r < n; r++) if (Math.abs(M[r][c]) > Math.abs(M[p][c])) p = r;
    [M[c], M[p]] = [M[p], M[c]];
    for (let r = c + 1; r < n; r++) { const f = M[r][c] / M[c][c]; for (let k = c; k <= n; k++) M[r][k] -= f * M[c][k]; }
  }
  const x = new Array(n).fill(0);
  for (let r = n - 1; r >= 0; r--) { let s = M[r][n]; for (let k = r + 1; k < n; k++) s -= M[r][k] * x[k]; x[r] = s / M[r][r]; }
  return x;
}

function lcg(seed) { let s = seed >>> 0; return () => { s = (Math.imul(1664525, s) + 1013904223) >>> 0; return s / 4294967296; }; }   // core/utils.ts:161-168

function randomDD(n, seed) {
  const rnd = lcg(seed), A = [];
  for (let i = 0; i < n; i++) {
    const row = new Array(n).fill(0);
    let s = 0;
    for (let j = 0; j < n; j++) if (j !== i && rnd() < 0.3) { row[j] = (rnd() - 0.5) * 2; s += Math.abs(row[j]); }
    row[i] = 2 * s + 1;
    A.push(row);
  }
  return A;
}

async function rejects(p, code, re) {
  try { await p; } catch (e) {
    assert(e instanceof SolverError, `expected SolverError, got ${e}`);
    if (code) assert.strictEqual(e.code, code, e.message);
    if (re) assert(re.test(e.message), e.message);
    return e;
  }
  assert.fail('expected a rejection');
}

(async () => {
  // ---- host logic (no GPU needed) ----
  assert.throws(() => new SublinearSolver({ method: 'bogus', epsilon: 1e-6, maxIterations: 10 }), SolverError);
  assert.throws(() => new SublinearSolver({ method: 'neumann', epsilon: -1, maxIterations: 10 }), /epsilon/);
  assert.throws(() => new SublinearSolver({ method: 'neumann', epsilon: 1e-6, maxIterations: 0 }), /maxIterations/);
  assert.throws(() => MatrixOperations.validateMatrix({ rows: 2, cols: 2, format: 'dense', data: [[1, 2]] }), /array of rows/);
  assert.throws(() => MatrixOperations.validateMatrix({ rows: 2, cols: 2, format: 'dense', data: [[1, 2], [3]] }), /Row 1 has invalid length/);
  assert.throws(() => MatrixOperations.validateMatrix({ rows: 2, cols: 2, format: 'coo', values: [1], rowIndices: [0, 1], colIndices: [0] }), /same length/);
  assert.throws(() => MatrixOperations.validateMatrix({ rows: 2, cols: 2, format: 'coo', values: [1], rowIndices: [2], colIndices: [0] }), /Invalid row index 2/);
  assert.throws(() => MatrixOperations.validateMatrix({ rows: 2, cols: 2, format: 'csr' }), /Unsupported matrix format/);
  const dense = { rows: 3, cols: 3, format: 'dense', data: [[4, 1, 0], [-1, 5, 2], [0, 0.5, 3]] };
  const a = MatrixOperations.analyzeMatrix(dense);
  assert.deepStrictEqual([a.isDiagonallyDominant, a.dominanceType, a.isSymmetric], [true, 'row', false]);
  assert(Math.abs(a.sparsity - 2 / 9) < 1e-15 && Math.abs(a.dominanceStrength - (5 - 3) / 5) < 1e-15);
  const notDD = { rows: 2, cols: 2, format: 'dense', data: [[1, 3], [2, 1]] };
  assert.strictEqual(MatrixOperations.analyzeMatrix(notDD).isDiagonallyDominant, false);
  const goldenPageRank = JSON.parse(require('fs').readFileSync(path.join(__dirname, '..', 'golden', 'reference_ts_pagerank_js.json'), 'utf8'));
  {   // G13: the system computePageRank assembles, entry for entry and bit for bit what the reference's own TypeScript assembled
      // (tests/golden/make_golden_ts_pagerank.py): weights, self loops on the diagonal, dangling nodes, left-to-right out-degrees
    assert(goldenPageRank.length === 3);
    for (const g of goldenPageRank) {
      const sys = pageRankSystem(g.adjacency, g.damping);
      const order = Array.from(sys.values.keys()).sort((a, b) => (sys.rowIndices[a] - sys.rowIndices[b]) || (sys.colIndices[a] - sys.colIndices[b]));
      assert.deepStrictEqual(order.map(k => sys.rowIndices[k]), g.system.rows, g.name);
      assert.deepStrictEqual(order.map(k => sys.colIndices[k]), g.system.cols, g.name);
      assert.deepStrictEqual(order.map(k => sys.values[k]), g.system.values, g.name);
    }
  }
  {   // G14: analyzeMatrix field by field (the bits of dominanceStrength included) against what the reference's own TypeScript returned for
      // these matrices (tests/golden/reference_ts_analyze.json, make_golden_ts_analyze.py): duplicated COO entries, storage order, zero diagonals
    const golden = JSON.parse(require('fs').readFileSync(path.join(__dirname, '..', 'golden', 'reference_ts_analyze.json'), 'utf8'));
    assert(golden.length >= 22);
    const codeOf = { INVALID_MATRIX: ErrorCodes.INVALID_MATRIX, INVALID_DIMENSIONS: ErrorCodes.INVALID_DIMENSIONS };
    for (const g of golden) {
      if (!g.invalid) { assert.deepStrictEqual(MatrixOperations.analyzeMatrix(g.matrix), g.analysis, g.name); continue; }
      let err = null;                                             // validateMatrix (core/matrix.ts:11-55): the reference's message and code
      try { MatrixOperations.analyzeMatrix(g.matrix); } catch (e) { err = e; }
      assert(err instanceof SolverError && err.message === g.error.message && err.code === codeOf[g.error.code], `${g.name}: ${err && err.message} / ${err && err.code}`);
    }
  }
  const s = new SublinearSolver({ method: 'neumann', epsilon: 1e-10, maxIterations: 1000 });
  await rejects(s.solve(notDD, [1, 1]), ErrorCodes.NOT_DIAGONALLY_DOMINANT, /not diagonally dominant/);
  await rejects(s.solve(dense, [1, 2]), ErrorCodes.INVALID_DIMENSIONS, /does not match matrix columns/);
  await rejects(s.estimateEntry(dense, [1, 2, 3], { row: 7, column: 0, epsilon: 1e-6, confidence: 0.95, method: 'neumann' }), ErrorCodes.INVALID_PARAMETERS, /Row index 7 out of bounds/);
  for (const f of ['createMatrix', 'neumannSolve', 'pushSolve', 'estimateEntry', 'estimateEntryRandomWalk', 'randomWalkSolve', 'cgSolve']) assert.strictEqual(typeof native[f], 'function');

  if (!onGpu) {
    if (native.deviceCount() === 0) {                       // no CPU fallback: the call must fail loudly
      const e = await rejects(s.solve(dense, [1, 2, -1]), null, /DeviceError/);
      assert.strictEqual(e.details.kind, 'DeviceError');
    }
    console.log('host logic ok');
    return;
  }

  // ---- on the GPU ----
  {   // a destroyed handle is dead, not dangling: second destroy is a no-op, any later use throws
    const h = native.createMatrix(2, 2, new Uint32Array([0, 1]), new Uint32Array([0, 1]), new Float64Array([2, 3]), false);
    assert.strictEqual(native.matrixInfo(h).nnz, 2);
    native.destroyMatrix(h);
    native.destroyMatrix(h);
    assert.throws(() => native.matrixInfo(h), /destroyed/);
    assert.throws(() => native.neumannSolve(h, new Float64Array([1, 1]), {}), /destroyed/);
  }
  const b3 = [1, 2, -1], x3 = gauss(dense.data, b3);
  for (const method of ['neumann', 'forward-push', 'backward-push', 'bidirectional']) {
    const r = await new SublinearSolver({ method, epsilon: 1e-12, maxIterations: 10000 }).solve(dense, b3);
    assert(r.converged && r.method === method && r.iterations > 0 && r.memoryUsed > 0 && r.computeTime >= 0);
    r.solution.forEach((v, i) => assert(Math.abs(v - x3[i]) < 1e-10, `${method} x[${i}]`));
  }
  await rejects(new SublinearSolver({ method: 'neumann', epsilon: 1e-12, maxIterations: 100, timeout: 1e-6 }).solve(dense, b3), ErrorCodes.TIMEOUT, /timed out after/);
  const n = 200, A = randomDD(n, 7), b = Array.from({ length: n }, (_, i) => Math.sin(i));
  const xr = gauss(A, b);
  const rows = [], cols = [], vals = [];
  A.forEach((row, i) => row.forEach((v, j) => { if (v !== 0) { rows.push(i); cols.push(j); vals.push(v); } }));
  const coo = { rows: n, cols: n, format: 'coo', values: vals, rowIndices: rows, colIndices: cols };
  const cli = { rows: n, cols: n, format: 'coo', data: { values: vals, rowIndices: rows, colIndices: cols } };       // bin/cli.js layout
  const rd = await s.solve({ rows: n, cols: n, format: 'dense', data: A }, b);
  const rc = await s.solve(coo, b), rl = await s.solve(cli, b);
  assert.deepStrictEqual(rd.solution, rc.solution);          // same entries, same arithmetic: bit-identical across input layouts
  assert.deepStrictEqual(rd.solution, rl.solution);
  rd.solution.forEach((v, i) => assert(Math.abs(v - xr[i]) < 1e-8));
  assert(rd.residual < 1e-9);
  const est = await s.estimateEntry(coo, b, { row: 17, column: 0, epsilon: 1e-9, confidence: 0.95, method: 'neumann' });
  assert(Math.abs(est.estimate - xr[17]) < 1e-8 && est.confidence === 1 && est.variance === 0);
  const seeded = new SublinearSolver({ method: 'neumann', epsilon: 1e-6, maxIterations: 1000, seed: 42 });
  const w1 = await seeded.estimateEntry(coo, b, { row: 17, column: 0, epsilon: 0.02, confidence: 0.95, method: 'random-walk' });
  const w2 = await seeded.estimateEntry(coo, b, { row: 17, column: 0, epsilon: 0.02, confidence: 0.95, method: 'random-walk' });
  assert.strictEqual(w1.estimate, w2.estimate);               // seeded: reproducible
  // the reference's random-walk estimator has its own semantics (solver.ts:390-432, 630-648): pinned on the case the
  // Python / oracle tests pin it on — tridiagonal (-1, 10, -1), b = 1: numSamples = max(100, ceil(1/eps^2)), value b_i / a_ii
  const tri = { rows: 10, cols: 10, format: 'coo', values: [], rowIndices: [], colIndices: [] };
  for (let i = 0; i < 10; i++) for (const [j, v] of [[i - 1, -1], [i, 10], [i + 1, -1]]) if (j >= 0 && j < 10) { tri.rowIndices.push(i); tri.colIndices.push(j); tri.values.push(v); }
  const wt = await seeded.estimateEntry(tri, new Array(10).fill(1), { row: 0, column: 0, epsilon: 0.05, confidence: 0.95, method: 'random-walk' });
  assert(wt.numSamples === 400 && Math.abs(wt.estimate - 0.1) < 1e-9 && wt.confidence === 0.95, JSON.stringify(wt));
  {   // the `random-walk` METHOD of solve() (solveRandomWalk, solver.ts:278-357): exact where every walk is (a diagonal system: each walk
      // returns b_i / a_ii), CONVERGENCE_FAILED where the residual misses epsilon (:335-341) — tridiag(-1, 10, -1), b = 1: x = 0.1, ||A x - b|| = 0.57
    const diag = { rows: 3, cols: 3, format: 'dense', data: [[4, 0, 0], [0, -5, 0], [0, 0, 8]] };
    const rw = await new SublinearSolver({ method: 'random-walk', epsilon: 0.1, maxIterations: 10, seed: 3 }).solve(diag, [1, 2, 3]);
    assert(rw.converged && rw.method === 'random-walk' && rw.iterations === 3 && rw.residual < 1e-14, JSON.stringify(rw));
    [0.25, -0.4, 0.375].forEach((v, i) => assert(Math.abs(rw.solution[i] - v) < 1e-15));
    await rejects(new SublinearSolver({ method: 'random-walk', epsilon: 0.05, maxIterations: 10, seed: 3 }).solve(tri, new Array(10).fill(1)),
                  ErrorCodes.CONVERGENCE_FAILED, /Random walk sampling failed to achieve desired accuracy/);
  }
  {   // { stream: 'reference' }: the reference's ONE serial LCG stream — estimate / variance (estimateEntry) and solution (solve) bit for
      // bit what the reference's own TypeScript printed for these inputs (tests/golden/reference_walk_js.json, make_golden_walk.py: G10 / G11)
    const golden = JSON.parse(require('fs').readFileSync(require('path').join(__dirname, '..', 'golden', 'reference_walk_js.json'), 'utf8'));
    assert(golden.length === 2);
    for (const g of golden) {
      const gm = { rows: g.n, cols: g.n, format: 'coo', values: g.values, rowIndices: g.rows, colIndices: g.cols };
      const rs = new SublinearSolver({ method: 'random-walk', epsilon: g.epsilon, maxIterations: 10, seed: g.seed, stream: 'reference' });
      if (g.kind === 'estimate') {
        const e = await rs.estimateEntry(gm, g.b, { row: g.row, column: 0, epsilon: g.epsilon, confidence: 0.9, method: 'random-walk' });
        assert.strictEqual(e.estimate, g.expect.mean);
        assert.strictEqual(e.variance, g.expect.variance);
        const eb = await new SublinearSolver({ method: 'random-walk', epsilon: g.epsilon, maxIterations: 10, seed: g.seed })
          .estimateEntry(gm, g.b, { row: g.row, column: 0, epsilon: g.epsilon, confidence: 0.9, method: 'random-walk' });
        assert(eb.estimate !== e.estimate && eb.numSamples === e.numSamples);       // the block form: another sample of the same estimator
      } else {
        let err = null;
        try { await rs.solve(gm, g.b); } catch (x) { err = x; }                     // this system misses epsilon: the reference throws there too
        assert(err && err.code === ErrorCodes.CONVERGENCE_FAILED && Math.abs(err.details.finalResidual - g.expect.residual) < 1e-12 * g.expect.residual, String(err));
        assert.strictEqual(err.details.variance, Math.sqrt(g.expect.totalVariance));
      }
    }
    assert.throws(() => new SublinearSolver({ method: 'random-walk', epsilon: 0.1, maxIterations: 10, stream: 'nonsense' }), /Unknown random-walk stream/);
  }
  {   // G12 on the device: solve() with method forward-push against the reference's own solveForwardPush (executed: make_golden_ts_push.py) —
      // pushes counted as iterations, the solution bit for bit
    const goldenPush = JSON.parse(require('fs').readFileSync(path.join(__dirname, '..', 'golden', 'reference_ts_push_js.json'), 'utf8'));
    assert(goldenPush.length >= 3);
    for (const g of goldenPush) {
      const r = await new SublinearSolver({ method: 'forward-push', epsilon: g.epsilon, maxIterations: g.maxIterations }).solve(g.matrix, g.b);
      assert.strictEqual(r.iterations, g.iterations, g.name);
      assert.deepStrictEqual(r.solution, g.solution, g.name);
      assert(Math.abs(r.residual - g.residual) <= 1e-12 * g.residual, g.name);
    }
  }
  for (const g of goldenPageRank) {   // G13 on the device: computePageRank with method forward-push returns the reference's solution, bit for bit
    const pr = new SublinearSolver({ method: 'forward-push', epsilon: 1e-6, maxIterations: 10 });
    const cfg = { damping: g.damping, epsilon: g.epsilon, maxIterations: g.maxIterations };
    if (g.personalized) cfg.personalized = g.personalized;
    assert.deepStrictEqual(Array.from(await pr.computePageRank(g.adjacency, cfg)), g.solution, g.name);
  }
  {   // G6 (tests/mcp/mcp-tool-tests.js:27-52): 10 x 10 tridiag(-1, 10, -1), b = e0 + e9, epsilon 1e-3 -> 12 pushes, ||r|| = 5.2915e-4
    const t = { rows: 10, cols: 10, format: 'coo', values: [], rowIndices: [], colIndices: [] };
    for (let i = 0; i < 10; i++) for (const [j, v] of [[i - 1, -1], [i, 10], [i + 1, -1]]) if (j >= 0 && j < 10) { t.rowIndices.push(i); t.colIndices.push(j); t.values.push(v); }
    const bb = new Array(10).fill(0); bb[0] = 1; bb[9] = 1;
    const g6 = await new SublinearSolver({ method: 'forward-push', epsilon: 1e-3, maxIterations: 1000 }).solve(t, bb);
    assert(g6.converged && g6.iterations === 12 && Math.abs(g6.residual - 5.2915e-4) < 1e-7, JSON.stringify(g6));
    await rejects(new SublinearSolver({ method: 'forward-push', epsilon: 1e-12, maxIterations: 3 }).solve(t, bb), ErrorCodes.CONVERGENCE_FAILED, /failed to converge after 3/);
  }
  // PageRank of a small graph against power iteration
  const adj = { rows: 5, cols: 5, format: 'dense', data: [[0, 1, 1, 0, 0], [0, 0, 1, 0, 0], [1, 0, 0, 1, 0], [0, 0, 0, 0, 1], [1, 0, 0, 0, 0]] };
  const pr = await new SublinearSolver({ method: 'forward-push', epsilon: 1e-13, maxIterations: 100000 }).computePageRank(adj, { damping: 0.85, epsilon: 1e-13, maxIterations: 100000 });
  let p = new Array(5).fill(0.2);
  for (let it = 0; it < 500; it++) {
    const q = new Array(5).fill(0.15 / 5);
    for (let i = 0; i < 5; i++) { const out = adj.data[i].reduce((u, v) => u + v, 0); for (let j = 0; j < 5; j++) if (adj.data[i][j]) q[j] += 0.85 * p[i] * adj.data[i][j] / out; }
    p = q;
  }
  pr.forEach((v, i) => assert(Math.abs(v - p[i]) < 1e-10, `pagerank[${i}] ${v} vs ${p[i]}`));
  assert(Math.abs(pr.reduce((u, v) => u + v, 0) - 1) < 1e-10);
  console.log('gpu surface ok');
})().catch((e) => { console.error(e); process.exit(1); });
