"""Seeded synthetic inputs.  numpy twins of csrc/sl_synth.hip — bit-identical.

Recipes follow the reference's own benchmark generators (cited per function); they are
counter-based so any row range can be produced independently on any rank.
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_K = np.uint64(0xD1B54A32D192ED03)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.copy()
    z ^= z >> np.uint64(30)
    z *= _M1
    z ^= z >> np.uint64(27)
    z *= _M2
    z ^= z >> np.uint64(31)
    return z


def sdd_rows(n: int, k: int, seed: int, half_bandwidth: int = 0, row_lo: int = 0, row_hi: int | None = None):
    """S-DD(n, k, seed, w): rows [row_lo, row_hi) as CSR (row_ptr u32, col_idx u32, values f64) + b.

    After src/ultra_fast.rs:221-248 / benches/performance_benchmarks.rs:12-43 (LCG columns, diagonal
    10 + 0.01 i, b = 1 + 0.001 i) but duplicate-free, self-excluding, exactly k entries per row
    (diagonal included), strictly row diagonally dominant (sum|offdiag| <= d/2), asymmetric,
    columns ascending.  half_bandwidth w > 0 draws columns from the band [i-w, i+w].
    """
    if row_hi is None:
        row_hi = n
    return sdd_rows_at(n, k, seed, half_bandwidth, np.arange(row_lo, row_hi, dtype=np.uint64))


def sdd_rows_at(n: int, k: int, seed: int, half_bandwidth: int, rows_wanted):
    """The rows `rows_wanted` (any list of global row numbers, in that order) of S-DD(n, k, seed, w) — the generator is counter-based,
    so a scattered set of rows costs what it holds (bench.py's parity gate regenerates the rows a sampled block gathers from)."""
    if not (2 <= k <= 64):
        raise ValueError("k must be in [2, 64]")
    m = k - 1
    banded = bool(half_bandwidth) and 2 * half_bandwidth + 1 < n
    if ((half_bandwidth + 1) if banded else n) // m < 2:
        raise ValueError("column window too narrow for k-1 distinct off-diagonals")
    with np.errstate(over="ignore"):
        i = np.ascontiguousarray(rows_wanted, dtype=np.uint64)
        rows = i.size
        if not banded:
            lo = np.zeros(rows, dtype=np.uint64)
            sw = np.full(rows, n // m, dtype=np.uint64)
        else:                                   # band: columns in [i-w, i+w], clipped at the matrix edge
            w = np.uint64(half_bandwidth)
            lo = np.where(i > w, i - w, np.uint64(0))
            hi = np.minimum(i + w + np.uint64(1), np.uint64(n))
            sw = (hi - lo) // np.uint64(m)
        sw = sw[:, None]
        j = np.arange(m, dtype=np.uint64)
        key = np.uint64(seed) * _G + (i[:, None] * np.uint64(64) + j[None, :] + np.uint64(1)) * _K
        h1 = _mix64(key)
        h2 = _mix64(h1 + _G)
        off = h1 % sw
        c = lo[:, None] + j[None, :] * sw + off
        hit = c == i[:, None]
        c = np.where(hit, np.where(off + np.uint64(1) < sw, c + np.uint64(1), c - np.uint64(1)), c)
    d = 10.0 + 0.01 * (i % np.uint64(1000)).astype(np.float64)
    scale = d / float(2 * m)
    u = (h2 >> np.uint64(11)).astype(np.float64) * 1.1102230246251565e-16
    v = (2.0 * u - 1.0) * scale[:, None]
    v = np.where(v == 0.0, scale[:, None], v)
    # insert the diagonal at its sorted position
    p = (c < i[:, None]).sum(axis=1)
    col = np.empty((rows, k), dtype=np.uint32)
    val = np.empty((rows, k), dtype=np.float64)
    slot = np.arange(k)[None, :]
    before = slot < p[:, None]
    at = slot == p[:, None]
    src = np.clip(np.where(before, slot, slot - 1), 0, m - 1)
    col[:] = np.take_along_axis(c, src, axis=1).astype(np.uint32)
    val[:] = np.take_along_axis(v, src, axis=1)
    col[at] = i.astype(np.uint32)
    val[at] = d
    row_ptr = (np.arange(rows + 1, dtype=np.uint64) * np.uint64(k)).astype(np.uint32)
    b = 1.0 + 0.001 * (i % np.uint64(1000)).astype(np.float64)
    return row_ptr, col.reshape(-1), val.reshape(-1), b


def ts_lcg(seed: int):
    """createSeededRandom, src/core/utils.ts:161-168."""
    state = seed

    def nxt() -> float:
        nonlocal state
        state = (state * 1664525 + 1013904223) % 0x100000000
        return state / 0x100000000

    return nxt


def gen1000_dense(size: int = 1000, strength: float = 2.0, seed: int = 42, density: float = 0.3):
    """S-GEN1000: `generate -t diagonally-dominant -s 1000` (src/cli/index.ts:308-352,
    src/mcp/tools/matrix.ts:297-322: each off-diagonal kept w.p. 0.3, U(-1,1), a_ii = strength*sum|off| + 1)
    with Math.random replaced by the TS LCG (core/utils.ts:161-168) so it is reproducible.
    Returns CSR (row_ptr, col_idx, values) and b = ones."""
    rnd = ts_lcg(seed)
    rp = [0]
    ci: list[int] = []
    va: list[float] = []
    for i in range(size):
        row_c: list[int] = []
        row_v: list[float] = []
        s = 0.0
        for j in range(size):
            if j == i:
                row_c.append(j)
                row_v.append(0.0)
                continue
            if rnd() < density:
                v = (rnd() - 0.5) * 2.0
                if v != 0.0:
                    row_c.append(j)
                    row_v.append(v)
                    s += abs(v)
        di = row_c.index(i)
        row_v[di] = strength * s + 1.0
        ci.extend(row_c)
        va.extend(row_v)
        rp.append(len(ci))
    return (np.asarray(rp, dtype=np.uint32), np.asarray(ci, dtype=np.uint32), np.asarray(va, dtype=np.float64),
            np.ones(size, dtype=np.float64))


def pagerank_graph(n: int, seed: int, mean_degree: int = 16, zipf_s: float = 2.1, max_degree: int = 10_000):
    """S-PR(n, seed): directed power-law graph as a weighted adjacency CSR (unit weights).
    Out-degree ~ Zipf(s) capped, targets drawn with a mild preferential bias; deterministic."""
    rng = np.random.Generator(np.random.PCG64(seed))
    deg = rng.zipf(zipf_s, size=n).astype(np.int64)
    deg = np.minimum(deg * max(1, mean_degree // 4), max_degree)
    deg = np.minimum(deg, n - 1)
    rp = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(deg, out=rp[1:])
    nnz = int(rp[-1])
    # preferential targets: square of a uniform pushes mass toward low ids
    u = rng.random(nnz)
    tgt = np.minimum((u * u * n).astype(np.int64), n - 1)
    src = np.repeat(np.arange(n, dtype=np.int64), deg)
    tgt = np.where(tgt == src, (tgt + 1) % n, tgt)
    # sort + unique per row
    order = np.lexsort((tgt, src))
    src, tgt = src[order], tgt[order]
    keep = np.ones(nnz, dtype=bool)
    keep[1:] = (src[1:] != src[:-1]) | (tgt[1:] != tgt[:-1])
    src, tgt = src[keep], tgt[keep]
    rp2 = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp2, src + 1, 1)
    np.cumsum(rp2, out=rp2)
    return rp2.astype(np.uint32), tgt.astype(np.uint32), np.ones(tgt.size, dtype=np.float64)


def adjacency_csr_first_match(r, c, v, n: int):
    """Triplets of an adjacency -> CSR (ascending columns) the way the reference READS a COO matrix: MatrixOperations.getEntry
    (src/core/matrix.ts:95-116) returns the FIRST stored match of (row, col) — later duplicates never enter its arithmetic, and a stored 0
    hides them too — so the first occurrence is kept, then exact zeros are dropped (a zero weight changes nothing in computePageRank)."""
    r = np.ascontiguousarray(r, dtype=np.int64)
    c = np.ascontiguousarray(c, dtype=np.int64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    key = r * n + c
    _, first = np.unique(key, return_index=True)                 # sorted by key = (row, column); index of the first occurrence
    first = first[v[first] != 0.0]
    rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp, r[first] + 1, 1)
    return np.cumsum(rp).astype(np.uint32), c[first].astype(np.uint32), v[first]


def pagerank_system(n: int, adj_rp, adj_ci, adj_w, damping: float = 0.85):
    """The system computePageRank assembles (src/core/solver.ts:664-722), in CSR, with the reference's OWN arithmetic — so that the values
    carry its bits (golden G13: the reference's TypeScript executed on the same adjacency):
        out_j      = adj[j][0] + adj[j][1] + ... left to right over the row            (:679-684)
        S[i][j]    = (i == j ? 1 : 0) - damping * (adj[j][i] / out_j)   for out_j > 0   (:689-698; a dangling node's column stays the identity's)
        rhs_i      = 1 * ((1 - damping) / n)                                            (:708)
    The adjacency is CSR with ascending, duplicate-free columns (the reference's getEntry reads the FIRST stored match of a COO entry:
    coalesce duplicates before calling).  Exact zeros are not stored (SparseMatrix::from_dense filters them)."""
    rp = np.ascontiguousarray(adj_rp, dtype=np.int64)
    ci = np.ascontiguousarray(adj_ci, dtype=np.int64)
    w = np.ascontiguousarray(adj_w, dtype=np.float64)
    lens = np.diff(rp)
    out = np.zeros(n)
    for k in range(int(lens.max()) if n and lens.size else 0):          # the k-th entry of every row that has one: a left-to-right sum per row
        rows = np.flatnonzero(lens > k)
        out[rows] = out[rows] + w[rp[rows] + k]
    src = np.repeat(np.arange(n, dtype=np.int64), lens)                 # edge src -> ci with weight w
    live = out[src] > 0
    src, dst, a = src[live], ci[live], w[live]
    prob = a / out[src]
    val = -(damping * prob)                                             # 0 - damping * prob
    loop = src == dst
    diag = np.ones(n)
    diag[dst[loop]] = 1.0 - damping * prob[loop]                         # 1 - damping * prob on a self loop
    rows_s = np.concatenate([dst[~loop], np.arange(n, dtype=np.int64)])
    cols_s = np.concatenate([src[~loop], np.arange(n, dtype=np.int64)])
    vals_s = np.concatenate([val[~loop], diag])
    keep = vals_s != 0.0
    rows_s, cols_s, vals_s = rows_s[keep], cols_s[keep], vals_s[keep]
    order = np.lexsort((cols_s, rows_s))
    rows_s, cols_s, vals_s = rows_s[order], cols_s[order], vals_s[order]
    s_rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(s_rp, rows_s + 1, 1)
    s_rp = np.cumsum(s_rp)
    return (s_rp.astype(np.uint32), cols_s.astype(np.uint32), vals_s.astype(np.float64), np.full(n, 1.0 * ((1.0 - damping) / n)))
