"""Graph-side mirror of the reference's push solvers over the C ABI.

  reference (orphan Rust spec)                                   here
  PushGraph::{from_matrix, from_edges, out_degree, in_degree}    PushGraph          src/graph/adjacency.rs:199-277
  ForwardPushConfig {alpha, epsilon, max_pushes, ...}            ForwardPushConfig  src/solver/forward_push.rs:26-49
  ForwardPushSolver::{solve_single_source, solve_multi_source,
      query_single_entry, extrapolated_solution}                 ForwardPushSolver  forward_push.rs:67-301
  BackwardPushSolver::{solve_single_target, query_transition_probability,
      combine_with_forward}                                      BackwardPushSolver src/solver/backward_push.rs:67-334

The reference pushes one node at a time from a priority queue (inherently sequential).  The device runs the
synchronous form of the same push: personalised PageRank pi_s = alpha e_s^T (I - (1-alpha) P)^-1 is the solution
of A x = alpha e_s with A = I - (1-alpha) P^T, and the reference's (estimate, residual) pair corresponds to
(x, r / alpha) of the residual push r = b - A x: `estimate[u] += alpha r[u]; r[v] += (1-alpha) r[u] w_uv / deg_u`
is exactly one Gauss-Southwell push on that system.  Same fixed point, same invariants (non-negativity, mass
conservation sum(estimate) + sum(residual) = 1), different visiting order.  The admission / skip rule is the spec's:
node u is pushed while residual[u] >= epsilon * max(out_degree(u), 1) (forward_push.rs:93-99; the queue admits
residual / degree >= queue_threshold = 1e-8 << epsilon, graph/mod.rs:171-181, so the skip rule is the one that binds) —
passed to the device as one threshold per row (sl_push_options.theta_rows); at exit every node is below ITS threshold,
as after the reference's loop.  ForwardPushConfig.degree_scaled = False gives the absolute threshold epsilon instead.
Dangling nodes keep their mass on themselves (forward_push.rs:210-215): a self loop of weight 1.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, Sequence, Tuple

import numpy as np

from .solver import PushSolver, SparseMatrix


@dataclass
class ForwardPushConfig:
    alpha: float = 0.15
    epsilon: float = 1e-6
    max_pushes: int = 1_000_000
    queue_threshold: float = 1e-8        # kept for signature parity; the synchronous form has no queue
    adaptive_threshold: bool = True
    degree_scaled: bool = True           # skip rule residual[u] < epsilon * max(deg_u, 1) (forward_push.rs:96-99); False: residual[u] < epsilon


BackwardPushConfig = ForwardPushConfig


@dataclass
class PushResult:
    estimate: np.ndarray
    residual: np.ndarray
    push_count: int
    nodes_visited: int
    residual_norm: float


class PushGraph:
    """adjacency CSR + degrees (row sums) + reverse degrees (column sums)"""

    def __init__(self, row_ptr, col_idx, weights, n: int):
        self.n = int(n)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        rows = np.repeat(np.arange(self.n), np.diff(self.row_ptr.astype(np.int64)))
        self.degrees = np.bincount(rows, weights=self.weights, minlength=self.n)                  # graph/mod.rs:81-89
        self.reverse_degrees = np.bincount(self.col_idx.astype(np.int64), weights=self.weights, minlength=self.n)
        self._rows = rows
        self._cache = {}

    @classmethod
    def from_edges(cls, num_nodes: int, edges: Iterable[Tuple[int, int, float]]):
        """PushGraph::from_edges, adjacency.rs:226-238: out-of-range endpoints are skipped"""
        e = [(int(a), int(b), float(w)) for a, b, w in edges if 0 <= a < num_nodes and 0 <= b < num_nodes]
        e.sort(key=lambda t: (t[0], t[1]))
        rp = np.zeros(num_nodes + 1, dtype=np.uint32)
        for a, _, _ in e:
            rp[a + 1] += 1
        rp = np.cumsum(rp).astype(np.uint32)
        return cls(rp, [t[1] for t in e], [t[2] for t in e], num_nodes)

    @classmethod
    def from_matrix(cls, row_ptr, col_idx, weights, n):
        return cls(row_ptr, col_idx, weights, n)

    def num_nodes(self) -> int:
        return self.n

    def num_edges(self) -> int:
        return int(self.weights.size)

    def out_degree(self, node: int) -> float:
        return float(self.degrees[node]) if 0 <= node < self.n else 0.0

    def in_degree(self, node: int) -> float:
        return float(self.reverse_degrees[node]) if 0 <= node < self.n else 0.0

    def system(self, alpha: float, backward: bool) -> SparseMatrix:
        """forward: A = I - (1-alpha) P^T ; backward: A = I - (1-alpha) P  (P_uv = w_uv / deg_u; dangling u: P_uu = 1)"""
        key = (float(alpha), bool(backward))
        if key not in self._cache:
            deg = self.degrees
            safe = np.where(deg > 0, deg, 1.0)
            pr, pc = self._rows, self.col_idx.astype(np.int64)
            pv = self.weights / safe[pr]
            dang = np.nonzero(deg <= 0)[0]
            pr = np.concatenate([pr, dang])
            pc = np.concatenate([pc, dang])
            pv = np.concatenate([pv, np.ones(dang.size)])
            if not backward:
                pr, pc = pc, pr                               # P^T
            import scipy.sparse as sp
            A = (sp.identity(self.n, format="csr") - (1.0 - alpha) * sp.csr_matrix((pv, (pr, pc)), shape=(self.n, self.n))).tocsr()
            A.sum_duplicates()
            A.sort_indices()
            self._cache[key] = SparseMatrix.from_csr(A.indptr, A.indices, A.data, self.n, self.n, with_transpose=True)
        return self._cache[key]


class _PushBase:
    backward = False

    def __init__(self, graph: PushGraph, config: ForwardPushConfig | None = None):
        self.graph, self.config = graph, config or ForwardPushConfig()

    def _solve(self, seeds: Sequence[int]) -> PushResult:
        n, c = self.graph.n, self.config
        valid = [s for s in seeds if 0 <= s < n]
        if not valid or n == 0:                                # forward_push.rs:75-83: out-of-range source -> empty result
            return PushResult(np.zeros(n), np.zeros(n), 0, 0, 0.0)
        b = np.zeros(n)
        np.add.at(b, valid, c.alpha / len(seeds))              # unit mass split over the sources, :131-137
        m = self.graph.system(c.alpha, self.backward)
        # admission / skip rule of the spec (forward_push.rs:93-99, graph/mod.rs:171-212): node u is pushed while
        # residual[u] >= epsilon * max(degree(u), 1); the device residual is alpha x the spec's, so theta_u = alpha * epsilon * max(deg_u, 1)
        deg = self.graph.reverse_degrees if self.backward else self.graph.degrees
        theta_rows = c.alpha * c.epsilon * np.maximum(deg, 1.0) if getattr(c, "degree_scaled", True) else None
        out = PushSolver(theta=c.alpha * c.epsilon, max_rounds=max(1, c.max_pushes), theta_rows=theta_rows).solve(m, b)
        residual = out["residual"] / c.alpha
        est = out["solution"]
        return PushResult(est, residual, out["pushes"], int(np.count_nonzero(est)), float(np.linalg.norm(residual)))

    def extrapolated_solution(self, result: PushResult) -> np.ndarray:
        """estimate + alpha * residual, forward_push.rs:292-301"""
        return result.estimate + self.config.alpha * result.residual


class ForwardPushSolver(_PushBase):
    def solve_single_source(self, source: int) -> PushResult:          # forward_push.rs:67-122
        return self._solve([source])

    def solve_multi_source(self, sources: Sequence[int]) -> PushResult:  # :125-177
        return self._solve(list(sources))

    def query_single_entry(self, source: int, target: int, via: str = "local") -> float:   # :224-231
        """pi_source(target).  via="local" (default): ONE entry of the solution of A x = alpha e_source — a local push on A^T from
        e_target on the device (sl_estimate_entry: cost = the rows that push touches, nothing of size n comes back);
        via="solve": the spec's own route, the whole single-source solve, then one entry of it."""
        n, c = self.graph.n, self.config
        if not (0 <= source < n) or not (0 <= target < n):            # :75-83 / :226-230
            return 0.0
        if via == "solve":
            return float(self.solve_single_source(source).estimate[target])
        from .solver import estimate_entry
        b = np.zeros(n)
        b[source] = c.alpha
        m = self.graph.system(c.alpha, self.backward)
        return float(estimate_entry(m, b, target, theta=c.alpha * c.epsilon, max_rounds=max(1, c.max_pushes)).estimate)


class BackwardPushSolver(_PushBase):
    backward = True

    def solve_single_target(self, target: int) -> PushResult:          # backward_push.rs:67-122
        return self._solve([target])

    def solve_multi_target(self, targets: Sequence[int]) -> PushResult:
        return self._solve(list(targets))

    def query_transition_probability(self, source: int, target: int) -> float:   # :228-235
        r = self.solve_single_target(target)
        return float(r.estimate[source]) if 0 <= source < r.estimate.size else 0.0

    def combine_with_forward(self, backward_result: PushResult, forward_estimate, forward_residual) -> float:
        """backward_push.rs:314-333"""
        a = self.config.alpha
        k = min(backward_result.estimate.size, len(forward_estimate))
        be, br = backward_result.estimate[:k], backward_result.residual[:k]
        fe, fr = np.asarray(forward_estimate)[:k], np.asarray(forward_residual)[:k]
        return float(np.sum(be * fe + br * fe * a + be * fr * a))
