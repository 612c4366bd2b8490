"""Graph-side mirror of the reference's push solvers over the C ABI.

  reference (orphan Rust spec)                                   here
  PushGraph::{from_matrix, from_edges, out_degree, in_degree}    PushGraph          src/graph/adjacency.rs:199-277
  ForwardPushConfig {alpha, epsilon, max_pushes, ...}            ForwardPushConfig  src/solver/forward_push.rs:26-49
  ForwardPushSolver::{solve_single_source, solve_multi_source,
      query_single_entry, extrapolated_solution}                 ForwardPushSolver  forward_push.rs:67-301
  BackwardPushSolver::{solve_single_target, solve_multi_target, solve_with_source,
      query_transition_probability, reachability_probabilities,
      extrapolated_solution, combine_with_forward}                                      BackwardPushSolver src/solver/backward_push.rs:67-334
  BidirectionalPushSolver::{solve_bidirectional, adaptive_solve}  BidirectionalPushSolver backward_push.rs:337-410

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

import ctypes as C

import numpy as np

from . import _lib as L
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
    """PushGraph (src/graph/adjacency.rs:199-277) on the device behind sl_push_graph_*: adjacency CSR, its transpose, degrees (row
    sums) and reverse degrees (column sums) are built by the library; the host keeps the CSR arrays it was given for inspection."""

    def __init__(self, row_ptr, col_idx, weights, n: int):
        self.n = int(n)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
        self.col_idx = np.ascontiguousarray(col_idx, dtype=np.uint32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self._h = L.vp()
        L.check(L.load().sl_push_graph_create(self.n, L.ptr(self.row_ptr), L.ptr(self.col_idx), L.ptr(self.weights), L.SL_MEM_HOST, C.byref(self._h)))
        self.degrees = np.zeros(self.n)                                                           # graph/mod.rs:81-89
        self.reverse_degrees = np.zeros(self.n)
        L.check(L.load().sl_push_graph_degrees(self._h, L.ptr(self.degrees), L.ptr(self.reverse_degrees), L.SL_MEM_HOST))
        self._cache = {}

    def __del__(self):
        try:
            if self._h:
                L.load().sl_push_graph_destroy(self._h)
                self._h = None
        except Exception:
            pass

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

    def forward_neighbors(self, node: int):
        """PushGraph::forward_neighbors, adjacency.rs:251-253: (target, weight) pairs of the node's out-edges in stored order"""
        if not (0 <= node < self.n):
            return iter(())
        k0, k1 = int(self.row_ptr[node]), int(self.row_ptr[node + 1])
        return iter(list(zip(self.col_idx[k0:k1].tolist(), self.weights[k0:k1].tolist())))

    def backward_neighbors(self, node: int):
        """PushGraph::backward_neighbors, adjacency.rs:256-258: (source, weight) pairs of the node's in-edges — a row of the transposed
        adjacency (CompressedSparseRow::transpose, graph/mod.rs:92-130: sources ascending, parallel edges in stored order), read from
        host arrays the graph was given (the device keeps its own transpose for the pushes)"""
        if not (0 <= node < self.n):
            return iter(())
        if "_rev" not in self._cache:                                    # built once on the host from the arrays the graph was given
            order = np.argsort(self.col_idx, kind="stable")              # stable: sources ascending, parallel edges in stored order
            src = np.repeat(np.arange(self.n, dtype=np.int64), np.diff(self.row_ptr.astype(np.int64)))
            ptr = np.zeros(self.n + 1, dtype=np.int64)
            np.add.at(ptr, self.col_idx.astype(np.int64) + 1, 1)
            self._cache["_rev"] = (np.cumsum(ptr), src[order], self.weights[order])
        ptr, src, w = self._cache["_rev"]
        k0, k1 = int(ptr[node]), int(ptr[node + 1])
        return iter(list(zip(src[k0:k1].tolist(), w[k0:k1].tolist())))

    def system(self, alpha: float, backward: bool, flags: int = L.SL_MATRIX_WITH_TRANSPOSE, dangling_identity: bool = False) -> SparseMatrix:
        """forward: A = I - (1-alpha) P^T ; backward: A = I - (1-alpha) P  (P_uv = w_uv / deg_u; dangling u: P_uu = 1, or — dangling_identity,
        TS computePageRank's rule — nothing) — assembled on the device by sl_push_graph_system"""
        key = (float(alpha), bool(backward), int(flags), bool(dangling_identity))
        if key not in self._cache:
            h = L.vp()
            mode = (L.SL_SYSTEM_BACKWARD if backward else L.SL_SYSTEM_FORWARD) | (L.SL_SYSTEM_DANGLING_IDENTITY if dangling_identity else 0)
            L.check(L.load().sl_push_graph_system(self._h, float(alpha), mode, int(flags), C.byref(h)))
            self._cache[key] = SparseMatrix(h, self.n, self.n)
        return self._cache[key]

    # ---- the ACL push in the spec's own visiting order (order-exact parity entries) ----
    def _acl(self, fn, args, config, log_cap):
        o = L.AclOptions()
        L.load().sl_acl_options_default(C.byref(o))
        o.alpha, o.epsilon, o.queue_threshold, o.max_pushes = config.alpha, config.epsilon, config.queue_threshold, int(config.max_pushes)
        o.adaptive_threshold, o.mem = (1 if config.adaptive_threshold else 0), L.SL_MEM_HOST
        est, res = np.zeros(max(self.n, 1)), np.zeros(max(self.n, 1))
        log = np.zeros(max(int(log_cap), 1), dtype=np.uint32)
        r = L.AclResult()
        L.check(fn(self._h, *args, C.byref(o), L.ptr(est), L.ptr(res), L.ptr(log) if log_cap else None, int(log_cap), C.byref(r)))
        out = PushResult(est[: self.n], res[: self.n], int(r.push_count), int(r.nodes_visited), float(r.residual_norm))
        out.push_log = log[: min(int(log_cap), int(r.push_count))].copy() if log_cap else None
        out.stopped_by = int(r.stopped_by)
        return out

    def acl_forward(self, sources, config, log_cap: int = 0):
        src = np.ascontiguousarray(list(sources), dtype=np.uint64)
        return self._acl(L.load().sl_forward_push_acl, (int(src.size), L.ptr(src)), config, log_cap)

    def acl_backward(self, targets, config, log_cap: int = 0):
        tg = np.ascontiguousarray(list(targets), dtype=np.uint64)
        return self._acl(L.load().sl_backward_push_acl, (int(tg.size), L.ptr(tg)), config, log_cap)

    def acl_forward_with_target(self, source: int, target: int, target_precision: float, config, log_cap: int = 0):
        return self._acl(L.load().sl_forward_push_acl_with_target, (int(source), int(target), float(target_precision)), config, log_cap)

    def acl_backward_with_source(self, source: int, target: int, source_precision: float, config, log_cap: int = 0):
        return self._acl(L.load().sl_backward_push_acl_with_source, (int(source), int(target), float(source_precision)), config, log_cap)

    def acl_reachability(self, target: int, config) -> np.ndarray:
        o = L.AclOptions()
        L.load().sl_acl_options_default(C.byref(o))
        o.alpha, o.epsilon, o.queue_threshold, o.max_pushes = config.alpha, config.epsilon, config.queue_threshold, int(config.max_pushes)
        o.adaptive_threshold, o.mem = (1 if config.adaptive_threshold else 0), L.SL_MEM_HOST
        out, r = np.zeros(max(self.n, 1)), L.AclResult()
        L.check(L.load().sl_backward_push_acl_reachability(self._h, int(target), C.byref(o), L.ptr(out), C.byref(r)))
        return out[: self.n]


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
        """estimate + alpha * residual, forward_push.rs:292-301 / backward_push.rs:302-311 (sl_acl_extrapolated_solution: on the device)"""
        e, r = np.ascontiguousarray(result.estimate, dtype=np.float64), np.ascontiguousarray(result.residual, dtype=np.float64)
        out = np.empty(max(e.size, 1))
        L.check(L.load().sl_acl_extrapolated_solution(int(e.size), float(self.config.alpha), L.ptr(e), L.ptr(r), L.ptr(out), L.SL_MEM_HOST))
        return out[: e.size]


class ForwardPushSolver(_PushBase):
    """order="synchronous" (default): the data-parallel push on the system matrix (same fixed point, per-row thresholds);
    order="reference": the spec's own visiting order — WorkQueue pops, push_count / nodes_visited / every bit as the reference's loop
    (sl_forward_push_acl; sequential across pushes, for parity and small graphs)"""

    def solve_single_source(self, source: int, order: str = "synchronous", log_cap: int = 0) -> PushResult:          # forward_push.rs:67-122
        if order == "reference":
            return self.graph.acl_forward([source], self.config, log_cap)
        return self._solve([source])

    def solve_multi_source(self, sources: Sequence[int], order: str = "synchronous", log_cap: int = 0) -> PushResult:  # :125-177
        if order == "reference":
            return self.graph.acl_forward(list(sources), self.config, log_cap)
        return self._solve(list(sources))

    def solve_with_target(self, source: int, target: int, target_precision: float, log_cap: int = 0) -> PushResult:   # :233-290
        """early termination once estimate[target] > target_precision and residual[target] < 0.1 target_precision — defined by the
        visiting order, so it always runs the spec's own"""
        return self.graph.acl_forward_with_target(source, target, target_precision, self.config, log_cap)

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

    def solve_single_target(self, target: int, order: str = "synchronous", log_cap: int = 0) -> PushResult:          # backward_push.rs:67-122
        if order == "reference":
            return self.graph.acl_backward([target], self.config, log_cap)
        return self._solve([target])

    def solve_multi_target(self, targets: Sequence[int], order: str = "synchronous", log_cap: int = 0) -> PushResult:   # backward_push.rs:125-176
        if order == "reference":
            return self.graph.acl_backward(list(targets), self.config, log_cap)
        return self._solve(list(targets))

    def solve_with_source(self, source: int, target: int, source_precision: float, log_cap: int = 0) -> PushResult:     # backward_push.rs:238-293
        """early termination once estimate[source] > source_precision and residual[source] < 0.1 source_precision (:262-264) — defined by
        the visiting order, so it always runs the spec's own; source or target out of range: the empty result (:243-251)"""
        return self.graph.acl_backward_with_source(source, target, source_precision, self.config, log_cap)

    def reachability_probabilities(self, target: int) -> np.ndarray:                                                    # backward_push.rs:296-299
        """solve_single_target(target) in the spec's order, then extrapolated_solution of its result — one call, on the device"""
        return self.graph.acl_reachability(target, self.config)

    def query_transition_probability(self, source: int, target: int) -> float:   # :228-235
        r = self.solve_single_target(target)
        return float(r.estimate[source]) if 0 <= source < r.estimate.size else 0.0

    def combine_with_forward(self, backward_result: PushResult, forward_estimate, forward_residual) -> float:
        """backward_push.rs:314-333: for i in 0..min(len): total += be_i * fe_i; total += br_i * fe_i * alpha; total += be_i * fr_i * alpha —
        three adds per node, one after the other, and so here: the 3 k terms interleaved in that order and folded left to right
        (np.add.accumulate is the sequential running sum), not a pairwise np.sum — the reference's bits"""
        return combine_with_forward(self.config.alpha, backward_result.estimate, backward_result.residual, forward_estimate, forward_residual)


def combine_with_forward(alpha: float, backward_estimate, backward_residual, forward_estimate, forward_residual) -> float:
    """BackwardPushSolver::combine_with_forward, backward_push.rs:314-333, in the reference's order of operations (host-side: the four
    vectors are the callers' host arrays, and a left-to-right fold has no parallel form that keeps its bits)"""
    be, br = np.asarray(backward_estimate, dtype=np.float64), np.asarray(backward_residual, dtype=np.float64)
    fe, fr = np.asarray(forward_estimate, dtype=np.float64), np.asarray(forward_residual, dtype=np.float64)
    k = min(be.size, fe.size)                                              # `.min(forward_estimate.len())`; the residuals are indexed alike
    if k == 0:
        return 0.0
    terms = np.empty(3 * k)
    terms[0::3] = be[:k] * fe[:k]
    terms[1::3] = (br[:k] * fe[:k]) * alpha
    terms[2::3] = (be[:k] * fr[:k]) * alpha
    return float(np.add.accumulate(terms)[-1] + 0.0)                       # 0.0 + t0 = t0 exactly (-0.0 + 0.0 = +0.0, as `0.0 + t0` gives)


class BidirectionalPushSolver:
    """BidirectionalPushSolver, backward_push.rs:337-410: a forward solve from the source and a backward solve from the target combined
    (solve_bidirectional), or whichever single direction starts from the node of much higher degree (adaptive_solve).
    order="reference" runs both pushes in the spec's visiting order (the bits of the reference's own loop), "synchronous" (default) the
    data-parallel pushes."""

    def __init__(self, graph: PushGraph, forward_config: ForwardPushConfig | None = None, backward_config: ForwardPushConfig | None = None,
                 order: str = "synchronous"):
        self.graph, self.order = graph, order
        self.forward_config, self.backward_config = forward_config or ForwardPushConfig(), backward_config or BackwardPushConfig()

    def solve_bidirectional(self, source: int, target: int) -> float:                                                  # :359-377
        f = ForwardPushSolver(self.graph, self.forward_config).solve_single_source(source, order=self.order)
        bs = BackwardPushSolver(self.graph, self.backward_config)
        b = bs.solve_single_target(target, order=self.order)
        return bs.combine_with_forward(b, f.estimate, f.residual)

    def adaptive_solve(self, source: int, target: int) -> float:                                                       # :380-410
        n = self.graph.num_nodes()
        if not (0 <= source < n) or not (0 <= target < n):
            return 0.0
        out_s, in_t = self.graph.out_degree(source), self.graph.in_degree(target)
        if out_s > in_t * 2.0:                                             # start from the target (backward push)
            r = BackwardPushSolver(self.graph, self.backward_config).solve_single_target(target, order=self.order)
            return float(r.estimate[source])                               # query_transition_probability, :228-235
        if in_t > out_s * 2.0:                                             # start from the source (forward push)
            return float(ForwardPushSolver(self.graph, self.forward_config).solve_single_source(source, order=self.order).estimate[target])   # query_single_entry, :224-231
        return self.solve_bidirectional(source, target)
