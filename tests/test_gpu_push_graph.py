"""Graph-side push solvers on the GPU (ForwardPushSolver / BackwardPushSolver mirror) against the reference's
property tests (tests/rust/push_tests.rs) and the oracle's sequential ACL restatement (forward_push.rs:67-216)."""
import numpy as np
import pytest

from sublinear_time_solver_amd.push_graph import BackwardPushSolver, ForwardPushConfig, ForwardPushSolver, PushGraph
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def simple_graph():
    # create_simple_graph, tests/rust/push_tests.rs:15-22
    return PushGraph([0, 2, 4, 6, 7], [1, 2, 0, 3, 0, 3, 1], [0.5, 0.5, 0.8, 0.2, 0.6, 0.4, 1.0], 4)


def random_graph(n, edges_per_node):
    # create_random_graph, push_tests.rs:25-46 (LCG 1103515245 / 12345, weight 1/edges_per_node, self edges skipped)
    seed, edges = 12345, []
    for i in range(n):
        for _ in range(edges_per_node):
            seed = (seed * 1103515245 + 12345) % 2 ** 64
            t = seed % n
            if t != i:
                edges.append((i, t, 1.0 / edges_per_node))
    return PushGraph.from_edges(n, edges)


def test_forward_push_fixture_properties(gpu):
    g = simple_graph()
    s = ForwardPushSolver(g, ForwardPushConfig(alpha=0.15, epsilon=1e-6))
    r = s.solve_single_source(0)
    assert (r.estimate >= 0).all() and (r.residual >= -1e-18).all()                 # push_tests.rs:77-104
    assert abs(r.estimate.sum() + r.residual.sum() - 1.0) < 1e-12                  # mass conservation :107-129
    P = np.zeros((4, 4))
    for i in range(4):
        P[i, g.col_idx[g.row_ptr[i]:g.row_ptr[i + 1]]] = g.weights[g.row_ptr[i]:g.row_ptr[i + 1]]
    pi = 0.15 * np.linalg.solve((np.eye(4) - 0.85 * P).T, np.eye(4)[0])
    np.testing.assert_allclose(r.estimate, pi, atol=3e-6)
    np.testing.assert_allclose(r.estimate, [0.431272, 0.276168, 0.183291, 0.109267], atol=3e-6)
    acl = O.acl_push(g.row_ptr, g.col_idx, g.weights, [0], alpha=0.15, epsilon=1e-6)
    np.testing.assert_allclose(r.estimate, acl["estimate"], atol=3e-6)             # same fixed point as the sequential queue
    np.testing.assert_allclose(s.extrapolated_solution(r), acl["estimate"] + 0.15 * acl["residual"], atol=3e-6)
    assert abs(s.query_single_entry(0, 2) - pi[2]) < 3e-6 and s.query_single_entry(0, 99) == 0.0


def test_forward_push_monotone_and_edge_cases(gpu):
    g = simple_graph()
    pushes = [ForwardPushSolver(g, ForwardPushConfig(epsilon=e)).solve_single_source(0).push_count for e in (1e-2, 1e-4, 1e-6)]
    assert pushes[0] <= pushes[1] <= pushes[2]                                      # push_tests.rs:132-162
    assert ForwardPushSolver(g, ForwardPushConfig(alpha=0.99)).solve_single_source(0).estimate[0] > 0.5   # :520-537
    r = ForwardPushSolver(g).solve_single_source(10)                                # out-of-bounds source :433-495
    assert r.push_count == 0 and r.estimate.sum() == 0.0
    path = PushGraph.from_edges(5, [(0, 1, 1.0), (1, 2, 1.0), (2, 3, 1.0), (7, 1, 1.0)])   # invalid edge skipped; node 4 isolated
    r = ForwardPushSolver(path).solve_single_source(0)
    assert r.estimate[4] == 0.0 and r.estimate[0] > 0 and abs(r.estimate.sum() + r.residual.sum() - 1.0) < 1e-12
    m = ForwardPushSolver(g).solve_multi_source([0, 3])
    assert abs(m.estimate.sum() + m.residual.sum() - 1.0) < 1e-12


def test_random_graph_forward_backward_bidirectional(gpu):
    g = random_graph(400, 5)
    f = ForwardPushSolver(g, ForwardPushConfig(epsilon=1e-9))
    b = BackwardPushSolver(g, ForwardPushConfig(epsilon=1e-9))
    fr = f.solve_single_source(3)
    acl = O.acl_push(g.row_ptr, g.col_idx, g.weights, [3], alpha=0.15, epsilon=1e-9, adaptive_threshold=False, queue_threshold=0.0)
    assert abs(fr.estimate.sum() + fr.residual.sum() - 1.0) < 1e-10
    np.testing.assert_allclose(fr.estimate, acl["estimate"], atol=5e-7)
    # pi_s(t) from the forward solve == transition probability from the backward solve (backward_push.rs:228-235)
    for t in (0, 17, 399):
        assert abs(b.query_transition_probability(3, t) - fr.estimate[t]) < 1e-7
    br = b.solve_single_target(17)
    combined = b.combine_with_forward(br, fr.estimate, fr.residual)                 # backward_push.rs:314-333
    assert np.isfinite(combined) and combined >= 0


def test_single_entry_query_is_local(gpu):
    """query_single_entry (forward_push.rs:224-231): the default route is ONE entry through a local push on the transposed system
    (nothing of size n comes back); it agrees with the spec's route — the whole single-source solve, then one entry — within the
    stop rule's error on a 5000-node graph"""
    g = random_graph(5000, 4)
    s = ForwardPushSolver(g, ForwardPushConfig(alpha=0.15, epsilon=1e-9))
    full = s.solve_single_source(11).estimate
    for t in (0, 11, 1234, 4999):
        assert abs(s.query_single_entry(11, t) - full[t]) < 1e-6
        assert s.query_single_entry(11, t, via="solve") == full[t]
    assert s.query_single_entry(11, 5000) == 0.0 and s.query_single_entry(-1, 3) == 0.0
