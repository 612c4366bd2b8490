"""BackwardPushSolver::combine_with_forward (backward_push.rs:314-333) is a left-to-right fold with three additions per node; the
package's host-side fold (push_graph.combine_with_forward: interleaved terms + a sequential running sum) must give the bits of the
reference's loop as the CPU checker restates it — on vectors where a pairwise or per-term-blocked sum would not.  No GPU needed."""
import numpy as np

from oracle import oracle as O
from sublinear_time_solver_amd.push_graph import combine_with_forward


def test_host_fold_has_the_references_bits():
    rng = np.random.default_rng(5)
    differs_from_blocked = 0
    for n in (0, 1, 2, 7, 1000, 4097):
        be, br, fe, fr = (rng.random(n) * 10.0 ** rng.integers(-8, 3, size=n) for _ in range(4))
        got, want = combine_with_forward(0.15, be, br, fe, fr), O.acl_combine_with_forward(0.15, be, br, fe, fr)
        assert np.float64(got).view(np.uint64) == np.float64(want).view(np.uint64), n
        blocked = float(np.sum(be * fe + br * fe * 0.15 + be * fr * 0.15))
        differs_from_blocked += blocked != want
    assert differs_from_blocked > 0                                     # the order is visible in the last bits
    # lengths: min(|backward estimate|, |forward estimate|) (backward_push.rs:322)
    be, br, fe, fr = np.arange(1.0, 6.0), np.ones(5), np.arange(1.0, 4.0), np.ones(3)
    assert combine_with_forward(0.5, be, br, fe, fr) == O.acl_combine_with_forward(0.5, be, br, fe, fr) == sum(be[i] * fe[i] + fe[i] * 0.5 + be[i] * 0.5 for i in range(3))
