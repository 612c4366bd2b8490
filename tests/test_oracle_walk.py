"""CPU pins of the random-walk oracle (a15 / f-3 and the `random-walk` method of the TS solve(), core/solver.ts:278-357, 390-432):
the jump-ahead that cuts the reference's ONE LCG stream (core/utils.ts:161-168, golden G7) into a block per walk, the relation between
the serial form (the reference as written) and the block form (what the device computes), and the solve's verdict rule."""
from pathlib import Path

import numpy as np

from oracle import oracle as O
from sublinear_time_solver_amd import generators as G


def test_jump_ahead_is_the_generator_stepped():
    for seed in (1, 42, 12345, 0xFFFFFFFF):
        u = O.ts_lcg(seed, 5000)
        states = [seed] + [int(round(x * 2 ** 32)) for x in u]
        for k in (0, 1, 2, 3, 17, 2047, 2048, 2049, 4096, 5000):
            assert O.ts_lcg_jump(seed, k) == states[k], (seed, k)
        assert O.ts_lcg_jump(seed, 2 ** 32) == seed                      # full period of the mixed generator mod 2^32
        assert O.ts_lcg_jump(O.ts_lcg_jump(seed, 2048 * 7), 2048 * 5) == O.ts_lcg_jump(seed, 2048 * 12)      # blocks compose


def _system(n=30, k=5, seed=4):
    rp, ci, va, _ = G.sdd_rows(n, k, seed=seed)
    return rp, ci, va, np.random.default_rng(seed).standard_normal(n) * 2.0


def test_walk_zero_of_the_block_form_is_the_references_first_walk():
    """with ONE walk per coordinate budgeted twice over (2 walks), coordinate 0's first walk reads the stream from the seed in both forms"""
    rp, ci, va, b = _system()
    for seed in (3, 99):
        vals = O.ts_random_walk_streams(rp, ci, va, b, 0, 2, seed)[0]
        blocks = O.ts_random_walk_solve(rp, ci, va, b, 0.5, seed, num_walks=2, per_walk_streams=True)
        assert blocks["x"][0] == (vals[0] + vals[1]) / 2
        # the serial form's coordinate 0: the same first walk, then the second from wherever the first stopped — replay it from the LCG
        serial = O.ts_random_walk_solve(rp, ci, va, b, 0.5, seed, num_walks=2, per_walk_streams=False)
        first_walk_draws = None
        for used in range(1, 2001):                                        # the second walk of the serial form starts `used` draws in
            if (O.ts_random_walk_streams(rp, ci, va, b, 0, 1, O.ts_lcg_jump(seed, used))[0][0] + vals[0]) / 2 == serial["x"][0]:
                first_walk_draws = used
                break
        assert first_walk_draws is not None and first_walk_draws <= 2000


def test_block_form_and_serial_form_agree_within_monte_carlo_error():
    rp, ci, va, b = _system(n=40)
    W = 2500
    a = O.ts_random_walk_solve(rp, ci, va, b, 0.02, 7, per_walk_streams=True)
    s = O.ts_random_walk_solve(rp, ci, va, b, 0.02, 7, per_walk_streams=False)
    z = np.abs(a["x"] - s["x"]) / np.sqrt((a["variances"] + s["variances"]) / W)
    assert z.max() < 5.0 and np.mean(z < 2.0) > 0.85, (z.max(), np.mean(z < 2.0))
    # the verdict rule: residual < epsilon, else ConvergenceFailure with everything filled (solver.ts:334-341)
    assert a["status"] == 3 and not a["converged"] and a["residual"] >= 0.02 and a["total_variance"] == sum(a["variances"].tolist())
    # coordinates are the single-entry estimates seeded where their walks begin in the stream
    for i in (0, 13, 39):
        _, m, v = O.ts_random_walk_streams(rp, ci, va, b, i, W, O.ts_lcg_jump(7, i * W * 2048))
        assert m == a["x"][i] and v == a["variances"][i]


def test_a_diagonal_system_is_solved_exactly_by_every_walk():
    """no off-diagonal weight: a walk that is not absorbed at once finds sum == 0 and returns b_i / a_ii all the same (solver.ts:404-412)"""
    d = np.array([4.0, -5.0, 8.0, 0.5])
    rp, ci, va = O.csr_from_triplets(range(4), range(4), d, 4, 4)
    r = O.ts_random_walk_solve(rp, ci, va, [1.0, 2.0, 3.0, 4.0], 0.1, 1, per_walk_streams=True)
    want = np.array([1.0, 2.0, 3.0, 4.0]) * (1.0 / d)                  # b[cur] * absorptionProbs[cur], absorptionProbs = 1 / a_ii
    assert r["status"] == 0 and r["converged"] and np.abs(r["x"] - want).max() <= 4e-16 * np.abs(want).max() and r["residual"] < 1e-14 and r["total_variance"] < 1e-28
