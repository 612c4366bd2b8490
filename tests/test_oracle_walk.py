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


# ---- G10 / G11: the reference's own TypeScript walk code executed on these inputs (tests/golden/make_golden_walk.py) -----------------------
GOLDEN_WALK = Path(__file__).resolve().parent / "golden" / "reference_walk.npz"


def golden_walk_cases():
    g = np.load(GOLDEN_WALK)
    for k in (str(x) for x in g["names"]):
        n, row, seed = (int(v) for v in g[k + "/params"])
        rp, ci, va = O.csr_from_triplets(g[k + "/rows"], g[k + "/cols"], g[k + "/values"], n, n)
        yield k, g, dict(n=n, row=row, seed=seed, eps=float(g[k + "/epsilon"][0]), rp=rp, ci=ci, va=va, b=g[k + "/b"])


def test_serial_forms_equal_the_executed_reference_bit_for_bit():
    """the oracle's serial forms — what SL_WALK_STREAM_SERIAL must equal — against the numbers the reference's createSeededRandom /
    createTransitionMatrix / performRandomWalk and its own reduce() calls produced: every walk value, mean, variance, x_i, totalVariance"""
    seen = set()
    for k, g, c in golden_walk_cases():
        if k + "/estimates" in g:
            want = g[k + "/estimates"]
            vals, mean, var = O.ts_random_walk_serial(c["rp"], c["ci"], c["va"], c["b"], c["row"], want.size, c["seed"])
            assert (vals.view(np.uint64) == want.view(np.uint64)).all(), k
            assert np.unique(want).size > 10, "a fixture whose walks all agree pins nothing"
            assert (mean, var) == tuple(g[k + "/mean_variance"]), k
            assert O.ts_random_walk_estimate(c["rp"], c["ci"], c["va"], c["b"], c["row"], c["eps"], c["seed"]) == (mean, var, want.size), k   # the count rule too
            seen.add("estimate")
        else:
            r = O.ts_random_walk_solve(c["rp"], c["ci"], c["va"], c["b"], c["eps"], c["seed"], per_walk_streams=False)
            assert (r["x"].view(np.uint64) == g[k + "/solution"].view(np.uint64)).all(), k
            assert (r["variances"].view(np.uint64) == g[k + "/variances"].view(np.uint64)).all(), k
            assert (r["total_variance"], r["residual"]) == tuple(g[k + "/total_variance_residual"]), k
            seen.add("solve")
    assert seen == {"estimate", "solve"}


def test_block_stride_follows_the_number_of_walks_of_a_call():
    """2048-draw blocks give 2^21 starting points in a generator of period 2^32: a call with more walks takes narrower blocks, so that walk
    s and walk s + 2^21 never read the same draws (with a fixed stride they were bit-identical copies)"""
    assert O.walk_stride(1) == O.walk_stride(1 << 21) == 2048
    assert O.walk_stride((1 << 21) + 1) == 1024 and O.walk_stride(1 << 22) == 1024 and O.walk_stride(10 ** 7) == 256
    assert O.walk_stride(1 << 28) == 16 and O.walk_stride(1 << 40) == 16          # (the device refuses more than 2^28 walks per call)
    for total in (1 << 21, (1 << 21) + 1, 5 * 10 ** 6, 1 << 26, 1 << 28):
        assert total * O.walk_stride(total) <= 1 << 32                            # every walk of the call starts at its own position
    # the failure the fixed stride had: the states of walk s and walk s + 2^21 coincide at stride 2048, and no longer at the call's stride
    seed, s = 42, 12345
    assert O.ts_lcg_jump(seed, s * 2048) == O.ts_lcg_jump(seed, (s + (1 << 21)) * 2048)
    st = O.walk_stride(1 << 22)
    assert O.ts_lcg_jump(seed, s * st) != O.ts_lcg_jump(seed, (s + (1 << 21)) * st)
    starts = {O.ts_lcg_jump(seed, w * st) for w in range(0, 1 << 22, 4099)}
    assert len(starts) == len(range(0, 1 << 22, 4099))
