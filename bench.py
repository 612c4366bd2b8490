#!/usr/bin/env python3
"""bench.py — headline benchmark: fused push/Neumann iterations on S-DD(n = 10M per GPU, 16 nnz/row), SURVEY §8(d)'s input:
uniformly random columns (`--bandwidth 0`, the default).  The same step on the banded variant of the recipe (half-width 4096, the
"locality" form of config 5) is measured in the same run and reported with its own full roofline block (`roofline_banded`).

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its own ranks, or runs under torch.distributed.run)

A "step" is one pass of the hot path over the whole (synthetic) system: the fused kernel
    t' = t - dinv .* (A t);  x += t';  ||t'||^2
(NeumannState::apply_iteration_matrix + compute_next_term, src/solver/neumann.rs:252-299), inputs
resident in HBM before the timed region.  value = nnz * K * N / max-over-ranks time.
Weak scaling: every rank owns `--n` rows of an N*n-row system (row-range partition, one exchange of the term vector per step).
N > 1 runs through the library's own communicator by default (`--exchange abi`: sl_comm + partitioned NeumannState behind the C ABI —
IPC-mapped vectors pulled over xGMI, the norm summed over all ranks every step); `--exchange p2p | allreduce` runs the exchange over
torch.distributed / RCCL instead (grouped send/recv of the halo strips, or the all-reduce form; all-gather for uniform columns).

Before the timed region every rank passes a PARITY GATE: two fused steps on the instance it is about to time, sampled
4096-row blocks of the term and the solution read back through the ABI and compared bit for bit, in a CPU child process, with the
reference's arithmetic over rows regenerated from the counter-based generator (`parity_gate` in the line; a gate that fails prints
`value: null`).  N > 1 lines also carry `scaling_reference`: rank 0 alone on the whole one-GPU system and on its own slice, same job.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline`, `cpu_baseline` (N = 1) and `parity_gate`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# the ranks of a fresh box reach their rendezvous tens of seconds apart (the first `import torch` pages the image in): the
# communicator's bounded waits get room for that; a rank that DIES is noticed at once all the same (marked communicator / the parent)
os.environ.setdefault("SL_COMM_TIMEOUT_MS", "180000")

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling
L2_REQUEST_RATE = 2.7e11       # L2 requests/s the eight XCDs serve together: 34.5 TB/s / 128 B, = the 265-280 G gathers/s tools/gather_bench.hip measures on L2-resident tables


LAYOUTS = {4: "order-free column stream: a CU's entries sorted by column, sums by LDS atomics (SL_ORDER_ANY; gathers from L2)",
           0: "row slices", 1: "column panels, dynamic tiles (gathers from L2)", 2: "column panels, paced persistent blocks (gathers from L2)",
           3: "narrow column panels over block-local rows, paced persistent blocks (wide band: gathers from L1)"}


def recorded_traffic(n, k, w):
    """HBM-side bytes per step of the step kernel from the committed rocprofv3 --pmc passes (profiles/pmc_traffic.json, written by
    tools/prof_summary.py) — a RECORDED figure of an earlier profiled run of this configuration, not a measurement of this run."""
    tf = ROOT / "profiles" / "pmc_traffic.json"
    try:
        rec = json.loads(tf.read_text())
        recs = rec.get("records", {"x": rec})
        import hashlib
        now = hashlib.sha256((ROOT / "sublinear_time_solver_amd" / "csrc" / "sl_kernels.hip").read_bytes()).hexdigest()[:16]
        stale = None
        for r in recs.values():
            if r.get("n") == n and r.get("k") == k and r.get("bandwidth") == w:
                if r.get("kernel_source_sha16") != now:      # counters of another kernel are not this kernel's traffic ("stale by construction")
                    stale = (f"profiles/pmc_traffic.json [{r.get('tag')}] was measured on sl_kernels.hip {r.get('kernel_source_sha16')} (commit {r.get('commit')}); the file is "
                             f"{now} now: traffic not reported — re-profile with `bash tools/profile.sh <tag> --bandwidth {w}`")
                    continue
                return r.get("hbm_bytes_per_step", r.get("hbm_bytes_per_launch")), (f"profiles/pmc_traffic.json [{r.get('tag')}, commit {r.get('commit')}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                                                   "passes of an earlier run of this configuration on this very kernel source; not measured in this run)")
        if stale:
            return None, stale
    except Exception:
        pass
    return None, None


def algorithmic_bytes(n_rows: int, nnz: int) -> int:
    """DESIGN.md §4 / SURVEY.md §8(d): 12 B per entry + 4 B row descriptor + 40 B of vectors per row."""
    return 12 * nnz + 4 * (n_rows + 1) + 40 * n_rows


def cpu_baseline(n_global: int, k: int, seed: int, w: int, threads_all: int, rows: int = 0):
    """The reference's CPU hot loop (oracle restatement of simd_ops.rs SpMV variants inside the Neumann step, a8 + a9) on the WHOLE
    system (rows = 0) or on its first `rows` rows against the full-length vector, with the time split SpMV / vector passes: the
    reference's vector passes are serial loops (neumann.rs:289-296, 264-266) beside its threaded SpMV.  A third,
    clearly labelled leg threads those passes too (not the reference).  The input is the same S-DD system, synthesised by the
    library's generator on the GPU when there is one (1.9 GB down the PCIe) and by generators.sdd_rows otherwise."""
    import numpy as np
    from oracle import oracle as O
    from sublinear_time_solver_amd import generators as G

    O.build(fast=True)
    n_s = n_global if rows <= 0 else min(n_global, rows)
    rp = ci = va = None
    try:
        import torch
        from sublinear_time_solver_amd import _lib as L
        if torch.cuda.is_available():
            lib, dev = L.load(), torch.device("cuda", 0)
            d_rp = torch.empty(n_s + 1, dtype=torch.int32, device=dev); d_ci = torch.empty(n_s * k, dtype=torch.int32, device=dev)
            d_va = torch.empty(n_s * k, dtype=torch.float64, device=dev); d_b = torch.empty(n_s, dtype=torch.float64, device=dev)
            L.check(lib.sl_synth_sdd_device(n_global, k, seed, w, 0, n_s, d_rp.data_ptr(), d_ci.data_ptr(), d_va.data_ptr(), d_b.data_ptr()))
            rp, ci, va = d_rp.cpu().numpy().view(np.uint32), d_ci.cpu().numpy().view(np.uint32), d_va.cpu().numpy()
            del d_rp, d_ci, d_va, d_b
            torch.cuda.empty_cache()
    except Exception:
        rp = None
    if rp is None:
        parts = [G.sdd_rows(n_global, k, seed, w, lo, min(n_s, lo + 1_000_000)) for lo in range(0, n_s, 1_000_000)]
        rp = np.concatenate([[0]] + [p[0][1:].astype(np.int64) + i * 1_000_000 * k for i, p in enumerate(parts)]).astype(np.uint32)
        ci = np.concatenate([p[1] for p in parts]); va = np.concatenate([p[2] for p in parts])
    dinv = 1.0 / (10.0 + 0.01 * (np.arange(n_s) % 1000))
    t_init = 1.0 + 0.001 * (np.arange(n_global) % 1000)
    t_init[:n_s] *= dinv
    out, split = {}, {}
    legs = (("simd4_1t", O.ORDER_SIMD4, 1, False), ("rowchunk_all", O.ORDER_SEQ, threads_all, False), ("rowchunk_all_parallel_passes", O.ORDER_SEQ, threads_all, True))
    for label, order, threads, par in legs:
        t, x = t_init.copy(), t_init[:n_s].copy()
        O.neumann_steps_split(rp, ci, va, dinv, t, x, 1, order, threads, par, fast=True)          # warm (page faults, thread pool)
        done, dt, sp, ve = 0, 0.0, 0.0, 0.0
        steps = 2 if n_s >= 4_000_000 else 16
        while dt < 4.0 and done < 8192:                                                # ~4 s of CPU work per leg
            t[:] = t_init                                                              # restart the series: no denormal tail
            x[:] = t_init[:n_s]
            t0 = time.perf_counter()
            _, a_, b_ = O.neumann_steps_split(rp, ci, va, dinv, t, x, steps, order, threads, par, fast=True)
            dt += time.perf_counter() - t0
            sp += a_; ve += b_
            done += steps
        out[label] = n_s * k * done / dt
        split[label] = {"spmv_s_per_step": sp / done, "vector_passes_s_per_step": ve / done}
    best = max(("simd4_1t", "rowchunk_all"), key=out.get)       # the faithful legs only
    whole = "the WHOLE system" if n_s == n_global else f"rows [0,{n_s}) of the same system (gathered vector full length {n_global})"
    return {"value": out[best], "unit": "nnz*iter/s", "cores": threads_all if best == "rowchunk_all" else 1,
            "kind": "port",
            "sample": f"{whole}: S-DD(n={n_global}, k={k}, seed={seed}), a8+a9 steps; simd_ops.rs 4-lane SpMV 1 thread = {out['simd4_1t']:.3e}, "
                      f"row-chunk threads x{threads_all} = {out['rowchunk_all']:.3e} nnz*iter/s; per step of the all-thread leg: SpMV "
                      f"{split['rowchunk_all']['spmv_s_per_step']:.3f} s (random gathers over the host's memory) + the reference's serial vector passes "
                      f"{split['rowchunk_all']['vector_passes_s_per_step']:.3f} s",
            "single_thread_simd4": out["simd4_1t"], "all_threads_rowchunk": out["rowchunk_all"], "time_split": split,
            "port_plus_parallel_passes": {"value": out["rowchunk_all_parallel_passes"], "cores": threads_all,
                                          "note": "NOT the reference: its vector passes threaded like its SpMV (simd_ops.rs:219 chunks) — what the same cores give once those loops are parallel"}}


GATE_BLOCK = 4096          # rows per sampled block of the parity gate
GATE_STEPS = 2             # fused steps before the blocks are read: the second one gathers what the FIRST one wrote (on a partition: what the exchange moved)


def gate_block_starts(n_local):
    """three blocks of a rank's rows: its first rows, a block in the middle, its last rows (the edges are what gathers from the peers)"""
    if n_local <= GATE_BLOCK:
        return [0]
    starts = {0, n_local - GATE_BLOCK}
    if n_local >= 3 * GATE_BLOCK:
        starts.add(n_local // 2 - 37)          # not aligned to slices, tiles or row groups
    return sorted(starts)


def cpu_parity_gate(path):
    """The checker side of the parity gate (part of bench.py's CPU leg: runs in its own process, never inside the timed region).
    `path` holds what one rank read back after GATE_STEPS fused steps from t0 = D^-1 b, x0 = t0: the rows of sampled blocks of its
    current term and of its solution.  Regenerates on the host — from the counter-based generator alone — the sampled rows, the rows
    those gather from and the rows THOSE gather from, runs the reference's arithmetic over them with the oracle's SpMV (sparse.rs:187-203
    order: product rounded, then added, left to right; neumann.rs:289-296: tmp *= dinv, term -= tmp; :264-266: x += term), and compares bits."""
    import numpy as np
    from oracle import oracle as O
    from sublinear_time_solver_amd import generators as G
    z = np.load(path)
    n_global, k, seed, w, lo, order = (int(z[key]) for key in ("n_global", "k", "seed", "w", "lo", "order"))
    starts, t_dev, x_dev = z["starts"], z["term"], z["x"]
    steps = int(z["steps"])
    assert steps == 2

    def closed_t0(idx):       # t0 = rhs = b * (1 / d), the generator's closed forms of b and the diagonal
        m = (idx % np.uint64(1000)).astype(np.float64)
        return (1.0 + 0.001 * m) * (1.0 / (10.0 + 0.01 * m)), 1.0 / (10.0 + 0.01 * m)

    def step_rows(rows, t_of):       # one fused step on `rows` given a function idx -> previous term; returns (new term rows, cols used)
        rp, ci, va, _ = G.sdd_rows_at(n_global, k, seed, w, rows)
        cols = np.unique(ci).astype(np.uint64)
        y = O.spmv(rp, np.searchsorted(cols, ci).astype(np.uint32), va, t_of(cols), order)     # a monotone renumbering keeps every row's order
        return rp, ci, va, cols, y

    rows_checked, bad, max_rel = 0, 0, 0.0
    for bi, s0 in enumerate(starts.tolist()):
        cnt = min(GATE_BLOCK, t_dev.shape[1])
        R = np.arange(lo + s0, lo + s0 + cnt, dtype=np.uint64)
        # step 1 on every row the block gathers from (its own rows included: the diagonal)
        _, ciR, _, U1, _ = step_rows(R, lambda c: closed_t0(c)[0])
        _, _, _, _, yU = step_rows(U1, lambda c: closed_t0(c)[0])
        t0U, dinvU = closed_t0(U1)
        t1U = t0U - yU * dinvU
        # step 2 on the block itself
        rpR, ciR, vaR, colsR, yR = step_rows(R, lambda c: t1U[np.searchsorted(U1, c)])
        pos = np.searchsorted(U1, R)
        t0R, dinvR = closed_t0(R)
        t2 = t1U[pos] - yR * dinvR
        xe = (t0R + t1U[pos]) + t2
        td, xd = t_dev[bi, :cnt], x_dev[bi, :cnt]
        bad += int((td.view(np.uint64) != t2.view(np.uint64)).sum()) + int((xd.view(np.uint64) != xe.view(np.uint64)).sum())
        max_rel = max(max_rel, float(np.max(np.abs(td - t2)) / np.max(np.abs(t2))), float(np.max(np.abs(xd - xe)) / np.max(np.abs(xe))))
        rows_checked += cnt
    return {"rows_checked": rows_checked, "bitwise_equal": bad == 0, "values_differing": bad, "max_rel_err": max_rel,
            "steps": steps, "blocks_at_local_rows": starts.tolist()}


def parity_gate(args, lib, L, st, n_local, n_global, lo, w):
    """BEFORE the timed region: GATE_STEPS fused steps on the instance the bench is about to time, three blocks of GATE_BLOCK rows of the
    term and of the solution read back through the ABI, checked bit for bit by the CPU checker in a child process; the state is reset
    afterwards (SolverState::reset, neumann.rs:367-378: current_term = rhs, solution = 0, counters 0): the timed steps start from the
    first term as they always did.  The solution then starts from 0 instead of the state's x0 — its VALUES do not enter the timing
    (x is read and written once per step whatever it holds) and the gate has already compared them."""
    import subprocess
    import tempfile
    import numpy as np
    if args.order not in (L.SL_ORDER_CSR_SEQUENTIAL, L.SL_ORDER_SIMD4):
        return {"skipped": "order-relaxed mode: results to rounding, no bitwise gate"}
    starts = gate_block_starts(n_local)
    cnt = min(GATE_BLOCK, n_local)
    nrm, ms = C.c_double(0.0), C.c_float(0.0)
    L.check(lib.sl_neumann_state_run_steps(st, GATE_STEPS, C.byref(nrm), C.byref(ms)))
    term, x = np.empty((len(starts), cnt)), np.empty((len(starts), cnt))
    for i, s0 in enumerate(starts):
        L.check(lib.sl_neumann_state_current_term(st, s0, cnt, term[i].ctypes.data, L.SL_MEM_HOST))
        L.check(lib.sl_neumann_state_solution_rows(st, s0, cnt, x[i].ctypes.data, L.SL_MEM_HOST))
    L.check(lib.sl_neumann_state_reset(st))
    return run_gate_checker(args, n_global, lo, w, starts, term, x)


def run_gate_checker(args, n_global, lo, w, starts, term, x):
    """the blocks a rank read back -> the CPU checker in a child process -> its verdict"""
    import subprocess
    import tempfile
    import numpy as np
    if os.environ.get("SL_BENCH_GATE_CORRUPT") == "1":        # test of the gate itself: one ulp in one value it read back
        term.view(np.uint64)[0, 5] ^= 1
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "gate.npz")
        np.savez(f, n_global=n_global, k=args.k, seed=args.seed, w=w, lo=lo, order=args.order, steps=GATE_STEPS, starts=np.asarray(starts, dtype=np.int64), term=term, x=x)
        try:
            r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--parity-gate-only", f], capture_output=True, text=True, timeout=600)
            if r.returncode != 0:
                raise RuntimeError(f"checker exited with {r.returncode}: {r.stderr[-400:]}")
            return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:
            return {"rows_checked": 0, "bitwise_equal": False, "error": str(e)}


def parity_gate_torch(args, torch, drv, part, n_global, w, t0_full, reduce_norm):
    """the same gate for the exchange ABOVE the ABI (main_torch: --exchange p2p | allreduce, the last fallback of N > 1): two steps of the
    torch.distributed driver, the same blocks read from its tensors, the same checker; then the driver starts over from t0"""
    import numpy as np
    if args.order not in (0, 1):
        return {"skipped": "order-relaxed mode: results to rounding, no bitwise gate"}
    n_local = part.n_local
    starts = gate_block_starts(n_local)
    cnt = min(GATE_BLOCK, n_local)
    t_start = t0_full.clone()
    for _ in range(GATE_STEPS):
        drv.step(reduce_norm)
    torch.cuda.synchronize()
    tc = drv.t[drv.cur]
    term = np.stack([tc[part.lo + s0:part.lo + s0 + cnt].cpu().numpy() for s0 in starts])
    x = np.stack([drv.x[s0:s0 + cnt].cpu().numpy() for s0 in starts])
    drv.restart(t_start, t_start[part.lo:part.hi])
    del t_start
    return run_gate_checker(args, n_global, part.lo, w, starts, term, x)


def single_rank_reference(args, lib, L, torch, dev, n_global, lo, hi, w, steps):
    """ms per fused step of ONE process alone on its GPU, no communicator, no peers' traffic: rows [lo, hi) of the n_global-row system
    against the full-length gathered vector (hi - lo = n_global: the whole system, ping-pong; otherwise one rank's slice of the
    partition, the gathered vector standing still — what the step costs before any exchange)."""
    k, rows = args.k, hi - lo
    rp = torch.empty(rows + 1, dtype=torch.int32, device=dev); ci = torch.empty(rows * k, dtype=torch.int32, device=dev)
    va = torch.empty(rows * k, dtype=torch.float64, device=dev); b = torch.empty(rows, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n_global, k, args.seed, w, lo, hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(rows, n_global, rows * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, lo, 0, C.byref(h)))
    del rp, ci, va
    torch.cuda.empty_cache()
    try:
        dinv = torch.empty(rows, dtype=torch.float64, device=dev)
        L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
        nrm = torch.zeros(2, dtype=torch.float64, device=dev)
        if rows == n_global:
            ta = b * dinv
            tb, x = torch.empty_like(ta), ta.clone()
            ms = C.c_float(0)
            L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), args.order, 5, C.byref(ms)))
            reps = []
            for _ in range(2):      # twice, the smaller figure: the peers' last state is still being unmapped when this starts (seen with ranks sharing a device)
                L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), tb.data_ptr(), ta.data_ptr(), x.data_ptr(), nrm.data_ptr(), args.order, steps, C.byref(ms)))
                reps.append(ms.value / steps)
            return min(reps)
        idx = torch.arange(n_global, device=dev, dtype=torch.float64)
        ta = (1.0 + 0.001 * torch.remainder(idx, 1000.0)) * (1.0 / (10.0 + 0.01 * torch.remainder(idx, 1000.0)))
        del idx
        tb, x = torch.zeros(rows, dtype=torch.float64, device=dev), ta[lo:hi].clone()
        stream = torch.cuda.current_stream(dev)
        L.check(lib.sl_set_stream(C.c_void_p(stream.cuda_stream)))
        try:
            for _ in range(5):
                L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), args.order))
            reps = []
            for _ in range(2):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(steps):
                    L.check(lib.sl_neumann_step(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), args.order))
                e1.record(stream)
                torch.cuda.synchronize(dev)
                reps.append(e0.elapsed_time(e1) / steps)
            return min(reps)
        finally:
            L.check(lib.sl_set_stream(None))
    finally:
        lib.sl_matrix_destroy(h)
        torch.cuda.empty_cache()


def column_structure_sweep(lib, L, torch, dev, n, k, seed, order, bandwidths, steps=30):
    """Secondary, clearly-labelled measurements on ONE GPU: the same fused step on the same S-DD recipe with
    other column structures (half-bandwidth w; 0 = uniform over all columns, the reference generators' recipe).
    Reported next to the headline so the dependence on gather locality is visible."""
    out = {}
    for w in bandwidths:
        rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
        ci = torch.empty(n * k, dtype=torch.int32, device=dev)
        va = torch.empty(n * k, dtype=torch.float64, device=dev)
        b = torch.empty(n, dtype=torch.float64, device=dev)
        L.check(lib.sl_synth_sdd_device(n, k, seed, w, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
        h = C.c_void_p()
        L.check(lib.sl_matrix_create_csr(n, n, n * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, 0, C.byref(h)))
        del rp, ci, va
        dinv = torch.empty(n, dtype=torch.float64, device=dev)
        L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
        ta = b * dinv
        tb = torch.empty_like(ta)
        x = ta.clone()
        nrm = torch.zeros(2, dtype=torch.float64, device=dev)
        ms = C.c_float(0)
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm.data_ptr(), order, 3, C.byref(ms)))
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), tb.data_ptr(), ta.data_ptr(), x.data_ptr(), nrm.data_ptr(), order, steps, C.byref(ms)))
        per = ms.value / steps
        mi = L.MatrixInfo()
        L.check(lib.sl_matrix_get_info(h, C.byref(mi)))
        ach = algorithmic_bytes(n, n * k) / (per * 1e-3) / 1e9
        out["uniform" if w == 0 else f"w{w}"] = {"ms_per_step": per, "nnz_iter_per_s": n * k / (per * 1e-3), "achieved_GBps": ach,
                                                   "roofline_frac": ach / HBM_PEAK_GBS,
                                                   "layout": LAYOUTS[int(mi.column_panels)] if mi.column_panels else "row slices"}
        lib.sl_matrix_destroy(h)
        del dinv, ta, tb, x, b
        torch.cuda.empty_cache()
    return out


def _abi_measure(args, lib, L, torch, dev, comm, world, rank, w, verify=True):
    """One column structure through the C ABI: this rank's rows synthesized in HBM, row-slice / panel layouts built, partitioned
    NeumannState over `comm`, W warm-up steps, then K timed steps bracketed by comm.barrier() + torch.cuda.synchronize() on both
    sides; MAX over ranks of wall time and of the library's own HIP-event time of the loop."""
    n_local, k = args.n, args.k
    n_global = n_local * world
    lo, hi = rank * n_local, (rank + 1) * n_local
    rp = torch.empty(n_local + 1, dtype=torch.int32, device=dev)
    ci = torch.empty(n_local * k, dtype=torch.int32, device=dev)
    va = torch.empty(n_local * k, dtype=torch.float64, device=dev)
    bb = torch.empty(n_local, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n_global, k, args.seed, w, lo, hi, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), bb.data_ptr()))
    h = C.c_void_p()
    L.check(lib.sl_matrix_create_csr(n_local, n_global, n_local * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, lo,
                                     L.SL_MATRIX_ORDER_ANY if args.order == L.SL_ORDER_ANY else 0, C.byref(h)))
    del rp, ci, va
    torch.cuda.empty_cache()
    info = L.MatrixInfo()
    L.check(lib.sl_matrix_get_info(h, C.byref(info)))
    o = L.NeumannOptions()
    lib.sl_neumann_options_default(C.byref(o))
    o.order, o.mem, o.start = args.order, L.SL_MEM_DEVICE, L.SL_START_REFERENCE_DEFAULT      # x0 = D^-1 b: x = t0, the bench loop's start
    st = C.c_void_p()
    L.check(lib.sl_neumann_state_create_partitioned(comm._h, h, bb.data_ptr(), None, C.byref(o), C.byref(st)))
    out = {}
    try:
        nrm, ms = C.c_double(0.0), C.c_float(0.0)
        gate = None
        if not args.no_parity_gate:       # every rank checks blocks of ITS rows; the verdicts are gathered below
            gate = parity_gate(args, lib, L, st, n_local, n_global, lo, w)
        if args.warmup:
            L.check(lib.sl_neumann_state_run_steps(st, args.warmup, C.byref(nrm), C.byref(ms)))
        comm.barrier()
        torch.cuda.synchronize(dev)
        t_start = time.perf_counter()
        L.check(lib.sl_neumann_state_run_steps(st, args.steps, C.byref(nrm), C.byref(ms)))     # returns when the K steps are done on this rank
        torch.cuda.synchronize(dev)
        comm.barrier()
        elapsed = max(comm.allgather_f64(time.perf_counter() - t_start))
        dev_ms = max(comm.allgather_f64(float(ms.value)))
        bad = C.c_uint64(0)
        if verify:      # did the transport move what the owners wrote?  (position-weighted checksums of every piece against its owner's copy)
            L.check(lib.sl_neumann_state_verify_exchange(st, C.byref(bad)))
        bad_all = sum(comm.allgather_u64(int(bad.value)))
        out = {"elapsed": elapsed, "dev_ms": dev_ms, "norm": float(nrm.value) ** 0.5, "panels": int(info.column_panels), "pieces_bad": bad_all,
               "verified": bool(verify)}
        if gate is not None and "skipped" not in gate:
            rows_all = sum(comm.allgather_u64(int(gate.get("rows_checked", 0))))
            ranks_ok = sum(comm.allgather_u64(1 if gate.get("bitwise_equal") else 0))
            worst = max(comm.allgather_f64(float(gate.get("max_rel_err", float("inf")))))
            out["gate"] = {"rows_checked": rows_all, "bitwise_equal": ranks_ok == world, "max_rel_err": worst, "ranks_checked": world, "ranks_equal": ranks_ok,
                           "steps_before_check": GATE_STEPS, "blocks_per_rank": len(gate_block_starts(n_local)), "block_rows": min(GATE_BLOCK, n_local),
                           "checker": "oracle SpMV (sparse.rs:187-203 order) over rows regenerated on the host by generators.sdd_rows_at, in a child process before the timed region"}
            if gate.get("error"):
                out["gate"]["error"] = gate["error"]
        elif gate is not None:
            out["gate"] = gate
        if world == 1:      # kernel-only duration: the same launches (fused step + closing reduction) through the library's HIP-event bracket, no ticket
            dinv = torch.empty(n_local, dtype=torch.float64, device=dev)
            L.check(lib.sl_matrix_diagonal_inverse(h, dinv.data_ptr(), L.SL_MEM_DEVICE))
            ta = bb * dinv
            tb, x = torch.empty_like(ta), ta.clone()
            nrm2 = torch.zeros(2, dtype=torch.float64, device=dev)
            km = C.c_float(0)
            L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), nrm2.data_ptr(), args.order, 3, C.byref(km)))
            L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), tb.data_ptr(), ta.data_ptr(), x.data_ptr(), nrm2.data_ptr(), args.order, args.steps, C.byref(km)))
            out["kern_ms"] = km.value / args.steps
            del dinv, ta, tb, x, nrm2
    finally:
        lib.sl_neumann_state_destroy(st)
        lib.sl_matrix_destroy(h)
        del bb
        torch.cuda.empty_cache()
    return out


def main_abi(args, world, rank, local_rank, attempt=0, emit=True):
    """Returns rank 0's line (None on the other ranks); prints it when `emit` (a caller that measures several exchange variants in one
    job merges them first: exchange_variants below).
    Every N (1 included) through the C ABI alone: sl_comm + partitioned NeumannState; torch only provides the device buffers the
    generator writes into and the device synchronisation.  Barrier = sl_comm_barrier (drains the stream, then all ranks meet), MAX
    over ranks through sl_comm_allgather.  The transport of the vectors and sums is the library's (SL_COMM_TRANSPORT: ipc | rccl)."""
    import torch
    from sublinear_time_solver_amd import _lib as L
    from sublinear_time_solver_amd import Communicator
    ndev = max(1, torch.cuda.device_count())
    local_rank %= ndev                                       # fewer devices than ranks (test boxes): ranks share a GPU, which the ipc transport supports
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = L.load()
    L.check(lib.sl_set_device(local_rank))
    name = f"bench_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', os.environ.get('SL_BENCH_JOB', 'x'))}_{attempt}"
    comm = Communicator(rank, world, name[:60].replace("/", "_"))
    try:
        ci_ = L.CommInfo()
        L.check(lib.sl_comm_info(comm._h, C.byref(ci_)))
        devices = comm.allgather_u64(int(ci_.device))
        transport = ("rccl" + ("+halo-allreduce" if ci_.halo_allreduce else "")) if ci_.transport == 1 else "ipc"
        n_local, k = args.n, args.k
        n_global = n_local * world
        # N = 1: BASELINE configs[2], uniform columns.  N > 1: BASELINE configs[4] in the form north_star and SURVEY 8(e) give it — the halo
        # only travels: columns within w = n / (4 P) = n_local / 4 of the row (gathers spread over 40 MB of the vector at n_local = 10^7,
        # far beyond any cache: the same kernel class and, at N = 1, the same time per step as uniform columns — 0.940 against 0.944 ms,
        # profiles/r03_c5_locality.txt); the uniform-column form (every rank needs every row: an all-gather) and the narrow band are
        # measured in the same job and reported beside it
        w_c5 = max(n_local // 4, 1)
        w_head = args.bandwidth if args.bandwidth >= 0 else (DEFAULT_BANDWIDTH if world == 1 else w_c5)
        m = _abi_measure(args, lib, L, torch, dev, comm, world, rank, w_head)
        if m["pieces_bad"]:
            raise RuntimeError(f"exchange verification failed: {m['pieces_bad']} pieces differ from their owners' copies (transport {transport})")
        if "gate" in m and "skipped" not in m["gate"] and not m["gate"]["bitwise_equal"]:
            # a fast kernel whose results differ is not measured: the line carries the gate and NO value (every rank sees the same verdict;
            # not a transport failure, so no fallback is tried)
            if rank != 0:
                return None
            failed_line = {"metric": "push_iterations_x_nnz_per_sec", "value": None, "unit": "nnz*iter/s", "n_gpus": len(set(devices)), "n_ranks": world,
                           "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                           "vs_baseline": None, "dtype": "f64", "data": "synthetic", "error": "parity gate failed: the step's results differ from the CPU checker's; nothing was timed",
                           "parity_gate": m["gate"], "config": {"workload": f"S-DD(n={n_local} rows/rank, nnz/row={k}, seed={args.seed}, half-bandwidth {w_head})", "transport": transport}}
            if emit:
                print(json.dumps(failed_line), flush=True)
            return failed_line
        variants = []
        if not args.no_sweep and world > 1:                                        # the other column structures of the recipe in the same job
            for w_other in (0, w_c5, BANDED_BANDWIDTH):
                if w_other == w_head:
                    continue
                mo = _abi_measure(args, lib, L, torch, dev, comm, world, rank, w_other)
                variants.append((w_other, mo))      # (a variant whose exchange does not verify is reported as such — every rank sees the same count — and does not take the headline with it)
        # N > 1: the like-for-like one-GPU reference, measured in THIS job by rank 0 alone while its peers wait at a barrier: (i) the whole
        # n_local-row system with the head's column structure — what `--gpus 1 --bandwidth w_head` times — and (ii) rank 0's own rows of the
        # N * n_local-row system with nobody exchanging.  efficiency = exchange cost; the change of workload between N = 1 and N > 1 is (i) vs (ii)
        ref = None
        if world > 1 and not args.no_scaling_reference:
            comm.barrier()
            torch.cuda.synchronize(dev)
            if rank == 0:
                try:
                    time.sleep(0.5)      # the peers have just released their vectors and mappings
                    rs = max(10, min(args.steps, 30))
                    ref = {"n1_ms_per_step": single_rank_reference(args, lib, L, torch, dev, n_local, 0, n_local, w_head, rs),
                           "slice_ms_per_step": single_rank_reference(args, lib, L, torch, dev, n_global, 0, n_local, w_head, rs), "steps": rs,
                           "what": f"rank 0 alone on its device, the other {world - 1} ranks idle at a barrier: n1 = the whole {n_local}-row system, half-bandwidth {w_head} "
                                   f"(one process, no communicator: sl_neumann_run_steps); slice = rows [0, {n_local}) of the {n_global}-row system against the standing "
                                   "full-length vector (sl_neumann_step): the step before any exchange"}
                except Exception as e:      # reported context, never fatal to the line
                    ref = {"error": str(e)}
            comm.barrier()
        if rank != 0:
            return None
        nnz_total = n_global * k
        per_launch_bytes = algorithmic_bytes(n_local, n_local * k)
        launch_ms = m.get("kern_ms", m["dev_ms"] / args.steps)
        achieved = per_launch_bytes / (launch_ms * 1e-3) / 1e9
        value = nnz_total * args.steps / m["elapsed"]
        name_of = lambda w: ("uniform columns" if w == 0 else
                             f"columns within n/(4P) = {w} of the row (config 5's locality-bounded form, SURVEY 8(e))" if (w == w_c5 and world > 1) else f"band half-width {w}")
        traffic, traffic_source = recorded_traffic(n_local, k, w_head) if world == 1 else (None, None)
        kernel = LAYOUTS.get(m["panels"], "row slices") if m["panels"] else ("LDS-window band kernel" if 0 < w_head <= 9400 else "row-slice general kernel")
        exchange = f"abi: sl_comm, transport {transport}" + (" (one rank: tickets of one, no peers)" if world == 1 else
                   (" — term all-gathered: every rank needs every row" if w_head == 0 else " — halo strips at the range boundaries only"))
        n_dev = len(set(devices))
        out = {
            "metric": "push_iterations_x_nnz_per_sec", "value": value, "unit": "nnz*iter/s", "n_gpus": n_dev, "n_ranks": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["elapsed"] * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"S-DD(n={n_local} rows/rank, nnz/row={k}, seed={args.seed}, {name_of(w_head)}) fused Neumann/push step, fp64, "
                                   + ("1xMI355X HBM roofline run (BASELINE configs[2])" if world == 1 else
                                      f"{world} ranks on {n_dev} device(s)" + (f" = {world}xMI355X" if n_dev == world else " — ranks SHARE a GPU: a functional run of the partitioned path, not a scaling figure")
                                      + ", row-partitioned (BASELINE configs[4] per-GPU shape)"),
                       "n_per_gpu": n_local, "n_global": n_global, "nnz_per_row": k, "half_bandwidth": w_head,
                       "order": {0: "csr_sequential", 1: "simd4", 2: "any (SL_ORDER_ANY)"}[args.order],
                       "exchange": exchange, "transport": transport, "partition": f"rows{world}",
                       "n_ranks_joined": int(ci_.ranks_joined), "devices": devices, "devices_visible": ndev,
                       "exchange_verified": bool(m["verified"] and not m["pieces_bad"]) if world > 1 else None,
                       "norm_allreduce_every": 1 if world > 1 else None, "rows_iter_per_s": value / k, "last_term_norm": m["norm"],
                       "bytes_received_per_rank_per_step": 0 if world == 1 else ((8 * n_local * (world - 1)) if w_head == 0 else 8 * w_head * min(2, world - 1))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "column_structure": "uniform over all columns (SURVEY 8(d) S-DD)" if w_head == 0 else f"band half-width {w_head}",
                         "kernel": kernel, "algorithmic_bytes_per_launch": per_launch_bytes, "launch_ms": launch_ms,
                         "timed_region_device_ms_per_step": m["dev_ms"] / args.steps,      # HIP events around the K timed steps, same stream (world 1: + the ticket of one)
                         "algorithmic_over_copy_ceiling_6290": achieved / 6290.0,
                         "note": ("launch_ms = fused step kernel + closing reduction, HIP events over the same K launches without the ticket" if world == 1 else
                                  "per GPU: one step = fused kernel + all-rank sum + exchange; launch_ms = the slowest rank's device time per step")},
        }
        if "gate" in m:
            out["parity_gate"] = m["gate"]
        if ref is not None:
            out["scaling_reference"] = ref
        if w_head == 0 and m["panels"]:      # the bound this kernel actually meets: one L2 request per gathered entry
            req = n_local * k
            floor_ms = req / L2_REQUEST_RATE * 1e3
            out["roofline"]["l2_request"] = {"bound": "l2 requests (one 128-byte request per 8-byte gather)", "requests_per_launch": req,
                                            "peak_requests_per_s": L2_REQUEST_RATE, "floor_ms": floor_ms, "frac": floor_ms / launch_ms,
                                            "source": "tools/gather_bench.hip: 265-280 G gathers/s from L2-resident tables = 34.5 TB/s / 128 B (profiles/r01_gather_bench.txt, r02_panel2_prototype.txt)"}
            # Which bound operates: with uniformly random columns a CU's rows meet ~0.5 entries per 128-byte line of the vector, so every
            # gather is an L2 request of its own whatever the layout, and the step cannot be faster than the chip's L2 request rate allows
            # — the HBM fraction this column structure can reach follows from that floor, and it is below BASELINE's 0.60 target.
            # (tools/l2_model, gate-checked against profiles/r03_uniform_pmc.txt: the step's L2 misses are already the compulsory ones —
            # the stream once, the vector once per L2 and round; and where XCD-local spans cut the fills by 38 % the time did not move.)
            out["roofline"]["operative_bound"] = "l2_request"
            out["roofline"]["hbm_frac_ceiling_for_this_column_structure"] = (per_launch_bytes / (floor_ms * 1e-3) / 1e9) / HBM_PEAK_GBS
            out["roofline"]["target_note"] = ("BASELINE north_star's 0.60 of the HBM roofline is not reachable on uniformly random columns at this n: the L2 request rate "
                                              "bounds the launch at floor_ms, i.e. hbm_frac_ceiling_for_this_column_structure; the banded structure of the recipe "
                                              "(roofline_banded) is the HBM-bound case")
        for w_o, mo in variants:
            key = "uniform_variant" if w_o == 0 else "halo_variant" if w_o == BANDED_BANDWIDTH else "locality_variant"
            out[key] = {
                "half_bandwidth": w_o, "column_structure": name_of(w_o) + (" — only the strips at the range boundaries travel" if w_o else " — every rank needs every row: all-gather"),
                "value": nnz_total * args.steps / mo["elapsed"], "unit": "nnz*iter/s", "ms_per_step": mo["elapsed"] * 1e3 / args.steps,
                "device_ms_per_step_slowest_rank": mo["dev_ms"] / args.steps, "exchange_verified": bool(mo["verified"] and not mo["pieces_bad"]),
                "roofline_frac_per_gpu": per_launch_bytes / (mo["dev_ms"] / args.steps * 1e-3) / 1e9 / HBM_PEAK_GBS, "last_term_norm": mo["norm"],
                "parity_gate": mo.get("gate"),
                "bytes_received_per_rank_per_step": (8 * n_local * (world - 1)) if w_o == 0 else 8 * w_o * min(2, world - 1)}
        if world == 1 and not args.no_sweep:
            others = [v for v in (0, BANDED_BANDWIDTH, 512, 32768, w_c5) if v != w_head]      # w_c5: the per-GPU structure of the N > 1 lines, on one GPU
            sweep = column_structure_sweep(lib, L, torch, dev, n_local, k, args.seed, args.order, others)
            out["config"]["other_column_structures"] = sweep
            bk = f"w{BANDED_BANDWIDTH}"
            if bk in sweep:     # the banded variant of the recipe: a full roofline block of its own, same run, same box
                tb_, tsrc = recorded_traffic(n_local, k, BANDED_BANDWIDTH)
                out["roofline_banded"] = {"bound": "hbm", "achieved": sweep[bk]["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": sweep[bk]["roofline_frac"], "traffic": tb_, "traffic_source": tsrc,
                                          "column_structure": f"band half-width {BANDED_BANDWIDTH}", "kernel": "LDS-window band kernel",
                                          "algorithmic_bytes_per_launch": per_launch_bytes, "launch_ms": sweep[bk]["ms_per_step"],
                                          "nnz_iter_per_s": sweep[bk]["nnz_iter_per_s"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(n_global, k, args.seed, w_head)
        if emit:
            emit_line(out)
        return out
    finally:
        comm.close()


# ---- N > 1 on one GPU per rank: the SAME line under both exchanges the library has ---------------------------------------------------------
# BASELINE's north_star words the exchange as "RCCL all-reduce over xGMI on the halo residual vector only"; the library's default
# transport pulls the halo strips out of IPC-mapped peer memory instead (same bytes, no collective).  Which is faster is a property of
# the node, so the N > 1 line measures the headline structure under BOTH — `ipc`, and `rccl` with SL_COMM_HALO=allreduce (ONE ncclAllReduce
# over the compact halo buffer) — reports both in `exchange_variants`, and takes the faster for `value`.  n_ranks_joined of the rccl variant
# comes from ncclCommCount (sl_comm_info): the line itself says whether RCCL saw N ranks.
EXCHANGE_VARIANTS = (("ipc", {"SL_COMM_TRANSPORT": "ipc", "SL_COMM_HALO": None}),
                     ("rccl_allreduce", {"SL_COMM_TRANSPORT": "rccl", "SL_COMM_HALO": "allreduce"}))


def apply_variant_env(env_spec, env=None):
    env = os.environ if env is None else env
    for key, val in env_spec.items():
        if val is None:
            env.pop(key, None)
        else:
            env[key] = val


def _measured(line):
    """a line that carries a measurement (or, in a rehearsal under the emulator, would carry one: its figures are removed, not absent)"""
    return bool(line) and not line.get("error") and (line.get("value") is not None or "dry_run" in line)


def variant_summary(line):
    """what exchange_variants holds of one variant's line"""
    cfg = line.get("config", {})
    rl = line.get("roofline", {})
    return {"ok": _measured(line), "value": line.get("value"), "unit": line.get("unit"), "ms_per_step": line.get("ms_per_step"),
            "device_ms_per_step_slowest_rank": rl.get("launch_ms"), "roofline_frac_per_gpu": rl.get("frac"), "transport": cfg.get("transport"),
            "n_ranks_joined": cfg.get("n_ranks_joined"), "exchange": cfg.get("exchange"), "exchange_verified": cfg.get("exchange_verified"),
            "bytes_received_per_rank_per_step": cfg.get("bytes_received_per_rank_per_step"),
            "parity_gate_bitwise_equal": (line.get("parity_gate") or {}).get("bitwise_equal"), "error": line.get("error")}


def merge_exchange_variants(lines, failures):
    """lines: {variant: rank 0's line}; failures: {variant: why}.  The headline is the faster measured variant (its whole line, the secondary
    measurements of the first variant carried over), both are stated in exchange_variants."""
    measured = {k: v for k, v in lines.items() if _measured(v)}
    if not measured:
        head = next((v for v in lines.values() if v), None)
        if head is None:
            return None
    else:
        best = max(measured, key=lambda k: measured[k]["value"] or 0.0)
        head = dict(measured[best])
        full = next((v for v in lines.values() if v and any(key in v for key in ("uniform_variant", "halo_variant", "locality_variant", "scaling_reference"))), None)
        if full is not None and full is not measured[best]:      # the light variant won: the other column structures and the one-GPU reference were measured once, under the first
            for key in ("uniform_variant", "halo_variant", "locality_variant", "scaling_reference"):
                if key in full:
                    head[key] = dict(full[key], measured_under=full.get("config", {}).get("transport")) if isinstance(full[key], dict) else full[key]
        head.setdefault("config", {})["exchange_headline"] = best
    ev = {k: variant_summary(v) for k, v in lines.items() if v}
    for k, why in failures.items():
        ev[k] = {"ok": False, "error": why}
    ev["note"] = ("the headline structure measured under each exchange the library has, same job, same ranks: ipc = halo strips pulled out of IPC-mapped peer memory; "
                  "rccl_allreduce = SL_COMM_TRANSPORT=rccl + SL_COMM_HALO=allreduce, ONE ncclAllReduce over the compact halo buffer (BASELINE north_star's wording); "
                  "value / ms_per_step of the line = the faster variant (config.exchange_headline)")
    head["exchange_variants"] = ev
    return head


def run_cpu_baseline(n_global, k, seed, w):
    """the oracle leg in its own process: a crash of the CPU checker must not take the GPU line with it"""
    import subprocess
    try:
        r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-only", "--rows", str(n_global), "--k", str(k),
                            "--seed", str(seed), "--bandwidth", str(w)], capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            raise RuntimeError(f"child exited with {r.returncode}: {r.stderr[-300:]}")
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    except Exception as e:  # the baseline is reported context; never lose the GPU line over it
        return {"value": None, "unit": "nnz*iter/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


# ---- N > 1 without a launcher: `python bench.py --gpus N` starts its own ranks -------------------------------------------------------
# Precedent: simd_ops::parallel_matrix_vector_multiply (src/simd_ops.rs:201-239) hides its row chunks behind ONE call; here one
# command hides the N processes.  The parent starts one child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
# environment, the contract torch.distributed.run would set up), waits with a time limit, and — being the one place that sees all
# ranks — decides the fallback: the library's ipc transport, then its rccl transport, then torch.distributed over RCCL.  An attempt
# counts when every rank exits 0, rank 0 printed its line, and the exchange verification inside the line is green.
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher(args, argv):
    import signal
    import subprocess
    import tempfile
    n = args.gpus
    ndev = 0
    try:
        from sublinear_time_solver_amd import _lib as L
        cnt = C.c_int(0)
        L.load().sl_device_count(C.byref(cnt))
        ndev = int(cnt.value)
    except Exception as e:
        print(f"[bench launcher] device count unavailable ({e})", file=sys.stderr, flush=True)
    order = [t for t in os.environ.get("SL_BENCH_TRANSPORTS", "ipc,rccl,torch").split(",") if t in ("ipc", "rccl", "torch")]
    if 0 < ndev < n:      # ranks share GPUs (test boxes): RCCL refuses two ranks on one device; only the ipc transport can run
        print(f"[bench launcher] {n} ranks on {ndev} device(s): ranks share GPUs, ipc transport only", file=sys.stderr, flush=True)
        order = [t for t in order if t == "ipc"] or ["ipc"]
    attempts = []
    t_launch = time.time()

    def run_attempt(att, transport, later, extra_env=None):
        """one job of n ranks under `transport`; returns rank 0's line (dict) or None; appends its record to `attempts`"""
        left = args.total_timeout - (time.time() - t_launch)
        if left < 60 and att:      # the whole job stays inside the driver's limit: what is left would not run an attempt
            attempts.append({"transport": transport, "ok": False, "seconds": 0.0, "why": f"not started: {left:.0f} s of the total budget of {args.total_timeout:.0f} s left"})
            return None
        # a later transport always gets its turn: an attempt may take its own limit, and never more than its share of what is left
        limit = max(30.0, min(args.attempt_timeout, left - 90.0 * later))
        port = _free_port()
        job = f"{os.getpid()}_{att}"
        procs, logs = [], []
        t0 = time.time()
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       SL_BENCH_CHILD="1", SL_BENCH_TRANSPORT=transport, SL_BENCH_JOB=job, SL_BENCH_ATTEMPT=str(att))
            env.pop("TORCHELASTIC_RUN_ID", None)
            env.update(extra_env or {})
            err = tempfile.TemporaryFile(mode="w+")
            logs.append(err)
            procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py")] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                          stderr=err, text=True, start_new_session=True, cwd=str(ROOT)))
        deadline, failed, why = t0 + limit, False, ""
        import threading
        buf = []
        th = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
        th.start()
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                failed, why = True, f"rank {[i for i, c in enumerate(codes) if c not in (None, 0)][0]} exited with {[c for c in codes if c not in (None, 0)][0]}"
                break
            if all(c == 0 for c in codes):
                break
            if time.time() > deadline:
                failed, why = True, f"time limit of {limit:.0f} s"
                break
            time.sleep(0.2)
        if failed:
            time.sleep(1.0)            # the peers' bounded waits notice a marked communicator at once; give them a moment to leave by themselves
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)      # the exact process groups started above
                    except ProcessLookupError:
                        pass
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                pass
        th.join(timeout=10)
        out0 = buf[0] if buf else ""
        lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
        tails = []
        for r, f in enumerate(logs):
            f.seek(0)
            txt = f.read()
            f.close()
            if failed or not lines:
                tails.append(f"--- rank {r} stderr tail ---\n{txt[-1500:]}")
            elif r == 0 and txt.strip():
                sys.stderr.write(txt[-4000:])
        rec = {"transport": transport + ("+halo-allreduce" if (extra_env or {}).get("SL_BENCH_HALO") == "allreduce" else ""), "ok": (not failed) and bool(lines),
               "seconds": round(time.time() - t0, 1)}
        if failed or not lines:
            rec["why"] = why or "rank 0 printed no line"
            attempts.append(rec)
            print(f"[bench launcher] attempt {att} (transport {rec['transport']}) failed: {rec['why']}\n" + "\n".join(tails), file=sys.stderr, flush=True)
            return None
        attempts.append(rec)
        return json.loads(lines[-1])

    # one GPU per rank and no transport forced: the headline structure under BOTH exchanges of the library, one line (EXCHANGE_VARIANTS)
    if ndev >= n and "SL_BENCH_TRANSPORTS" not in os.environ and os.environ.get("SL_BENCH_ONE_EXCHANGE") != "1" and not os.environ.get("SL_COMM_TRANSPORT"):
        lines, failures = {}, {}
        first = run_attempt(0, "ipc", 2)
        if first is not None:
            lines["ipc"] = first
        else:
            failures["ipc"] = attempts[-1].get("why", "failed")
        second = run_attempt(1, "rccl", 1, {"SL_BENCH_HALO": "allreduce", **({"SL_BENCH_LIGHT": "1"} if first is not None else {})})
        if second is not None:
            lines["rccl_allreduce"] = second
        else:
            failures["rccl_allreduce"] = attempts[-1].get("why", "failed")
        merged = merge_exchange_variants(lines, failures) if lines else None
        if merged is not None:
            merged.setdefault("config", {})["launcher"] = {"self_launched_ranks": n, "attempts": attempts}
            print(json.dumps(merged), flush=True)
            return 0
        order = [t for t in order if t == "torch"]
    base = len(attempts)
    for i, transport in enumerate(order):
        line = run_attempt(base + i, transport, len(order) - i - 1)
        if line is None:
            continue
        line.setdefault("config", {})["launcher"] = {"self_launched_ranks": n, "attempts": attempts}
        print(json.dumps(line), flush=True)
        return 0
    print(f"[bench launcher] every transport failed: {attempts}", file=sys.stderr, flush=True)
    return 1


def refuse_emulator():
    """bench.py measures an MI355X.  The SIMT emulator of tests/simt/ (the kernels as host fibers, test infrastructure for a container
    without a GPU) exports `simt_counters`; a library that has it is not a device and nothing may be timed on it.  The one exception is a
    REHEARSAL of this file's code path in the CPU suite (SL_BENCH_DRY_RUN=1, tests/test_simt_emulated.py): its plumbing lives with the
    emulator (tests/simt/bench_rehearsal.py: host tensors where this file asks for device ones, and a line with every measured figure
    removed) and is loaded from there only then — nothing of it is part of a measuring run."""
    path = os.environ.get("SUBLINEAR_HIP_LIB")
    if not path:
        return
    try:
        if hasattr(C.CDLL(path), "simt_counters"):
            if os.environ.get("SL_BENCH_DRY_RUN") == "1":
                import importlib.util
                spec = importlib.util.spec_from_file_location("bench_rehearsal", str(ROOT / "tests" / "simt" / "bench_rehearsal.py"))
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.install()
                global REHEARSAL
                REHEARSAL = mod
                return
            print(f"bench.py: {path} is the SIMT emulator (tests only): refusing to measure anything on it", file=sys.stderr, flush=True)
            sys.exit(2)
    except OSError:
        pass


REHEARSAL = None      # tests/simt/bench_rehearsal.py under SL_BENCH_DRY_RUN=1 + the emulator library; None in every measuring run


def emit_line(line):
    """rank 0's ONE JSON line (a rehearsal prints its structure with every measured figure removed)"""
    print(json.dumps(REHEARSAL.dry_run_line(line) if REHEARSAL else line), flush=True)


def main():
    refuse_emulator()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=10_000_000,
                    help="rows per GPU (use --rows under torch.distributed.run: its own parser rejects `--n` as an ambiguous prefix)")
    ap.add_argument("--k", type=int, default=16, help="entries per row (diagonal included)")
    ap.add_argument("--bandwidth", type=int, default=-1, help="half bandwidth w of the column window; 0 = uniform columns; -1 = default")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--order", type=int, default=0, help="0 = CSR sequential (sparse.rs), 1 = simd4 (simd_ops.rs), 2 = any order (SL_ORDER_ANY: results to rounding)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)   # child process of the cpu_baseline leg
    ap.add_argument("--no-sweep", action="store_true", help="skip the secondary column-structure measurements")
    ap.add_argument("--exchange", choices=["abi", "p2p", "allreduce"], default="abi",
                    help="abi = the library's own communicator and partitioned state behind the C ABI (default, every N; transport ipc | rccl: "
                         "SL_COMM_TRANSPORT); p2p / allreduce = the exchange ABOVE the ABI over torch.distributed / RCCL: neighbour sends of the boundary strips, "
                         "or ONE all-reduce over a zero-filled compact strip buffer")
    ap.add_argument("--no-overlap", action="store_true", help="multi-GPU (torch path): do not split boundary / interior rows")
    ap.add_argument("--force-split", action="store_true", help="testing (torch path): use the boundary / interior split even on one GPU")
    ap.add_argument("--attempt-timeout", type=float, default=float(os.environ.get("SL_BENCH_ATTEMPT_TIMEOUT", "420")),
                    help="self-launched N > 1: seconds one transport attempt may take before the parent ends it and tries the next")
    ap.add_argument("--total-timeout", type=float, default=float(os.environ.get("SL_BENCH_TOTAL_TIMEOUT", "1440")),
                    help="self-launched N > 1: seconds all attempts together may take (the driver allows a bench run 1800 s)")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the bitwise check of sampled row blocks against the CPU checker before the timed region")
    ap.add_argument("--parity-gate-only", default=None, help=argparse.SUPPRESS)          # child process of the parity gate (CPU checker)
    ap.add_argument("--no-scaling-reference", action="store_true", help="N > 1: skip rank 0's one-GPU reference measurements")
    args = ap.parse_args()
    if args.parity_gate_only:       # the CPU checker of the parity gate, in its own process
        print(json.dumps(cpu_parity_gate(args.parity_gate_only)), flush=True)
        return
    if args.cpu_baseline_only:      # runs in its own process: a crash of the CPU checker must not take the GPU line with it
        w0 = args.bandwidth if args.bandwidth >= 0 else DEFAULT_BANDWIDTH
        print(json.dumps(cpu_baseline(args.n, args.k, args.seed, w0, os.cpu_count() or 1)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:      # `python bench.py --gpus N`: no launcher around us, so be one
        sys.exit(launcher(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    child = os.environ.get("SL_BENCH_CHILD") == "1"
    use_abi = args.exchange == "abi" and not args.force_split and os.environ.get("SL_BENCH_FORCE_DIST") != "1"
    if use_abi and child:                                   # one attempt with the transport the parent chose; the parent decides what comes next
        transport = os.environ.get("SL_BENCH_TRANSPORT", "ipc")
        if os.environ.get("SL_BENCH_FAIL_ATTEMPT") == os.environ.get("SL_BENCH_ATTEMPT", "0") and rank == world - 1:
            raise SystemExit("SL_BENCH_FAIL_ATTEMPT: this rank leaves before the rendezvous (test of the launcher's fallback)")
        if transport in ("ipc", "rccl"):
            os.environ["SL_COMM_TRANSPORT"] = transport
            if os.environ.get("SL_BENCH_HALO") == "allreduce":      # the rccl_allreduce variant of the parent's two-exchange measurement
                os.environ["SL_COMM_HALO"] = "allreduce"
            if os.environ.get("SL_BENCH_LIGHT") == "1":             # second variant: the headline structure only
                args.no_sweep, args.no_scaling_reference, args.no_cpu_baseline = True, True, True
            main_abi(args, world, rank, local_rank, int(os.environ.get("SL_BENCH_ATTEMPT", "0")))
            return
        args.exchange = "p2p"
    elif use_abi and world == 1:
        return main_abi(args, world, rank, local_rank)
    elif use_abi:
        # under torch.distributed.run: the ranks agree on the outcome of every attempt (a gloo all-reduce) before anyone moves on
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("cpu:gloo,cuda:nccl" if os.environ.get("SL_BENCH_BACKEND", "nccl") == "nccl" else "gloo", rank=rank, world_size=world)
        import torch
        one_gpu_per_rank = max(1, torch.cuda.device_count()) >= world
        if one_gpu_per_rank and not os.environ.get("SL_COMM_TRANSPORT") and os.environ.get("SL_BENCH_ONE_EXCHANGE") != "1":
            # both exchanges, one line (EXCHANGE_VARIANTS above): the first variant that measures carries the secondary measurements, the other runs light
            import copy
            lines, failures, have_full = {}, {}, False
            for att, (vname, env_spec) in enumerate(EXCHANGE_VARIANTS):
                apply_variant_env(env_spec)
                a = copy.copy(args)
                if have_full:
                    a.no_sweep, a.no_scaling_reference, a.no_cpu_baseline = True, True, True
                ok, err, line = 1, None, None
                # A measured variant is never lost to the one after it: ncclCommInitRank and the collectives of a first contact with RCCL have
                # no time limit of their own, and a rank that hangs there would take the whole job — and the line already measured — into the
                # driver's kill.  Once a variant is in hand, the next one runs under a watchdog on EVERY rank (same limit everywhere): when it
                # fires, rank 0 prints the line with that variant marked as timed out, and every rank leaves with status 0.
                dog = None
                if have_full:
                    import threading
                    def fire(vname=vname):
                        if rank == 0:
                            f = dict(failures)
                            f[vname] = f"no result within {args.attempt_timeout:.0f} s (watchdog): the variant measured before it stands"
                            emit_line(merge_exchange_variants(lines, f))
                        sys.stdout.flush(); sys.stderr.flush()
                        os._exit(0)
                    dog = threading.Timer(args.attempt_timeout, fire)
                    dog.daemon = True
                    dog.start()
                try:
                    if os.environ.get("SL_BENCH_HANG_VARIANT") == vname and rank == world - 1:      # test of the watchdog: this rank never comes back
                        time.sleep(10 ** 6)
                    line = main_abi(a, world, rank, local_rank, att, emit=False)
                except Exception as e:                      # every wait in the communicator is bounded: all ranks get here
                    ok, err = 0, e
                flag = torch.tensor([ok], dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # (still under the watchdog: the ranks that finished wait here for the one that hangs)
                if dog is not None:
                    dog.cancel()
                if int(flag[0]) == 1:
                    have_full = True
                    if rank == 0:
                        lines[vname] = line
                else:
                    failures[vname] = f"failed on {'rank ' + str(rank) + ': ' + repr(err) if err else 'another rank'}"
                    print(f"[bench rank {rank}] exchange variant {vname}: {failures[vname]}", file=sys.stderr, flush=True)
            apply_variant_env({"SL_COMM_TRANSPORT": None, "SL_COMM_HALO": None})
            if have_full:
                if rank == 0:
                    merged = merge_exchange_variants(lines, failures)
                    emit_line(merged)
                dist.destroy_process_group()
                return
            print(f"[bench rank {rank}] falling back to the exchange over torch.distributed / RCCL", file=sys.stderr, flush=True)
            args.exchange = "p2p"
            return main_torch(args, world, rank, local_rank)
        transports = [os.environ["SL_COMM_TRANSPORT"]] if os.environ.get("SL_COMM_TRANSPORT") else ["ipc", "rccl"]
        if not one_gpu_per_rank:
            transports = [t for t in transports if t == "ipc"] or ["ipc"]
        for att, transport in enumerate(transports):
            os.environ["SL_COMM_TRANSPORT"] = transport
            ok, err = 1, None
            try:
                main_abi(args, world, rank, local_rank, att)
            except Exception as e:                          # every wait in the communicator is bounded: all ranks get here
                ok, err = 0, e
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag[0]) == 1:
                dist.destroy_process_group()
                return
            print(f"[bench rank {rank}] ABI path with transport {transport} failed on {'this rank: ' + repr(err) if err else 'another rank'}", file=sys.stderr, flush=True)
        print(f"[bench rank {rank}] falling back to the exchange over torch.distributed / RCCL", file=sys.stderr, flush=True)
        args.exchange = "p2p"
    return main_torch(args, world, rank, local_rank)


def main_torch(args, world, rank, local_rank):
    """The exchange ABOVE the ABI, over torch.distributed (backend "nccl" = RCCL): --exchange p2p | allreduce, --force-split, and the
    last fallback of N > 1.  One GPU without --force-split never comes here."""
    import torch
    import torch.distributed as dist
    from sublinear_time_solver_amd import _lib as L
    from sublinear_time_solver_amd import distributed as D

    if args.exchange == "abi":
        args.exchange = "p2p"
    # SL_BENCH_BACKEND=gloo is a TEST mode: ranks may share a GPU and the exchanges are staged through the host
    backend = os.environ.get("SL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("SL_BENCH_FORCE_DIST") == "1"        # TEST: initialise the process group even at world size 1
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if dist.is_initialized():
            pass                                                     # the agreed fallback of the ABI attempts: the group exists already
        elif backend == "nccl":
            os.environ.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")   # the small transfer kernels must not queue behind a 20 K-block launch
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    lib = L.load()
    L.check(lib.sl_set_device(local_rank))
    stream = torch.cuda.current_stream(dev)
    L.check(lib.sl_set_stream(C.c_void_p(stream.cuda_stream)))

    n_local, k = args.n, args.k
    n_global = n_local * world
    w = args.bandwidth if args.bandwidth >= 0 else DEFAULT_BANDWIDTH
    part = D.RowPartition(n_global, world, rank)
    assert part.n_local == n_local

    # ---- synthesize this rank's rows directly in HBM, build the row-slice layout ---------------------
    def build_piece(lo_l, hi_l):
        """rows [part.lo + lo_l, part.lo + hi_l) as their own device matrix (row slice, global column ids)"""
        rows = hi_l - lo_l
        rp = torch.empty(rows + 1, dtype=torch.int32, device=dev)
        ci = torch.empty(rows * k, dtype=torch.int32, device=dev)
        va = torch.empty(rows * k, dtype=torch.float64, device=dev)
        bb = torch.empty(rows, dtype=torch.float64, device=dev)
        L.check(lib.sl_synth_sdd_device(n_global, k, args.seed, w, part.lo + lo_l, part.lo + hi_l, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), bb.data_ptr()))
        hh = C.c_void_p()
        L.check(lib.sl_matrix_create_csr(rows, n_global, rows * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(),
                                         L.SL_MEM_DEVICE, part.lo + lo_l, 0, C.byref(hh)))
        del rp, ci, va, bb
        L.check(lib.sl_matrix_diagonal_inverse(hh, dinv[lo_l:hi_l].data_ptr(), L.SL_MEM_DEVICE))
        return hh

    dinv = torch.empty(n_local, dtype=torch.float64, device=dev)
    overlap = (world > 1 or args.force_split) and w > 0 and not args.no_overlap and n_local >= 4 * w
    handles = []
    if overlap:   # boundary rows first, halo transfer in flight under the interior rows
        bnd, inter = D.split_bounds(n_local, w, rank > 0 or args.force_split, rank < world - 1 or args.force_split)
        pieces_b = [(lo_l, hi_l, build_piece(lo_l, hi_l)) for lo_l, hi_l in bnd if hi_l > lo_l]
        pieces_i = [(lo_l, hi_l, build_piece(lo_l, hi_l)) for lo_l, hi_l in inter if hi_l > lo_l]
        handles = [hh for _, _, hh in pieces_b + pieces_i]
        h = pieces_i[0][2]
        local_step = D.HipSplitStep(pieces_b, pieces_i, dinv, args.order)
    else:
        h = build_piece(0, n_local)
        handles = [h]
        local_step = D.hip_local_step(h, dinv, args.order)
    torch.cuda.empty_cache()
    info = L.MatrixInfo()
    L.check(lib.sl_matrix_get_info(h, C.byref(info)))

    # t0 = D^-1 b on every rank (b is a closed form of the row index, so no exchange is needed to start)
    idx = torch.arange(n_global, device=dev, dtype=torch.float64)
    t0 = torch.zeros(part.n_padded, dtype=torch.float64, device=dev)
    t0[:n_global] = (1.0 + 0.001 * torch.remainder(idx, 1000.0)) * (1.0 / (10.0 + 0.01 * torch.remainder(idx, 1000.0)))
    del idx
    x = t0[part.lo:part.hi].clone()
    # SL_BENCH_LOOPBACK=1 (with SL_BENCH_FORCE_DIST=1 --force-split): MEASUREMENT mode on one GPU — the rank exchanges both
    # boundary strips with itself over RCCL and all-reduces the norm, i.e. the full per-step enqueue path of an inner rank
    loopback = force_dist and world == 1 and os.environ.get("SL_BENCH_LOOPBACK") == "1" and backend == "nccl"
    if w == 0:
        exchange = D.AllGatherExchange(part)
    elif args.exchange == "allreduce":
        exchange = D.HaloAllReduceExchange(part, w)
    else:
        exchange = D.HaloExchange(part, w, loopback=loopback)
    # the norm log is all-reduced once per batch of 10 steps (= SL_SOLVE_BATCH of the speculative solve loop); 1 = every step
    reduce_every = int(os.environ.get("SL_BENCH_REDUCE_EVERY", "10"))
    drv = D.PartitionedNeumann(part, local_step, exchange, t0, x, reduce_every=reduce_every)
    drv.reduce_always = loopback and os.environ.get("SL_BENCH_NO_ALLREDUCE") != "1"
    if overlap and os.environ.get("SL_BENCH_HALO_WAIT") == "main":          # A/B knob: main stream waits for the strips every step
        local_step.halo_wait_on_side = False
    if overlap and os.environ.get("SL_BENCH_SERIAL_BOUNDARY") == "1":       # A/B knob: boundary pieces on the main stream, ahead of the interior
        local_step.concurrent = False
    reduce_norm = os.environ.get("SL_BENCH_NO_ALLREDUCE") != "1"             # A/B knob (measurement only)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    gate = None
    if not args.no_parity_gate:
        gate = parity_gate_torch(args, torch, drv, part, n_global, w, t0, reduce_norm)
        if "skipped" not in gate:
            agg = torch.tensor([float(gate.get("rows_checked", 0)), 1.0 if gate.get("bitwise_equal") else 0.0, float(gate.get("max_rel_err", float("inf")))], dtype=torch.float64, device=dev)
            lo_ok = agg[1:2].clone()
            if world > 1 or force_dist:
                D.all_reduce_scalar(agg[0:1], dist.ReduceOp.SUM); D.all_reduce_scalar(lo_ok, dist.ReduceOp.MIN); D.all_reduce_scalar(agg[2:3], dist.ReduceOp.MAX)
            gate = {"rows_checked": int(agg[0]), "bitwise_equal": bool(lo_ok[0] > 0.5), "max_rel_err": float(agg[2]), "ranks_checked": world,
                    "steps_before_check": GATE_STEPS, "checker": "oracle SpMV over rows regenerated on the host, in a child process before the timed region (exchange over torch.distributed)"}
    for _ in range(args.warmup):
        drv.step(reduce_norm)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # library launches go to THIS stream
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        drv.step(reduce_norm)
    ev1.record(stream)
    enqueue_s = time.perf_counter() - t_start        # host time to enqueue the K steps (no waiting unless the queue is full)
    barrier()
    elapsed = time.perf_counter() - t_start
    dev_ms = ev0.elapsed_time(ev1)
    tmax = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device=dev)
    if world > 1 or force_dist:
        D.all_reduce_scalar(tmax, dist.ReduceOp.MAX)
    elapsed, dev_ms = float(tmax[0]), float(tmax[1])
    term_norm = drv.term_norm()

    # kernel-only duration on one GPU: the same launches through the library's own HIP-event bracket
    kern_ms = None
    if world == 1:
        ta, tb = drv.t[drv.cur], drv.t[1 - drv.cur]
        ms = C.c_float(0)
        L.check(lib.sl_neumann_run_steps(h, dinv.data_ptr(), ta.data_ptr(), tb.data_ptr(), x.data_ptr(), drv.norm2.data_ptr(),
                                         args.order, args.steps, C.byref(ms)))
        kern_ms = ms.value / args.steps

    if rank == 0:
        nnz_total = n_global * k
        ms_per_step = elapsed * 1e3 / args.steps
        value = nnz_total * args.steps / elapsed
        per_launch_bytes = algorithmic_bytes(n_local, n_local * k)
        launch_ms = kern_ms if kern_ms is not None else dev_ms / args.steps
        achieved = per_launch_bytes / (launch_ms * 1e-3) / 1e9
        traffic, traffic_source = recorded_traffic(n_local, k, w)
        out = {
            "metric": "push_iterations_x_nnz_per_sec", "value": value, "unit": "nnz*iter/s",
            "n_gpus": world if backend == "nccl" else min(world, max(1, torch.cuda.device_count())), "n_ranks": world,      # gloo test mode: ranks share devices
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"S-DD(n={n_local} rows/GPU, nnz/row={k}, seed={args.seed}, "
                                   f"{'uniform columns' if w == 0 else f'band half-width {w}'}) fused Neumann/push step, fp64, "
                                   "1xMI355X HBM roofline run (BASELINE configs[2])",
                       "n_per_gpu": n_local, "n_global": n_global, "nnz_per_row": k, "half_bandwidth": w,
                       "order": {0: "csr_sequential", 1: "simd4", 2: "any (SL_ORDER_ANY)"}[args.order],
                       "transport": "torch.distributed (" + backend + ")", "exchange": (exchange.name + ("+overlap" if overlap else "") + ("+loopback" if loopback else "")) if (world > 1 or overlap) else "none", "partition": f"rows{world}",
                       "norm_allreduce_every": reduce_every if (world > 1 or loopback) else None,
                       "rows_iter_per_s": value / k, "last_term_norm": term_norm},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "column_structure": "uniform over all columns (SURVEY 8(d) S-DD)" if w == 0 else f"band half-width {w}",
                         "kernel": LAYOUTS.get(int(info.column_panels), "row slices") if info.column_panels else ("LDS-window band kernel" if 0 < w <= 9400 else "row-slice general kernel"),
                         "algorithmic_bytes_per_launch": per_launch_bytes, "launch_ms": launch_ms,
                         "timed_region_device_ms_per_step": dev_ms / args.steps,     # HIP events around the K timed steps, same stream
                         "host_enqueue_ms_per_step": enqueue_s * 1e3 / args.steps,
                         # algorithmic bytes count 12 B per entry; 16-bit column offsets move 10, so this ratio can exceed 1 on banded inputs
                         "algorithmic_over_copy_ceiling_6290": achieved / 6290.0},
        }
        if gate is not None:
            out["parity_gate"] = gate
            if "skipped" not in gate and not gate["bitwise_equal"]:      # a step whose results differ is not reported as a measurement
                out["value"], out["ms_per_step"], out["error"] = None, None, "parity gate failed: the step's results differ from the CPU checker's"
        if world == 1 and not args.no_sweep:
            others = [v for v in (0, BANDED_BANDWIDTH, 512, 32768) if v != w]
            sweep = column_structure_sweep(lib, L, torch, dev, n_local, k, args.seed, args.order, others)
            out["config"]["other_column_structures"] = sweep
            bk = f"w{BANDED_BANDWIDTH}"
            if bk in sweep:     # the banded variant of the recipe: a full roofline block of its own, same run, same box
                tb_, tsrc = recorded_traffic(n_local, k, BANDED_BANDWIDTH)
                out["roofline_banded"] = {"bound": "hbm", "achieved": sweep[bk]["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": sweep[bk]["roofline_frac"], "traffic": tb_, "traffic_source": tsrc,
                                          "column_structure": f"band half-width {BANDED_BANDWIDTH}", "kernel": "LDS-window band kernel",
                                          "algorithmic_bytes_per_launch": per_launch_bytes, "launch_ms": sweep[bk]["ms_per_step"],
                                          "nnz_iter_per_s": sweep[bk]["nnz_iter_per_s"]}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = run_cpu_baseline(n_global, k, args.seed, w)
        emit_line(out)
    for hh in handles:
        lib.sl_matrix_destroy(hh)
    if world > 1 or force_dist:
        dist.destroy_process_group()

# default column structure of the headline run: 0 = uniform over all columns — S-DD exactly as SURVEY §8(d) writes it (the reference
# generators' recipe j = s mod n).  The banded variant (half-width 4096 ~ 1.3 sqrt(n), a naturally ordered 2-D grid operator; the
# "locality" form of config 5) gets its own full roofline block in the same line.
DEFAULT_BANDWIDTH = 0
BANDED_BANDWIDTH = 4096      # the banded variant reported next to the headline (roofline_banded)

if __name__ == "__main__":
    main()
