#!/usr/bin/env python3
"""Random campaign over tests/c/dist_smoke.c (the partitioned solve through the C ABI, one process per rank, ranks sharing the box's
GPU): random world sizes 1..8, sizes, bandwidths on both sides of every halo form (neighbour halo, reach beyond the neighbour, all
columns), equal and unequal row ranges, the boundary-first step on and off.  dist_smoke itself compares with the one-GPU solve through
the same ABI (iteration count, convergence flag, solution bit for bit) and checks the exchange with sl_neumann_state_verify_exchange.

Round 4: half of the cases run with SL_STAGING=pageable (the round-3 calls: asynchronous copies to and from pageable std::vector memory
in the layout build), the other half on the library's pinned staging; a layout misfit the library repairs is recorded with the three
counts of its diagnosis (which link of widths read-back / row-length kernel / uploaded pointers was off) and the mode it happened in.
--parallel P runs P cases side by side (more processes contending for the one GPU: the condition the misfits were seen under).

usage: python tests/fuzz_dist.py --seconds 300 [--seed0 S] [--parallel P]      prints one JSON line; exit status 1 on any failure"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def build(tmp, simt_lib=None):
    exe = Path(tmp) / "dist_smoke"
    pkg, name = (ROOT / "sublinear_time_solver_amd", "sublinear_hip") if not simt_lib else (Path(simt_lib).resolve().parent, "sublinear_hip_simt")
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c" / "dist_smoke.c"),
                        "-o", str(exe), f"-L{pkg}", f"-l{name}", "-lm", f"-Wl,-rpath,{pkg}"], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stderr)
    return exe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed0", type=int, default=int(time.time()) & 0xFFFFFF)
    ap.add_argument("--min-world", type=int, default=1)
    ap.add_argument("--sizes", default="", help="comma-separated n to draw from instead of the default mix")
    ap.add_argument("--parallel", type=int, default=1, help="cases run side by side")
    ap.add_argument("--transports", default="ipc", help="comma-separated, drawn per case: ipc, rccl (grouped send / receive, all-gather), rccl-allreduce (the compact halo buffer)")
    ap.add_argument("--simt", default="", help="path of tests/simt/_build/libsublinear_hip_simt.so: the campaign on the SIMT emulator (no GPU; one process per rank over "
                                               "memfd-backed IPC, RCCL = tests/simt's stand-in) instead of the device")
    args = ap.parse_args()
    transports = [t for t in args.transports.split(",") if t in ("ipc", "rccl", "rccl-allreduce")] or ["ipc"]
    import threading
    lock = threading.Lock()
    failures, forms, repaired, by_staging = [], {}, [], {"pinned": 0, "pageable": 0}
    counter = [0]
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp, args.simt or None)
        t_end = time.time() + args.seconds

        def worker(wid):
            rng = np.random.default_rng([args.seed0, wid])
            while time.time() < t_end and len(failures) < 10:
                one_case(rng)

        def one_case(rng):
            world = int(rng.integers(max(1, min(args.min_world, 8)), 9))
            n = int(rng.choice([int(v) for v in args.sizes.split(",")] if args.sizes else [64 * world + 1, 5000, 20011, 90000, 400000, 1200000]))
            n = max(n, 64 * world)
            w = int(rng.choice([1, 40, 300, 5000, 15000, n // max(world, 1), 10**9]))
            uneven = bool(rng.random() < 0.5)
            overlap = str(rng.choice(["0", "1"]))
            cmd = [str(exe), str(world), str(n), str(w)] + (["uneven"] if uneven else [])
            staging = "pageable" if rng.random() < 0.5 else "pinned"
            env = dict(os.environ, SL_COMM_TIMEOUT_MS="60000", SL_DIST_OVERLAP=overlap, SL_LOG="1")
            transport = str(rng.choice(transports))
            env["SL_COMM_TRANSPORT"] = "rccl" if transport.startswith("rccl") else "ipc"
            env.pop("SL_COMM_HALO", None)
            if transport == "rccl-allreduce":
                env["SL_COMM_HALO"] = "allreduce"
            if args.simt:
                d = str(Path(args.simt).resolve().parent)
                env.update(SIMT_IPC="1", SIMT_THREADS=env.get("SIMT_THREADS", "1"), LD_LIBRARY_PATH=d + ":" + os.environ.get("LD_LIBRARY_PATH", ""), SL_COMM_TIMEOUT_MS="180000")
            if staging == "pageable":
                env["SL_STAGING"] = "pageable"
            if n >= 400000 and rng.random() < 0.5:            # the paced layout with XCD-local spans forced on every rank (a pretended small device)
                cus = int(rng.choice([4, 8, 12]))
                env.update(SL_COLUMN_PANELS="1", SL_PW_FORCE="1", SL_PW_CUS=str(cus), SL_PW_XCD=str(int(rng.choice([2, 4]))))
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env)
                ok = r.returncode == 0 and "dist_smoke ok" in r.stdout
                tail = (r.stdout[-600:] + r.stderr[-1200:]) if not ok else ""
                form = ("edge blocks first" if "runs its edge blocks first" in r.stderr else
                        "edge rounds first (paced layout)" if "runs its edge rounds first" in r.stderr else "exchange after the step")
                if "paced column panels" in r.stderr and "rounds first" not in form:
                    form += " (paced layout)"
            except subprocess.TimeoutExpired:
                ok, tail, form, r = False, "timeout", "?", None
            lock.acquire()
            counter[0] += 1
            by_staging[staging] += 1
            if r is not None:      # faults the library noticed and repaired by itself: recorded with its own words, whether or not the case then passed
                for ln in r.stderr.splitlines():
                    if "did not fit their slices" in ln or "IPC Attach" in ln or "succeeded on try" in ln:
                        repaired.append({"cmd": " ".join(cmd[1:]), "staging": staging, "line": ln[:700]})
            key = f"world {world}, {transport}: {form}"
            forms[key] = forms.get(key, 0) + 1
            if not ok:
                failures.append({"cmd": " ".join(cmd[1:]), "overlap": overlap, "staging": staging, "transport": transport, "forced": {k: env[k] for k in ("SL_PW_CUS", "SL_PW_XCD") if k in env}, "tail": tail})
                print("FAIL", " ".join(cmd[1:]), "overlap", overlap, "staging", staging, "\n", tail, file=sys.stderr)
            lock.release()

        threads = [threading.Thread(target=worker, args=(w,)) for w in range(max(1, args.parallel))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    cases = counter[0]
    print(json.dumps({"seed0": args.seed0, "on": "simt emulator (tests/simt; no GPU)" if args.simt else "device", "transports": transports, "cases": cases, "parallel": args.parallel, "cases_by_staging": by_staging, "forms_seen": dict(sorted(forms.items())), "failures": failures, "noticed_and_repaired": repaired}))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
