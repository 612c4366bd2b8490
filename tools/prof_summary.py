#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile.sh into profiles/<tag>_*.txt/json.

usage: python tools/prof_summary.py gpurun_out/prof_<tag> <tag> [n k bandwidth]
  stats/stats_results.db      -> profiles/<tag>_kernel_stats.txt   (top_kernels view = `--stats`)
  pmc_fetch/, pmc_write/      -> profiles/<tag>_pmc.txt and one record of profiles/pmc_traffic.json (keyed by column structure)
  pmc_tcc/                    -> L2 hit rate of the step kernel, same text file
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE tallies the 128-B requests
of a wide coalesced read stream at 64 B, i.e. exactly half the bytes (MI355X_MICROARCH.md §HBM), so the
corrected HBM read bytes are 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken as reported.
"""
import json
import sqlite3
import sys
from pathlib import Path


STEP_KERNELS = ("sl_band_kernel", "sl_rows_kernel", "sl_panel_kernel", "sl_pw_kernel", "sl_mpass_kernel")


def q(db, sql):
    return sqlite3.connect(db).execute(sql).fetchall()


def main():
    src, tag = Path(sys.argv[1]), sys.argv[2]
    cfg = [int(v) for v in sys.argv[3:6]] if len(sys.argv) >= 6 else None
    import os
    # PROF_OUT: where the summaries go.  On the GPU box only gpurun_out/ travels back (<= 64 MiB, the .db files are 22 MB each), so
    # tools/profile.sh summarises there into gpurun_out/prof_<tag>/summary and deletes the databases; copy the summaries to profiles/.
    out = Path(os.environ.get("PROF_OUT") or (Path(__file__).resolve().parent.parent / "profiles"))
    out.mkdir(parents=True, exist_ok=True)
    lines = [f"# rocprofv3 --kernel-trace --stats : {src}/stats  (top_kernels view)", ""]
    rows = q(src / "stats" / "stats_results.db", "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc")
    lines.append(f"{'kernel':<92} {'calls':>6} {'total_ms':>10} {'avg_us':>10} {'pct':>7}")
    for name, calls, tot, avg, pct in rows:          # the view reports microseconds
        lines.append(f"{name[:92]:<92} {calls:>6} {tot / 1e3:>10.3f} {avg:>10.2f} {pct:>7.2f}")
    # steady-state average of the dominant kernel: drop the first launches (cold caches / clocks)
    k = q(src / "stats" / "stats_results.db",
          "select name, duration from kernels where name like '%sl_band_kernel%' or name like '%sl_rows_kernel%' or name like '%sl_panel_kernel%' "
          "or name like '%sl_pw_kernel%' or name like '%sl_mpass_kernel%' order by start")
    by = {}
    for name, d in k:
        by.setdefault(name, []).append(d)
    lines.append("")
    for name, ds in by.items():
        tail = ds[len(ds) // 4:]
        lines.append(f"steady-state {name[:80]}: n={len(tail)} avg={sum(tail) / len(tail) / 1e3:.2f} us min={min(tail) / 1e3:.2f} us max={max(tail) / 1e3:.2f} us")
    (out / f"{tag}_kernel_stats.txt").write_text("\n".join(lines) + "\n")
    print("\n".join(lines))

    pm = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) : {src}", ""]
    rec = {}
    step_kernel = {}
    best_tcc = [0, 0.0, 0.0]
    for sub, db, cname in (("pmc_fetch", "fetch_results.db", "FETCH_SIZE"), ("pmc_write", "write_results.db", "WRITE_SIZE")):
        p = src / sub / db
        if not p.exists():
            continue
        cols = [r[1] for r in q(p, "pragma table_info('counters_collection')")]
        pm.append(f"{cname}: columns of counters_collection = {cols}")
        namecol = "kernel_name" if "kernel_name" in cols else "name"
        rows = q(p, f"select {namecol}, counter_name, avg(value), count(*) from counters_collection "
                    f"where counter_name='{cname}' group by {namecol} order by avg(value) desc limit 6")
        for kn, cn, avg, n in rows:
            pm.append(f"  {kn[:90]:<90} {cn} avg={avg:.1f} KiB over {n} dispatches")
        # the step kernel (most dispatches among the row kernels), whatever its rank in the list above
        best_n = 0
        for kn, cn, avg, n in q(p, f"select {namecol}, counter_name, avg(value), count(*) from counters_collection "
                                   f"where counter_name='{cname}' group by {namecol}"):
            if any(t in kn for t in STEP_KERNELS) and n > best_n:
                rec[cn] = avg
                best_n = n
                step_kernel[cname] = kn
    p = src / "pmc_tcc" / "tcc_results.db"
    if p.exists():
        rows = q(p, "select kernel_name, counter_name, avg(value), count(*) from counters_collection where counter_name in "
                    "('TCC_HIT_sum','TCC_MISS_sum') group by kernel_name, counter_name")
        tcc = {}
        for kn, cn, avg, n in rows:
            if any(t in kn for t in STEP_KERNELS):
                tcc.setdefault(kn, {})[cn] = (avg, n)
        pm.append("")
        for kn, d in tcc.items():
            if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
                h, m = d["TCC_HIT_sum"][0], d["TCC_MISS_sum"][0]
                pm.append(f"  {kn[:90]:<90} TCC_HIT {h:.4e} TCC_MISS {m:.4e} per launch over {d['TCC_HIT_sum'][1]} dispatches -> L2 hit rate {h / (h + m):.3f}")
                if d["TCC_HIT_sum"][1] >= best_tcc[0]:
                    best_tcc[:] = [d["TCC_HIT_sum"][1], h, m]
    if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
        rd = 2.0 * rec.get("FETCH_SIZE", 0.0) * 1024.0
        wr = rec.get("WRITE_SIZE", 0.0) * 1024.0
        pm += ["", f"step kernel: {step_kernel.get('FETCH_SIZE', step_kernel.get('WRITE_SIZE', '?'))[:100]}",
               f"dominant kernel per launch: FETCH_SIZE {rec.get('FETCH_SIZE', 0):.0f} KiB -> corrected read bytes {rd:.4e} (x2, gfx950; raw {rd / 2:.4e} — the x2 rule is calibrated on wide streaming reads, gather line fills may be counted differently)",
               f"                            WRITE_SIZE {rec.get('WRITE_SIZE', 0):.0f} KiB -> write bytes {wr:.4e}",
               f"                            HBM-side traffic per launch = {rd + wr:.4e} B  (Infinity-Cache hits are counted: this is L2 <-> fabric traffic)"]
        if cfg:
            n, kk, w = cfg
            alg = 12 * n * kk + 4 * (n + 1) + 40 * n
            pm.append(f"                            algorithmic bytes per launch = {alg:.4e} B  (traffic / algorithmic = {(rd + wr) / alg:.3f})")
            tf = out / "pmc_traffic.json"
            try:
                allrec = json.loads(tf.read_text())
                if "records" not in allrec:
                    allrec = {"records": {f"w{allrec.get('bandwidth')}" if allrec.get("bandwidth") else "uniform": allrec}}
            except Exception:
                allrec = {"records": {}}
            allrec["records"]["uniform" if w == 0 else f"w{w}"] = {
                "tag": tag, "n": n, "k": kk, "bandwidth": w, "fetch_size_kib": rec.get("FETCH_SIZE"), "write_size_kib": rec.get("WRITE_SIZE"),
                "hbm_bytes_per_launch": rd + wr, "algorithmic_bytes_per_launch": alg,
                "l2_hit_rate": (best_tcc[1] / (best_tcc[1] + best_tcc[2])) if best_tcc[0] else None,
                # which kernel these counters belong to: bench.py reports a record only while sl_kernels.hip is still that file
                "kernel_source_sha16": kernel_source_sha16(), "commit": head_commit()}
            tf.write_text(json.dumps(allrec, indent=1) + "\n")
    (out / f"{tag}_pmc.txt").write_text("\n".join(pm) + "\n")
    print("\n".join(pm))


def kernel_source_sha16():
    import hashlib
    src = Path(__file__).resolve().parent.parent / "sublinear_time_solver_amd" / "csrc" / "sl_kernels.hip"
    return hashlib.sha256(src.read_bytes()).hexdigest()[:16]


def head_commit():
    import subprocess
    try:
        return subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=Path(__file__).resolve().parent.parent, capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None          # (the GPU box has no .git: the hash of the source is what counts)


if __name__ == "__main__":
    main()
