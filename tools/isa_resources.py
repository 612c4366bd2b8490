#!/usr/bin/env python3
"""Register / LDS / scratch budget of every gfx950 kernel of csrc/, read from the code object metadata hipcc emits — no GPU needed.

    python tools/isa_resources.py                     # compile csrc/*.hip with the Makefile's flags, print a table
    python tools/isa_resources.py --json out.json     # ... and write {kernel: {vgpr, agpr, sgpr, scratch, lds, spills, waves_per_simd, ...}}
    python tools/isa_resources.py --asm-dir DIR       # reuse DIR/<file>.s where newer than the source
    python tools/isa_resources.py --skeleton 'sl_pw_kernel<1'   # loop headers, vector loads / stores and s_waitcnt of a kernel, in order:
                                                      # how many loads a software pipeline really keeps in flight

What a blind edit can break without failing a CPU test — a spill to scratch, an occupancy tier lost to a few registers, an LDS
array that no longer leaves room for a second block — shows up here; tests/test_isa_host.py pins the hot kernels against
profiles/isa_resources.json.

waves_per_simd follows MI355X_MICROARCH.md: 512 VGPRs per SIMD lane, allocation granule 8, unified with AGPRs on gfx950,
at most 8 waves per SIMD; a launch bound of W waves per block on 4 SIMDs needs ceil(W / 4) of them per SIMD at once.
"""
import argparse
import json
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import yaml

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "sublinear_time_solver_amd" / "csrc"
HIPCC = "/opt/rocm/bin/hipcc"
CXXFILT = "/usr/bin/c++filt"


def makefile_flags():
    flags = re.search(r"^CXXFLAGS = (.*)$", (CSRC / "Makefile").read_text(), flags=re.M).group(1)
    return flags.replace("$(ARCH)", "gfx950").split()


def compile_asm(src: Path, out: Path):
    r = subprocess.run([HIPCC, *makefile_flags(), "--cuda-device-only", "-S", "-o", str(out), str(src)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S {src.name} failed:\n{r.stderr[-3000:]}")
    return out


def demangle(names):
    r = subprocess.run([CXXFILT], input="\n".join(names), capture_output=True, text=True, check=True)
    return r.stdout.split("\n")[: len(names)]


def short_name(full: str) -> str:
    """void sl_pw_kernel<1, false, 0, false>(sl_row_args) -> sl_pw_kernel<1, false, 0, false>"""
    s = re.sub(r"^void\s+", "", full)
    depth = 0
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return s[:i]
    return s


def waves_per_simd(vgpr: int, agpr: int) -> int:
    total = -(-(vgpr + agpr) // 8) * 8                # unified register file on gfx950: arch + acc VGPRs, granule 8
    return 8 if total == 0 else max(0, min(8, 512 // total))


def kernel_bodies(asm: str):
    """mangled name -> the instruction text of the kernel (label .. .Lfunc_end)"""
    out = {}
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, flags=re.M | re.S):
        out[m.group(1)] = m.group(2)
    return out


def drained_loops(body: str):
    """loops that issue at least four vector memory loads AND wait for `vmcnt(0)`: a software pipeline that drains — every load in flight
    is waited for inside the loop that is meant to keep some ahead.  Loop membership is the compiler's own block annotation
    (`; =>This Inner Loop Header` / `;   in Loop: Header=BBn_m`, innermost loop of a block).  [(header, loads, waits)].  A loop that waits by
    design (a staging loop in front of a barrier) shows up too: read the kernel before believing the number."""
    cur, stat = None, {}
    for ln in body.split("\n"):
        m = re.match(r"^\.LBB(\d+_\d+):(.*)$", ln)
        if m:
            rest = m.group(2)
            h = re.search(r"in Loop: Header=BB(\d+_\d+)", rest)
            cur = m.group(1) if "Loop Header" in rest else (h.group(1) if h else None)
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        st = stat.setdefault(cur, [0, 0])
        if re.search(r"\b(global|buffer|flat)_load_", ln) and "lds" not in ln:
            st[0] += 1
        if re.search(r"s_waitcnt[^\n]*vmcnt\(0\)", ln):
            st[1] += 1
    return [(h, l, w) for h, (l, w) in sorted(stat.items()) if l >= 4 and w]


def parse_asm(path: Path):
    asm = path.read_text()
    md = re.search(r"^\s*\.amdgpu_metadata\n(.*?)^\s*\.end_amdgpu_metadata", asm, flags=re.M | re.S)
    if not md:
        return {}
    meta = yaml.safe_load(md.group(1))
    kernels = meta.get("amdhsa.kernels") or []
    bodies = kernel_bodies(asm)
    names = demangle([k[".name"] for k in kernels])
    res = {}
    for k, full in zip(kernels, names):
        body = bodies.get(k[".name"], "")
        ins = [ln.split()[0] for ln in body.split("\n") if ln.startswith("\t") and not ln.startswith("\t.") and not ln.strip().startswith(";")]
        vg, ag = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
        name = short_name(full.replace("(anonymous namespace)::", ""))
        if "rocprim" in name or "hipcub" in name or name == "":      # rocPRIM's sort / scan kernels (sl_sort.hip, sl_matrix.hip, sl_acl.hip): the library's, not ours
            continue
        res[name] = {
            "file": path.stem + ".hip",
            "vgpr": vg, "agpr": ag, "sgpr": int(k.get(".sgpr_count", 0)),
            "vgpr_spills": int(k.get(".vgpr_spill_count", 0)), "sgpr_spills": int(k.get(".sgpr_spill_count", 0)),
            "scratch_bytes": int(k.get(".private_segment_fixed_size", 0)), "dynamic_stack": bool(k.get(".uses_dynamic_stack", False)),
            "lds_static_bytes": int(k.get(".group_segment_fixed_size", 0)),
            "max_workgroup": int(k.get(".max_flat_workgroup_size", 0)),
            "waves_per_simd": waves_per_simd(vg, ag),
            "instructions": len(ins),
            "v_fma_f64": sum(1 for i in ins if i.startswith("v_fma_f64") or i.startswith("v_fmac_f64")),
            "f64_divisions": sum(1 for i in ins if i.startswith("v_div_fixup_f64")),      # each IEEE division expands to 5 v_fma_f64 of its own
            "scratch_ops": sum(1 for i in ins if i.startswith("scratch_") or i.startswith("buffer_store") and "offen" in i),
            "drained_loops": len(drained_loops(body)),
            "lds_dma_loads": sum(1 for ln in body.split("\n") if re.search(r"\b(global|buffer)_load_(lds_)?dword.*\blds\b|global_load_lds", ln)),
        }
    return res


def collect(sources=None, asm_dir=None, jobs=8):
    sources = [CSRC / s for s in sources] if sources else sorted(CSRC.glob("*.hip"))
    import tempfile
    tmp = Path(asm_dir) if asm_dir else Path(tempfile.mkdtemp(prefix="sl_isa_"))
    tmp.mkdir(parents=True, exist_ok=True)
    todo = []
    for s in sources:
        out = tmp / (s.stem + ".s")
        dep = max(s.stat().st_mtime, (CSRC / "sl_internal.hpp").stat().st_mtime, (ROOT / "include" / "sublinear_hip.h").stat().st_mtime)
        if not out.exists() or out.stat().st_mtime < dep:
            todo.append((s, out))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(lambda a: compile_asm(*a), todo))
    res = {}
    for s in sources:
        res.update(parse_asm(tmp / (s.stem + ".s")))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json")
    ap.add_argument("--asm-dir")
    ap.add_argument("--match", default="", help="only kernels whose name contains this")
    ap.add_argument("--skeleton", default="", help="print the memory skeleton (loop headers, vector loads / stores, s_waitcnt) of the kernels whose name contains this")
    ap.add_argument("sources", nargs="*")
    a = ap.parse_args()
    res = collect(a.sources or None, a.asm_dir)
    rows = sorted(res.items())
    if not a.skeleton:
        print(f"{'kernel':78s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'scr':>5s} {'lds':>6s} {'w/simd':>6s} {'instr':>6s}")
    for name, r in rows:
        if a.match in name and not a.skeleton:
            print(f"{name[:78]:78s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['scratch_bytes']:5d} {r['lds_static_bytes']:6d} {r['waves_per_simd']:6d} {r['instructions']:6d}")
    if a.skeleton:
        import tempfile
        tmp = Path(a.asm_dir) if a.asm_dir else None
        if tmp is None:
            tmp = Path(tempfile.mkdtemp(prefix="sl_isa_"))
            collect(a.sources or None, tmp)
        for path in sorted(tmp.glob("*.s")):
            bodies = kernel_bodies(path.read_text())
            for mangled, full in zip(bodies, demangle(list(bodies))):
                if a.skeleton in full:
                    print(f"== {short_name(full)}   drained loops: {drained_loops(bodies[mangled])}")
                    for i, ln in enumerate(bodies[mangled].split("\n")):
                        if re.search(r"Loop Header|\b(global|buffer|flat|scratch)_(load|store)|s_waitcnt|s_barrier", ln):
                            print(f"{i:6d}  {ln.strip()[:110]}")
    if a.json:
        Path(a.json).write_text(json.dumps(dict(rows), indent=1, sort_keys=True) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
