#!/usr/bin/env python3
"""Build tests/simt/_build/libsublinear_hip_simt.so: the UNCHANGED sources of sublinear_time_solver_amd/csrc/*.hip compiled as host C++
against the SIMT emulator (tests/simt/hip/hip_runtime.h + simt_rt.cpp).  TEST INFRASTRUCTURE — the product never loads this library.

The sources are copied into the build directory with the handful of textual rewrites that a host compiler needs and that change no
arithmetic and no control flow:
  __builtin_amdgcn_X(...)                         -> simt_amdgcn_X(...)          (the emulator's versions of the gfx950 builtins)
  __attribute__((address_space(N)))               -> removed                     (the casts of global_load_lds' pointer arguments)
  extern __shared__ ... double NAME[];            -> double *NAME = the block's dynamic LDS window
  asm volatile("s_waitcnt ..." ::: "memory")      -> a wave barrier              (where the wave waits for ITS memory operations on the hardware, the
                                                                                  emulated lanes — which run one after the other — wait for each other: LDS written
                                                                                  by one lane before the wait is read by another lane after it)
  asm volatile("v_mov_b32 %0, 0" : "=v"(x))       -> x = 0
  asm volatile("" : "+v"(a), ...) / "+s"(p)       -> a compiler barrier          (scheduling / register-class hints for the real ISA)
Compile flags carry -ffp-contract=off like the device build: a product is rounded before it is added."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
# SIMT_CSRC / SIMT_BUILD: another tree's kernels (e.g. `git worktree add /tmp/r03 <commit>`): does an older, hardware-verified version of a
# kernel behave the same under the emulator?  — tells an emulator artefact from a regression
CSRC = Path(os.environ["SIMT_CSRC"]) if os.environ.get("SIMT_CSRC") else ROOT / "sublinear_time_solver_amd" / "csrc"
# SIMT_SANITIZE=1: the same library under AddressSanitizer (its own build directory): "device" allocations get their exact size, so a kernel
# that reads or writes past a device buffer — harmless-looking on hardware, where allocations are padded to pages — is reported with the
# kernel's source line.  Run with LD_PRELOAD=<libclang_rt.asan-x86_64.so> (tests/test_simt_asan.py does).
# SIMT_SANITIZE=2: AddressSanitizer + UndefinedBehaviorSanitizer (shift counts, signed overflow, float -> integer conversions out of range,
# static array bounds, null dereferences; not `alignment`: the hardware's vector loads need dword alignment only) in `_build_ubsan/`.
SANITIZE = os.environ.get("SIMT_SANITIZE") in ("1", "2")
UBSAN = os.environ.get("SIMT_SANITIZE") == "2"
BUILD = Path(os.environ["SIMT_BUILD"]) if os.environ.get("SIMT_BUILD") else HERE / ("_build_ubsan" if UBSAN else "_build_asan" if SANITIZE else "_build")
CXX = os.environ.get("SIMT_CXX", "/opt/rocm/lib/llvm/bin/clang++")
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-g1", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fno-strict-aliasing", "-pthread", "-Wno-unused-value",
         "-Wno-unused-result", "-Wno-unknown-attributes", "-Wno-ignored-attributes", f"-I{HERE}", f"-I{BUILD}"]
if SANITIZE:
    FLAGS += ["-fsanitize=address", "-fno-omit-frame-pointer", "-DSIMT_EXACT_ALLOC", "-shared-libasan"]
if UBSAN:
    FLAGS += ["-fsanitize=undefined,float-cast-overflow", "-fno-sanitize=alignment,vptr,function", "-fno-sanitize-recover=undefined"]


def asan_runtime() -> str:
    """path of the shared ASAN runtime of the compiler that builds the emulator (to LD_PRELOAD into the python that loads the library)"""
    r = subprocess.run([CXX, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True)
    return r.stdout.strip()

REWRITES = [
    (re.compile(r"__builtin_amdgcn_"), "simt_amdgcn_"),
    (re.compile(r"__attribute__\(\(address_space\(\d+\)\)\)\s*"), ""),
    (re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s*)?(\w+)\s+(\w+)\[\];"), r"\1 *\2 = static_cast<\1 *>(::simt::block().dyn_lds);"),
    (re.compile(r'asm volatile\("s_waitcnt[^"]*"\s*:::\s*"memory"\);'), 'simt_amdgcn_wave_barrier();'),
    (re.compile(r'asm volatile\("v_mov_b32 %0, 0"\s*:\s*"=v"\((\w+)\)\);'), r"\1 = 0;"),
    (re.compile(r'asm volatile\(""\s*:\s*"\+[vs]"[^;]*\);'), 'asm volatile("" ::: "memory");'),
]


# Places where a kernel relies on the hardware's wave-synchronous execution WITHOUT any instruction the emulator can see: LDS written by
# one lane and read by another lane of the same wave with nothing but program order in between ("LDS accesses of one wave execute in
# issue order").  The emulated lanes run one after the other, so the order has to be made explicit there: a wave barrier is inserted
# in the emulated copy (on the hardware it would be a no-op).  (file, anchor line that must exist exactly once, text inserted before it)
SYNC_POINTS = [
    # (none at present.  Until round 6 the dynamic-tile panel kernel read its rows' slots right behind the last group's LDS updates; its epilogue
    #  — sl_tile_epilogue — now opens with `s_waitcnt vmcnt(0)`, which the rewrite above turns into the wave barrier this list used to insert.)
]


def rewrite(text: str, name: str = "") -> str:
    for rx, rep in REWRITES:
        text = rx.sub(rep, text)
    for fname, anchor, ins in SYNC_POINTS:
        if fname == name:
            if text.count(anchor) != 1:
                raise RuntimeError(f"{name}: the anchor of an emulator sync point is gone or ambiguous:\n{anchor}")
            text = text.replace(anchor, ins + anchor)
    if os.environ.get("SIMT_PATCH"):        # debugging: a python file defining patch(name, text) -> text (printf probes in a scratch build)
        ns = {}
        exec(Path(os.environ["SIMT_PATCH"]).read_text(), ns)
        text = ns["patch"](name, text)
    return text


def sources():
    return sorted(CSRC.glob("*.hip"))


def stale(out: Path) -> bool:
    if not out.exists():
        return True
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + [CSRC.parent.parent / "include" / "sublinear_hip.h", HERE / "simt_rt.cpp", HERE / "simt_debug.cpp", HERE / "fake_rccl.cpp", HERE / "hip" / "hip_runtime.h",
                                                                    HERE / "rocprim" / "device" / "device_radix_sort.hpp", Path(__file__)]
    return out.stat().st_mtime < max(d.stat().st_mtime for d in deps)


def build(force: bool = False) -> Path:
    out = BUILD / "libsublinear_hip_simt.so"
    if not force and not stale(out):
        return out
    (BUILD / "csrc").mkdir(parents=True, exist_ok=True)
    (BUILD / "include").mkdir(parents=True, exist_ok=True)
    # mirror the tree so that "../../include/sublinear_hip.h" of sl_internal.hpp resolves: _build/pkg/csrc + _build/include
    pkg = BUILD / "pkg" / "csrc"
    pkg.mkdir(parents=True, exist_ok=True)
    (BUILD / "include" / "sublinear_hip.h").write_text((CSRC.parent.parent / "include" / "sublinear_hip.h").read_text())
    for f in list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")):
        (pkg / f.name).write_text(rewrite(f.read_text(), f.name))
    objs = []

    def cc(src: Path):
        obj = BUILD / (src.stem + ".o")
        r = subprocess.run([CXX, *FLAGS, "-c", str(src), "-o", str(obj)], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{src.name}:\n{r.stderr[-6000:]}")
        return obj

    todo = [pkg / s.name for s in sources()] + [HERE / "simt_rt.cpp", HERE / "simt_debug.cpp"]
    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, todo))
    r = subprocess.run([CXX, "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", *(["-fsanitize=address", "-shared-libasan"] if SANITIZE else []), *(["-fsanitize=undefined", "-shared-libsan"] if UBSAN else []), "-o", str(out), *map(str, objs),
                        "-ldl", "-lrt"], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])
    # the stand-in the library's dlopen("librccl.so.1") finds when this directory leads LD_LIBRARY_PATH (multi-process tests only)
    r = subprocess.run([CXX, "-std=c++17", "-O2", "-fPIC", "-shared", "-pthread", "-Wl,-soname,librccl.so.1", "-o", str(BUILD / "librccl.so.1"), str(HERE / "fake_rccl.cpp")],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-4000:])
    return out


if __name__ == "__main__":
    try:
        print(build(force="--force" in sys.argv))
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
