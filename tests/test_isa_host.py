"""The register / LDS / scratch budget of every gfx950 kernel, pinned without a GPU.

hipcc cross-compiles csrc/ to gfx950 assembly here; tools/isa_resources.py reads each kernel's resources from the code-object
metadata.  profiles/isa_resources.json is the committed record (regenerate: `python tools/isa_resources.py --json
profiles/isa_resources.json`).  A blind edit that spills, loses an occupancy tier, grows a static LDS array or lets an FMA
contraction into a parity kernel fails HERE instead of on the driver's GPU box."""
import importlib.util
import re
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
RECORD = ROOT / "profiles" / "isa_resources.json"

spec = importlib.util.spec_from_file_location("isa_resources", ROOT / "tools" / "isa_resources.py")
ISA = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ISA)

# kernels allowed to touch scratch: none.  (Until round 6: the 16-wave band kernel with 12 bytes and the multi-pass push kernel with 20 —
# both lost them when the wave's number became a scalar and the stream addresses left the VGPRs.)
SCRATCH_ALLOWED = {}
# files whose kernels carry the reference's arithmetic: a product is rounded before it is added (no contraction)
PARITY_FILES = {"sl_kernels.hip", "sl_frontier.hip", "sl_acl.hip", "sl_southwell.hip", "sl_cg.hip", "sl_walk.hip", "sl_matrix.hip"}


ASM_DIR = []


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    ASM_DIR.append(tmp_path_factory.mktemp("isa"))
    return ISA.collect(asm_dir=ASM_DIR[0])


def test_every_kernel_is_in_the_record_and_within_its_budget(isa):
    rec = json.loads(RECORD.read_text())
    missing = sorted(set(isa) - set(rec))
    gone = sorted(set(rec) - set(isa))
    assert not missing and not gone, f"profiles/isa_resources.json is stale: new {missing[:8]}, removed {gone[:8]} — regenerate it and look at the diff"
    worse = []
    for name, now in isa.items():
        was = rec[name]
        if now["waves_per_simd"] < was["waves_per_simd"]:
            worse.append(f"{name}: {was['vgpr']} -> {now['vgpr']} VGPRs loses an occupancy tier ({was['waves_per_simd']} -> {now['waves_per_simd']} waves/SIMD)")
        if now["scratch_bytes"] > was["scratch_bytes"] or now["vgpr_spills"] > was["vgpr_spills"]:
            worse.append(f"{name}: scratch {was['scratch_bytes']} -> {now['scratch_bytes']} B, VGPR spills {was['vgpr_spills']} -> {now['vgpr_spills']}")
        if now["lds_static_bytes"] != was["lds_static_bytes"]:
            worse.append(f"{name}: static LDS {was['lds_static_bytes']} -> {now['lds_static_bytes']} B (the dynamic window is sized around it)")
        if now["max_workgroup"] != was["max_workgroup"]:
            worse.append(f"{name}: launch bound {was['max_workgroup']} -> {now['max_workgroup']}")
    assert not worse, "\n".join(worse)


def test_no_kernel_spills_or_uses_a_dynamic_stack(isa):
    bad = {k: (v["scratch_bytes"], v["vgpr_spills"]) for k, v in isa.items()
           if (v["scratch_bytes"] or v["vgpr_spills"] or v["dynamic_stack"]) and v["scratch_bytes"] > SCRATCH_ALLOWED.get(k, 0)}
    assert not bad, bad
    assert all(v["agpr"] == 0 for v in isa.values()), "no MFMA work here: accumulation registers in use means the allocator ran out of VGPRs"


def test_hot_kernels_keep_their_occupancy_and_lds_geometry(isa):
    # the headline: paced column panels, one 16-wave block per CU with the 160 KiB accumulator window => 4 waves/SIMD are needed and enough
    for epi in range(4):
        k = isa[f"sl_pw_kernel<{epi}, false, 0, false>"]
        assert k["vgpr"] <= 128 and k["waves_per_simd"] >= 4 and k["max_workgroup"] == 1024 and k["scratch_bytes"] == 0, (epi, k)
        assert k["lds_static_bytes"] <= 512                       # progress words + reduction scratch; the rest of the 160 KiB is the dynamic window
    assert isa["sl_pw_kernel<3, false, 0, true>"]["vgpr"] <= 112                   # index-only stream: no value registers; its peak is the epilogue's four rows in flight (one 16-wave block per CU: 128 is the bound)
    # order-free stream, row slices, long rows, frontier machinery: never below 4 (16-wave blocks) resp. 8 (plain 256-thread kernels)
    assert isa["sl_pwr_kernel<1, 0>"]["waves_per_simd"] >= 4
    for name, k in isa.items():
        if name.startswith("sl_rows_kernel") or name.startswith("sl_long_rows_kernel") or name == "sl_rows_add_kernel":
            assert k["waves_per_simd"] >= 5 and k["scratch_bytes"] == 0, (name, k["vgpr"])
        if name.startswith("sl_band_kernel") and ", 16>" in name:                # 16-wave band blocks: launch bound asks for 4 waves/SIMD
            assert k["vgpr"] <= 128, (name, k["vgpr"])
        if name.startswith("sl_band_kernel"):                                     # stream addresses live in SGPRs: no spill anywhere in the family
            assert k["scratch_bytes"] == 0 and k["vgpr_spills"] == 0, (name, k)
    # no pipelined uniform-width band kernel has a loop that issues loads and waits for all of them (tools/isa_resources.py, drained_loops);
    # the ragged-row batches still do (their loads sit under wave-uniform branches: DESIGN 5.5) — listed, not hidden
    for name, k in isa.items():
        if name.startswith("sl_band_kernel") and re.search(r"<\d, \d, (8|16), true,", name):       # (every epilogue: the push's optional per-row threshold /
            assert k["drained_loops"] == 0, name                                                   #  column value travel with the slice's other vectors)
        if name.startswith("sl_band_kernel") and re.search(r"<\d, \d, 0, true,", name):
            assert k["drained_loops"] == 1, name
    # the paced headline kernel: the only loop that waits for everything in flight is the ROUND (tile pointers, span, one wait in front of the
    # epilogue): the chunk pipeline never did, and since round 6 the per-row epilogue loop does not either (4 rows' vectors in flight together)
    for name, k in isa.items():
        if name.startswith("sl_pw_kernel"):
            assert k["drained_loops"] == 1, (name, k["drained_loops"])
    # the pipelined uniform-width band kernels (the 0.90 kernel and its 16-wave form): two slices of matrix bytes in registers and room to spare
    for nw in (4, 8, 16):
        assert isa[f"sl_band_kernel<0, 1, 16, true, true, {nw}>"]["vgpr"] <= 116, nw
        assert isa[f"sl_band_kernel<0, 1, 8, true, true, {nw}>"]["waves_per_simd"] >= 6, nw
    assert isa["sl_small_rounds_kernel"]["waves_per_simd"] >= 4 and isa["sl_small_rounds_kernel"]["scratch_bytes"] == 0
    assert isa["sl_long_rows_kernel<0, 1, false>"]["lds_static_bytes"] == 2048   # one 64-entry product line per wave


def test_parity_kernels_round_the_product_before_the_add(isa):
    """-ffp-contract=off must hold for every kernel that carries the reference's arithmetic: the only v_fma_f64 allowed are the five
    each correctly rounded IEEE division expands to (v_div_scale / v_rcp / 5 x v_fma / v_div_fmas / v_div_fixup)"""
    bad = {k: (v["v_fma_f64"], v["f64_divisions"]) for k, v in isa.items() if v["file"] in PARITY_FILES and v["v_fma_f64"] != 5 * v["f64_divisions"]}
    assert not bad, bad
    hot = [k for k in isa if k.startswith(("sl_pw_kernel", "sl_pwr_kernel", "sl_rows_kernel", "sl_band_kernel", "sl_long_rows_kernel", "sl_panel_kernel",
                                           "sl_mpass_kernel", "sl_small_rounds_kernel", "sl_rows_add_kernel", "acl_kernel"))]
    assert len(hot) > 100 and all(isa[k]["v_fma_f64"] == 5 * isa[k]["f64_divisions"] for k in hot)
    assert all(isa[k]["v_fma_f64"] == 0 for k in hot if not k.startswith("acl_kernel"))


def test_band_kernel_keeps_two_slices_of_loads_in_flight(isa):
    """The pipelined band kernel is written to have slice j + 1's matrix bytes in flight while slice j is reduced and to issue slice
    j + 2's before it waits for slice j + 1's.  What the compiler makes of it decides whether that happens: until round 6 the loads sat
    under `if (next slice exists)` and their addresses in VGPR pairs that doubled as load destinations — the waits were placed for the
    path that skips the loads (s_waitcnt vmcnt(0) in front of every prefetch and a few instructions into every reduction).  Pinned here on
    the ISA of the 0.90 kernel (w = 4096, 16 entries per row): in the steady-state loop every stream load takes the scalar-base form,
    nothing waits for vmcnt(0), and the waits leave at least the 10 newest stream loads outstanding."""
    import re, subprocess
    text = (ASM_DIR[0] / "sl_kernels.s").read_text()
    names = re.findall(r"^(_Z\w*sl_band_kernel\w*):", text, flags=re.M)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    for nw in (8, 16):
        sym = next(n for n, d in zip(names, dem) if f"<0, 1, 16, true, true, {nw}>" in d)
        body = text[text.index(sym + ":"):]
        body = body[: body.index(".end_amdhsa_kernel")].split("\n")
        # the steady-state loop: the first loop header after the window barrier whose body issues 20 stream loads (two slices of 10)
        heads = [i for i, l in enumerate(body) if "Loop Header" in l]
        loops = []
        for h in heads:
            label = body[h].split(":")[0].strip()
            back = [i for i, l in enumerate(body) if i > h and re.search(r"s_c?branch\w*\s+" + re.escape(label) + r"\b", l)]
            if back:
                seg = body[h:back[-1]]
                if sum("global_load_dwordx4" in l and " nt" in l for l in seg) == 20:
                    loops.append(seg)
        assert len(loops) == 1, (nw, len(loops))
        seg = loops[0]
        stream = [l for l in seg if "global_load_dwordx4" in l]
        assert all(re.search(r"global_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\]", l) for l in stream), [l for l in stream if " off" in l][:3]
        waits = [int(m.group(1)) for l in seg for m in [re.search(r"s_waitcnt vmcnt\((\d+)\)", l)] if m]
        assert waits and min(waits) >= 10, (nw, waits)
