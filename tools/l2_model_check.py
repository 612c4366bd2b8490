"""Does tools/l2_model restate the paced layout the LIBRARY builds?  Run as a child with SUBLINEAR_HIP_LIB = the SIMT emulator library
(tests/test_l2_model_host.py does): the matrix is built by the library's own kernels (as host fibers), its stream is read through the
emulator's test-only window (tests/simt/simt_debug.cpp), decoded (chunk-transposed index words: slot << 21 | super-panel step << 20 |
column & 0xfffff), and compared with the model's `--dump-tile` output tile by tile: the rows a tile owns, in slot order, and the columns
of its stream, in stream order.

    SUBLINEAR_HIP_LIB=tests/simt/_build/libsublinear_hip_simt.so SIMT_ALLOW=1 python tools/l2_model_check.py --n 200000 --k 16 --cus 8 [--w 0] [--xcd-spans 0]
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--w", type=int, default=0)
    ap.add_argument("--cus", type=int, default=8)
    ap.add_argument("--xcds", type=int, default=8)
    ap.add_argument("--xcd-spans", type=int, default=0)
    ap.add_argument("--tiles", type=int, default=12, help="how many tiles to compare (spread over the launch)")
    a = ap.parse_args()
    os.environ["SL_PW_FORCE"], os.environ["SL_PW_CUS"] = "1", str(a.cus)
    if a.xcd_spans:
        os.environ["SL_PW_XCD"] = str(a.xcds)
    else:
        os.environ["SL_PW_XCD"] = "0"
    import sublinear_time_solver_amd as S
    from sublinear_time_solver_amd import _lib as L
    from sublinear_time_solver_amd import generators as G
    lib = L.load()
    assert hasattr(lib, "simt_debug_pw_layout"), "needs the SIMT emulator library (SUBLINEAR_HIP_LIB)"
    rp, ci, va, _ = G.sdd_rows(a.n, a.k, seed=1, half_bandwidth=a.w)
    m = S.SparseMatrix.from_csr(rp, ci, va, a.n, a.n, column_panels=True)
    idx_p, tp_p = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    out = (C.c_uint64 * 8)()
    lib.simt_debug_pw_layout.restype = C.c_int
    assert lib.simt_debug_pw_layout(C.c_void_p(m._h), C.byref(idx_p), C.byref(tp_p), out), "the library built no paced layout for this matrix"
    n_tiles, chunks, rpw, blocks, deal, pbits, slack, xcd = (int(v) for v in out)
    tile_ptr = np.ctypeslib.as_array(tp_p, shape=(n_tiles + 1,)).copy()
    idx = np.ctypeslib.as_array(idx_p, shape=(chunks * 256,))
    exe = ROOT / "tools" / "l2_model"
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", str(ROOT / "tools" / "l2_model.cpp"), "-o", str(exe)], check=True)
    report = {"n": a.n, "k": a.k, "w": a.w, "cus": a.cus, "library": {"tiles": n_tiles, "rows_per_tile": rpw, "deal": deal, "pbits": pbits, "slack": slack, "xcd": xcd},
              "tiles_compared": 0, "entries_compared": 0}
    for t in sorted(set(np.linspace(0, n_tiles - 1, a.tiles).astype(int).tolist())):
        with tempfile.NamedTemporaryFile(suffix=".bin") as f:
            subprocess.run([str(exe), "--n", str(a.n), "--k", str(a.k), "--w", str(a.w), "--cus", str(a.cus), "--xcds", str(a.xcds), "--pbits", str(pbits),
                            "--xcd-spans", str(a.xcd_spans), "--dump-tile", str(t), f.name], check=True)
            raw = np.fromfile(f.name, dtype=np.uint8)
        hdr = raw[:48].view(np.uint64)
        m_tiles, m_rpw, m_deal, m_slack, n_rows, n_ent = (int(v) for v in hdr)
        rows = raw[48:48 + 8 * n_rows].view(np.uint64)
        cols = raw[48 + 8 * n_rows:48 + 8 * n_rows + 4 * n_ent].view(np.uint32)
        assert (m_tiles, m_rpw, m_deal, m_slack) == (n_tiles, rpw, deal, slack), ("geometry", (m_tiles, m_rpw, m_deal, m_slack), (n_tiles, rpw, deal, slack))
        # the library's stream of tile t: entry e of chunk c sits at idx[c * 256 + (e % 64) * 4 + e // 64]
        c0, c1 = int(tile_ptr[t]), int(tile_ptr[t + 1])
        words = idx[c0 * 256:c1 * 256].reshape(c1 - c0, 64, 4).transpose(0, 2, 1).reshape(-1)
        slot, step, low = words >> 21, (words >> 20) & 1, words & 0xFFFFF
        sp = np.cumsum(step)                                                       # the super-panel after every entry's own step
        real = slot < rpw                                                          # padding and bridging entries carry the spare slot
        lib_cols = ((sp[real].astype(np.uint64) << 20) | low[real]).astype(np.uint32)
        lib_slots = slot[real]
        assert lib_cols.size == cols.size, (t, lib_cols.size, cols.size)
        assert (lib_cols == cols).all(), f"tile {t}: the stream's columns differ at entry {int(np.argmax(lib_cols != cols))}"
        # slots -> rows: slot s of the tile is row rows[s] of the model's list (groups ascending, 16 rows each); every entry's row holds its column
        ent_rows = rows[lib_slots]
        k = a.k
        assert all(int(c) in set(ci[int(r) * k:(int(r) + 1) * k].tolist()) for r, c in list(zip(ent_rows, lib_cols))[:: max(1, lib_cols.size // 500)]), f"tile {t}: slot -> row mapping"
        report["tiles_compared"] += 1
        report["entries_compared"] += int(cols.size)
    report["equal"] = True
    print(json.dumps(report))


if __name__ == "__main__":
    main()
