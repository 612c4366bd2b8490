"""Golden vectors of the reference's `generate -t diagonally-dominant` AS WRITTEN: tests/golden/reference_ts_generate.npz  (G15).

BASELINE config 0 is "n=1000 diagonally-dominant CSR (generate -t diagonally-dominant -s 1000)".  The reference's generator
(MatrixTools.generateDiagonallyDominantMatrix, src/mcp/tools/matrix.ts:297-322, behind src/cli/index.ts:308-352) draws from Math.random, so its
output is not reproducible; this repository's twin (generators.gen1000_dense) replaces Math.random by the reference's own seeded generator
createSeededRandom (core/utils.ts:161-168).  Here the reference's method body is READ from /root/reference, its TypeScript annotations
removed, and evaluated by node with Math.random = createSeededRandom(seed) — the same substitution — so the twin can be compared with the
reference's own loop entry for entry: the full matrix at size 150, and at size 1000 (config 0) the entry count, the exact sums of the diagonal
and of all values, and a SHA-256 of the row-major float64 table.

    python tests/golden/make_golden_ts_generate.py
"""
import hashlib
import json
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_golden_walk import REF, method_body, strip_types      # noqa: E402

CASES = [dict(name="dd150_strength2_seed42", size=150, strength=2.0, seed=42, full=True),
         dict(name="dd150_strength1.5_seed7", size=150, strength=1.5, seed=7, full=True),
         dict(name="dd1000_strength2_seed42", size=1000, strength=2.0, seed=42, full=False)]      # BASELINE config 0


def main():
    tools = (REF / "src" / "mcp" / "tools" / "matrix.ts").read_text()
    utils = (REF / "src" / "core" / "utils.ts").read_text()
    gen = strip_types(method_body(tools, r"private static generateDiagonallyDominantMatrix\s*"))
    seeded = strip_types(method_body(utils, r"export function createSeededRandom\s*"))
    runner = f"""
function createSeededRandom(seed) {{ {seeded} }}
function generateDiagonallyDominantMatrix(size, strength) {{ {gen} }}
const out = [];
for (const c of require('./cases.json')) {{
  Math.random = createSeededRandom(c.seed);
  out.push(generateDiagonallyDominantMatrix(c.size, c.strength).data);
}}
process.stdout.write(JSON.stringify(out));
"""
    with tempfile.TemporaryDirectory(prefix="golden_ts_gen_") as d:
        scratch = Path(d)
        (scratch / "cases.json").write_text(json.dumps(CASES))
        (scratch / "run.js").write_text(runner)
        p = subprocess.run(["node", "--max-old-space-size=4096", "run.js"], cwd=scratch, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        res = json.loads(p.stdout)
    out = {"names": np.array([c["name"] for c in CASES])}
    for c, data in zip(CASES, res):
        a = np.asarray(data, dtype=np.float64)
        k = c["name"]
        out[k + "/params"] = np.array([c["size"], c["seed"], int((a != 0).sum())], dtype=np.int64)
        out[k + "/strength"] = np.array([c["strength"]], dtype=np.float64)
        out[k + "/sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)
        import math
        out[k + "/sums"] = np.array([math.fsum(np.diag(a).tolist()), math.fsum(a.ravel().tolist())], dtype=np.float64)
        if c["full"]:
            rr, cc = np.nonzero(a)
            out[k + "/rows"], out[k + "/cols"], out[k + "/values"] = rr.astype(np.uint32), cc.astype(np.uint32), a[rr, cc]
    path = ROOT / "tests" / "golden" / "reference_ts_generate.npz"
    np.savez_compressed(path, **out)
    print(path, {str(k): int(out[str(k) + "/params"][2]) for k in out["names"]})


if __name__ == "__main__":
    main()
