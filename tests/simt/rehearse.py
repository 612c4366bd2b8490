"""python tests/simt/rehearse.py <script.py> [args...] — run one of the GPU-session tools (tools/*.py, tests/full_solve_report.py) against the
SIMT emulator with torch stood in for by tests/simt/fake_torch.py (host arrays as "device" buffers): a REHEARSAL of the script's code path at
a small size, so that the first GPU call of a session is not spent on a typo.  Needs SUBLINEAR_HIP_LIB = the emulator library.  Whatever
the script prints is not a measurement."""
import importlib.util
import os
import runpy
import sys
from pathlib import Path

here = Path(__file__).resolve().parent
lib = os.environ.get("SUBLINEAR_HIP_LIB", "")
if "simt" not in Path(lib).name:
    sys.exit("rehearse.py: SUBLINEAR_HIP_LIB must name the emulator library (tests/simt/_build/libsublinear_hip_simt.so)")
spec = importlib.util.spec_from_file_location("torch", str(here / "fake_torch.py"))
sys.modules["torch"] = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sys.modules["torch"])
sys.path.insert(0, str(here.parent.parent))
script = sys.argv[1]
sys.argv = sys.argv[1:]
print(f"[rehearsal under the SIMT emulator — no figure below is a measurement] {script}", file=sys.stderr)
runpy.run_path(script, run_name="__main__")
