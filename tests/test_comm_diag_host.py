"""The ipc self-test must NAME what a wrong page is.  The classifier is host arithmetic; it is reached here through
the hooked test build of the library (libsublinear_hip_hooks.so, -DSL_DEBUG_HOOKS — never the product), with pages forged the way each of
the four stories would leave them.  No GPU needed."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HOOKED = ROOT / "sublinear_time_solver_amd" / "libsublinear_hip_hooks.so"


@pytest.fixture(scope="module")
def hooks():
    if not HOOKED.exists():
        import subprocess
        subprocess.run(["make", "-C", str(ROOT / "sublinear_time_solver_amd" / "csrc"), "-j8"], check=True, capture_output=True)
    lib = C.CDLL(str(HOOKED))
    lib.sl_hook_classify_ipc_page.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
    lib.sl_hook_ipc_pattern.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
    return lib


def _classify(lib, words, nonce, world, peer):
    w = np.ascontiguousarray(words[:8], dtype=np.uint64)
    out = C.create_string_buffer(512)
    lib.sl_hook_classify_ipc_page(w.ctypes.data, nonce, world, peer, out, 512)
    return out.value.decode()


def _page(lib, seed, words=512):
    p = np.zeros(words, dtype=np.uint64)
    lib.sl_hook_ipc_pattern(seed & (2 ** 64 - 1), words, p.ctypes.data)
    return p


def test_the_product_library_does_not_export_the_hooks():
    prod = C.CDLL(str(ROOT / "sublinear_time_solver_amd" / "libsublinear_hip.so"))
    assert not hasattr(prod, "sl_hook_classify_ipc_page") and not hasattr(prod, "sl_hook_ipc_pattern")


def test_each_story_of_a_wrong_page_is_told_apart(hooks):
    nonce, world = (4242 << 40) ^ (77 << 20) ^ 123456, 8
    # the pattern the kernels write: affine in the index, seed * K1 at word 0 (what the classifier inverts)
    p3 = _page(hooks, nonce + 3)
    assert int(p3[1]) - int(p3[0]) in (0xBF58476D1CE4E5B9, 0xBF58476D1CE4E5B9 - 2 ** 64)
    # (1) never written
    assert "zeros" in _classify(hooks, np.zeros(8, dtype=np.uint64), nonce, world, 3) and "ordering" in _classify(hooks, np.zeros(8, dtype=np.uint64), nonce, world, 3)
    # (2) another rank's page of this job
    msg = _classify(hooks, _page(hooks, nonce + 5), nonce, world, 3)
    assert "rank 5 of THIS job" in msg and "rank 3's was meant" in msg
    # (3) another job's page (round 4's suspicion: two jobs side by side on one device)
    other = (999 << 40) ^ (78 << 20) ^ 5
    msg = _classify(hooks, _page(hooks, other + 3), nonce, world, 3)
    assert "ANOTHER communicator" in msg and f"{other + 3:016x}" in msg and f"{nonce:016x}" in msg
    # (4) torn: right pattern with words of something else inside the first eight
    torn = _page(hooks, nonce + 3).copy()
    torn[2] = 0xDEADBEEF
    assert "no rank's pattern" in _classify(hooks, torn, nonce, world, 3)
    # the right first words but a wrong count elsewhere in the page
    assert "torn copy" in _classify(hooks, _page(hooks, nonce + 3), nonce, world, 3)
    # seeds at the edges of the job's range: world - 1 is ours, world is not
    assert "rank 7 of THIS job" in _classify(hooks, _page(hooks, nonce + 7), nonce, world, 0)
    assert "ANOTHER communicator" in _classify(hooks, _page(hooks, nonce + 8), nonce, world, 0)
