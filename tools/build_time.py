import ctypes as C, time, sys, os
sys.path.insert(0, os.getcwd())
import torch
from sublinear_time_solver_amd import _lib as L
lib = L.load(); dev = torch.device("cuda", 0)
n, k = 10_000_000, 16
for w, flags, label in ((0, 8, "uniform, no panels"), (0, 0, "uniform, auto (panels)"), (4096, 0, "band 4096")):
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev); ci = torch.empty(n * k, dtype=torch.int32, device=dev)
    va = torch.empty(n * k, dtype=torch.float64, device=dev); b = torch.empty(n, dtype=torch.float64, device=dev)
    L.check(lib.sl_synth_sdd_device(n, k, 1, w, 0, n, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), b.data_ptr()))
    torch.cuda.synchronize()
    for rep in range(2):
        h = C.c_void_p(); t0 = time.perf_counter()
        L.check(lib.sl_matrix_create_csr(n, n, n * k, rp.data_ptr(), ci.data_ptr(), va.data_ptr(), L.SL_MEM_DEVICE, 0, flags, C.byref(h)))
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        mi = L.MatrixInfo(); L.check(lib.sl_matrix_get_info(h, C.byref(mi)))
        if rep: print(f"{label}: create {dt*1e3:.1f} ms, device bytes {mi.device_bytes/1e9:.2f} GB, panels {mi.column_panels}")
        lib.sl_matrix_destroy(h)
    del rp, ci, va, b
