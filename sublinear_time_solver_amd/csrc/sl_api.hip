// sl_api.hip — the C ABI of include/sublinear_hip.h: context, matrix lifetime, primitives,
// and the host-side iteration control of NeumannSolver::solve (neumann.rs:469-555).
// Host control only decides WHEN kernels run; all arithmetic on vectors happens on the device.
#include "sl_internal.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstring>
#include <dlfcn.h>
#include <initializer_list>
#include <numeric>
#include <vector>

// ---- context ---------------------------------------------------------------------------------
sl_ctx &sl_context()
{
    static thread_local sl_ctx ctx;
    return ctx;
}
// ---- tracing / logging ------------------------------------------------------------------------
namespace {
struct roctx_api {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    roctx_api()
    {
        const char *off = getenv("SL_ROCTX");
        if (off && off[0] == '0') return;
        for (const char *lib : {"libroctx64.so", "libroctx64.so.4", "librocprofiler-sdk-roctx.so"}) {
            if (void *h = dlopen(lib, RTLD_LAZY | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const roctx_api &roctx() { static const roctx_api api; return api; }     // initialised once, thread-safe (C++11 static)
}
void sl_range_push(const char *name) { if (roctx().push) roctx().push(name); sl_log(2, "range > %s", name); }
void sl_range_pop() { if (roctx().pop) roctx().pop(); }
int sl_log_level()
{
    static const int level = [] { const char *e = getenv("SL_LOG"); return e && *e ? atoi(e) : 0; }();
    return level;
}
void sl_log(int level, const char *fmt, ...)
{
    if (level > sl_log_level()) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[sublinear_hip] %s\n", buf);
}

sl_status sl_fail(sl_status s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    sl_context().last_error = buf;
    sl_log(1, "status %d: %s", (int)s, buf);
    return s;
}
static bool poison_on() { static const bool on = [] { const char *e = getenv("SL_POISON_ALLOC"); return e && *e == '1'; }(); return on; }
void sl_poison(void *p, size_t bytes)
{
    if (!poison_on() || !p || !bytes) return;
    (void)hipDeviceSynchronize();                           // (debug mode: whatever still reads an old block of the cache finishes first)
    (void)hipMemset(p, 0xA5, bytes);
    (void)hipDeviceSynchronize();
}
hipError_t sl_malloc_checked(void **p, size_t bytes)
{
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) sl_poison(*p, bytes);
    return e;
}

void *sl_scratch(size_t bytes)
{
    sl_ctx &c = sl_context();
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (c.scratch && c.scratch_device == dev && c.scratch_bytes >= bytes) return c.scratch;
    if (c.scratch) { hipFree(c.scratch); c.scratch = nullptr; c.scratch_bytes = 0; }
    size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    if (sl_malloc(&c.scratch, want) != hipSuccess) return nullptr;
    c.scratch_bytes = want;
    c.scratch_device = dev;
    return c.scratch;
}

static bool staging_pageable() { static const bool on = [] { const char *e = getenv("SL_STAGING"); return e && !strcmp(e, "pageable"); }(); return on; }
static void *pinned_staging(size_t *cap)
{
    sl_ctx &c = sl_context();
    const size_t want = 4u << 20;
    if (!c.pinned) {
        if (c.pinned_failed) return nullptr;                         // asked once, refused: the pageable calls from here on, no retry per transfer
        if (hipHostMalloc(&c.pinned, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();                                 // the fallback below works: do not leave this failure for the next SL_HIP(hipGetLastError()) to report
            c.pinned = nullptr; c.pinned_failed = true;
            return nullptr;
        }
        c.pinned_bytes = want;
    }
    *cap = c.pinned_bytes;
    return c.pinned;
}
sl_status sl_read_back(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st)
{
    if (!bytes) return SL_OK;
    size_t cap = 0;
    void *pin = staging_pageable() ? nullptr : pinned_staging(&cap);
    if (!pin) {                                                    // SL_STAGING=pageable, or no page-locked memory to be had
        SL_HIP(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, st));
        SL_HIP(hipStreamSynchronize(st));
        return SL_OK;
    }
    for (size_t off = 0; off < bytes; off += cap) {
        const size_t len = bytes - off < cap ? bytes - off : cap;
        SL_HIP(hipMemcpyAsync(pin, static_cast<const char *>(dev_src) + off, len, hipMemcpyDeviceToHost, st));
        SL_HIP(hipStreamSynchronize(st));
        memcpy(static_cast<char *>(host_dst) + off, pin, len);
    }
    return SL_OK;
}
sl_status sl_upload(void *dev_dst, const void *host_src, size_t bytes, hipStream_t st)
{
    if (!bytes) return SL_OK;
    size_t cap = 0;
    void *pin = staging_pageable() ? nullptr : pinned_staging(&cap);
    if (!pin) {
        SL_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, st));
        SL_HIP(hipStreamSynchronize(st));                               // the contract of sl_upload: host_src is reusable and the data is on the device on return
        return SL_OK;
    }
    for (size_t off = 0; off < bytes; off += cap) {
        const size_t len = bytes - off < cap ? bytes - off : cap;
        memcpy(pin, static_cast<const char *>(host_src) + off, len);
        SL_HIP(hipMemcpyAsync(static_cast<char *>(dev_dst) + off, pin, len, hipMemcpyHostToDevice, st));
        SL_HIP(hipStreamSynchronize(st));                           // the staging buffer is free again, the table is on the device
    }
    return SL_OK;
}

bool sl_side_stream(sl_ctx &c)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (c.side && c.side_device == dev) return true;
    if (c.side) { (void)hipStreamDestroy(c.side); (void)hipEventDestroy(c.ev_fork); (void)hipEventDestroy(c.ev_join); c.side = nullptr; }
    // the side stream carries the short chains that others wait for (hub rows beside the slice kernel; a partition's edge blocks, halo
    // ticket and pulls beside the interior): highest priority, so that its blocks are dispatched ahead of the long kernel's
    int pr_least = 0, pr_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
    if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, pr_greatest) != hipSuccess) { c.side = nullptr; return false; }
    if (hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(c.side); c.side = nullptr; return false;
    }
    c.side_device = dev;
    return true;
}

// ---- workspace pool --------------------------------------------------------------------------
static size_t ws_cache_limit()
{
    static size_t lim = 0;
    if (!lim) { const char *e = getenv("SL_WORKSPACE_CACHE_MB"); lim = (size_t)(e ? atoll(e) : 16384) << 20; if (!lim) lim = 1; }
    return lim;
}
void *sl_ws_alloc(size_t bytes)
{
    sl_ctx &c = sl_context();
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    bytes = (bytes + 255) & ~(size_t)255;
    int best = -1;
    for (size_t i = 0; i < c.ws.size(); ++i) {        // smallest cached block that fits and is not grossly oversized
        const auto &b = c.ws[i];
        if (b.in_use || b.device != dev || b.bytes < bytes || b.bytes > 4 * bytes + (1u << 20)) continue;
        if (best < 0 || b.bytes < c.ws[best].bytes) best = (int)i;
    }
    if (best >= 0) { c.ws[best].in_use = true; sl_poison(c.ws[best].p, c.ws[best].bytes); return c.ws[best].p; }
    void *p = nullptr;
    if (sl_malloc(&p, bytes) != hipSuccess) {
        sl_release_workspace();                        // give the cache back and try once more
        if (sl_malloc(&p, bytes) != hipSuccess) return nullptr;
    }
    c.ws.push_back({p, bytes, dev, true});
    return p;
}
void sl_ws_free(void *p)
{
    sl_ctx &c = sl_context();
    size_t cached = 0;
    for (const auto &b : c.ws) if (!b.in_use) cached += b.bytes;
    for (size_t i = 0; i < c.ws.size(); ++i) {
        if (c.ws[i].p != p) continue;
        if (cached + c.ws[i].bytes > ws_cache_limit()) { hipFree(p); c.ws.erase(c.ws.begin() + i); }
        else c.ws[i].in_use = false;
        return;
    }
    hipFree(p);                                        // not ours (allocated by another thread): plain free
}
extern "C" void sl_release_workspace(void)
{
    sl_ctx &c = sl_context();
    // the thread's page-locked staging buffer goes with its workspace: a thread that built a matrix and calls this before it ends leaks nothing
    if (c.pinned) { (void)hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_bytes = 0; }
    c.pinned_failed = false;
    for (size_t i = 0; i < c.ws.size();) {
        if (!c.ws[i].in_use) { hipFree(c.ws[i].p); c.ws.erase(c.ws.begin() + i); }
        else ++i;
    }
}

namespace {

sl_status require_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return sl_fail(SL_DEVICE_ERROR, "no HIP device available (%s); libsublinear_hip has no CPU fallback",
                       e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return SL_OK;
}

// bring a vector to the device if it lives on the host; returns the device pointer
sl_status stage_in(const double *src, uint64_t n, sl_mem where, DevBuf &tmp, const double **dev)
{
    if (where == SL_MEM_DEVICE) { *dev = src; return SL_OK; }
    SL_TRY(tmp.alloc(n * sizeof(double)));
    SL_HIP(hipMemcpyAsync(tmp.p, src, n * sizeof(double), hipMemcpyHostToDevice, sl_context().stream));
    *dev = tmp.as<double>();
    return SL_OK;
}
sl_status read_scalars(const double *d, double *h, int count)
{
    hipStream_t st = sl_context().stream;
    SL_HIP(hipMemcpyAsync(h, d, count * sizeof(double), hipMemcpyDeviceToHost, st));
    SL_HIP(hipStreamSynchronize(st));
    return SL_OK;
}
sl_row_args row_args(const sl_matrix *m) { return sl_matrix_row_args(m); }
// squared-norm thresholds for the device-side stop rules: for every h >= 0
//   (sqrt(h) <  tol) == (h <  sq_threshold_lt(tol))      and      (sqrt(h) <= tol) == (h <= sq_threshold_le(tol))
// (sqrt is monotone and correctly rounded; the loops move a few ulps at most)
double sq_threshold_lt(double tol)
{
    if (tol != tol) return tol;                     // NaN: never true on either side
    if (!(tol > 0.0)) return 0.0;                   // sqrt(h) < tol never holds; h < 0 never holds
    double t = tol * tol;
    while (t > 0.0 && std::sqrt(t) >= tol) t = std::nextafter(t, 0.0);
    while (std::sqrt(t) < tol) t = std::nextafter(t, INFINITY);      // smallest t with sqrt(t) >= tol
    return t;
}
double sq_threshold_le(double tol)
{
    if (tol != tol) return tol;
    if (tol < 0.0) return -1.0;
    double t = tol * tol;
    while (std::sqrt(t) <= tol && t < INFINITY) t = std::nextafter(t, INFINITY);
    while (t > 0.0 && std::sqrt(t) > tol) t = std::nextafter(t, 0.0);  // largest t with sqrt(t) <= tol
    return t;
}
size_t partial_bytes(const sl_matrix *m) { return (((size_t)sl_row_grid(m->n_slices) + m->n_long) * 2 + 4096) * sizeof(double); }
} // namespace

sl_row_args sl_matrix_row_args(const sl_matrix *m)
{
    sl_row_args a;
    memset(&a, 0, sizeof(a));
    a.slice_ptr = m->d_slice_ptr; a.row_len = m->d_row_len; a.cols = m->d_cols; a.cols16 = m->d_cols16; a.vals = m->d_vals;
    a.n_rows = m->n_rows; a.n_cols = m->n_cols; a.n_slices = m->n_slices; a.row_offset = m->row_offset;
    a.bandwidth = m->bandwidth; a.uniform_width = m->uniform_width; a.max_row_nnz = m->max_row_nnz;
    a.csr_ptr = m->d_row_ptr; a.csr_idx = m->d_col_idx; a.csr_val = m->d_values;
    a.long_rows = m->d_long_rows; a.n_long = (uint32_t)m->n_long;
    a.pan_tile_ptr = m->d_pan_tile_ptr; a.pan_row = m->d_pan_row; a.pan_col = m->d_pan_col; a.pan_val = m->d_pan_val;
    a.n_pan_tiles = (uint32_t)m->n_pan_tiles; a.pan_balanced = m->pan_balanced ? 1u : 0u;
    a.pw_idx = m->d_pw_idx; a.pw_val = m->d_pw_val; a.pw_tile_ptr = m->d_pw_tile_ptr;
    a.pw_tiles = (uint32_t)m->n_pw_tiles; a.pw_rpw = m->pw_rpw; a.pw_blocks = m->pw_blocks; a.pw_slack = m->pw_slack;
    a.pw_deal = m->pw_deal; a.pw_pbits = m->pw_pbits; a.pw_xcd = m->pw_xcd; a.pw_span_tab = m->d_pw_span_tab;
    a.pwr_idx = m->d_pwr_idx; a.pwr_val = m->d_pwr_val; a.pwr_base = m->d_pwr_base; a.pwr_tile_ptr = m->d_pwr_tile_ptr; a.pwr_diag = m->d_pwr_diag;
    a.pwr_tiles = (uint32_t)m->n_pwr_tiles; a.pwr_rpb = m->pwr_rpb; a.pwr_blocks = m->pwr_blocks;
    return a;
}

// f64::powi as rustc emits it for a run-time exponent (llvm.powi.f64 -> compiler-rt __powidf2): square and multiply, not libm pow —
// its rounding sequence is part of estimate_error_bounds' result (neumann.rs:336)
static double sl_powi(double a, int b)
{
    const bool recip = b < 0;
    double r = 1.0;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0 / r : r;
}

// ---- library -----------------------------------------------------------------------------------
extern "C" {

int sl_abi_version(void) { return SL_ABI_VERSION; }
const char *sl_last_error_message(void) { return sl_context().last_error.c_str(); }
const char *sl_status_string(sl_status s)
{
    switch (s) {
    case SL_OK: return "OK";
    case SL_NOT_DIAGONALLY_DOMINANT: return "MatrixNotDiagonallyDominant";
    case SL_NUMERICAL_INSTABILITY: return "NumericalInstability";
    case SL_CONVERGENCE_FAILURE: return "ConvergenceFailure";
    case SL_INVALID_INPUT: return "InvalidInput";
    case SL_DIMENSION_MISMATCH: return "DimensionMismatch";
    case SL_UNSUPPORTED_FORMAT: return "UnsupportedMatrixFormat";
    case SL_ALLOCATION: return "MemoryAllocationError";
    case SL_INDEX_OUT_OF_BOUNDS: return "IndexOutOfBounds";
    case SL_INVALID_SPARSE_MATRIX: return "InvalidSparseMatrix";
    case SL_ALGORITHM_ERROR: return "AlgorithmError";
    case SL_DEVICE_ERROR: return "DeviceError";
    }
    return "Unknown";
}
sl_status sl_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return SL_OK;
}
sl_status sl_device_name(int device, char *name, uint64_t capacity)
{
    if (!name || !capacity) return sl_fail(SL_INVALID_INPUT, "null argument");
    name[0] = 0;
    SL_TRY(require_device());
    hipDeviceProp_t prop;
    SL_HIP(hipGetDeviceProperties(&prop, device));
    snprintf(name, (size_t)capacity, "%s", prop.gcnArchName);
    return SL_OK;
}
sl_status sl_set_device(int device)
{
    SL_TRY(require_device());
    SL_HIP(hipSetDevice(device));
    return SL_OK;
}
sl_status sl_set_stream(void *hip_stream)
{
    sl_context().stream = static_cast<hipStream_t>(hip_stream);
    return SL_OK;
}
sl_status sl_synchronize(void)
{
    SL_TRY(require_device());
    SL_HIP(hipStreamSynchronize(sl_context().stream));
    return SL_OK;
}

// ---- matrices ------------------------------------------------------------------------------------
void sl_matrix_destroy(sl_matrix *m)
{
    if (!m) return;
    hipFree(m->d_slice_ptr); hipFree(m->d_row_len); hipFree(m->d_cols); hipFree(m->d_cols16); hipFree(m->d_vals);
    hipFree(m->d_row_ptr); hipFree(m->d_col_idx); hipFree(m->d_values);
    hipFree(m->d_tptr); hipFree(m->d_trow); hipFree(m->d_tval); hipFree(m->d_tent); hipFree(m->d_long_rows);
    hipFree(m->d_pan_tile_ptr); hipFree(m->d_pan_row); hipFree(m->d_pan_col); hipFree(m->d_pan_val);
    hipFree(m->d_pw_idx); hipFree(m->d_pw_val); hipFree(m->d_pw_tile_ptr); hipFree(m->d_pw_span_tab);
    hipFree(m->d_pwr_idx); hipFree(m->d_pwr_val); hipFree(m->d_pwr_base); hipFree(m->d_pwr_tile_ptr); hipFree(m->d_pwr_diag);
    hipFree(m->d_colval);
    delete m;
}

sl_status sl_matrix_create_csr(uint64_t n_rows, uint64_t n_cols, uint64_t nnz, const uint32_t *row_ptr,
                               const uint32_t *col_idx, const double *values, sl_mem where,
                               uint64_t row_offset, uint32_t flags, sl_matrix **out)
{
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!row_ptr || (nnz && (!col_idx || !values))) return sl_fail(SL_INVALID_INPUT, "null CSR array");
    if (n_rows > 0xffffffffull || n_cols > 0xffffffffull || nnz > 0xffffffffull)
        return sl_fail(SL_INVALID_INPUT, "dimensions exceed IndexType = u32 (types.rs:22)");
    if (row_offset + n_rows > n_cols && row_offset != 0)
        return sl_fail(SL_DIMENSION_MISMATCH, "row slice [%llu, %llu) exceeds the global dimension %llu",
                       (unsigned long long)row_offset, (unsigned long long)(row_offset + n_rows), (unsigned long long)n_cols);
    SL_TRY(require_device());
    SL_ABI_BEGIN
    sl_matrix *m = new sl_matrix();
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz; m->row_offset = row_offset; m->flags = flags;
    m->caller_device_arrays = where != SL_MEM_HOST;
    hipGetDevice(&m->device);
    const bool keep = (flags & (SL_MATRIX_KEEP_CSR | SL_MATRIX_WITH_TRANSPOSE)) != 0;
    sl_status st;
    if (where == SL_MEM_HOST) {
        DevBuf rp, ci, va;
        hipStream_t s = sl_context().stream;
        st = rp.alloc_owned((n_rows + 1) * sizeof(uint32_t));
        if (st == SL_OK) st = ci.alloc_owned(nnz * sizeof(uint32_t));
        if (st == SL_OK) st = va.alloc_owned(nnz * sizeof(double));
        if (st == SL_OK) {
            hipError_t e = hipMemcpyAsync(rp.p, row_ptr, (n_rows + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess && nnz) e = hipMemcpyAsync(ci.p, col_idx, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, s);
            if (e == hipSuccess && nnz) e = hipMemcpyAsync(va.p, values, nnz * sizeof(double), hipMemcpyHostToDevice, s);
            if (e != hipSuccess) st = sl_fail(SL_DEVICE_ERROR, "CSR upload failed: %s", hipGetErrorString(e));
        }
        if (st == SL_OK) {
            if (keep) { // hand the uploaded buffers over instead of copying them again
                st = sl_build_from_device_csr(m, rp.as<uint32_t>(), ci.as<uint32_t>(), va.as<double>(), false);
                if (st == SL_OK && !m->d_row_ptr) {      // (long rows make the build keep its own copy already)
                    m->d_row_ptr = static_cast<uint32_t *>(rp.release()); m->d_col_idx = static_cast<uint32_t *>(ci.release());
                    m->d_values = static_cast<double *>(va.release());
                    m->device_bytes += (n_rows + 1) * sizeof(uint32_t) + nnz * 12;
                }
            } else {
                st = sl_build_from_device_csr(m, rp.as<uint32_t>(), ci.as<uint32_t>(), va.as<double>(), false);
            }
        }
    } else {
        st = sl_build_from_device_csr(m, row_ptr, col_idx, values, keep);
    }
    if (st != SL_OK) { sl_matrix_destroy(m); return st; }
    *out = m;
    return SL_OK;
    SL_ABI_END
}

sl_status sl_matrix_create_from_triplets(uint64_t n_triplets, const uint64_t *rows, const uint64_t *cols,
                                         const double *values, uint64_t n_rows, uint64_t n_cols,
                                         uint32_t flags, sl_matrix **out)
{
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (n_triplets && (!rows || !cols || !values)) return sl_fail(SL_INVALID_INPUT, "null triplet array");
    if (n_rows > 0xffffffffull || n_cols > 0xffffffffull || n_triplets > 0xffffffffull)      // before any host allocation of that size
        return sl_fail(SL_INVALID_INPUT, "dimensions exceed IndexType = u32 (types.rs:22)");
    SL_ABI_BEGIN
    // matrix/mod.rs:165-187: validation in input order
    for (uint64_t k = 0; k < n_triplets; ++k) {
        if (rows[k] >= n_rows)
            return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "row index %llu in triplet %llu (max %llu)", (unsigned long long)rows[k],
                           (unsigned long long)k, (unsigned long long)(n_rows ? n_rows - 1 : 0));
        if (cols[k] >= n_cols)
            return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "column index %llu in triplet %llu (max %llu)", (unsigned long long)cols[k],
                           (unsigned long long)k, (unsigned long long)(n_cols ? n_cols - 1 : 0));
        if (!std::isfinite(values[k]))
            return sl_fail(SL_INVALID_INPUT, "Non-finite value at (%llu, %llu)", (unsigned long long)rows[k], (unsigned long long)cols[k]);
    }
    // sparse.rs:536-542 drop exact zeros; sparse.rs:91-96 stable sort by (row, col)
    std::vector<uint64_t> idx;
    idx.reserve(n_triplets);
    for (uint64_t k = 0; k < n_triplets; ++k) if (values[k] != 0.0) idx.push_back(k);
    std::stable_sort(idx.begin(), idx.end(), [&](uint64_t a, uint64_t b) {
        if (rows[a] != rows[b]) return rows[a] < rows[b];
        return cols[a] < cols[b];
    });
    const uint64_t nnz = idx.size();
    std::vector<uint32_t> rp(n_rows + 1, 0), ci(nnz);
    std::vector<double> va(nnz);
    for (uint64_t k = 0; k < nnz; ++k) { rp[rows[idx[k]] + 1] += 1; ci[k] = (uint32_t)cols[idx[k]]; va[k] = values[idx[k]]; }
    for (uint64_t i = 0; i < n_rows; ++i) rp[i + 1] += rp[i];
    return sl_matrix_create_csr(n_rows, n_cols, nnz, rp.data(), ci.data(), va.data(), SL_MEM_HOST, 0, flags, out);
    SL_ABI_END
}

sl_status sl_matrix_get_info(const sl_matrix *m, sl_matrix_info *info)
{
    if (!m || !info) return sl_fail(SL_INVALID_INPUT, "null argument");
    info->n_rows = m->n_rows; info->n_cols = m->n_cols; info->nnz = m->nnz; info->row_offset = m->row_offset;
    info->padded_nnz = m->padded_nnz; info->n_slices = m->n_slices; info->device_bytes = m->device_bytes; info->bandwidth = m->bandwidth;
    info->max_row_nnz = m->max_row_nnz; info->min_row_nnz = m->min_row_nnz; info->uniform_width = m->uniform_width;
    info->has_transpose = m->d_tptr != nullptr;
    info->long_row_threshold = m->long_row;
    info->column_panels = m->d_pwr_idx ? 4u : m->d_pw_idx ? (m->pw_band ? 3u : 2u) : (m->d_pan_tile_ptr ? 1u : 0u);      // 2 = the paced layout, 3 = its wide-band form, 4 = the order-free column stream
    info->reserved = 0;
    info->n_long_rows = (uint32_t)m->n_long;
    return SL_OK;
}

sl_status sl_matrix_download_csr(const sl_matrix *m, uint32_t *row_ptr, uint32_t *col_idx, double *values)
{
    if (!m) return sl_fail(SL_INVALID_INPUT, "null matrix");
    hipStream_t s = sl_context().stream;
    if (!m->d_row_ptr) {      // no raw copy kept: the rows are written back from the slice layout (SparseMatrix::as_csr / to_triplets always work)
        SL_ABI_BEGIN
        DevBuf rp, ci, va;
        SL_TRY(rp.alloc((m->n_rows + 1) * sizeof(uint32_t))); SL_TRY(ci.alloc((m->nnz ? m->nnz : 1) * sizeof(uint32_t))); SL_TRY(va.alloc((m->nnz ? m->nnz : 1) * sizeof(double)));
        SL_TRY(sl_matrix_slices_to_csr(m, rp.as<uint32_t>(), ci.as<uint32_t>(), va.as<double>()));
        SL_HIP(hipMemcpyAsync(row_ptr, rp.p, (m->n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        if (m->nnz) {
            SL_HIP(hipMemcpyAsync(col_idx, ci.p, m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            SL_HIP(hipMemcpyAsync(values, va.p, m->nnz * sizeof(double), hipMemcpyDeviceToHost, s));
        }
        SL_HIP(hipStreamSynchronize(s));
        return SL_OK;
        SL_ABI_END
    }
    SL_HIP(hipMemcpyAsync(row_ptr, m->d_row_ptr, (m->n_rows + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    if (m->nnz) {
        SL_HIP(hipMemcpyAsync(col_idx, m->d_col_idx, m->nnz * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        SL_HIP(hipMemcpyAsync(values, m->d_values, m->nnz * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}

sl_status sl_matrix_is_diagonally_dominant(const sl_matrix *m, int *is_dd)
{
    if (!m || !is_dd) return sl_fail(SL_INVALID_INPUT, "null argument");
    unsigned long long hs[4];
    SL_TRY(sl_matrix_diag_pass(m, nullptr, hs));
    *is_dd = (hs[0] & 1ull) ? 0 : 1;
    return SL_OK;
}

sl_status sl_matrix_diagonal_dominance_factor(const sl_matrix *m, int *has_factor, double *factor)
{
    if (!m || !has_factor || !factor) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    double h[2];
    SL_TRY(sl_matrix_cond_pass(m, h));
    *has_factor = std::isfinite(h[0]) ? 1 : 0;                          // matrix/mod.rs:508-512: Some(min) only when finite
    *factor = *has_factor ? h[0] : 0.0;
    return SL_OK;
    SL_ABI_END
}

sl_status sl_matrix_spectral_radius_estimate(const sl_matrix *m, double *radius)
{
    if (!m || !radius) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    double h[2];
    SL_TRY(sl_matrix_cond_pass(m, h));
    *radius = h[1];
    return SL_OK;
    SL_ABI_END
}

// ---- trait Matrix: get / row_iter / col_iter / frobenius_norm / sparsity_info (matrix/mod.rs:33-41, 74-82, 523-545) ----
sl_status sl_matrix_get(const sl_matrix *m, uint64_t row, uint64_t col, int *found, double *value)
{
    if (!m || !found || !value) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    return sl_matrix_get_entry(m, row, col, found, value);
    SL_ABI_END
}

sl_status sl_matrix_row(const sl_matrix *m, uint64_t row, uint64_t capacity, uint32_t *cols, double *values, uint64_t *count)
{
    if (!m || !count || (capacity && (!cols || !values))) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    return sl_matrix_fetch_row(m, row, capacity, cols, values, count);
    SL_ABI_END
}

sl_status sl_matrix_col(const sl_matrix *m, uint64_t col, uint64_t capacity, uint32_t *rows, double *values, uint64_t *count)
{
    if (!m || !count || (capacity && (!rows || !values))) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    return sl_matrix_fetch_col(m, col, capacity, rows, values, count);
    SL_ABI_END
}

sl_status sl_matrix_frobenius_norm(const sl_matrix *m, double *norm)
{
    if (!m || !norm) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    double sq = 0.0;
    SL_TRY(sl_matrix_frobenius_sq(m, &sq));
    *norm = std::sqrt(sq);
    return SL_OK;
    SL_ABI_END
}

sl_status sl_matrix_sparsity_info(const sl_matrix *m, sl_sparsity_info *info)
{
    if (!m || !info) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    info->nnz = m->nnz; info->rows = m->n_rows; info->cols = m->n_cols;
    const uint64_t total = m->n_rows * m->n_cols;                                          // SparsityInfo::new, types.rs:346-358
    info->sparsity_ratio = total > 0 ? (double)m->nnz / (double)total : 0.0;
    info->avg_nnz_per_row = m->n_rows > 0 ? (double)m->nnz / (double)m->n_rows : 0.0;
    info->max_nnz_per_row = m->max_row_nnz;
    uint64_t bw = 0;
    SL_TRY(sl_matrix_entry_bandwidth(m, &bw));                                             // Some(max |r - c|) over ALL entries, matrix/mod.rs:535-541
    info->bandwidth = bw;
    info->is_banded = bw < m->n_rows / 4 ? 1 : 0;                                          // :542
    info->reserved = 0;
    return SL_OK;
    SL_ABI_END
}

// ---- the `&mut self` methods of SparseMatrix: scale / add_diagonal (matrix/mod.rs:346-372) ----
sl_status sl_matrix_scale(sl_matrix *m, double factor)
{
    if (!m) return sl_fail(SL_INVALID_INPUT, "null matrix");
    SL_ABI_BEGIN
    SL_TRY(require_device());
    return sl_matrix_scale_values(m, factor);
    SL_ABI_END
}

sl_status sl_matrix_add_diagonal(sl_matrix *m, double alpha)
{
    if (!m) return sl_fail(SL_INVALID_INPUT, "null matrix");
    SL_ABI_BEGIN
    // matrix/mod.rs:356-361: only square matrices — a row slice of a square system (global column ids) counts as part of one
    const bool row_slice = (m->row_offset > 0 || (m->flags & SL_MATRIX_ROW_SLICE)) && m->row_offset + m->n_rows <= m->n_cols;
    if (m->n_rows != m->n_cols && !row_slice) return sl_fail(SL_INVALID_INPUT, "Cannot add diagonal to non-square matrix");
    SL_TRY(require_device());
    return sl_matrix_shift_diagonal(m, alpha);
    SL_ABI_END
}

sl_status sl_matrix_diagonal_inverse(const sl_matrix *m, double *dinv, sl_mem where)
{
    if (!m || !dinv) return sl_fail(SL_INVALID_INPUT, "null argument");
    DevBuf tmp;
    double *d = dinv;
    if (where == SL_MEM_HOST) { SL_TRY(tmp.alloc(m->n_rows * sizeof(double))); d = tmp.as<double>(); }
    unsigned long long hs[4];
    SL_TRY(sl_matrix_diag_pass(m, d, hs));
    if (hs[0] & 2ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (hs[0] & 4ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    if (where == SL_MEM_HOST) {
        SL_HIP(hipMemcpyAsync(dinv, d, m->n_rows * sizeof(double), hipMemcpyDeviceToHost, sl_context().stream));
        SL_HIP(hipStreamSynchronize(sl_context().stream));
    }
    return SL_OK;
}

// ---- primitives ------------------------------------------------------------------------------------
sl_status sl_spmv(const sl_matrix *m, const double *x, double *y, sl_order order, sl_mem where)
{
    if (!m || !x || !y) return sl_fail(SL_INVALID_INPUT, "null argument");
    hipStream_t s = sl_context().stream;
    DevBuf xin, yout;
    const double *dx;
    SL_TRY(stage_in(x, m->n_cols, where, xin, &dx));
    double *dy = y;
    if (where == SL_MEM_HOST) { SL_TRY(yout.alloc(m->n_rows * sizeof(double))); dy = yout.as<double>(); }
    sl_row_args a = row_args(m);
    a.gather = dx; a.out = dy;
    SL_TRY(sl_launch_rows(a, order, SL_EPI_SPMV, s));
    if (where == SL_MEM_HOST) SL_HIP(hipMemcpyAsync(y, dy, m->n_rows * sizeof(double), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}

// Matrix::multiply_vector_add (matrix/mod.rs:441-465) over CSRStorage::multiply_vector_add (sparse.rs:192-203): y += A x with the
// running sum of row i STARTING FROM y_i — (y_i + a_0 x_0) + a_1 x_1 ..., every product rounded before it is added.
sl_status sl_spmv_add(const sl_matrix *m, const double *x, double *y, sl_order order, sl_mem where)
{
    if (!m || !x || !y) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (order == SL_ORDER_SIMD4)
        return sl_fail(SL_INVALID_INPUT, "multiply_vector_add exists in the CSR order only (sparse.rs:192-203); simd_ops.rs has no accumulating form");
    if (x == y) return sl_fail(SL_INVALID_INPUT, "x and y must not alias");
    SL_ABI_BEGIN
    hipStream_t s = sl_context().stream;
    DevBuf xin, yio;
    const double *dx;
    SL_TRY(stage_in(x, m->n_cols, where, xin, &dx));
    double *dy = y;
    if (where == SL_MEM_HOST) {
        SL_TRY(yio.alloc((m->n_rows ? m->n_rows : 1) * sizeof(double)));
        dy = yio.as<double>();
        if (m->n_rows) SL_HIP(hipMemcpyAsync(dy, y, m->n_rows * sizeof(double), hipMemcpyHostToDevice, s));
    }
    sl_row_args a = row_args(m);
    a.gather = dx; a.out = dy;
    SL_TRY(sl_launch_rows_add(a, s));
    if (where == SL_MEM_HOST && m->n_rows) SL_HIP(hipMemcpyAsync(y, dy, m->n_rows * sizeof(double), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
    SL_ABI_END
}

static sl_status reduce_common(int mode, uint64_t n, const double *x, const double *y, double *out, sl_mem where)
{
    if (!x || !out || ((mode == 1 || mode == 4) && !y)) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    SL_TRY(require_device());
    hipStream_t s = sl_context().stream;
    DevBuf xin, yin;
    const double *dx, *dy = nullptr;
    SL_TRY(stage_in(x, n, where, xin, &dx));
    if (mode == 1 || mode == 4) SL_TRY(stage_in(y, n, where, yin, &dy));
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    double *res = scr + 4000;
    if (mode == 0) SL_TRY(sl_launch_sumsq(n, dx, scr, res, s));
    else if (mode == 1) SL_TRY(sl_launch_dot(n, dx, dy, scr, res, s));
    else if (mode == 3) SL_TRY(sl_launch_abs_max(n, dx, scr, res, s));
    else if (mode == 4) SL_TRY(sl_launch_diff_sumsq(n, dx, dy, scr, res, s));
    else SL_TRY(sl_launch_abs_sum(n, dx, scr, res, s));
    double h;
    SL_TRY(read_scalars(res, &h, 1));
    *out = (mode == 0) ? std::sqrt(h) : h;
    return SL_OK;
    SL_ABI_END
}
// solver::utils (solver/mod.rs:374-461): l1_norm, linf_norm, compute_norm, compute_residual, check_convergence
sl_status sl_l1_norm(uint64_t n, const double *x, double *out, sl_mem where) { return reduce_common(2, n, x, nullptr, out, where); }
sl_status sl_linf_norm(uint64_t n, const double *x, double *out, sl_mem where) { return reduce_common(3, n, x, nullptr, out, where); }
sl_status sl_compute_norm(uint64_t n, const double *x, sl_norm_type norm_type, double *out, sl_mem where)
{
    switch (norm_type) {
    case SL_NORM_L1: return sl_l1_norm(n, x, out, where);
    case SL_NORM_LINF: return sl_linf_norm(n, x, out, where);
    case SL_NORM_L2: case SL_NORM_WEIGHTED: return sl_l2_norm(n, x, out, where);      // "Default to L2 for weighted", solver/mod.rs:389
    }
    return sl_fail(SL_INVALID_INPUT, "unknown sl_norm_type %d", (int)norm_type);
}
sl_status sl_compute_residual(const sl_matrix *m, const double *x, const double *b, double *residual, sl_order order, sl_mem where)
{
    if (!m || !x || !b || !residual) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_ABI_BEGIN
    hipStream_t s = sl_context().stream;
    DevBuf xin, bin, rout;
    const double *dx, *db;
    SL_TRY(stage_in(x, m->n_cols, where, xin, &dx));
    SL_TRY(stage_in(b, m->n_rows, where, bin, &db));
    double *dr = residual;
    if (where == SL_MEM_HOST) { SL_TRY(rout.alloc((m->n_rows ? m->n_rows : 1) * sizeof(double))); dr = rout.as<double>(); }
    sl_row_args a = row_args(m);
    a.gather = dx; a.out = dr;
    SL_TRY(sl_launch_rows(a, order, SL_EPI_SPMV, s));                       // matrix.multiply_vector(x, residual)
    if (m->n_rows) SL_TRY(sl_launch_sub(m->n_rows, dr, db, dr, s));         // *r -= b_val
    if (where == SL_MEM_HOST && m->n_rows) SL_HIP(hipMemcpyAsync(residual, dr, m->n_rows * sizeof(double), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
    SL_ABI_END
}
sl_status sl_check_convergence(double residual_norm, double tolerance, sl_convergence_mode mode, double b_norm, uint64_t n,
                               const double *prev_solution, const double *current_solution, sl_mem where, int *converged)
{
    if (!converged) return sl_fail(SL_INVALID_INPUT, "null argument");
    *converged = 0;
    switch (mode) {
    case SL_CONV_RESIDUAL_NORM: *converged = residual_norm <= tolerance; return SL_OK;
    case SL_CONV_RELATIVE_RESIDUAL: *converged = b_norm > 0.0 ? (residual_norm / b_norm) <= tolerance : residual_norm <= tolerance; return SL_OK;
    case SL_CONV_COMBINED: *converged = residual_norm <= tolerance && (b_norm == 0.0 || (residual_norm / b_norm) <= tolerance); return SL_OK;
    case SL_CONV_SOLUTION_CHANGE: case SL_CONV_RELATIVE_SOLUTION_CHANGE: {
        if (!prev_solution) return SL_OK;                                    // None => false
        if (!current_solution) return sl_fail(SL_INVALID_INPUT, "null current_solution");
        double change = 0.0, prev = 0.0;
        SL_TRY(reduce_common(4, n, current_solution, prev_solution, &change, where));
        if (mode == SL_CONV_SOLUTION_CHANGE) { *converged = std::sqrt(change) <= tolerance; return SL_OK; }
        SL_TRY(reduce_common(0, n, prev_solution, nullptr, &prev, where));   // sqrt(sum prev^2)
        *converged = prev > 0.0 ? (std::sqrt(change) / prev) <= tolerance : std::sqrt(change) <= tolerance;
        return SL_OK;
    }
    }
    return sl_fail(SL_INVALID_INPUT, "unknown sl_convergence_mode %d", (int)mode);
}
sl_status sl_dot(uint64_t n, const double *x, const double *y, double *out, sl_mem where) { return reduce_common(1, n, x, y, out, where); }
sl_status sl_l2_norm(uint64_t n, const double *x, double *out, sl_mem where) { return reduce_common(0, n, x, nullptr, out, where); }

sl_status sl_axpy(uint64_t n, double alpha, const double *x, double *y, sl_mem where)
{
    if (!x || !y) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_TRY(require_device());
    hipStream_t s = sl_context().stream;
    DevBuf xin, yio;
    const double *dx;
    SL_TRY(stage_in(x, n, where, xin, &dx));
    double *dy = y;
    if (where == SL_MEM_HOST) {
        SL_TRY(yio.alloc(n * sizeof(double)));
        dy = yio.as<double>();
        SL_HIP(hipMemcpyAsync(dy, y, n * sizeof(double), hipMemcpyHostToDevice, s));
    }
    SL_TRY(sl_launch_axpy(n, alpha, dx, dy, s));
    if (where == SL_MEM_HOST) SL_HIP(hipMemcpyAsync(y, dy, n * sizeof(double), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}

// ---- fused Neumann step ------------------------------------------------------------------------------
sl_status sl_neumann_step(const sl_matrix *m, const double *dinv, const double *t_in, double *t_out,
                          double *x, double *norm2, sl_order order)
{
    if (!m || !dinv || !t_in || !t_out || !x) return sl_fail(SL_INVALID_INPUT, "null argument");
    double *scr = static_cast<double *>(sl_scratch(partial_bytes(m)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    sl_row_args a = row_args(m);
    a.gather = t_in; a.dinv = dinv; a.out = t_out; a.x = x; a.partials = scr; a.partials_slack = 4096; a.result = norm2;
    return sl_launch_rows(a, order, SL_EPI_NEUMANN, sl_context().stream);
}

// the same step for callers that cut one rank's rows into several matrices (boundary rows first, so that their halo
// can travel while the interior computes): every piece leaves its per-block partial sums in the caller's buffer, one
// fixed-order reduction over all of them closes the step — 3 + 1 launches instead of 3 x 2 plus glue
sl_status sl_matrix_partials_capacity(const sl_matrix *m, uint64_t *count)
{
    if (!m || !count) return sl_fail(SL_INVALID_INPUT, "null argument");
    *count = (uint64_t)sl_row_grid(m->n_slices) + m->n_long;
    return SL_OK;
}
sl_status sl_neumann_step_partials(const sl_matrix *m, const double *dinv, const double *t_in, double *t_out, double *x,
                                   double *partials, uint32_t *n_partials, sl_order order)
{
    if (!m || !dinv || !t_in || !t_out || !x || !partials || !n_partials) return sl_fail(SL_INVALID_INPUT, "null argument");
    sl_row_args a = row_args(m);
    a.gather = t_in; a.dinv = dinv; a.out = t_out; a.x = x; a.partials = partials; a.result = nullptr;
    return sl_launch_rows(a, order, SL_EPI_NEUMANN, sl_context().stream, n_partials);
}
sl_status sl_reduce_partials(const double *partials, uint32_t n, double *norm2)
{
    if (!partials || !norm2) return sl_fail(SL_INVALID_INPUT, "null argument");
    return sl_launch_final_reduce(partials, n, norm2, sl_context().stream);
}

sl_status sl_residual_norm2(const sl_matrix *m, const double *x_full, const double *rhs, double *r_out, double *norm2, sl_order order)
{
    if (!m || !x_full || !rhs || !norm2) return sl_fail(SL_INVALID_INPUT, "null argument");
    double *scr = static_cast<double *>(sl_scratch(partial_bytes(m)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    sl_row_args a = row_args(m);
    a.gather = x_full; a.aux = rhs; a.out = r_out; a.partials = scr; a.partials_slack = 4096; a.result = norm2;
    return sl_launch_rows(a, order, SL_EPI_RESIDUAL, sl_context().stream);
}

sl_status sl_neumann_run_steps(const sl_matrix *m, const double *dinv, double *t_a, double *t_b, double *x,
                               double *norm2, sl_order order, uint64_t steps, float *elapsed_ms)
{
    if (!m || !dinv || !t_a || !t_b || !x) return sl_fail(SL_INVALID_INPUT, "null argument");
    hipStream_t s = sl_context().stream;
    sl_timer timer;
    SL_TRY(timer.start(s));
    sl_status st = SL_OK;
    for (uint64_t k = 0; k < steps && st == SL_OK; ++k)
        st = sl_neumann_step(m, dinv, (k & 1) ? t_b : t_a, (k & 1) ? t_a : t_b, x, norm2, order);
    const float ms = timer.stop();
    if (elapsed_ms) *elapsed_ms = ms;
    return st;
}

// ---- NeumannSolver::solve ---------------------------------------------------------------------------
void sl_neumann_options_default(sl_neumann_options *o);
void sl_neumann_options_streaming(sl_neumann_options *o)      // SolverOptions::streaming, solver/mod.rs:101-116
{
    sl_neumann_options_default(o);
    o->tolerance = 1e-4; o->max_iterations = 1000; o->collect_stats = 1; o->compute_error_bounds = 0;
}
int sl_neumann_result_meets_quality_criteria(const sl_neumann_result *r, double tolerance)      // SolverResult::meets_quality_criteria, solver/mod.rs:192-195
{
    return r && r->converged && r->residual_norm <= tolerance;
}
void sl_neumann_options_default(sl_neumann_options *o)
{
    memset(o, 0, sizeof(*o));
    o->tolerance = 1e-6;          // solver/mod.rs:47-62
    o->max_iterations = 1000;
    o->max_terms = 50;            // neumann.rs:58-60
    o->series_tolerance = 1e-8;
    o->order = SL_ORDER_CSR_SEQUENTIAL;
    o->start = SL_START_ZERO;
    o->residual = SL_RESIDUAL_TRUE;
    o->mem = SL_MEM_HOST;
}

// ---- NeumannState (neumann.rs:95-249) as an object behind the ABI: initialize / update_rhs / run / extract_solution -----------
// sl_neumann_solve is create + run + solution on a state that lives for one call (NeumannSolver::solve, neumann.rs:469-555).
} // extern "C"

struct sl_neumann_state {
    const sl_matrix *m = nullptr;
    sl_neumann_options o{};
    uint64_t n = 0;
    int device = 0;
    DevBuf b, dinv, rhs, x, ta, tb, scal, ctlbuf;      // b: the state's own device copy of the right-hand side (update_rhs changes it)
    bool owned = false;                                // true: the vectors are allocations of their own (a state that outlives the call), not pool loans
    // partitioned state (sl_neumann_state_create_partitioned): n = this rank's rows; the term vectors are the full-length gathered
    // vectors of the partition (dist->t[0/1]); x / rhs / dinv / b stay local
    sl_dist *dist = nullptr;
    double *tpair[2] = {nullptr, nullptr};             // the two term buffers t_cur / t_nxt alternate between
    // boundary-first step of a partition (dist_step): the step launch's blocks [0, ov_edge) and [ov_tail, ov_blocks) — the rows the
    // peers pull — run on the side stream with the "halo ready" ticket and the pulls behind them, the interior blocks beside them on
    // the main stream.  ov_edge = 0: off (one launch, then ticket and pulls)
    uint32_t ov_edge = 0, ov_tail = 0, ov_blocks = 0;
    bool ov_rounds = false;       // the paced layout with edge-first rounds: ov_edge = rounds of the first launch, one launch for the rest
    hipEvent_t ev_main = nullptr, ev_side = nullptr;
    ~sl_neumann_state()
    {
        if (ev_main) (void)hipEventDestroy(ev_main);
        if (ev_side) (void)hipEventDestroy(ev_side);
        sl_dist_destroy(dist);
    }
    double *t_cur = nullptr, *t_nxt = nullptr;
    double resn = INFINITY, tn = 0.0;
    bool series_conv = false;
    uint64_t terms = 0, matvec = 0, step_launches = 0, resid_launches = 0;
};

namespace {

bool state_converged(const sl_neumann_state &st)      // is_converged, neumann.rs:422-430
{
    return (st.resn <= st.o.tolerance) || (st.series_conv && !(st.terms >= st.o.max_terms));
}

// NeumannState::new, neumann.rs:139-249 — order of checks preserved
sl_status state_init(sl_neumann_state &st, const sl_matrix *m, const double *b, const double *initial_guess, const sl_neumann_options *o)
{
    if (m->row_offset == 0 && m->n_rows != m->n_cols)
        return sl_fail(SL_INVALID_INPUT, "Matrix must be square for Neumann series");
    if (m->row_offset != 0 || m->n_rows != m->n_cols)
        return sl_fail(SL_UNSUPPORTED_FORMAT, "sl_neumann_solve needs the whole matrix; drive row slices with sl_neumann_step");
    st.m = m; st.o = *o; st.n = m->n_rows;
    (void)hipGetDevice(&st.device);
    const uint64_t n = st.n;
    const sl_mem where = (sl_mem)o->mem;
    hipStream_t s = sl_context().stream;
    auto get = [&](DevBuf &d, size_t bytes) { return st.owned ? d.alloc_owned(bytes) : d.alloc(bytes); };
    SL_TRY(get(st.b, n * 8)); SL_TRY(get(st.dinv, n * 8)); SL_TRY(get(st.rhs, n * 8)); SL_TRY(get(st.x, n * 8));
    SL_TRY(get(st.ta, n * 8)); SL_TRY(get(st.tb, n * 8)); SL_TRY(get(st.scal, 64)); SL_TRY(get(st.ctlbuf, sizeof(sl_solve_ctl)));
    if (n) SL_HIP(hipMemcpyAsync(st.b.p, b, n * 8, where == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    unsigned long long hs[4];
    SL_TRY(sl_matrix_diag_pass(m, st.dinv.as<double>(), hs));
    if (hs[0] & 1ull) {
        double dv[2];
        sl_matrix_row_dominance(m, hs[1], dv);
        return sl_fail(SL_NOT_DIAGONALLY_DOMINANT, "matrix is not row diagonally dominant (first failing row %llu: |a_ii| = %.17g, sum of |a_ij| = %.17g)", hs[1], dv[0], dv[1]);
    }
    if (hs[0] & 2ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (hs[0] & 4ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    SL_TRY(sl_launch_scale_rows(n, st.b.as<double>(), st.dinv.as<double>(), st.rhs.as<double>(), s));     // rhs = b * dinv  (:191-194)
    if (o->start == SL_START_INITIAL_GUESS) {
        if (!initial_guess) return sl_fail(SL_INVALID_INPUT, "initial_guess is null");
        SL_HIP(hipMemcpyAsync(st.x.p, initial_guess, n * 8, where == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    } else if (o->start == SL_START_REFERENCE_DEFAULT) {
        SL_HIP(hipMemcpyAsync(st.x.p, st.rhs.p, n * 8, hipMemcpyDeviceToDevice, s));     // :197-208
    } else {
        SL_HIP(hipMemsetAsync(st.x.p, 0, n * 8, s));
    }
    SL_HIP(hipMemcpyAsync(st.ta.p, st.rhs.p, n * 8, hipMemcpyDeviceToDevice, s));        // current_term = rhs (:211)
    st.tpair[0] = st.ta.as<double>(); st.tpair[1] = st.tb.as<double>();
    st.t_cur = st.tpair[0]; st.t_nxt = st.tpair[1];
    return SL_OK;
}

// Collective: does every rank have an interior to hide the exchange behind?  A rank's edge = the blocks holding the rows within the
// largest reach of any rank from either end of its range (what a neighbour pulls); the step kernel must be one that launches in
// ranges (band / general kernel, no hub rows), and the edges must leave at least one interior block.  SL_DIST_OVERLAP=0 turns the
// boundary-first step off, =2 keeps it on at world size 1 (tests).
sl_status dist_plan_overlap(sl_neumann_state &st)
{
    static const int env = [] { const char *e = getenv("SL_DIST_OVERLAP"); return e && *e ? atoi(e) : 1; }();
    sl_dist *d = st.dist;
    uint32_t R = 0, NB = 0;
    SL_TRY(sl_rows_geometry(sl_matrix_row_args(st.m), (sl_order)st.o.order, SL_EPI_NEUMANN, &R, &NB));
    const uint64_t W = d->max_reach;
    uint64_t edge = 0, real_blocks = 0;
    // the paced layout with edge-first rounds (sl_matrix::pw_edge_rounds): what any neighbour pulls lies in the rows its first rounds cover
    const bool rounds_form = (sl_order)st.o.order == SL_ORDER_CSR_SEQUENTIAL && st.m->d_pw_idx && st.m->pw_edge_rounds && !st.m->n_long;
    bool mine = env != 0 && (R != 0 || rounds_form) && W > 0 && W < d->n_global && (d->c->world > 1 || env == 2) && sl_side_stream(sl_context());
    if (mine && rounds_form) {
        mine = W <= st.m->pw_edge_rows;
        edge = st.m->pw_edge_rounds;
    } else
    if (mine) {
        edge = (W + R - 1) / R + 1;                                     // + 1: the last block may be ragged
        real_blocks = (st.n + R - 1) / R;
        mine = 2 * edge + 1 <= real_blocks;
    }
    if (mine && (!st.ev_main || !st.ev_side))
        mine = hipEventCreateWithFlags(&st.ev_main, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&st.ev_side, hipEventDisableTiming) == hipSuccess;
    std::vector<uint64_t> all((size_t)d->c->world);
    const uint64_t flag = mine ? 1 : 0;
    SL_TRY(sl_comm_allgather_blob(d->c, &flag, sizeof(flag), all.data()));
    for (uint64_t f : all) if (!f) mine = false;
    if (mine && rounds_form) { st.ov_edge = (uint32_t)edge; st.ov_rounds = true; }
    else if (mine) { st.ov_edge = (uint32_t)edge; st.ov_tail = (uint32_t)(real_blocks - edge); st.ov_blocks = NB; }
    if (mine && rounds_form)
        sl_log(1, "partition: rank %d runs its edge rounds first, the exchange beside the interior (reach %llu, the first %u rounds of the paced layout cover %llu rows from either end)",
               d->c->rank, (unsigned long long)W, st.ov_edge, (unsigned long long)st.m->pw_edge_rows);
    else
    sl_log(1, "partition: rank %d %s (reach %llu, %u rows per block, %llu edge blocks of %llu)", d->c->rank,
           mine ? "runs its edge blocks first, the exchange beside the interior" : "exchanges after the whole step", (unsigned long long)W, R,
           (unsigned long long)(2 * edge), (unsigned long long)real_blocks);
    return SL_OK;
}

// One fused step t_nxt = (I - D^-1 A) t_cur, x += t_nxt on a partition, enqueued on `s`: this rank's share of ||t_nxt||^2 into
// d_loc, the sum over all ranks (rank order) into `result` / judged against the stop rule, the pieces of the new term this rank's
// columns reach pulled into `vec`.  `a` carries the step's vectors and — inside a speculative batch — the gate.
// Boundary-first form (ov_edge != 0), per step k:
//   side:  wait(main so far: the sum ticket of step k-1)  edge blocks  ->  ticket "halo ready" (all ranks)  ->  pulls
//   main:  interior blocks                                 wait(side)  ->  reduction  ->  sum ticket
// The side chain starts behind the previous sum ticket, so every rank gates the same launches off when a stop rule has fired (a
// "halo ready" ticket that one rank skipped and another waits for cannot happen); edge blocks write rows the peers pull only after
// the peers' previous pulls (their halo ticket k follows those pulls in stream order, and this rank's edge blocks of step k+1 follow
// its own wait on halo ticket k); interior and edge blocks write disjoint rows; the pulls write outside this rank's rows.
sl_status dist_step(sl_neumann_state &st, sl_row_args a, sl_dist_vector *vec, double *d_loc, double *result, sl_solve_ctl *ctl, uint32_t rel,
                    uint32_t slot, int mode, double thr, hipStream_t s)
{
    sl_dist *D = st.dist;
    const sl_order order = (sl_order)st.o.order;
    a.result = d_loc;
    a.ctl = ctl; a.gate_it = rel; a.ctl_slot = slot; a.ctl_mode = SL_JUDGE_LOCAL; a.ctl_threshold = thr;
    if (!st.ov_edge) {
        SL_TRY(sl_launch_rows(a, order, SL_EPI_NEUMANN, s));
        SL_TRY(sl_comm_launch_ticket(D->c, d_loc, result, ctl, rel, slot, mode, thr, s));
        return sl_dist_pull(D, vec, s);
    }
    sl_ctx &c = sl_context();
    if (!sl_side_stream(c)) return sl_fail(SL_DEVICE_ERROR, "no side stream for the boundary-first step");
    uint32_t nparts = 0;
    SL_HIP(hipEventRecord(st.ev_main, s));
    SL_HIP(hipStreamWaitEvent(c.side, st.ev_main, 0));
    sl_row_args e = a;
    e.blk_lo = 0; e.blk_cnt = st.ov_edge;
    SL_TRY(sl_launch_rows(e, order, SL_EPI_NEUMANN, c.side, &nparts));
    if (!st.ov_rounds) {
        e.blk_lo = st.ov_tail; e.blk_cnt = st.ov_blocks - st.ov_tail;   // to the end of the grid: the padding blocks write their zero partials
        SL_TRY(sl_launch_rows(e, order, SL_EPI_NEUMANN, c.side));
    }
    SL_TRY(sl_comm_launch_ticket(D->c, nullptr, nullptr, ctl, rel, 0, SL_JUDGE_LOCAL, 0.0, c.side, 1));
    SL_TRY(sl_dist_pull(D, vec, c.side));
    SL_HIP(hipEventRecord(st.ev_side, c.side));
    sl_row_args in = a;
    in.blk_lo = st.ov_edge; in.blk_cnt = st.ov_rounds ? 0xffffu : st.ov_tail - st.ov_edge;      // (rounds: to the last one)
    SL_TRY(sl_launch_rows(in, order, SL_EPI_NEUMANN, s));
    SL_HIP(hipStreamWaitEvent(s, st.ev_side, 0));
    SL_TRY(sl_launch_rows_reduce(a, SL_EPI_NEUMANN, nparts, s));
    return sl_comm_launch_ticket(D->c, d_loc, result, ctl, rel, slot, mode, thr, s);
}

// NeumannState::new on a row partition: `m` = this rank's rows [lo, hi) with global column ids (row_offset = lo, n_cols = n_global)
sl_status state_init_partitioned(sl_neumann_state &st, sl_comm *c, const sl_matrix *m, const double *b, const double *initial_guess,
                                 const sl_neumann_options *o)
{
    st.m = m; st.o = *o; st.n = m->n_rows; st.owned = true;
    (void)hipGetDevice(&st.device);
    const uint64_t n = st.n;
    const sl_mem where = (sl_mem)o->mem;
    hipStream_t s = sl_context().stream;
    SL_TRY(sl_dist_create(c, m, &st.dist));                             // collective: row ranges, reach, pull plan
    sl_dist *d = st.dist;
    // each of these is collective and ends in an agreement: a rank whose allocation / IPC export / import failed still takes part in
    // every exchange, and all ranks leave with the same verdict (the communicator's counters stay in step)
    SL_TRY(sl_dist_vector_create(c, d->n_global, &d->t[0]));
    SL_TRY(sl_dist_vector_create(c, d->n_global, &d->t[1]));
    SL_TRY(sl_dist_vector_create(c, d->n_global, &d->x));
    sl_status mine = SL_OK;
    auto get = [&](DevBuf &dv, size_t bytes) { return dv.alloc_owned(bytes); };
    do {
        if ((mine = get(st.b, n * 8)) != SL_OK || (mine = get(st.dinv, n * 8)) != SL_OK || (mine = get(st.rhs, n * 8)) != SL_OK
            || (mine = get(st.x, n * 8)) != SL_OK || (mine = get(st.scal, 64)) != SL_OK || (mine = get(st.ctlbuf, sizeof(sl_solve_ctl))) != SL_OK) break;
        if (n && hipMemcpyAsync(st.b.p, b, n * 8, where == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s) != hipSuccess) { mine = sl_fail(SL_DEVICE_ERROR, "upload of b failed"); break; }
        unsigned long long hs[4];
        if ((mine = sl_matrix_diag_pass(m, st.dinv.as<double>(), hs)) != SL_OK) break;
        if (hs[0] & 1ull) {
            double dv[2];
            sl_matrix_row_dominance(m, hs[1], dv);
            mine = sl_fail(SL_NOT_DIAGONALLY_DOMINANT, "matrix is not row diagonally dominant (first failing row %llu: |a_ii| = %.17g, sum of |a_ij| = %.17g)", hs[1] + d->lo, dv[0], dv[1]);
            break;
        }
        if (hs[0] & 2ull) { mine = sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2] + d->lo); break; }
        if (hs[0] & 4ull) { mine = sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3] + d->lo); break; }
        if ((mine = sl_launch_scale_rows(n, st.b.as<double>(), st.dinv.as<double>(), st.rhs.as<double>(), s)) != SL_OK) break;
        if (o->start == SL_START_INITIAL_GUESS) {
            if (!initial_guess) { mine = sl_fail(SL_INVALID_INPUT, "initial_guess is null"); break; }
            if (hipMemcpyAsync(st.x.p, initial_guess, n * 8, where == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s) != hipSuccess) { mine = sl_fail(SL_DEVICE_ERROR, "upload of the initial guess failed"); break; }
        } else if (o->start == SL_START_REFERENCE_DEFAULT) {
            if (hipMemcpyAsync(st.x.p, st.rhs.p, n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) { mine = sl_fail(SL_DEVICE_ERROR, "copy failed"); break; }
        } else if (hipMemsetAsync(st.x.p, 0, n * 8, s) != hipSuccess) { mine = sl_fail(SL_DEVICE_ERROR, "memset failed"); break; }
        if (hipMemcpyAsync(d->t[0].mine + d->lo, st.rhs.p, n * 8, hipMemcpyDeviceToDevice, s) != hipSuccess) { mine = sl_fail(SL_DEVICE_ERROR, "copy failed"); break; }   // current_term = rhs
    } while (0);
    SL_TRY(sl_comm_agree(c, mine));
    st.tpair[0] = d->t[0].mine; st.tpair[1] = d->t[1].mine;
    st.t_cur = st.tpair[0]; st.t_nxt = st.tpair[1];
    // the first term on every rank: "my rows are written" ticket, then the pieces this rank's columns reach.  A failure here is
    // local: the communicator is marked, so that the peers' waits end at once instead of running into their time limit
    mine = sl_comm_launch_ticket(c, nullptr, nullptr, nullptr, 0, 0, SL_JUDGE_NONE, 0.0, s);
    if (mine == SL_OK) mine = sl_dist_pull(d, &d->t[0], s);
    if (mine == SL_OK && hipStreamSynchronize(s) != hipSuccess) mine = sl_fail(SL_DEVICE_ERROR, "the first exchange failed on the device");
    if (mine != SL_OK) { sl_comm_poison(c); return mine; }
    if (sl_comm_failed(c)) return sl_fail(SL_DEVICE_ERROR, "a rank of the communicator did not arrive (first exchange)");
    return dist_plan_overlap(st);
}

// the loop of NeumannSolver::solve (neumann.rs:477-555) from the state's current position; iteration count starts at 0
sl_status state_run(sl_neumann_state &st, double *term_norms, sl_neumann_result *res)
{
    const sl_matrix *m = st.m;
    const sl_neumann_options *o = &st.o;
    const uint64_t n = st.n;
    const sl_order order = (sl_order)o->order;
    hipStream_t s = sl_context().stream;
    const auto wall0 = std::chrono::steady_clock::now();
    double *scr = static_cast<double *>(sl_scratch(partial_bytes(m)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    double *d_res = st.scal.as<double>();
    DevBuf &dinv = st.dinv, &rhs = st.rhs, &x = st.x;
    double *&t_cur = st.t_cur, *&t_nxt = st.t_nxt;
    const double *res_rhs = (o->residual == SL_RESIDUAL_REFERENCE_SCALED) ? rhs.as<double>() : st.b.as<double>();
    double &resn = st.resn, &tn = st.tn;
    bool &series_conv = st.series_conv;
    uint64_t &terms = st.terms, &matvec = st.matvec, &step_launches = st.step_launches, &resid_launches = st.resid_launches;
    uint64_t it = 0;
    sl_status status = SL_OK;
    sl_range_push("neumann solve loop");
    sl_dist *D = st.dist;                                    // partitioned: sums over all ranks, pulls after every new term / before every residual
    const uint64_t lo = D ? D->lo : 0;
    double *d_loc = d_res + 2;                               // this rank's share of a sum (partitioned)
    auto vec_of = [&](double *p) { return p == D->t[0].mine ? &D->t[0] : &D->t[1]; };
    // gather x for a residual: my rows into the gathered solution, "written" ticket, pull
    auto gather_x = [&](sl_solve_ctl *ctl, uint32_t rel) -> sl_status {
        SL_HIP(hipMemcpyAsync(D->x.mine + lo, x.p, n * 8, hipMemcpyDeviceToDevice, s));
        SL_TRY(sl_comm_launch_ticket(D->c, nullptr, nullptr, ctl, rel, 0, SL_JUDGE_LOCAL, 0.0, s));
        return sl_dist_pull(D, &D->x, s);
    };

    auto is_converged = [&]() { return state_converged(st); };
    auto update_residual = [&]() -> sl_status {                                         // neumann.rs:302-318
        sl_row_args a = row_args(m);
        a.gather = x.as<double>(); a.aux = res_rhs; a.out = nullptr; a.partials = scr; a.partials_slack = 4096; a.result = d_res;
        if (D) { SL_TRY(gather_x(nullptr, 0)); a.gather = D->x.mine; a.result = d_loc; }
        SL_TRY(sl_launch_rows(a, order, SL_EPI_RESIDUAL, s));
        if (D) SL_TRY(sl_comm_launch_ticket(D->c, d_loc, d_res, nullptr, 0, 0, SL_JUDGE_NONE, 0.0, s));
        double h;
        SL_TRY(read_scalars(d_res, &h, 1));
        resn = std::sqrt(h);
        ++matvec; ++resid_launches;
        return SL_OK;
    };

    // The loop is enqueued speculatively, `batch` iterations at a time (sl_solve_ctl, sl_internal.hpp): nothing is
    // read back inside a batch, the reducing launches apply the stop rules on the device and later launches gate
    // themselves off; the host then replays the reference's control flow over the logged sums.  Same decisions and
    // bits as an iteration-by-iteration loop, one host round trip per batch instead of one or two per iteration.
    static const int batch_env = [] { const char *e = getenv("SL_SOLVE_BATCH"); int v = e ? atoi(e) : 10; return v < 1 ? 1 : (v > 25 ? 25 : v); }();
    sl_solve_ctl *d_ctl = st.ctlbuf.as<sl_solve_ctl>();
    const double thr_series = sq_threshold_lt(o->series_tolerance), thr_tol = sq_threshold_le(o->tolerance);
    struct planned { int kind; double *t_after; };      // kind 0: term k = 0, 1: fused step, 2: residual
    std::vector<planned> plan;

    sl_timer timer;
    SL_TRY(timer.start(s));
    bool done = false;
    while (!done && !is_converged() && it < o->max_iterations) {
        // ---- enqueue iterations [it, it_end) as if no stop rule fired ----
        const uint64_t it_end = std::min<uint64_t>(it + (uint64_t)batch_env, o->max_iterations);
        plan.clear();
        status = sl_launch_ctl_reset(d_ctl, s);
        uint64_t p_terms = terms;
        double *p_cur = t_cur, *p_nxt = t_nxt;
        for (uint64_t pit = it; pit < it_end && status == SL_OK; ++pit) {
            const uint32_t rel = (uint32_t)(pit - it);
            if (p_terms < o->max_terms) {                                                // compute_next_term :252-277
                if (p_terms > 0) {
                    sl_row_args a = row_args(m);
                    a.gather = p_cur; a.dinv = dinv.as<double>(); a.out = p_nxt + lo; a.x = x.as<double>();
                    a.partials = scr; a.partials_slack = 4096; a.result = D ? d_loc : nullptr;
                    a.ctl = d_ctl; a.gate_it = rel; a.ctl_slot = (uint32_t)plan.size(); a.ctl_mode = SL_JUDGE_LT; a.ctl_threshold = thr_series;
                    if (D) status = dist_step(st, a, vec_of(p_nxt), d_loc, nullptr, d_ctl, rel, (uint32_t)plan.size(), SL_JUDGE_LT, thr_series, s);
                    else status = sl_launch_rows(a, order, SL_EPI_NEUMANN, s);
                    std::swap(p_cur, p_nxt);
                    plan.push_back({1, p_cur});
                } else {
                    status = sl_launch_axpy(n, 1.0, p_cur + lo, x.as<double>(), s);      // x += term (k = 0)
                    if (status == SL_OK)
                        status = sl_launch_sumsq_judged(n, p_cur + lo, scr, d_ctl, rel, (uint32_t)plan.size(), D ? SL_JUDGE_LOCAL : SL_JUDGE_LT, thr_series, s,
                                                        D ? d_loc : nullptr);
                    if (D && status == SL_OK) status = sl_comm_launch_ticket(D->c, d_loc, nullptr, d_ctl, rel, (uint32_t)plan.size(), SL_JUDGE_LT, thr_series, s);
                    plan.push_back({0, p_cur});
                }
                ++p_terms;
            }
            if (pit % 5 == 0 && status == SL_OK) {                                       // update_residual :302-318, :489-491
                sl_row_args a = row_args(m);
                a.gather = x.as<double>(); a.aux = res_rhs; a.out = nullptr; a.partials = scr; a.partials_slack = 4096; a.result = nullptr;
                a.ctl = d_ctl; a.gate_it = rel; a.ctl_slot = (uint32_t)plan.size(); a.ctl_mode = SL_JUDGE_LE_OR_NONFINITE; a.ctl_threshold = thr_tol;
                if (D) { status = gather_x(d_ctl, rel); a.gather = D->x.mine; a.result = d_loc; a.ctl_mode = SL_JUDGE_LOCAL; }
                if (status == SL_OK) status = sl_launch_rows(a, order, SL_EPI_RESIDUAL, s);
                if (D && status == SL_OK)
                    status = sl_comm_launch_ticket(D->c, d_loc, nullptr, d_ctl, rel, (uint32_t)plan.size(), SL_JUDGE_LE_OR_NONFINITE, thr_tol, s);
                plan.push_back({2, nullptr});
            }
        }
        if (status != SL_OK) break;
        sl_solve_ctl h_ctl;
        if (hipMemcpyAsync(&h_ctl, d_ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            status = sl_fail(SL_DEVICE_ERROR, "solve-loop readback failed");
            break;
        }
        // ---- replay neumann.rs:477-513 over the log ----
        size_t used = 0;
        auto next_logged = [&](double *h) -> bool {
            if (used >= h_ctl.n_done || used >= plan.size()) return false;
            *h = h_ctl.log[used++];
            return true;
        };
        bool in_sync = true;
        while (!is_converged() && it < it_end) {
            if (terms < o->max_terms) {
                double h;
                if (!(in_sync = next_logged(&h))) break;
                const planned &pl = plan[used - 1];
                if (pl.kind == 1) { t_cur = pl.t_after; ++matvec; ++step_launches; }
                tn = std::sqrt(h);
                if (term_norms) term_norms[terms] = tn;
                ++terms;
                if (tn < o->series_tolerance) series_conv = true;                        // :270-274
            }
            if (it % 5 == 0) {
                double h;
                if (!(in_sync = next_logged(&h))) break;
                resn = std::sqrt(h);
                ++matvec; ++resid_launches;
            }
            ++it;
            if (!std::isfinite(resn)) {                                                  // :501-507
                status = sl_fail(SL_NUMERICAL_INSTABILITY, "Non-finite residual norm at iteration %llu", (unsigned long long)it);
                done = true;
                break;
            }
            if (series_conv) { done = true; break; }                                     // :510-512
        }
        t_nxt = (t_cur == st.tpair[0]) ? st.tpair[1] : st.tpair[0];
        if (D && sl_comm_failed(D->c)) { status = sl_fail(SL_DEVICE_ERROR, "a rank of the communicator did not arrive within the time limit"); break; }
        if (!in_sync || used != h_ctl.n_done) {
            status = sl_fail(SL_DEVICE_ERROR, "speculative solve loop out of step with the device (%zu of %u reductions consumed)",
                             used, h_ctl.n_done);
            break;
        }
    }
    const float loop_ms = timer.stop();

    if (status == SL_OK) {
        status = update_residual();                                                      // :516
        if (status == SL_OK) {
            res->converged = is_converged() ? 1 : 0;
            if (!res->converged && it >= o->max_iterations)                              // :523-530
                status = sl_fail(SL_CONVERGENCE_FAILURE, "neumann: %llu iterations, residual %.6e > tolerance %.6e",
                                 (unsigned long long)it, resn, o->tolerance);
        }
    }
    // estimate_error_bounds, neumann.rs:321-347
    if (D && status == SL_OK && sl_comm_failed(D->c)) status = sl_fail(SL_DEVICE_ERROR, "a rank of the communicator did not arrive within the time limit");
    if (o->compute_error_bounds && series_conv && terms > 0 && status == SL_OK && !D) {      // (needs ||rhs|| over all ranks: not offered for partitions)
        double h;
        if (sl_launch_sumsq(n, rhs.as<double>(), scr, d_res, s) == SL_OK && read_scalars(d_res, &h, 1) == SL_OK) {
            const double rhs_norm = std::sqrt(h);
            double est = 0.0;                                                                // one term: the estimate stays 0.0 => Some(0.0), :327-332
            if (terms > 1) est = std::pow(tn / rhs_norm, 1.0 / (double)(terms - 1));         // f64::powf
            if (est < 1.0) res->error_bound = sl_powi(est, (int)terms) / (1.0 - est) * rhs_norm;   // :334-344 (a NaN estimate keeps None)
        }
    }
    res->iterations = it; res->terms_computed = terms; res->matvec_count = matvec;
    res->residual_norm = resn; res->last_term_norm = tn; res->series_converged = series_conv ? 1 : 0;
    res->device_time_ms = loop_ms;
    res->bytes_moved = (step_launches + resid_launches) * (12ull * m->nnz + 4ull * (n + 1)) + step_launches * 40ull * n
                       + resid_launches * 16ull * n;
    res->total_time_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    sl_range_pop();
    sl_log(1, "neumann: %llu iterations, %llu terms, residual %.3e, %s, %.3f ms on the device", (unsigned long long)it, (unsigned long long)terms, resn,
           res->converged ? "converged" : "not converged", (double)loop_ms);
    return status;
}

sl_status state_solution(const sl_neumann_state &st, double *x_out, sl_mem where)
{
    hipStream_t s = sl_context().stream;
    hipError_t ce = st.n ? hipMemcpyAsync(x_out, st.x.p, st.n * 8, where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s) : hipSuccess;
    if (ce == hipSuccess) ce = hipStreamSynchronize(s);
    if (ce != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "result download failed: %s", hipGetErrorString(ce));
    return SL_OK;
}

// NeumannSolver::update_rhs, neumann.rs:436-462: in list order, one after the other (the same index may come twice)
__global__ void sl_update_rhs_kernel(uint64_t count, const uint64_t *idx, const double *delta, const double *dinv, double *rhs, double *x, double *b)
{
    if (blockIdx.x || threadIdx.x) return;
    for (uint64_t k = 0; k < count; ++k) {
        const uint64_t i = idx[k];
        const double scaled = __dmul_rn(delta[k], dinv[i]);             // :448
        rhs[i] = __dadd_rn(rhs[i], scaled);                             // :449
        x[i] = __dadd_rn(x[i], scaled);                                 // :453
        b[i] = __dadd_rn(b[i], delta[k]);                               // the unscaled right-hand side the TRUE residual is measured against
    }
}

// current_term = rhs (neumann.rs:211, 372, 457) — on a partition: my rows, "written" ticket, the pieces my columns reach
sl_status state_restart_series(sl_neumann_state &st)
{
    hipStream_t s = sl_context().stream;
    st.t_cur = st.tpair[0]; st.t_nxt = st.tpair[1];
    const uint64_t lo = st.dist ? st.dist->lo : 0;
    SL_HIP(hipMemcpyAsync(st.t_cur + lo, st.rhs.p, st.n * 8, hipMemcpyDeviceToDevice, s));
    if (st.dist) {
        SL_TRY(sl_comm_launch_ticket(st.dist->c, nullptr, nullptr, nullptr, 0, 0, SL_JUDGE_NONE, 0.0, s));
        SL_TRY(sl_dist_pull(st.dist, &st.dist->t[0], s));
        SL_HIP(hipStreamSynchronize(s));
        if (sl_comm_failed(st.dist->c)) return sl_fail(SL_DEVICE_ERROR, "a rank of the communicator did not arrive within the time limit");
    }
    return SL_OK;
}

} // namespace

extern "C" {

sl_status sl_neumann_state_create_partitioned(sl_comm *comm, const sl_matrix *local, const double *b_local, const double *initial_guess_local,
                                              const sl_neumann_options *o, sl_neumann_state **out)
{
    SL_ABI_BEGIN
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!comm || !local || !b_local || !o) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_TRY(require_device());
    sl_range trace_range("partitioned neumann state");
    sl_neumann_state *st = new sl_neumann_state();
    const sl_status s0 = state_init_partitioned(*st, comm, local, b_local, initial_guess_local, o);
    if (s0 != SL_OK) { sl_neumann_state_destroy(st); return s0; }
    *out = st;
    return SL_OK;
    SL_ABI_END
}

// K fused steps t <- (I - D^-1 A) t, x += t, ||t||^2 (a8 + a9) from the state's current term, no stop rule — the measurement loop
// (bench.py); on a partition every step ends with the sum over all ranks and the pull of the new term's pieces.
sl_status sl_neumann_state_run_steps(sl_neumann_state *st, uint64_t steps, double *last_norm2, float *elapsed_ms)
{
    SL_ABI_BEGIN
    if (!st) return sl_fail(SL_INVALID_INPUT, "null argument");
    hipStream_t s = sl_context().stream;
    double *scr = static_cast<double *>(sl_scratch(partial_bytes(st->m)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    sl_dist *D = st->dist;
    const uint64_t lo = D ? D->lo : 0;
    double *d_res = st->scal.as<double>(), *d_loc = d_res + 2;
    sl_timer timer;
    SL_TRY(timer.start(s));
    for (uint64_t k = 0; k < steps; ++k) {
        sl_row_args a = sl_matrix_row_args(st->m);
        a.gather = st->t_cur; a.dinv = st->dinv.as<double>(); a.out = st->t_nxt + lo; a.x = st->x.as<double>();
        a.partials = scr; a.partials_slack = 4096; a.result = D ? d_loc : d_res;
        if (D) SL_TRY(dist_step(*st, a, st->t_nxt == D->t[0].mine ? &D->t[0] : &D->t[1], d_loc, d_res, nullptr, 0, 0, SL_JUDGE_NONE, 0.0, s));
        else SL_TRY(sl_launch_rows(a, (sl_order)st->o.order, SL_EPI_NEUMANN, s));
        std::swap(st->t_cur, st->t_nxt);
    }
    const float ms = timer.stop();
    if (elapsed_ms) *elapsed_ms = ms;
    if (last_norm2) SL_TRY(read_scalars(d_res, last_norm2, 1));
    if (D && sl_comm_failed(D->c)) return sl_fail(SL_DEVICE_ERROR, "a rank of the communicator did not arrive within the time limit");
    return SL_OK;
    SL_ABI_END
}

// Collective check of the partition's last exchange of the current term: every piece this rank holds of its peers' rows against the
// owner's own copy (position-weighted wrapping sums of the bit patterns).  *pieces_bad = 0 on every rank when the exchange moved
// what the owners wrote — the first thing to ask of a transport on hardware it has not seen.  One GPU / world 1: nothing to compare.
sl_status sl_neumann_state_verify_exchange(sl_neumann_state *st, uint64_t *pieces_bad)
{
    SL_ABI_BEGIN
    if (!st || !pieces_bad) return sl_fail(SL_INVALID_INPUT, "null argument");
    *pieces_bad = 0;
    if (!st->dist) return SL_OK;
    SL_HIP(hipStreamSynchronize(sl_context().stream));
    sl_dist *D = st->dist;
    return sl_dist_verify(D, st->t_cur == D->t[0].mine ? &D->t[0] : &D->t[1], pieces_bad);
    SL_ABI_END
}

sl_status sl_neumann_solve(const sl_matrix *m, const double *b, const double *initial_guess,
                           const sl_neumann_options *o, double *x_out, double *term_norms,
                           sl_neumann_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !o || !x_out || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    res->residual_norm = INFINITY;
    res->error_bound = -1.0;
    const auto wall0 = std::chrono::steady_clock::now();
    sl_neumann_state st;
    SL_TRY(state_init(st, m, b, initial_guess, o));
    sl_status status = state_run(st, term_norms, res);
    const sl_status cs = state_solution(st, x_out, (sl_mem)o->mem);       // on CONVERGENCE_FAILURE x_out is still filled
    if (status == SL_OK) status = cs;
    res->total_time_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return status;
    SL_ABI_END
}

sl_status sl_neumann_state_create(const sl_matrix *m, const double *b, const double *initial_guess, const sl_neumann_options *o,
                                  sl_neumann_state **out)
{
    SL_ABI_BEGIN
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!m || !b || !o) return sl_fail(SL_INVALID_INPUT, "null argument");
    SL_TRY(require_device());
    sl_neumann_state *st = new sl_neumann_state();
    st->owned = true;          // the state outlives the call (and may die on another thread): no loans from the per-thread workspace pool
    const sl_status s0 = state_init(*st, m, b, initial_guess, o);
    if (s0 != SL_OK) { sl_neumann_state_destroy(st); return s0; }
    *out = st;
    return SL_OK;
    SL_ABI_END
}

void sl_neumann_state_destroy(sl_neumann_state *st)
{
    if (!st) return;
    (void)hipStreamSynchronize(sl_context().stream);
    // collective: no peer still pulls from the vectors this frees.  Not on a communicator that has failed (nobody waits there any
    // more) or that the host already closed (sl_comm_destroy with states alive: the ranks are past their last collective)
    if (st->dist && !st->dist->c->closed && !sl_comm_failed(st->dist->c)) (void)sl_comm_host_barrier(st->dist->c);
    delete st;
}

sl_status sl_neumann_state_update_rhs(sl_neumann_state *st, uint64_t count, const uint64_t *indices, const double *deltas)
{
    SL_ABI_BEGIN
    if (!st || (count && (!indices || !deltas))) return sl_fail(SL_INVALID_INPUT, "null argument");
    hipStream_t s = sl_context().stream;
    // neumann.rs:438-445: the first index out of range ends the call with IndexOutOfBounds — the updates before it stay applied and
    // the series state is NOT reset (the reference returns from inside the loop)
    // partitioned state: a collective call with the SAME list of GLOBAL row indices on every rank; a rank applies the pairs of its rows
    const uint64_t dim = st->dist ? st->dist->n_global : st->n, lo = st->dist ? st->dist->lo : 0, hi = lo + st->n;
    uint64_t good = count;
    for (uint64_t k = 0; k < count; ++k) if (indices[k] >= dim) { good = k; break; }
    std::vector<uint64_t> li; std::vector<double> ld;
    for (uint64_t k = 0; k < good; ++k) if (indices[k] >= lo && indices[k] < hi) { li.push_back(indices[k] - lo); ld.push_back(deltas[k]); }
    if (!li.empty()) {
        const uint64_t cnt = li.size();
        DevBuf di, dd;
        SL_TRY(di.alloc(cnt * 8)); SL_TRY(dd.alloc(cnt * 8));
        SL_HIP(hipMemcpyAsync(di.p, li.data(), cnt * 8, hipMemcpyHostToDevice, s));
        SL_HIP(hipMemcpyAsync(dd.p, ld.data(), cnt * 8, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(sl_update_rhs_kernel, dim3(1), dim3(1), 0, s, cnt, di.as<uint64_t>(), dd.as<double>(), st->dinv.as<double>(),
                           st->rhs.as<double>(), st->x.as<double>(), st->b.as<double>());
        SL_HIP(hipGetLastError());
        SL_HIP(hipStreamSynchronize(s));                                   // the staging buffers go back to the pool
    }
    if (good < count)
        return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "Index %llu out of bounds (max %llu) in rhs_update", (unsigned long long)indices[good],
                       (unsigned long long)(dim ? dim - 1 : 0));
    // :456-459 reset series computation state
    SL_TRY(state_restart_series(*st));                                  // current_term = rhs
    st->terms = 0;
    st->series_conv = false;
    return SL_OK;
    SL_ABI_END
}

sl_status sl_neumann_state_run(sl_neumann_state *st, double *term_norms, sl_neumann_result *res)
{
    SL_ABI_BEGIN
    if (!st || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    res->residual_norm = INFINITY;
    res->error_bound = -1.0;
    return state_run(*st, term_norms, res);
    SL_ABI_END
}

sl_status sl_neumann_state_solution(const sl_neumann_state *st, double *x_out, sl_mem where)
{
    SL_ABI_BEGIN
    if (!st || !x_out) return sl_fail(SL_INVALID_INPUT, "null argument");
    return state_solution(*st, x_out, where);
    SL_ABI_END
}

// rows of current_term / solution (neumann.rs:104-107) without moving the whole vector
static sl_status state_rows(const sl_neumann_state *st, const double *src, uint64_t first, uint64_t count, double *out, sl_mem where)
{
    if (!st || (count && !out)) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (first > st->n || count > st->n - first)
        return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "rows [%llu, %llu) of a state of %llu rows", (unsigned long long)first, (unsigned long long)(first + count), (unsigned long long)st->n);
    if (!count) return SL_OK;
    hipStream_t s = sl_context().stream;
    SL_HIP(hipMemcpyAsync(out, src + first, count * 8, where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}
sl_status sl_neumann_state_current_term(const sl_neumann_state *st, uint64_t first_row, uint64_t count, double *t_out, sl_mem where)
{
    SL_ABI_BEGIN
    if (!st) return sl_fail(SL_INVALID_INPUT, "null argument");
    return state_rows(st, st->t_cur + (st->dist ? st->dist->lo : 0), first_row, count, t_out, where);
    SL_ABI_END
}
sl_status sl_neumann_state_solution_rows(const sl_neumann_state *st, uint64_t first_row, uint64_t count, double *x_out, sl_mem where)
{
    SL_ABI_BEGIN
    if (!st) return sl_fail(SL_INVALID_INPUT, "null argument");
    return state_rows(st, st->x.as<double>(), first_row, count, x_out, where);
    SL_ABI_END
}

sl_status sl_neumann_state_reset(sl_neumann_state *st)                      // SolverState::reset, neumann.rs:367-378
{
    SL_ABI_BEGIN
    if (!st) return sl_fail(SL_INVALID_INPUT, "null argument");
    hipStream_t s = sl_context().stream;
    SL_HIP(hipMemsetAsync(st->x.p, 0, st->n * 8, s));
    SL_TRY(state_restart_series(*st));
    st->resn = INFINITY; st->tn = 0.0; st->terms = 0; st->matvec = 0; st->step_launches = 0; st->resid_launches = 0; st->series_conv = false;
    return SL_OK;
    SL_ABI_END
}

} // extern "C"
