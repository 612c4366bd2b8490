// sl_synth.hip — S-DD(n, k, seed, w): the seeded synthetic diagonally dominant system of
// DESIGN.md §6, generated directly in HBM (bench / tests).  Bit-identical to the numpy
// generator sublinear_time_solver_amd/generators.py::sdd_rows (tests assert it).
//
// Recipe after the reference's own benchmark generators — LCG columns, diagonal 10 + 0.01 i,
// b = 1 + 0.001 i (src/ultra_fast.rs:221-248, benches/performance_benchmarks.rs:12-43) — but
// counter-based (splitmix64 of (seed, row, slot)) so that any row range can be produced
// independently on any rank, duplicate-free and self-excluding so nnz/row is exact:
//   row i: diagonal d_i = 10 + 0.01 (i mod 1000); k-1 off-diagonals, one per stratum of the
//   column window [lo, lo+span) (whole matrix when w = 0, else the band i-w .. i+w clipped at the edges),
//   value U(-1,1) * d_i / (2 (k-1))  =>  sum |offdiag| <= d_i / 2  (strictly row dominant,
//   asymmetric);  columns ascending;  b_i = 1 + 0.001 (i mod 1000).
#include "sl_internal.hpp"

__host__ __device__ static inline uint64_t sl_mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

__global__ __launch_bounds__(256) void sl_synth_sdd_kernel(uint64_t n, uint32_t k, uint64_t seed, uint64_t w,
                                                           uint64_t row_lo, uint64_t row_hi, uint32_t *row_ptr,
                                                           uint32_t *col_idx, double *values, double *b)
{
    const uint64_t li = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t rows = row_hi - row_lo;
    if (li > rows) return;
    row_ptr[li] = (uint32_t)(li * k);
    if (li == rows) return;
    const uint64_t i = row_lo + li;
    const uint32_t m = k - 1;
    uint64_t lo = 0, span = n;
    if (w != 0 && 2 * w + 1 < n) {            // band: columns in [i-w, i+w], clipped at the matrix edge
        lo = i > w ? i - w : 0;
        const uint64_t hi = i + w + 1 < n ? i + w + 1 : n;
        span = hi - lo;
    }
    const uint64_t sw = span / m;
    const double d = __dadd_rn(10.0, __dmul_rn(0.01, (double)(i % 1000)));
    const double scale = d / (double)(2 * m);
    uint32_t *crow = col_idx + li * k;
    double *vrow = values + li * k;
    uint32_t pos = 0;
    bool diag_done = false;
    for (uint32_t j = 0; j < m; ++j) {
        const uint64_t key = seed * 0x9E3779B97F4A7C15ull + (i * 64 + j + 1) * 0xD1B54A32D192ED03ull;
        const uint64_t h1 = sl_mix64(key);
        const uint64_t h2 = sl_mix64(h1 + 0x9E3779B97F4A7C15ull);
        const uint64_t off = h1 % sw;
        uint64_t c = lo + (uint64_t)j * sw + off;
        if (c == i) c = (off + 1 < sw) ? c + 1 : c - 1;
        const double u = __dmul_rn((double)(h2 >> 11), 1.1102230246251565e-16); // 2^-53
        double v = __dmul_rn(__dsub_rn(__dmul_rn(2.0, u), 1.0), scale);
        if (v == 0.0) v = scale;
        if (!diag_done && c > i) { crow[pos] = (uint32_t)i; vrow[pos] = d; ++pos; diag_done = true; }
        crow[pos] = (uint32_t)c; vrow[pos] = v; ++pos;
    }
    if (!diag_done) { crow[pos] = (uint32_t)i; vrow[pos] = d; }
    b[li] = __dadd_rn(1.0, __dmul_rn(0.001, (double)(i % 1000)));
}

extern "C" sl_status sl_synth_sdd_device(uint64_t n, uint32_t k, uint64_t seed, uint64_t half_bandwidth,
                                         uint64_t row_lo, uint64_t row_hi, uint32_t *row_ptr, uint32_t *col_idx,
                                         double *values, double *b)
{
    if (k < 2 || k > 64) return sl_fail(SL_INVALID_INPUT, "k must be in [2, 64]");
    if (row_hi > n || row_lo > row_hi) return sl_fail(SL_INVALID_INPUT, "bad row range");
    uint64_t span = n;
    if (half_bandwidth != 0 && 2 * half_bandwidth + 1 < n) span = half_bandwidth + 1;   // narrowest (edge) window
    if (span / (k - 1) < 2) return sl_fail(SL_INVALID_INPUT, "column window too narrow for k-1 distinct off-diagonals");
    if ((row_hi - row_lo) * (uint64_t)k > 0xffffffffull) return sl_fail(SL_INVALID_INPUT, "slice nnz exceeds u32");
    const uint64_t rows = row_hi - row_lo;
    hipLaunchKernelGGL(sl_synth_sdd_kernel, dim3((uint32_t)((rows + 1 + 255) / 256)), dim3(256), 0, sl_context().stream,
                       n, k, seed, half_bandwidth, row_lo, row_hi, row_ptr, col_idx, values, b);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

// ---- S-PR: power-law digraph -> M = I - alpha P (row i: diagonal 1, then -alpha/d_i per out-link) ----------
// integer-only degree law (reproducible everywhere): d = dmin << g, P(g >= j) = r^j with r = 2^-1.1
__device__ __forceinline__ uint32_t sl_pr_degree(uint64_t i, uint64_t seed, uint32_t dmin, uint32_t dmax, uint64_t n)
{
    const uint64_t h = sl_mix64(seed * 0x9E3779B97F4A7C15ull + (i + 1) * 0xD1B54A32D192ED03ull);
    // thresholds r^j * 2^64 for r = 0.46651649576840371 (2^-1.1)
    const uint64_t thr[14] = {8605710403603548160ull, 4014705861086822400ull, 1872926509855096064ull, 873751112209346048ull, 407619307041649408ull, 190161130728615296ull, 88713304338870928ull, 41386219868205984ull, 19307354266016144ull, 9007199254740991ull, 4202007033009544ull, 1960305596233800ull, 914514897390183ull, 426636285258469ull};
    uint32_t g = 0;
    while (g < 14 && h < thr[g]) ++g;
    uint64_t d = (uint64_t)dmin << g;
    if (d > dmax) d = dmax;
    if (d > n - 1) d = n - 1;
    return (uint32_t)d;
}
__global__ __launch_bounds__(256) void sl_pr_degree_kernel(uint64_t n, uint64_t seed, uint32_t dmin, uint32_t dmax, uint32_t *row_len)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) row_len[i + 1] = 1u + sl_pr_degree(i, seed, dmin, dmax, n);
    if (i == 0) row_len[0] = 0u;
}
__global__ __launch_bounds__(256) void sl_pr_fill_kernel(uint64_t n, uint64_t seed, double alpha, const uint32_t *row_ptr,
                                                         uint32_t *col_idx, double *values)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = row_ptr[i], d = row_ptr[i + 1] - s - 1u;
    col_idx[s] = (uint32_t)i; values[s] = 1.0;
    const double wgt = d ? -(alpha / (double)d) : 0.0;
    for (uint32_t k = 0; k < d; ++k) {
        const uint64_t h = sl_mix64(seed * 0x9E3779B97F4A7C15ull + (i * 16384ull + k + 7) * 0xBF58476D1CE4E5B9ull);
        const double u = __dmul_rn((double)(h >> 11), 1.1102230246251565e-16);
        uint64_t v = (uint64_t)__dmul_rn((double)n, __dmul_rn(u, u));
        if (v >= n) v = n - 1;
        if (v == i) v = (i + 1) % n;
        col_idx[s + 1 + k] = (uint32_t)v; values[s + 1 + k] = wgt;
    }
}
__global__ void sl_scan_u32_inplace_kernel(uint64_t n, uint32_t *a, unsigned long long *total)
{   // single thread block serial-chunk scan; one-off generator helper
    __shared__ unsigned long long part[1024];
    const uint64_t chunk = (n + 1023) / 1024, lo = threadIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    unsigned long long s = 0;
    for (uint64_t k = lo; k < hi; ++k) s += a[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long run = 0; for (int t = 0; t < 1024; ++t) { const unsigned long long v = part[t]; part[t] = run; run += v; } *total = run; }
    __syncthreads();
    unsigned long long run = part[threadIdx.x];
    for (uint64_t k = lo; k < hi; ++k) { run += a[k]; a[k] = (uint32_t)run; }
}

extern "C" sl_status sl_synth_pagerank_device(uint64_t n, uint64_t seed, double alpha, uint32_t dmin, uint32_t dmax,
                                              uint32_t *row_ptr, uint32_t *col_idx, double *values, uint64_t *nnz)
{
    if (n < 2 || dmin < 1 || dmax < dmin || dmax > 16000) return sl_fail(SL_INVALID_INPUT, "bad S-PR parameters");
    hipStream_t st = sl_context().stream;
    if (!col_idx) {
        hipLaunchKernelGGL(sl_pr_degree_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, seed, dmin, dmax, row_ptr);
        unsigned long long *d_total = nullptr;
        SL_HIP(sl_malloc(&d_total, 8));
        hipLaunchKernelGGL(sl_scan_u32_inplace_kernel, dim3(1), dim3(1024), 0, st, n + 1, row_ptr, d_total);
        unsigned long long h = 0;
        SL_HIP(hipMemcpyAsync(&h, d_total, 8, hipMemcpyDeviceToHost, st));
        SL_HIP(hipStreamSynchronize(st));
        hipFree(d_total);
        if (h > 0xffffffffull) return sl_fail(SL_INVALID_INPUT, "S-PR nnz exceeds u32");
        if (nnz) *nnz = h;
        return SL_OK;
    }
    hipLaunchKernelGGL(sl_pr_fill_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, seed, alpha, row_ptr, col_idx, values);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
