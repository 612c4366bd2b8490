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
