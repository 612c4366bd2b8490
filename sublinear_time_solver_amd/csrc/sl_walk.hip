// sl_walk.hip — Monte-Carlo branch of estimateEntry on the GPU.
// Spec: TS estimateEntry random-walk branch (src/core/solver.ts:585-601, 630-648) over performRandomWalk
// (:390-432) and createTransitionMatrix (:359-385): absorb[i] = 1/a_ii, T[i][j] = -a_ij/a_ii; a walk from
// `row` absorbs with probability |absorb[cur]| (value += b[cur] * absorb[cur]) or moves to the first j whose
// cumulative |T[cur][.]| reaches rand; at most 1000 steps; numSamples = max(100, ceil(1/eps^2)).
// The reference draws every walk from ONE LCG stream (serial by construction); here walk s owns the stream
// createSeededRandom(seed + s) (src/core/utils.ts:161-168) — one lane per walk, CSR rows instead of the
// dense n x n transition table.  Per-walk values are bit-identical to the CPU restatement of the same rule
// (tests/test_gpu_walk.py); mean / variance are tree reductions (1e-12 relative).
// (The estimator is mirrored as behaviour; SURVEY.md Appendix A10 explains why it is not unbiased in general.)
#include "sl_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

__device__ __forceinline__ double walk_lcg(uint64_t &state)
{
    state = (state * 1664525ull + 1013904223ull) & 0xffffffffull;
    return __dmul_rn((double)state, 2.3283064365386963e-10);          // / 2^32
}

__global__ __launch_bounds__(256) void sl_walk_kernel(uint64_t n_walks, uint32_t seed, uint32_t start_row, const uint32_t *row_ptr,
                                                      const uint32_t *col_idx, const double *val, const double *b, double *values)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_walks) return;
    uint64_t state = (uint32_t)(seed + (uint32_t)s);
    uint32_t cur = start_row;
    double value = 0.0;
    for (int step = 0; step < 1000; ++step) {
        const uint32_t k0 = row_ptr[cur], k1 = row_ptr[cur + 1];
        double d = 0.0;
        for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] == cur) d = val[k];
        const double absorb = 1.0 / d;
        if (walk_lcg(state) < fabs(absorb)) { value = __dadd_rn(value, __dmul_rn(b[cur], absorb)); break; }
        double sum = 0.0;
        for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] != cur) sum = __dadd_rn(sum, fabs(-val[k] / d));
        if (sum == 0.0) { value = __dadd_rn(value, __dmul_rn(b[cur], absorb)); break; }
        const double rnd = __dmul_rn(walk_lcg(state), sum);
        if (rnd <= 0.0) { cur = 0; continue; }
        double cum = 0.0;
        const uint32_t row = cur;
        for (uint32_t k = k0; k < k1; ++k) {
            if (col_idx[k] == row) continue;
            cum = __dadd_rn(cum, fabs(-val[k] / d));
            if (rnd <= cum) { cur = col_idx[k]; break; }
        }
    }
    values[s] = value;
}

// sum of (v - mean)^2 partials
__global__ __launch_bounds__(256) void sl_walk_var_kernel(uint64_t n, const double *v, double mean, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double q = v[i] - mean;
        acc += q * q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void sl_walk_sum_kernel(uint64_t n, const double *v, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

extern "C" sl_status sl_estimate_entry_random_walk(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double epsilon,
                                                   uint32_t seed, uint64_t num_samples, double *walk_values, sl_walk_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (row >= m->n_rows) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)row, (unsigned long long)m->n_rows);
    if (!(epsilon > 0.0) && num_samples == 0) return sl_fail(SL_INVALID_INPUT, "epsilon must be positive");
    if (!m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "random-walk estimation needs the raw CSR (create with SL_MATRIX_KEEP_CSR)");
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    unsigned long long hs[4];
    SL_HIP(hipGetLastError());
    {   // createTransitionMatrix rejects a zero diagonal anywhere (solver.ts:368-371)
        sl_status st = sl_matrix_diag_pass(m, nullptr, hs);
        if (st != SL_OK) return st;
        if (hs[0] & 6ull) return sl_fail(SL_NUMERICAL_INSTABILITY, "Zero diagonal at position %llu", (hs[0] & 2ull) ? hs[2] : hs[3]);
    }
    if (num_samples == 0) {
        const double ns = std::ceil(1.0 / (epsilon * epsilon));             // solver.ts:586
        num_samples = ns > 100.0 ? (uint64_t)ns : 100;
    }
    DevBuf bbuf, vbuf;
    const double *db = b;
    if (where == SL_MEM_HOST) {
        SL_TRY(bbuf.alloc(n * 8));
        SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, hipMemcpyHostToDevice, s));
        db = bbuf.as<double>();
    }
    SL_TRY(vbuf.alloc(num_samples * 8));
    double *d_vals = vbuf.as<double>();
    sl_timer timer;
    SL_TRY(timer.start(s));
    hipLaunchKernelGGL(sl_walk_kernel, dim3((uint32_t)((num_samples + 255) / 256)), dim3(256), 0, s, num_samples, seed, (uint32_t)row,
                       m->d_row_ptr, m->d_col_idx, m->d_values, db, d_vals);
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    sl_status st = SL_OK;
    double h_sum = 0.0, h_var = 0.0;
    if (!scr) st = sl_fail(SL_ALLOCATION, "scratch");
    if (st == SL_OK) {
        const uint32_t g = (uint32_t)std::min<uint64_t>((num_samples + 255) / 256, 2048);
        hipLaunchKernelGGL(sl_walk_sum_kernel, dim3(g), dim3(256), 0, s, num_samples, d_vals, scr);
        std::vector<double> part(g);
        hipMemcpyAsync(part.data(), scr, g * 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        for (uint32_t k = 0; k < g; ++k) h_sum += part[k];
        const double mean = h_sum / (double)num_samples;
        hipLaunchKernelGGL(sl_walk_var_kernel, dim3(g), dim3(256), 0, s, num_samples, d_vals, mean, scr);
        hipMemcpyAsync(part.data(), scr, g * 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        for (uint32_t k = 0; k < g; ++k) h_var += part[k];
        res->estimate = mean;
        res->variance = num_samples > 1 ? h_var / (double)(num_samples - 1) : 0.0;
        res->num_samples = num_samples;
    }
    const float ms = timer.stop();
    res->device_time_ms = ms;
    if (st == SL_OK && walk_values) {
        hipMemcpyAsync(walk_values, d_vals, num_samples * 8, where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s);
        hipStreamSynchronize(s);
    }
    hipError_t le = hipGetLastError();
    if (st == SL_OK && le != hipSuccess) st = sl_fail(SL_DEVICE_ERROR, "random-walk kernels failed: %s", hipGetErrorString(le));
    return st;
    SL_ABI_END
}
