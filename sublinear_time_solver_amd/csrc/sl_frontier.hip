// sl_frontier.hip — frontier machinery of the synchronous thresholded push:
//   * ordered stream compaction by wavefront ballot + popcount + prefix scan (ascending frontier),
//   * sparse rounds: candidate-row expansion over columns, pull update of candidate rows only,
//   * the host round loop (sl_push_solve) switching between sparse rounds and the dense
//     row-slice kernel, and sl_estimate_entry (local push on A^T).
//
// Spec: ForwardPushSolver::push_node (solver/forward_push.rs:179-216) and TS solveForwardPush
// (src/core/solver.ts:437-522) — invariant r = b - A x; here every round pushes ALL rows above
// the threshold at once; each candidate row's update is a pull over its own CSR row in column
// order (product rounded, then added), so the result does not depend on scheduling: frontier
// sets and values are bit-identical to the sequential CPU restatement.
#include "sl_internal.hpp"
#include <chrono>
#include <cmath>
#include <cstring>
#include <vector>

sl_status sl_sort_keys_u32(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n, hipStream_t s);

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

namespace {
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    sl_status alloc(size_t bytes)
    {
        if (p) { hipFree(p); p = nullptr; }
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) return sl_fail(SL_ALLOCATION, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return SL_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
#define SL_TRY(expr) do { sl_status s_ = (expr); if (s_ != SL_OK) return s_; } while (0)

// operator view: CSR rows of the operator B, and CSR rows of B^T (= columns of B)
struct op_view {
    const uint32_t *ptr, *idx; const double *val;   // rows of B
    const uint32_t *tptr, *tidx;                    // columns of B (pattern)
};
} // namespace

// ---- ordered compaction ---------------------------------------------------------------------
// tile = 2048 consecutive indices per 256-thread block; wave w owns 512 of them, 8 passes of 64.
#define SL_CTILE 2048

__device__ __forceinline__ bool sl_pred(const double *delta, double theta, uint64_t i)
{
    return theta <= 0.0 ? true : (delta[i] != 0.0);
}

__global__ __launch_bounds__(256) void sl_compact_count_kernel(uint64_t n, const double *delta, double theta, uint32_t *block_count)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SL_CTILE + (uint64_t)wave * 512;
    uint32_t cnt = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        const bool f = i < n && sl_pred(delta, theta, i);
        cnt += (uint32_t)__popcll(__ballot(f));
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-block exclusive scan of block counts (nb <= a few million); total -> *total_out
__global__ __launch_bounds__(1024) void sl_scan_kernel(uint32_t nb, const uint32_t *in, uint32_t *out, uint32_t *total_out)
{
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (nb + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * chunk;
    const uint32_t hi = lo + chunk < nb ? lo + chunk : nb;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += in[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
        *total_out = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = in[k]; out[k] = run; run += v; }
}

__global__ __launch_bounds__(256) void sl_compact_write_kernel(uint64_t n, const double *delta, double theta,
                                                               const uint32_t *block_off, uint32_t *list)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SL_CTILE + (uint64_t)wave * 512;
    unsigned long long masks[8];
    uint32_t cnt = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        const bool f = i < n && sl_pred(delta, theta, i);
        masks[p] = __ballot(f);
        cnt += (uint32_t)__popcll(masks[p]);
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    uint32_t off = block_off[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        if ((masks[p] >> lane) & 1ull) list[off + (uint32_t)__popcll(masks[p] & lt)] = (uint32_t)i;
        off += (uint32_t)__popcll(masks[p]);
    }
}

// ---- dense select (round 0): delta_i = r_i*dinv_i if |.| >= theta else 0 --------------------
__global__ __launch_bounds__(256) void sl_select_kernel(uint64_t n, const double *r, const double *dinv, double theta,
                                                        double *delta)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double p = DMUL(r[i], dinv[i]);
        delta[i] = (fabs(p) >= theta) ? p : 0.0;
    }
}

// ---- sparse round ------------------------------------------------------------------------------
// (1) per frontier column j: x_j += delta_j, and every row i with B_ij != 0 becomes a candidate
//     (atomicExch de-duplicates; the candidate LIST is unordered, its use is order-free).
__global__ __launch_bounds__(256) void sl_expand_kernel(uint32_t nf, const uint32_t *frontier, op_view op, const double *delta,
                                                        double *x, uint32_t *cand_flag, uint32_t *cand, uint32_t *cand_count)
{
    // one WAVE per frontier column: hub columns (power-law in-degree, 10^4..10^5 entries) are walked by 64
    // lanes with coalesced index loads instead of serialising one thread for milliseconds
    const uint32_t t = blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    if (t >= nf) return;
    const uint32_t j = frontier[t];
    if (lane == 0) x[j] = DADD(x[j], delta[j]);
    const uint32_t k1 = op.tptr[j + 1];
    for (uint32_t k = op.tptr[j] + lane; k < k1; k += 64) {
        const uint32_t i = op.tidx[k];
        if (atomicExch(&cand_flag[i], 1u) == 0u) cand[atomicAdd(cand_count, 1u)] = i;
    }
}

__device__ __forceinline__ double sl_csr_row_dot(const op_view &op, uint32_t i, const double *v, int order)
{
    const uint32_t s = op.ptr[i], e = op.ptr[i + 1];
    const uint32_t len = e - s;
    if (order == SL_ORDER_SIMD4 && len >= 8u) {       // simd_ops.rs:41-77
        const uint32_t chunks = len >> 2;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        for (uint32_t q = 0; q < chunks; ++q) {
            const uint32_t k = s + 4 * q;
            l0 = DADD(l0, DMUL(op.val[k], v[op.idx[k]]));
            l1 = DADD(l1, DMUL(op.val[k + 1], v[op.idx[k + 1]]));
            l2 = DADD(l2, DMUL(op.val[k + 2], v[op.idx[k + 2]]));
            l3 = DADD(l3, DMUL(op.val[k + 3], v[op.idx[k + 3]]));
        }
        double y = DADD(DADD(DADD(l0, l1), l2), l3);
        for (uint32_t k = s + 4 * chunks; k < e; ++k) y = DADD(y, DMUL(op.val[k], v[op.idx[k]]));
        return y;
    }
    double acc = 0.0;                                   // sparse.rs:194-202
    for (uint32_t k = s; k < e; ++k) acc = DADD(acc, DMUL(op.val[k], v[op.idx[k]]));
    return acc;
}

// (2) pull update of candidate rows: r_i -= (B delta_old)_i ; next frontier from |r_i dinv_i| >= theta
__device__ __forceinline__ void sl_pull_finish(uint32_t i, double acc, const double *dinv, double theta, double *r, double *delta_new,
                                               uint32_t *next, uint32_t *next_count)
{
    const double rn = DSUB(r[i], acc);
    r[i] = rn;
    const double p = DMUL(rn, dinv[i]);
    if (fabs(p) >= theta) {
        delta_new[i] = p;
        next[atomicAdd(next_count, 1u)] = i;
    }
}

// thread per candidate row; rows with more than SL_LONG_ROW entries are deferred to sl_pull_long_kernel
// (cand is reused as the deferred list: slot indices [0, *long_count) are overwritten only after being read)
__global__ __launch_bounds__(256) void sl_pull_kernel(uint32_t nc, const uint32_t *cand, op_view op, const double *delta_old,
                                                      const double *dinv, double theta, int order, double *r,
                                                      double *delta_new, uint32_t *cand_flag, uint32_t *next, uint32_t *next_count,
                                                      uint32_t *long_list, uint32_t *long_count)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nc) return;
    const uint32_t i = cand[t];
    cand_flag[i] = 0u;
    if (op.ptr[i + 1] - op.ptr[i] > SL_LONG_ROW) { long_list[atomicAdd(long_count, 1u)] = i; return; }
    sl_pull_finish(i, sl_csr_row_dot(op, i, delta_old, order), dinv, theta, r, delta_new, next, next_count);
}

// one wave per deferred long row (persistent: waves stride over the list).  Lanes form the products of 64
// consecutive entries in parallel; only the NON-ZERO products are then added, in entry order — skipping an
// exactly-zero product cannot change the running sum (s + (+-0) == s, and s is never -0), so the value equals
// the sequential reference sum bit for bit while the cost follows the (small) number of frontier columns hit.
__global__ __launch_bounds__(256) void sl_pull_long_kernel(const uint32_t *long_list, const uint32_t *long_count, op_view op,
                                                           const double *delta_old, const double *dinv, double theta, int order,
                                                           double *r, double *delta_new, uint32_t *next, uint32_t *next_count)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    const uint32_t n_long = *long_count;
    for (uint32_t t = wave; t < n_long; t += nwaves) {
        const uint32_t i = long_list[t];
        const uint32_t s = op.ptr[i], e = op.ptr[i + 1], len = e - s;
        const uint32_t chunks4 = (order == SL_ORDER_SIMD4) ? ((len >> 2) << 2) : 0u;
        double sum = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        bool merged = false;
        for (uint32_t base = s; base < e; base += 64) {
            const uint32_t k = base + lane;
            const double p = k < e ? DMUL(op.val[k], delta_old[op.idx[k]]) : 0.0;
            unsigned long long mask = __ballot(p != 0.0);
            while (mask) {
                const int l = __builtin_ctzll(mask);
                mask &= mask - 1;
                const double pv = __shfl(p, l);
                const uint32_t q = base - s + (uint32_t)l;
                if (q < chunks4) {
                    const uint32_t ln = q & 3u;
                    if (ln == 0) l0 = DADD(l0, pv); else if (ln == 1) l1 = DADD(l1, pv); else if (ln == 2) l2 = DADD(l2, pv); else l3 = DADD(l3, pv);
                } else {
                    if (order == SL_ORDER_SIMD4 && !merged) { sum = DADD(DADD(DADD(l0, l1), l2), l3); merged = true; }
                    sum = DADD(sum, pv);
                }
            }
        }
        if (order == SL_ORDER_SIMD4 && !merged) sum = DADD(DADD(DADD(l0, l1), l2), l3);
        if (lane == 0) sl_pull_finish(i, sum, dinv, theta, r, delta_new, next, next_count);
    }
}

// (3) delta_old[j] = 0 for the frontier just consumed (keeps the buffer all-zero outside a frontier)
__global__ __launch_bounds__(256) void sl_clear_kernel(uint32_t nf, const uint32_t *frontier, double *delta)
{
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    if (t < nf) delta[frontier[t]] = 0.0;
}

// generic (any operator given as CSR) dense round, one thread per row — used by estimate_entry
// when the frontier of A^T grows past the dense switch (A^T has no row-slice layout).
__global__ __launch_bounds__(256) void sl_dense_csr_round_kernel(uint64_t n, op_view op, const double *delta_old, const double *dinv,
                                                                 double theta, int order, double *r, double *x, double *delta_new)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double dself = delta_old[i];
    if (dself != 0.0) x[i] = DADD(x[i], dself);
    const double acc = sl_csr_row_dot(op, (uint32_t)i, delta_old, order);
    const double rn = DSUB(r[i], acc);
    r[i] = rn;
    const double p = DMUL(rn, dinv[i]);
    delta_new[i] = (fabs(p) >= theta) ? p : 0.0;
}

namespace {

struct push_state {
    uint64_t n = 0;
    op_view op{};
    double *x = nullptr, *r = nullptr, *dinv = nullptr;
    double *delta[2] = {nullptr, nullptr};
    uint32_t *frontier[2] = {nullptr, nullptr}; // cur / next
    uint32_t *cand = nullptr, *cand_flag = nullptr;
    uint32_t *counters = nullptr;               // [0] cand_count, [1] next_count, [2] compaction total, [3] deferred long rows
    uint32_t *long_list = nullptr;
    uint32_t *block_count = nullptr, *block_off = nullptr;
    uint32_t nblocks = 0;
};

sl_status compact(push_state &ps, const double *delta, double theta, uint32_t *list, uint32_t *h_count, hipStream_t s)
{
    if (ps.n == 0) { *h_count = 0; return SL_OK; }
    hipLaunchKernelGGL(sl_compact_count_kernel, dim3(ps.nblocks), dim3(256), 0, s, ps.n, delta, theta, ps.block_count);
    hipLaunchKernelGGL(sl_scan_kernel, dim3(1), dim3(1024), 0, s, ps.nblocks, ps.block_count, ps.block_off, ps.counters + 2);
    hipLaunchKernelGGL(sl_compact_write_kernel, dim3(ps.nblocks), dim3(256), 0, s, ps.n, delta, theta, ps.block_off, list);
    SL_HIP(hipMemcpyAsync(h_count, ps.counters + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}

struct push_log {
    uint32_t *log = nullptr; uint64_t cap = 0, words = 0;
    std::vector<uint32_t> tmp;
    sl_status append(push_state &ps, uint32_t *d_list, uint32_t nf, bool sorted, hipStream_t s)
    {
        if (!log) return SL_OK;
        if (words < cap) log[words] = nf;
        ++words;
        if (nf == 0) return SL_OK;
        uint32_t *src = d_list;
        DevBuf sorted_buf;
        if (!sorted && nf > 1) {
            SL_TRY(sorted_buf.alloc((size_t)nf * 4));
            SL_TRY(sl_sort_keys_u32(d_list, sorted_buf.as<uint32_t>(), nf, s));
            src = sorted_buf.as<uint32_t>();
        }
        const uint64_t room = words < cap ? cap - words : 0;
        const uint64_t take = nf < room ? nf : room;
        if (take) {
            SL_HIP(hipMemcpyAsync(log + words, src, take * 4, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
        }
        words += nf;
        (void)ps;
        return SL_OK;
    }
};

struct round_stats { uint64_t rounds = 0, pushes = 0, rows_touched = 0, dense_rounds = 0; bool converged = false; };

// The round loop.  `m` != null enables the row-slice dense kernel (operator = A itself).
sl_status run_push(push_state &ps, const sl_matrix *m, double theta, uint64_t max_rounds, int order, double dense_switch,
                   push_log &plog, round_stats &rs, float *device_ms)
{
    hipStream_t s = sl_context().stream;
    const uint64_t n = ps.n;
    hipEvent_t e0, e1;
    SL_HIP(hipEventCreate(&e0));
    SL_HIP(hipEventCreate(&e1));
    SL_HIP(hipEventRecord(e0, s));

    int cur = 0;                 // delta[cur] holds the frontier values, delta[1-cur] is all zero
    // round 0 frontier from r (dense select), ascending list by compaction
    hipLaunchKernelGGL(sl_select_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, n, ps.r, ps.dinv, theta, ps.delta[cur]);
    SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s));
    uint32_t nf = 0;
    SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s));
    bool list_valid = true, list_sorted = true;
    double *scr = nullptr;
    if (m) { scr = static_cast<double *>(sl_scratch((((size_t)sl_row_grid(m->n_slices) + m->n_long) * 2 + 4096) * sizeof(double))); if (!scr) return sl_fail(SL_ALLOCATION, "scratch"); }
    DevBuf resbuf;
    SL_TRY(resbuf.alloc(64));

    sl_status st = SL_OK;
    while (rs.rounds < max_rounds) {
        if (list_valid) SL_TRY(plog.append(ps, ps.frontier[0], nf, list_sorted, s));
        if (nf == 0) { rs.converged = true; break; }
        const bool dense = (double)nf > dense_switch * (double)n;
        if (dense) {
            uint32_t nf_next = 0;
            if (m) {
                sl_row_args a = sl_matrix_row_args(m);
                a.gather = ps.delta[cur]; a.dinv = ps.dinv; a.out = ps.delta[1 - cur]; a.x = ps.x; a.r = ps.r; a.theta = theta;
                a.partials = scr; a.result = resbuf.as<double>();
                st = sl_launch_rows(a, (sl_order)order, SL_EPI_PUSH, s);
                if (st != SL_OK) break;
                double h[2];
                SL_HIP(hipMemcpyAsync(h, resbuf.p, 16, hipMemcpyDeviceToHost, s));
                SL_HIP(hipStreamSynchronize(s));
                nf_next = (uint32_t)h[1];
            } else {
                hipLaunchKernelGGL(sl_dense_csr_round_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, ps.op, ps.delta[cur],
                                   ps.dinv, theta, order, ps.r, ps.x, ps.delta[1 - cur]);
            }
            rs.rounds += 1; rs.pushes += nf; rs.rows_touched += n; rs.dense_rounds += 1;
            cur = 1 - cur;                       // delta[cur] now dense-valid; delta[1-cur] holds stale values
            const bool next_dense_likely = m && (double)nf_next > dense_switch * (double)n && !plog.log;
            if (next_dense_likely) { nf = nf_next; list_valid = false; continue; } // no list needed: the dense kernel overwrites everything
            SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s));
            SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s));
            list_valid = true; list_sorted = true;
        } else {
            if (!list_valid) { SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s)); list_valid = true; list_sorted = true;
                               SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s)); }
            SL_HIP(hipMemsetAsync(ps.counters, 0, 2 * sizeof(uint32_t), s));
            SL_HIP(hipMemsetAsync(ps.counters + 3, 0, sizeof(uint32_t), s));
            hipLaunchKernelGGL(sl_expand_kernel, dim3((nf + 3) / 4), dim3(256), 0, s, nf, ps.frontier[0], ps.op, ps.delta[cur], ps.x,
                               ps.cand_flag, ps.cand, ps.counters);
            uint32_t nc = 0;
            SL_HIP(hipMemcpyAsync(&nc, ps.counters, 4, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
            if (nc) {
                hipLaunchKernelGGL(sl_pull_kernel, dim3((nc + 255) / 256), dim3(256), 0, s, nc, ps.cand, ps.op, ps.delta[cur], ps.dinv,
                                   theta, order, ps.r, ps.delta[1 - cur], ps.cand_flag, ps.frontier[1], ps.counters + 1, ps.long_list, ps.counters + 3);
                hipLaunchKernelGGL(sl_pull_long_kernel, dim3(nc < 4096u ? (nc + 3) / 4 : 1024u), dim3(256), 0, s, ps.long_list, ps.counters + 3, ps.op,
                                   ps.delta[cur], ps.dinv, theta, order, ps.r, ps.delta[1 - cur], ps.frontier[1], ps.counters + 1);
            }
            hipLaunchKernelGGL(sl_clear_kernel, dim3((nf + 255) / 256), dim3(256), 0, s, nf, ps.frontier[0], ps.delta[cur]);
            uint32_t nn = 0;
            SL_HIP(hipMemcpyAsync(&nn, ps.counters + 1, 4, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
            rs.rounds += 1; rs.pushes += nf; rs.rows_touched += nc;
            std::swap(ps.frontier[0], ps.frontier[1]);
            cur = 1 - cur;
            nf = nn; list_valid = true; list_sorted = false;
        }
    }
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    if (device_ms) *device_ms = ms;
    return st;
}

sl_status alloc_state(push_state &ps, uint64_t n, DevBuf bufs[16])
{
    ps.n = n;
    ps.nblocks = (uint32_t)((n + SL_CTILE - 1) / SL_CTILE);
    if (ps.nblocks == 0) ps.nblocks = 1;
    size_t k = 0;
    SL_TRY(bufs[k].alloc(n * 8)); ps.x = bufs[k++].as<double>();
    SL_TRY(bufs[k].alloc(n * 8)); ps.r = bufs[k++].as<double>();
    SL_TRY(bufs[k].alloc(n * 8)); ps.dinv = bufs[k++].as<double>();
    SL_TRY(bufs[k].alloc(n * 8)); ps.delta[0] = bufs[k++].as<double>();
    SL_TRY(bufs[k].alloc(n * 8)); ps.delta[1] = bufs[k++].as<double>();
    SL_TRY(bufs[k].alloc(n * 4)); ps.frontier[0] = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc(n * 4)); ps.frontier[1] = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc(n * 4)); ps.cand = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc(n * 4)); ps.cand_flag = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc(n * 4)); ps.long_list = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc(64)); ps.counters = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc((size_t)ps.nblocks * 4)); ps.block_count = bufs[k++].as<uint32_t>();
    SL_TRY(bufs[k].alloc((size_t)ps.nblocks * 4)); ps.block_off = bufs[k++].as<uint32_t>();
    SL_HIP(hipMemsetAsync(ps.cand_flag, 0, n * 4, sl_context().stream));
    SL_HIP(hipMemsetAsync(ps.counters, 0, 64, sl_context().stream));
    return SL_OK;
}

} // namespace

extern "C" {

void sl_push_options_default(sl_push_options *o)
{
    memset(o, 0, sizeof(*o));
    o->theta = 1e-6;               // ForwardPushConfig.epsilon, forward_push.rs:40-49
    o->max_rounds = 1000000;       // ForwardPushConfig.max_pushes
    o->order = SL_ORDER_CSR_SEQUENTIAL;
    o->mem = SL_MEM_HOST;
    o->dense_switch = 1.0 / 16.0;
}

sl_status sl_push_solve(const sl_matrix *m, const double *b, const sl_push_options *o, double *x, double *r_out,
                        uint32_t *frontier_log, uint64_t frontier_cap, uint64_t *frontier_words, sl_push_result *res)
{
    if (!m || !b || !o || !x || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (frontier_words) *frontier_words = 0;
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (!m->d_tptr || !m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "push needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    const hipMemcpyKind in_kind = o->mem == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind out_kind = o->mem == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;

    push_state ps;
    DevBuf bufs[16], bbuf, ax;
    SL_TRY(alloc_state(ps, n, bufs));
    ps.op = op_view{m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_trow};
    SL_TRY(bbuf.alloc(n * 8));
    SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, in_kind, s));
    SL_HIP(hipMemcpyAsync(ps.x, x, n * 8, in_kind, s));
    unsigned long long hs[4];
    SL_TRY(sl_matrix_diag_pass(m, ps.dinv, hs));
    if (hs[0] & 2ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (hs[0] & 4ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    // r = b - A x0
    SL_TRY(ax.alloc(n * 8));
    {
        sl_row_args a = sl_matrix_row_args(m);
        a.gather = ps.x; a.out = ax.as<double>();
        SL_TRY(sl_launch_rows(a, (sl_order)o->order, SL_EPI_SPMV, s));
        SL_TRY(sl_launch_sub(n, bbuf.as<double>(), ax.as<double>(), ps.r, s));
    }
    push_log plog;
    plog.log = frontier_log; plog.cap = frontier_cap;
    round_stats rs;
    float ms = 0.f;
    sl_status st = run_push(ps, m, o->theta, o->max_rounds, o->order, o->dense_switch > 0 ? o->dense_switch : 1.0 / 16.0, plog, rs, &ms);
    if (st != SL_OK) return st;
    res->rounds = rs.rounds; res->pushes = rs.pushes; res->rows_touched = rs.rows_touched; res->dense_rounds = rs.dense_rounds;
    res->converged = rs.converged ? 1 : 0; res->device_time_ms = ms;
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch");
    SL_TRY(sl_launch_sumsq(n, ps.r, scr, scr + 4000, s));
    double h = 0.0;
    SL_HIP(hipMemcpyAsync(&h, scr + 4000, 8, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(x, ps.x, n * 8, out_kind, s));
    if (r_out) SL_HIP(hipMemcpyAsync(r_out, ps.r, n * 8, out_kind, s));
    SL_HIP(hipStreamSynchronize(s));
    res->residual_norm = std::sqrt(h);
    if (frontier_words) *frontier_words = plog.words;
    return SL_OK;
}

// shared body: `given_is_transpose` = the matrix holds A^T (its rows are the operator of the push)
static sl_status estimate_entry_impl(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double theta,
                                     uint64_t max_rounds, bool given_is_transpose, sl_estimate_result *res)
{
    if (!m || !b || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (row >= m->n_rows) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)row, (unsigned long long)m->n_rows);
    if (!m->d_tptr || !m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "estimate_entry needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    push_state ps;
    DevBuf bufs[16], bbuf;
    SL_TRY(alloc_state(ps, n, bufs));
    unsigned long long hs[4];
    if (given_is_transpose) {
        // operator B = the matrix itself (= A^T); its dense rounds can use the row-slice kernels
        ps.op = op_view{m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_trow};
        SL_TRY(sl_matrix_diag_pass(m, ps.dinv, hs));
    } else {
        // operator B = A^T: rows of B = columns of A (sorted transpose), columns of B = rows of A
        ps.op = op_view{m->d_tptr, m->d_trow, m->d_tval, m->d_row_ptr, m->d_col_idx};
        SL_TRY(sl_csr_diag_pass(n, m->d_tptr, m->d_trow, m->d_tval, ps.dinv, hs));
    }
    if (hs[0] & 2ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (hs[0] & 4ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    // y0 = 0, r = e_row
    SL_HIP(hipMemsetAsync(ps.x, 0, n * 8, s));
    SL_HIP(hipMemsetAsync(ps.r, 0, n * 8, s));
    const double one = 1.0;
    SL_HIP(hipMemcpyAsync(ps.r + row, &one, 8, hipMemcpyHostToDevice, s));
    push_log plog;
    round_stats rs;
    float ms = 0.f;
    SL_TRY(run_push(ps, given_is_transpose ? m : nullptr, theta, max_rounds, SL_ORDER_CSR_SEQUENTIAL, given_is_transpose ? 1.0 / 16.0 : 0.25, plog, rs, &ms));
    // estimate = y . b ; residual_l1 = ||r_y||_1
    const double *db = b;
    if (where == SL_MEM_HOST) { SL_TRY(bbuf.alloc(n * 8)); SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, hipMemcpyHostToDevice, s)); db = bbuf.as<double>(); }
    double *scr = static_cast<double *>(sl_scratch(8192 * sizeof(double)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch");
    double h[2];
    SL_TRY(sl_launch_dot(n, ps.x, db, scr, scr + 4000, s));
    SL_HIP(hipMemcpyAsync(&h[0], scr + 4000, 8, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    SL_TRY(sl_launch_abs_sum(n, ps.r, scr, scr + 4000, s));
    SL_HIP(hipMemcpyAsync(&h[1], scr + 4000, 8, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    res->estimate = h[0]; res->residual_l1 = h[1];
    res->rounds = rs.rounds; res->pushes = rs.pushes; res->rows_touched = rs.rows_touched;
    res->device_time_ms = ms; res->converged = rs.converged ? 1 : 0;
    return SL_OK;
}

sl_status sl_estimate_entry(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double theta,
                            uint64_t max_rounds, sl_estimate_result *res)
{
    return estimate_entry_impl(m, b, where, row, theta, max_rounds, false, res);
}

sl_status sl_estimate_entry_transposed(const sl_matrix *mt, const double *b, sl_mem where, uint64_t row, double theta,
                                       uint64_t max_rounds, sl_estimate_result *res)
{
    return estimate_entry_impl(mt, b, where, row, theta, max_rounds, true, res);
}

sl_status sl_matrix_transpose(const sl_matrix *m, uint32_t flags, sl_matrix **out)
{
    if (!m || !out) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (!m->d_tptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "sl_matrix_transpose needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    if (m->row_offset != 0) return sl_fail(SL_UNSUPPORTED_FORMAT, "cannot transpose a row slice");
    return sl_matrix_create_csr(m->n_cols, m->n_rows, m->nnz, m->d_tptr, m->d_trow, m->d_tval, SL_MEM_DEVICE, 0, flags, out);
}

} // extern "C"
