// sl_southwell.hip — a14: the shipped TypeScript solveForwardPush (src/core/solver.ts:437-522) in ITS visiting order.
//
// Gauss-Southwell on r = b - A x: every step pushes the FIRST index of largest |r_i| (`>` scan from 0, solver.ts:456-462),
// stops when that maximum is below epsilon (:464-467), p = r_i / a_ii (a division, :475), x_i += p, r_i = 0 exactly (:476-477),
// r_j -= a_ji p for the other rows of COLUMN i (:480-485; rows whose entry is zero keep their bits, so the sweep runs over the
// stored column = row i of the transpose, in ascending row order).  `iterations` = pushes (:487).
//
// This is the |F| = 1 member of the push family (SURVEY §8 a-P): inherently sequential across pushes, so the device only
// parallelises INSIDE a push — the argmax over n (one block, fixed reduction tree with ties broken towards the smaller index:
// exactly the reference's "first maximum") and the column update.  It exists for order-exact parity with the reference (push
// sequence, iteration count, solution bits); the throughput path is the synchronous thresholded push (sl_push_solve).
// One launch runs up to a budget of pushes (bounded kernel time), the host relaunches until the stop rule fires.
#include "sl_internal.hpp"
#include <cmath>
#include <cstring>

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

struct sl_gs_ctl {
    unsigned long long pushes;   // pushes done so far
    uint32_t state;              // 0 running, 1 converged (max < epsilon), 2 zero diagonal, 3 nothing to push although max >= epsilon
    uint32_t node;               // state 2: the offending row
    double last_max;
};

#define SL_GS_THREADS 1024
__global__ __launch_bounds__(SL_GS_THREADS) void sl_gs_kernel(uint32_t n, const uint32_t *__restrict__ row_ptr, const uint32_t *__restrict__ col_idx,
                                                              const double *__restrict__ values, const uint32_t *__restrict__ tptr,
                                                              const uint32_t *__restrict__ trow, const double *__restrict__ tval, double *r, double *x,
                                                              double epsilon, unsigned long long budget, unsigned long long max_pushes,
                                                              uint32_t *log, unsigned long long log_cap, sl_gs_ctl *ctl)
{
    __shared__ double smax[SL_GS_THREADS / 64];
    __shared__ uint32_t sidx[SL_GS_THREADS / 64];
    __shared__ double bc_p;
    __shared__ uint32_t bc_node, bc_state;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    unsigned long long done = ctl->pushes;
    if (ctl->state != 0) return;
    for (unsigned long long step = 0; step < budget && done < max_pushes; ++step, ++done) {
        // first index of the largest |r_i|: a thread walks its indices upwards with `>`, ties between threads go to the smaller index
        double best = 0.0;
        uint32_t bi = 0xffffffffu;
        for (uint32_t i = tid; i < n; i += SL_GS_THREADS) {
            const double a = fabs(r[i]);
            if (a > best) { best = a; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ob = __shfl_xor(best, o);
            const uint32_t oi = __shfl_xor(bi, o);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { smax[wave] = best; sidx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < SL_GS_THREADS / 64; ++w)
                if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
            uint32_t st = 0;
            double p = 0.0;
            if (best < epsilon) st = 1;                                      // solver.ts:464-467
            else if (bi == 0xffffffffu) st = 3;                              // all |r_i| are 0 or NaN and epsilon <= 0: no node to push
            else {
                // MatrixOperations.getDiagonal = the stored (i, i) entry: same midpoint search as CSRStorage::get (sparse.rs:142-155)
                double d = 0.0;
                uint32_t lo = row_ptr[bi], hi = row_ptr[bi + 1];
                while (lo < hi) {
                    const uint32_t mid = lo + ((hi - lo) >> 1);
                    const uint32_t c = col_idx[mid];
                    if (c == bi) { d = values[mid]; break; }
                    if (c < bi) lo = mid + 1; else hi = mid;
                }
                if (fabs(d) < 1e-15) st = 2;                                 // :471-473
                else {
                    p = r[bi] / d;                                           // :475
                    x[bi] = DADD(x[bi], p);
                    r[bi] = 0.0;
                }
            }
            bc_p = p; bc_node = bi; bc_state = st;
            if (st) { ctl->state = st; ctl->node = bi; }
            ctl->last_max = best;
        }
        __syncthreads();
        if (bc_state) break;
        const uint32_t node = bc_node;
        const double p = bc_p;
        const uint32_t c0 = tptr[node], c1 = tptr[node + 1];
        // rows of the column ascend; the same row stored twice sits in neighbouring entries: then the updates of that row must
        // happen one after the other — such a column is walked by one thread
        int dup = 0;
        for (uint32_t k = c0 + 1 + tid; k < c1; k += SL_GS_THREADS) dup |= (trow[k] == trow[k - 1]) ? 1 : 0;
        dup = __syncthreads_or(dup);
        if (!dup) {
            for (uint32_t k = c0 + tid; k < c1; k += SL_GS_THREADS) {
                const uint32_t j = trow[k];
                if (j != node) r[j] = DSUB(r[j], DMUL(tval[k], p));             // :480-485
            }
        } else if (tid == 0) {
            for (uint32_t k = c0; k < c1; ++k) {
                const uint32_t j = trow[k];
                if (j != node) r[j] = DSUB(r[j], DMUL(tval[k], p));
            }
        }
        if (tid == 0 && log && done < log_cap) log[done] = node;
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) ctl->pushes = done;
}

extern "C" {

void sl_southwell_options_default(sl_southwell_options *o)
{
    memset(o, 0, sizeof(*o));
    o->epsilon = 1e-6;             // SolverConfig.epsilon
    o->max_iterations = 1000;      // SolverConfig.maxIterations
    o->mem = SL_MEM_HOST;
}

sl_status sl_forward_push_southwell(const sl_matrix *m, const double *b, const sl_southwell_options *o, double *x_out, double *r_out,
                                    uint32_t *push_log, uint64_t log_cap, sl_southwell_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !o || !x_out || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    res->residual_norm = INFINITY;
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (!m->d_tptr || !m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "the Gauss-Southwell push walks columns: create the matrix with SL_MATRIX_WITH_TRANSPOSE");
    sl_range trace_range("gauss-southwell push");
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    const hipMemcpyKind in_kind = o->mem == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind out_kind = o->mem == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    DevBuf r, x, ctlb, logb;
    SL_TRY(r.alloc((n ? n : 1) * 8)); SL_TRY(x.alloc((n ? n : 1) * 8)); SL_TRY(ctlb.alloc(sizeof(sl_gs_ctl)));
    const uint64_t log_dev = push_log ? std::min<uint64_t>(log_cap, o->max_iterations) : 0;
    if (log_dev) SL_TRY(logb.alloc(log_dev * 4));
    SL_HIP(hipMemsetAsync(x.p, 0, (n ? n : 1) * 8, s));                            // approximate = zeros (solver.ts:439)
    if (n) SL_HIP(hipMemcpyAsync(r.p, b, n * 8, in_kind, s));                      // residual = [...vector]
    SL_HIP(hipMemsetAsync(ctlb.p, 0, sizeof(sl_gs_ctl), s));
    sl_timer timer;
    SL_TRY(timer.start(s));
    sl_gs_ctl h;
    memset(&h, 0, sizeof(h));
    // a launch is bounded: ~8 us per push at n = 10^5 (one block scans r), so 4096 pushes stay well below a second
    const unsigned long long budget = n > (1u << 20) ? 256 : 4096;
    while (h.state == 0 && h.pushes < o->max_iterations) {
        hipLaunchKernelGGL(sl_gs_kernel, dim3(1), dim3(SL_GS_THREADS), 0, s, (uint32_t)n, m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_trow,
                           m->d_tval, r.as<double>(), x.as<double>(), o->epsilon, budget, (unsigned long long)o->max_iterations,
                           logb.as<uint32_t>(), (unsigned long long)log_dev, ctlb.as<sl_gs_ctl>());
        SL_HIP(hipGetLastError());
        SL_HIP(hipMemcpyAsync(&h, ctlb.p, sizeof(h), hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
    }
    res->device_time_ms = timer.stop();
    res->iterations = h.pushes;
    res->converged = h.state == 1 ? 1 : 0;
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch");
    double hsum = 0.0;
    if (n) {
        SL_TRY(sl_launch_sumsq(n, r.as<double>(), scr, scr + 4000, s));
        SL_HIP(hipMemcpyAsync(&hsum, scr + 4000, 8, hipMemcpyDeviceToHost, s));
        SL_HIP(hipMemcpyAsync(x_out, x.p, n * 8, out_kind, s));
        if (r_out) SL_HIP(hipMemcpyAsync(r_out, r.p, n * 8, out_kind, s));
    }
    if (log_dev && h.pushes) SL_HIP(hipMemcpyAsync(push_log, logb.p, std::min<uint64_t>(log_dev, h.pushes) * 4, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    res->residual_norm = h.pushes ? std::sqrt(hsum) : INFINITY;                        // state.residual = norm2(residual) after a push (:488), Infinity before the first (:446)
    if (h.state == 2) return sl_fail(SL_NUMERICAL_INSTABILITY, "Zero diagonal at position %u", h.node);
    if (h.state == 3) return sl_fail(SL_NUMERICAL_INSTABILITY, "no finite residual to push (epsilon = %g)", o->epsilon);
    if (!res->converged)                                                              // :509-515 (x_out / r_out / result are filled all the same)
        return sl_fail(SL_CONVERGENCE_FAILURE, "Forward push failed to converge after %llu iterations", (unsigned long long)o->max_iterations);
    return SL_OK;
    SL_ABI_END
}

} // extern "C"
