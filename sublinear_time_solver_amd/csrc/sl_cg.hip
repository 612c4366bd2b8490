// sl_cg.hip — conjugate gradient behind the same SpMV:
// OptimizedConjugateGradientSolver::solve (src/optimized_solver.rs:182-295) ==
// FastConjugateGradient::solve (src/fast_solver.rs:126-178).  x0 = 0, r = p = b; stop when
// rsold <= tol^2; break when |p.Ap| < 1e-16.  Per iteration on the device:
//   (1) Ap = A p              row kernel (band / general, same layouts as the Neumann step)
//   (2) pAp = p . Ap          fixed-tree reduction                       -> host: alpha
//   (3) x += alpha p ; r -= alpha Ap ; rsnew = r . r   one fused pass    -> host: beta
//   (4) p = r + beta p
// Element-wise arithmetic matches the reference loops (product rounded, then added); the two dot
// products are tree reductions, so alpha/beta agree with the sequential CPU sums to rounding and the
// solution to ~1e-13 relative (tests use 1e-10).
#include "sl_internal.hpp"
#include <chrono>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <algorithm>

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

__device__ __forceinline__ double cg_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// The loop runs without host round trips (same scheme as sl_neumann_solve, sl_solve_ctl in sl_internal.hpp): alpha, beta and
// r.r live in device memory, the one-block kernels that finish the two reductions compute them (IEEE division, the same
// bits as on the host), log the reduced sums and close the gate when a stop rule of optimized_solver.rs:221-263 fires;
// the host enqueues a batch of iterations and replays the control flow over the log.  Gate index: 2 * iteration for the
// first half of an iteration (A p, p.Ap), 2 * iteration + 1 for the second (updates, r.r, new direction).
struct sl_cg_scalars { double rsold, pap, alpha, beta; };

__device__ __forceinline__ bool cg_gated(const sl_solve_ctl *c, uint32_t g) { return g > c->stop_after; }

__global__ void sl_cg_set_rsold_kernel(sl_cg_scalars *sc, double rsold) { sc->rsold = rsold; }

// partial sums of p . ap
__global__ __launch_bounds__(256) void sl_cg_dot_kernel(const sl_solve_ctl *c, uint32_t g, uint64_t n, const double *__restrict__ p,
                                                        const double *__restrict__ ap, double *partials)
{
    __shared__ double red[4];
    if (cg_gated(c, g)) return;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc = DADD(acc, DMUL(p[i], ap[i]));
    acc = cg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

__device__ __forceinline__ double cg_block_total(const double *partials, uint32_t nparts, double *red)
{
    double acc = 0.0;
    for (uint32_t j = threadIdx.x; j < nparts; j += 1024) acc += partials[j];
    acc = cg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    double t = red[0];
    for (int w = 1; w < 16; ++w) t += red[w];
    return t;
}

// p.Ap -> alpha = rsold / pAp; breakdown rule |pAp| < 1e-16 (optimized_solver.rs:236-238)
__global__ __launch_bounds__(1024) void sl_cg_pap_kernel(sl_solve_ctl *c, uint32_t g, const double *partials, uint32_t nparts, sl_cg_scalars *sc)
{
    __shared__ double red[16];
    if (cg_gated(c, g)) return;
    const double pap = cg_block_total(partials, nparts, red);
    if (threadIdx.x != 0) return;
    sc->pap = pap;
    sc->alpha = sc->rsold / pap;
    c->log[c->n_done] = pap;
    c->n_done += 1;
    if (fabs(pap) < 1e-16 && g < c->stop_after) c->stop_after = g;           // the second half of this iteration does not run
}

// x += alpha p ; r -= alpha ap ; partial sum of r^2   (optimized_solver.rs:244-257)
__global__ __launch_bounds__(256) void sl_cg_update_kernel(const sl_solve_ctl *c, uint32_t g, uint64_t n, const sl_cg_scalars *sc,
                                                           const double *__restrict__ p, const double *__restrict__ ap, double *x, double *r,
                                                           double *partials)
{
    __shared__ double red[4];
    if (cg_gated(c, g)) return;
    const double alpha = sc->alpha;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        x[i] = DADD(x[i], DMUL(alpha, p[i]));
        const double rn = DSUB(r[i], DMUL(alpha, ap[i]));
        r[i] = rn;
        acc = DADD(acc, DMUL(rn, rn));
    }
    acc = cg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

// r.r -> beta = rsnew / rsold, rsold = rsnew; stop rules: non-finite (error, the direction update is skipped), rsnew <= tol^2
__global__ __launch_bounds__(1024) void sl_cg_rs_kernel(sl_solve_ctl *c, uint32_t g, const double *partials, uint32_t nparts, sl_cg_scalars *sc,
                                                        double tol_sq)
{
    __shared__ double red[16];
    if (cg_gated(c, g)) return;
    const double rsnew = cg_block_total(partials, nparts, red);
    if (threadIdx.x != 0) return;
    c->log[c->n_done] = rsnew;
    c->n_done += 1;
    if (rsnew != rsnew || fabs(rsnew) == INFINITY) { if (g - 1 < c->stop_after) c->stop_after = g - 1; return; }
    sc->beta = rsnew / sc->rsold;
    sc->rsold = rsnew;
    if (rsnew <= tol_sq && g < c->stop_after) c->stop_after = g;
}

// The fused form of the two vector passes (round 4, SL_CG_FUSED_DOT=1): p is read ONCE per iteration — the first pass keeps to the
// residual (r -= alpha Ap, r.r: 24 n bytes), the second does both updates that read the old direction (x += alpha p, then p = r + beta p:
// 40 n bytes) — 64 n instead of 72 n bytes, element-wise the same operations on the same values.  The second pass runs in every
// iteration whose r.r was finite (gate g), also the one the tolerance rule ends: x has its update as in the reference (:244-246).
__global__ __launch_bounds__(256) void sl_cg_residual_kernel(const sl_solve_ctl *c, uint32_t g, uint64_t n, const sl_cg_scalars *sc,
                                                             const double *__restrict__ ap, double *r, double *partials)
{
    __shared__ double red[4];
    if (cg_gated(c, g)) return;
    const double alpha = sc->alpha;
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double rn = DSUB(r[i], DMUL(alpha, ap[i]));
        r[i] = rn;
        acc = DADD(acc, DMUL(rn, rn));
    }
    acc = cg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void sl_cg_x_direction_kernel(const sl_solve_ctl *c, uint32_t g, uint64_t n, const sl_cg_scalars *sc,
                                                                const double *__restrict__ r, double *x, double *p)
{
    if (cg_gated(c, g)) return;
    const double alpha = sc->alpha, beta = sc->beta;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double pi = p[i];
        x[i] = DADD(x[i], DMUL(alpha, pi));
        p[i] = DADD(r[i], DMUL(beta, pi));
    }
}

// p = r + beta p   (optimized_solver.rs:261-263)
__global__ __launch_bounds__(256) void sl_cg_direction_kernel(const sl_solve_ctl *c, uint32_t g, uint64_t n, const sl_cg_scalars *sc,
                                                              const double *__restrict__ r, double *p)
{
    if (cg_gated(c, g)) return;
    const double beta = sc->beta;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        p[i] = DADD(r[i], DMUL(beta, p[i]));
}

namespace {
} // namespace

extern "C" {

void sl_cg_options_default(sl_cg_options *o)
{
    memset(o, 0, sizeof(*o));
    o->tolerance = 1e-6;          // OptimizedSolverConfig::default, optimized_solver.rs:119-127
    o->max_iterations = 1000;
    o->order = SL_ORDER_CSR_SEQUENTIAL;
    o->mem = SL_MEM_HOST;
}

sl_status sl_cg_solve(const sl_matrix *m, const double *b, const sl_cg_options *o, double *x_out, sl_cg_result *res)
{
    SL_ABI_BEGIN
    sl_range trace_range("cg solve");
    if (!m || !b || !o || !x_out || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    const auto wall0 = std::chrono::steady_clock::now();
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");   // :187-190
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    const hipMemcpyKind in_kind = o->mem == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind out_kind = o->mem == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    DevBuf x, r, p, ap, scal;
    SL_TRY(x.alloc(n * 8)); SL_TRY(r.alloc(n * 8)); SL_TRY(p.alloc(n * 8)); SL_TRY(ap.alloc(n * 8)); SL_TRY(scal.alloc(64));
    const size_t pbytes = (((size_t)sl_row_grid(m->n_slices) + m->n_long) * 2 + 8192) * sizeof(double);
    double *scr = static_cast<double *>(sl_scratch(pbytes));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch allocation failed");
    double *d_res = scal.as<double>();
    SL_HIP(hipMemsetAsync(x.p, 0, n * 8, s));
    SL_HIP(hipMemcpyAsync(r.p, b, n * 8, in_kind, s));                       // r = b  (:207)
    SL_HIP(hipMemcpyAsync(p.p, r.p, n * 8, hipMemcpyDeviceToDevice, s));     // p = r  (:218)
    auto read = [&](double *h) -> sl_status {
        SL_HIP(hipMemcpyAsync(h, d_res, 8, hipMemcpyDeviceToHost, s));
        SL_HIP(hipStreamSynchronize(s));
        return SL_OK;
    };
    const uint32_t vgrid = (uint32_t)std::min<uint64_t>((n + 255) / 256 ? (n + 255) / 256 : 1, 2048);
    double rsold = 0.0;
    SL_TRY(sl_launch_sumsq(n, r.as<double>(), scr, d_res, s));
    SL_TRY(read(&rsold));
    const double tol_sq = o->tolerance * o->tolerance;
    uint64_t it = 0, mv = 0;
    bool converged = false;
    DevBuf ctlbuf, scbuf;
    SL_TRY(ctlbuf.alloc(sizeof(sl_solve_ctl)));
    SL_TRY(scbuf.alloc(sizeof(sl_cg_scalars)));
    sl_solve_ctl *d_ctl = ctlbuf.as<sl_solve_ctl>();
    sl_cg_scalars *d_sc = scbuf.as<sl_cg_scalars>();
    hipLaunchKernelGGL(sl_cg_set_rsold_kernel, dim3(1), dim3(1), 0, s, d_sc, rsold);
    static const bool fused_dot = [] { const char *e = getenv("SL_CG_FUSED_DOT"); return e && *e == '1'; }();      // TODO(after the GPU run of tests/test_gpu_cg.py with it): on by default
    static const int batch_env = [] { const char *e = getenv("SL_SOLVE_BATCH"); const int v = e ? atoi(e) : 10; return v < 1 ? 1 : (v > 25 ? 25 : v); }();
    sl_timer timer;
    SL_TRY(timer.start(s));
    sl_status st = SL_OK;
    bool done = false;
    while (!done && it < o->max_iterations) {
        if (rsold <= tol_sq) { converged = true; break; }                    // :221-224
        // ---- enqueue a batch of iterations as if no stop rule fired ----
        const uint64_t batch = std::min<uint64_t>((uint64_t)batch_env, o->max_iterations - it);
        st = sl_launch_ctl_reset(d_ctl, s);
        for (uint64_t k = 0; k < batch && st == SL_OK; ++k) {
            const uint32_t ga = (uint32_t)(2 * k), gb = ga + 1;
            sl_row_args a = sl_matrix_row_args(m);
            a.gather = p.as<double>(); a.out = ap.as<double>();
            a.ctl = d_ctl; a.gate_it = ga;
            if (fused_dot) {
                // ap = A p and p . Ap in one launch (:227-233): the residual epilogue in its "product + dot" form, gated like the rest;
                // its closing reduction leaves the sum in d_res (SL_JUDGE_LOCAL: no log entry, no stop rule — sl_cg_pap_kernel owns both)
                a.aux = p.as<double>(); a.aux_dot = 1u;
                a.partials = scr; a.partials_slack = 8192; a.result = d_res; a.ctl_mode = SL_JUDGE_LOCAL;
                st = sl_launch_rows(a, (sl_order)o->order, SL_EPI_RESIDUAL, s);
                if (st != SL_OK) break;
                hipLaunchKernelGGL(sl_cg_pap_kernel, dim3(1), dim3(1024), 0, s, d_ctl, ga, d_res, 1u, d_sc);
            } else {
                st = sl_launch_rows(a, (sl_order)o->order, SL_EPI_SPMV, s);      // ap = A p  (:227)
                if (st != SL_OK) break;
                hipLaunchKernelGGL(sl_cg_dot_kernel, dim3(vgrid), dim3(256), 0, s, d_ctl, ga, n, p.as<double>(), ap.as<double>(), scr);
                hipLaunchKernelGGL(sl_cg_pap_kernel, dim3(1), dim3(1024), 0, s, d_ctl, ga, scr, vgrid, d_sc);
            }
            if (fused_dot) {
                hipLaunchKernelGGL(sl_cg_residual_kernel, dim3(vgrid), dim3(256), 0, s, d_ctl, gb, n, d_sc, ap.as<double>(), r.as<double>(), scr);
                hipLaunchKernelGGL(sl_cg_rs_kernel, dim3(1), dim3(1024), 0, s, d_ctl, gb, scr, vgrid, d_sc, tol_sq);
                hipLaunchKernelGGL(sl_cg_x_direction_kernel, dim3(vgrid), dim3(256), 0, s, d_ctl, gb, n, d_sc, r.as<double>(), x.as<double>(), p.as<double>());
            } else {
                hipLaunchKernelGGL(sl_cg_update_kernel, dim3(vgrid), dim3(256), 0, s, d_ctl, gb, n, d_sc, p.as<double>(), ap.as<double>(), x.as<double>(),
                                   r.as<double>(), scr);
                hipLaunchKernelGGL(sl_cg_rs_kernel, dim3(1), dim3(1024), 0, s, d_ctl, gb, scr, vgrid, d_sc, tol_sq);
                hipLaunchKernelGGL(sl_cg_direction_kernel, dim3(vgrid), dim3(256), 0, s, d_ctl, gb, n, d_sc, r.as<double>(), p.as<double>());
            }
        }
        if (st != SL_OK) break;
        sl_solve_ctl h_ctl;
        if (hipMemcpyAsync(&h_ctl, d_ctl, sizeof(h_ctl), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
            st = sl_fail(SL_DEVICE_ERROR, "CG loop readback failed");
            break;
        }
        // ---- replay optimized_solver.rs:221-263 over the log ----
        uint32_t used = 0;
        for (uint64_t k = 0; k < batch; ++k) {
            if (rsold <= tol_sq) { converged = true; done = true; break; }
            if (used >= h_ctl.n_done) { st = sl_fail(SL_DEVICE_ERROR, "CG loop out of step with the device"); done = true; break; }
            const double pap = h_ctl.log[used++];
            ++mv;
            if (std::fabs(pap) < 1e-16) { done = true; break; }               // :236-238
            if (used >= h_ctl.n_done) { st = sl_fail(SL_DEVICE_ERROR, "CG loop out of step with the device"); done = true; break; }
            const double rsnew = h_ctl.log[used++];
            if (!std::isfinite(rsnew)) { st = sl_fail(SL_NUMERICAL_INSTABILITY, "Non-finite residual in CG at iteration %llu", (unsigned long long)it); done = true; break; }
            rsold = rsnew;
            ++it;
        }
        if (st == SL_OK && used != h_ctl.n_done) { st = sl_fail(SL_DEVICE_ERROR, "CG loop out of step with the device (%u of %u sums consumed)", used, h_ctl.n_done); break; }
    }
    const float ms = timer.stop();
    res->iterations = it; res->matvec_count = mv; res->residual_norm = std::sqrt(rsold); res->converged = converged ? 1 : 0;
    res->device_time_ms = ms;
    hipError_t ce = hipMemcpyAsync(x_out, x.p, n * 8, out_kind, s);
    if (ce == hipSuccess) ce = hipStreamSynchronize(s);
    if (ce != hipSuccess && st == SL_OK) st = sl_fail(SL_DEVICE_ERROR, "result download failed: %s", hipGetErrorString(ce));
    res->total_time_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return st;
    SL_ABI_END
}

} // extern "C"
