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

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

__device__ __forceinline__ double cg_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// x += alpha p ; r -= alpha ap ; partial sum of r^2   (optimized_solver.rs:244-257)
__global__ __launch_bounds__(256) void sl_cg_update_kernel(uint64_t n, double alpha, const double *__restrict__ p,
                                                           const double *__restrict__ ap, double *x, double *r, double *partials)
{
    __shared__ double red[4];
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

// p = r + beta p   (optimized_solver.rs:261-263)
__global__ __launch_bounds__(256) void sl_cg_direction_kernel(uint64_t n, double beta, const double *__restrict__ r, double *p)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256)
        p[i] = DADD(r[i], DMUL(beta, p[i]));
}

__global__ __launch_bounds__(1024) void sl_cg_final_reduce_kernel(const double *partials, uint32_t nparts, double *result)
{
    __shared__ double red[16];
    double acc = 0.0;
    for (uint32_t j = threadIdx.x; j < nparts; j += 1024) acc += partials[j];
    acc = cg_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { double t = red[0]; for (int w = 1; w < 16; ++w) t += red[w]; *result = t; }
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
    sl_timer timer;
    SL_TRY(timer.start(s));
    sl_status st = SL_OK;
    while (it < o->max_iterations) {
        if (rsold <= tol_sq) { converged = true; break; }                    // :221-224
        sl_row_args a = sl_matrix_row_args(m);
        a.gather = p.as<double>(); a.out = ap.as<double>();
        st = sl_launch_rows(a, (sl_order)o->order, SL_EPI_SPMV, s);          // ap = A p  (:227)
        if (st != SL_OK) break;
        ++mv;
        double pap = 0.0;
        st = sl_launch_dot(n, p.as<double>(), ap.as<double>(), scr, d_res, s);
        if (st == SL_OK) st = read(&pap);
        if (st != SL_OK) break;
        if (std::fabs(pap) < 1e-16) break;                                   // :236-238
        const double alpha = rsold / pap;
        hipLaunchKernelGGL(sl_cg_update_kernel, dim3(vgrid), dim3(256), 0, s, n, alpha, p.as<double>(), ap.as<double>(), x.as<double>(),
                           r.as<double>(), scr);
        hipLaunchKernelGGL(sl_cg_final_reduce_kernel, dim3(1), dim3(1024), 0, s, scr, vgrid, d_res);
        double rsnew = 0.0;
        st = read(&rsnew);
        if (st != SL_OK) break;
        if (!std::isfinite(rsnew)) { st = sl_fail(SL_NUMERICAL_INSTABILITY, "Non-finite residual in CG at iteration %llu", (unsigned long long)it); break; }
        const double beta = rsnew / rsold;
        hipLaunchKernelGGL(sl_cg_direction_kernel, dim3(vgrid), dim3(256), 0, s, n, beta, r.as<double>(), p.as<double>());
        rsold = rsnew;
        ++it;
    }
    const float ms = timer.stop();
    res->iterations = it; res->matvec_count = mv; res->residual_norm = std::sqrt(rsold); res->converged = converged ? 1 : 0;
    res->device_time_ms = ms;
    hipError_t ce = hipMemcpyAsync(x_out, x.p, n * 8, out_kind, s);
    if (ce == hipSuccess) ce = hipStreamSynchronize(s);
    if (ce != hipSuccess && st == SL_OK) st = sl_fail(SL_DEVICE_ERROR, "result download failed: %s", hipGetErrorString(ce));
    res->total_time_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - wall0).count();
    return st;
}

} // extern "C"
