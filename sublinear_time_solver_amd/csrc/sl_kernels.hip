// sl_kernels.hip — gfx950 kernels of the push / Neumann hot path.
//
// Roofline: every kernel here is HBM-bound irregular-gather / streaming work (no MFMA).
// Algorithmic bytes of the dominant kernel (fused Neumann step), DESIGN.md §4:
//     12 B per stored entry (8 value + 4 column) + 44 B per row
//     (row_len 4, dinv 8, t_i 8, t_out 8, x read 8 + write 8)  [+ 8 B/row gathered vector, counted once]
//
// Arithmetic parity: a row's dot product is accumulated by ONE lane, left to right, product
// rounded then added (the file is built with -ffp-contract=off; tests disassemble the code
// object and assert that no v_fma_f64 was emitted), so results are bit-identical to
// CSRStorage::multiply_vector (matrix/sparse.rs:187-203) resp. the 4-lane order of
// simd_ops::matrix_vector_multiply_simd (simd_ops.rs:20-88).
#include "sl_internal.hpp"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave = one 64-row slice; lane = row.  ORDER 0: sequential, 1: simd4 lanes.
// UW > 0: every row of the matrix has exactly UW entries (UW % 4 == 0): no slice_ptr /
// row_len reads, fully unrolled, all loads issued before the dependent add chain.
template <int ORDER, int EPI, int UW>
__global__ __launch_bounds__(SL_BLOCK) void sl_rows_kernel(sl_row_args a, uint32_t nb8)
{
    __shared__ double red[2 * SL_WAVES_PER_BLOCK];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    // XCD-aware mapping: physical block b is dispatched to XCD b % 8; give every XCD a
    // contiguous range of row blocks so that its private L2 sees one band of the gathered
    // vector (speed only — correctness does not depend on placement).
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    const uint64_t s = (uint64_t)lb * SL_WAVES_PER_BLOCK + wave;
    const uint64_t i = s * SL_SLICE + lane;
    const bool live_slice = s < a.n_slices;
    const bool live = live_slice && i < a.n_rows;
    const double *__restrict__ g = a.gather;
    const u32x4 *__restrict__ cq = reinterpret_cast<const u32x4 *>(a.cols);
    const f64x2 *__restrict__ vq = reinterpret_cast<const f64x2 *>(a.vals);

    // epilogue operands are fetched up front so their latency hides under the row walk
    double e_t = 0.0, e_d = 0.0, e_x = 0.0;
    if (live) {
        if constexpr (EPI == SL_EPI_NEUMANN) { e_t = g[a.row_offset + i]; e_d = a.dinv[i]; e_x = a.x[i]; }
        else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t = a.aux[i]; }
        else if constexpr (EPI == SL_EPI_PUSH) { e_t = a.r[i]; e_d = a.dinv[i]; e_x = a.x[i]; }
    }

    double sum = 0.0;
    if (live_slice) {
        if constexpr (UW > 0) {
            constexpr int NQ = UW / 4;
            const uint64_t qb = s * NQ;
            u32x4 c[NQ];
            f64x2 va[NQ], vb[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                c[q] = __builtin_nontemporal_load(&cq[(qb + q) * 64 + lane]);
                va[q] = __builtin_nontemporal_load(&vq[((qb + q) * 2) * 64 + lane]);
                vb[q] = __builtin_nontemporal_load(&vq[((qb + q) * 2 + 1) * 64 + lane]);
            }
            double t[UW];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                t[4 * q + 0] = g[c[q].x];
                t[4 * q + 1] = g[c[q].y];
                t[4 * q + 2] = g[c[q].z];
                t[4 * q + 3] = g[c[q].w];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                sum = DADD(sum, DMUL(va[q].x, t[4 * q + 0]));
                sum = DADD(sum, DMUL(va[q].y, t[4 * q + 1]));
                sum = DADD(sum, DMUL(vb[q].x, t[4 * q + 2]));
                sum = DADD(sum, DMUL(vb[q].y, t[4 * q + 3]));
            }
        } else {
            const uint32_t q0 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s]);
            const uint32_t q1 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s + 1]);
            const uint32_t len = a.row_len[i];
            if constexpr (ORDER == 0) {
                for (uint32_t q = q0; q < q1; ++q) {
                    const u32x4 c = __builtin_nontemporal_load(&cq[(uint64_t)q * 64 + lane]);
                    const f64x2 va = __builtin_nontemporal_load(&vq[((uint64_t)q * 2) * 64 + lane]);
                    const f64x2 vb = __builtin_nontemporal_load(&vq[((uint64_t)q * 2 + 1) * 64 + lane]);
                    const double t0 = g[c.x], t1 = g[c.y], t2 = g[c.z], t3 = g[c.w];
                    const uint32_t k = (q - q0) * 4;
                    const double s0 = DADD(sum, DMUL(va.x, t0));
                    sum = (k < len) ? s0 : sum;
                    const double s1 = DADD(sum, DMUL(va.y, t1));
                    sum = (k + 1 < len) ? s1 : sum;
                    const double s2 = DADD(sum, DMUL(vb.x, t2));
                    sum = (k + 2 < len) ? s2 : sum;
                    const double s3 = DADD(sum, DMUL(vb.y, t3));
                    sum = (k + 3 < len) ? s3 : sum;
                }
            } else {
                // simd_ops.rs:41-77: rows with >= 8 entries: four lane sums over the full
                // chunks of 4, then ((l0+l1)+l2)+l3, then the tail sequentially; shorter
                // rows: sequential from 0.0 (chunks_eff = 0, the horizontal sum of zeros is 0.0).
                const uint32_t chunks = (len >= 8u) ? (len >> 2) : 0u;
                double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
                bool merged = false;
                for (uint32_t q = q0; q < q1; ++q) {
                    const u32x4 c = __builtin_nontemporal_load(&cq[(uint64_t)q * 64 + lane]);
                    const f64x2 va = __builtin_nontemporal_load(&vq[((uint64_t)q * 2) * 64 + lane]);
                    const f64x2 vb = __builtin_nontemporal_load(&vq[((uint64_t)q * 2 + 1) * 64 + lane]);
                    const double p0 = DMUL(va.x, g[c.x]), p1 = DMUL(va.y, g[c.y]);
                    const double p2 = DMUL(vb.x, g[c.z]), p3 = DMUL(vb.y, g[c.w]);
                    const uint32_t qi = q - q0;
                    if (qi < chunks) {
                        l0 = DADD(l0, p0); l1 = DADD(l1, p1); l2 = DADD(l2, p2); l3 = DADD(l3, p3);
                    } else {
                        if (!merged) { sum = DADD(DADD(DADD(l0, l1), l2), l3); merged = true; }
                        const uint32_t k = qi * 4;
                        if (k < len) sum = DADD(sum, p0);
                        if (k + 1 < len) sum = DADD(sum, p1);
                        if (k + 2 < len) sum = DADD(sum, p2);
                        if (k + 3 < len) sum = DADD(sum, p3);
                    }
                }
                if (!merged) sum = DADD(DADD(DADD(l0, l1), l2), l3);
            }
        }
    }

    double part0 = 0.0, part1 = 0.0;
    if (live) {
        if constexpr (EPI == SL_EPI_SPMV) {
            a.out[i] = sum;
        } else if constexpr (EPI == SL_EPI_NEUMANN) {
            // neumann.rs:289-296: tmp *= dinv ; term -= tmp ; :264-266: solution += term
            const double tmp = DMUL(sum, e_d);
            const double tn = DSUB(e_t, tmp);
            a.out[i] = tn;
            a.x[i] = DADD(e_x, tn);
            part0 = DMUL(tn, tn);
        } else if constexpr (EPI == SL_EPI_RESIDUAL) {
            // neumann.rs:303-309: residual = A x - rhs
            const double rr = DSUB(sum, e_t);
            if (a.out) a.out[i] = rr;
            part0 = DMUL(rr, rr);
        } else { // SL_EPI_PUSH (dense round of the thresholded push):
            // x_i += delta_i (the frontier being consumed); r -= A delta; next frontier value
            // delta'_i = r_i * dinv_i where |.| >= theta, else 0.
            const double dself = g[a.row_offset + i];
            if (dself != 0.0) a.x[i] = DADD(e_x, dself);
            const double rn = DSUB(e_t, sum);
            a.r[i] = rn;
            const double p = DMUL(rn, e_d);
            const bool f = fabs(p) >= a.theta;
            a.out[i] = f ? p : 0.0;
            part0 = DMUL(rn, rn);
            part1 = f ? 1.0 : 0.0;
        }
    }
    if constexpr (EPI != SL_EPI_SPMV) {
        part0 = wave_sum(part0);
        if constexpr (EPI == SL_EPI_PUSH) part1 = wave_sum(part1);
        if (lane == 0) { red[wave] = part0; red[SL_WAVES_PER_BLOCK + wave] = part1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double p0 = red[0], p1 = red[SL_WAVES_PER_BLOCK];
#pragma unroll
            for (int w = 1; w < SL_WAVES_PER_BLOCK; ++w) { p0 += red[w]; p1 += red[SL_WAVES_PER_BLOCK + w]; }
            a.partials[lb] = p0;
            if constexpr (EPI == SL_EPI_PUSH) a.partials[(uint64_t)nb8 * 8 + lb] = p1;
        }
    }
}

// fixed-order final reduction of per-block partials: thread j sums partials j, j+1024, ...
// sequentially, then a fixed butterfly.  nsets independent sets laid out back to back.
__global__ __launch_bounds__(1024) void sl_final_reduce_kernel(const double *partials, uint32_t nparts,
                                                               double *result, int nsets)
{
    __shared__ double red[16];
    for (int set = 0; set < nsets; ++set) {
        const double *p = partials + (uint64_t)set * nparts;
        double acc = 0.0;
        for (uint32_t j = threadIdx.x; j < nparts; j += 1024) acc += p[j];
        acc = wave_sum(acc);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = red[0];
            for (int w = 1; w < 16; ++w) t += red[w];
            result[set] = t;
        }
        __syncthreads();
    }
}

uint32_t sl_row_grid(uint64_t n_slices)
{
    uint64_t nb = (n_slices + SL_WAVES_PER_BLOCK - 1) / SL_WAVES_PER_BLOCK;
    uint64_t nb8 = (nb + 7) / 8;
    if (nb8 == 0) nb8 = 1;
    return (uint32_t)(nb8 * 8);
}

template <int ORDER, int EPI>
static void launch_by_width(const sl_row_args &a, uint32_t grid, hipStream_t s)
{
    const uint32_t nb8 = grid / 8;
    if (ORDER == 0 && a.uniform_width == 16)
        hipLaunchKernelGGL((sl_rows_kernel<0, EPI, 16>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
    else if (ORDER == 0 && a.uniform_width == 8)
        hipLaunchKernelGGL((sl_rows_kernel<0, EPI, 8>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
    else
        hipLaunchKernelGGL((sl_rows_kernel<ORDER, EPI, 0>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
}

sl_status sl_launch_rows(const sl_row_args &a, sl_order order, sl_epilogue epi, hipStream_t s)
{
    if (a.n_slices == 0) {
        if (epi != SL_EPI_SPMV && a.result) SL_HIP(hipMemsetAsync(a.result, 0, 2 * sizeof(double), s));
        return SL_OK;
    }
    const uint32_t grid = sl_row_grid(a.n_slices);
    const bool simd4 = (order == SL_ORDER_SIMD4);
    switch (epi) {
    case SL_EPI_SPMV:
        simd4 ? launch_by_width<1, SL_EPI_SPMV>(a, grid, s) : launch_by_width<0, SL_EPI_SPMV>(a, grid, s);
        break;
    case SL_EPI_NEUMANN:
        simd4 ? launch_by_width<1, SL_EPI_NEUMANN>(a, grid, s) : launch_by_width<0, SL_EPI_NEUMANN>(a, grid, s);
        break;
    case SL_EPI_RESIDUAL:
        simd4 ? launch_by_width<1, SL_EPI_RESIDUAL>(a, grid, s) : launch_by_width<0, SL_EPI_RESIDUAL>(a, grid, s);
        break;
    case SL_EPI_PUSH:
        simd4 ? launch_by_width<1, SL_EPI_PUSH>(a, grid, s) : launch_by_width<0, SL_EPI_PUSH>(a, grid, s);
        break;
    }
    SL_HIP(hipGetLastError());
    if (epi != SL_EPI_SPMV && a.result) {
        hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, a.partials, grid, a.result,
                           epi == SL_EPI_PUSH ? 2 : 1);
        SL_HIP(hipGetLastError());
    }
    return SL_OK;
}

// ---- vector primitives (a5) -----------------------------------------------------------------
#define SL_VEC_BLOCKS 2048

template <int MODE> // 0: sum x^2, 1: sum x*y, 2: sum |x|
__global__ __launch_bounds__(256) void sl_reduce_kernel(uint64_t n, const double *__restrict__ x,
                                                        const double *__restrict__ y, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double v = x[i];
        if (MODE == 0) acc = DADD(acc, DMUL(v, v));
        else if (MODE == 1) acc = DADD(acc, DMUL(v, y[i]));
        else acc = DADD(acc, fabs(v));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

static uint32_t vec_grid(uint64_t n)
{
    uint64_t b = (n + 255) / 256;
    if (b > SL_VEC_BLOCKS) b = SL_VEC_BLOCKS;
    if (b == 0) b = 1;
    return (uint32_t)b;
}

sl_status sl_launch_sumsq(uint64_t n, const double *x, double *partials, double *result, hipStream_t s)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL((sl_reduce_kernel<0>), dim3(g), dim3(256), 0, s, n, x, x, partials);
    hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, g, result, 1);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
sl_status sl_launch_dot(uint64_t n, const double *x, const double *y, double *partials, double *result, hipStream_t s)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL((sl_reduce_kernel<1>), dim3(g), dim3(256), 0, s, n, x, y, partials);
    hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, g, result, 1);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
sl_status sl_launch_abs_sum(uint64_t n, const double *x, double *partials, double *result, hipStream_t s)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL((sl_reduce_kernel<2>), dim3(g), dim3(256), 0, s, n, x, x, partials);
    hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, g, result, 1);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

template <int MODE> // 0: y = alpha*x + y ; 1: out = a*b ; 2: out = a-b
__global__ __launch_bounds__(256) void sl_ewise_kernel(uint64_t n, double alpha, const double *__restrict__ a,
                                                       const double *__restrict__ b, double *out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (MODE == 0) out[i] = DADD(DMUL(alpha, a[i]), out[i]);
        else if (MODE == 1) out[i] = DMUL(a[i], b[i]);
        else out[i] = DSUB(a[i], b[i]);
    }
}
sl_status sl_launch_axpy(uint64_t n, double alpha, const double *x, double *y, hipStream_t s)
{
    hipLaunchKernelGGL((sl_ewise_kernel<0>), dim3(vec_grid(n)), dim3(256), 0, s, n, alpha, x, x, y);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
sl_status sl_launch_scale_rows(uint64_t n, const double *a, const double *b, double *out, hipStream_t s)
{
    hipLaunchKernelGGL((sl_ewise_kernel<1>), dim3(vec_grid(n)), dim3(256), 0, s, n, 0.0, a, b, out);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
sl_status sl_launch_sub(uint64_t n, const double *a, const double *b, double *out, hipStream_t s)
{
    hipLaunchKernelGGL((sl_ewise_kernel<2>), dim3(vec_grid(n)), dim3(256), 0, s, n, 0.0, a, b, out);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
