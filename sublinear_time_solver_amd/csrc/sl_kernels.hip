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
#include <cstdlib>
#include <mutex>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
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

// ---- shared pieces of the row kernels ----------------------------------------------------------
// Row walk: lane = row of slice s.  GATHER(c) returns the gathered-vector entry of column c —
// from HBM/L2 (global variant) or from the LDS-staged window (band variant).
// ORDER 0: sequential, 1: simd4 lanes.  UW > 0: every row has exactly UW entries (UW % 4 == 0):
// no slice_ptr / row_len reads, fully unrolled, all loads issued before the dependent add chain.
// `seed` = the value the running sum starts from: 0.0 everywhere except the accumulating product y += A x
// (CSRStorage::multiply_vector_add, sparse.rs:192-203, which adds into result[row] itself; ORDER 0 only).
template <int ORDER, int UW, class GATHER>
__device__ __forceinline__ double sl_row_walk(const sl_row_args &a, uint64_t s, uint32_t lane, uint64_t i, GATHER gather, double seed = 0.0)
{
    const u32x4 *__restrict__ cq = reinterpret_cast<const u32x4 *>(a.cols);
    const f64x2 *__restrict__ vq = reinterpret_cast<const f64x2 *>(a.vals);
    double sum = seed;
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
            t[4 * q + 0] = gather(c[q].x);
            t[4 * q + 1] = gather(c[q].y);
            t[4 * q + 2] = gather(c[q].z);
            t[4 * q + 3] = gather(c[q].w);
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            sum = DADD(sum, DMUL(va[q].x, t[4 * q + 0]));
            sum = DADD(sum, DMUL(va[q].y, t[4 * q + 1]));
            sum = DADD(sum, DMUL(vb[q].x, t[4 * q + 2]));
            sum = DADD(sum, DMUL(vb[q].y, t[4 * q + 3]));
        }
    } else {
        // slice_ptr counts pair blocks (128 entries): two consecutive blocks form a quad, an odd last block stands alone
        const uint32_t h0 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s]);
        const uint32_t h1 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s + 1]);
        const uint32_t len_raw = a.row_len[i];
        const uint32_t len = len_raw == SL_LONG_SENTINEL ? 0u : len_raw;     // long rows: sl_long_rows_kernel
        // simd_ops.rs:41-77: rows with >= 8 entries: four lane sums over the full chunks of 4, then ((l0+l1)+l2)+l3,
        // then the tail sequentially; shorter rows (and ORDER 0): sequential from 0.0
        const uint32_t chunks = (ORDER == 1 && len >= 8u) ? (len >> 2) : 0u;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        bool merged = false;
#pragma unroll 2
        for (uint32_t h = h0; h < h1; h += 2) {
            u32x4 c;
            f64x2 vb;
            const f64x2 va = __builtin_nontemporal_load(&vq[(uint64_t)h * 64 + lane]);
            if (h + 1 < h1) {                                                  // wave-uniform
                c = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.cols + (uint64_t)h * 128) + lane);
                vb = __builtin_nontemporal_load(&vq[(uint64_t)(h + 1) * 64 + lane]);
            } else {                                                           // the odd last pair block: entries k, k+1 only
                const u32x2 c2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.cols + (uint64_t)h * 128) + lane);
                c.x = c2.x; c.y = c2.y; c.z = c2.x; c.w = c2.x;
                vb.x = 0.0; vb.y = 0.0;
            }
            const double p0 = DMUL(va.x, gather(c.x)), p1 = DMUL(va.y, gather(c.y));
            const double p2 = DMUL(vb.x, gather(c.z)), p3 = DMUL(vb.y, gather(c.w));
            const uint32_t k = (h - h0) * 2;
            if (ORDER == 1 && (k >> 2) < chunks) {
                l0 = DADD(l0, p0); l1 = DADD(l1, p1); l2 = DADD(l2, p2); l3 = DADD(l3, p3);
            } else {
                if (ORDER == 1 && !merged) { sum = DADD(DADD(DADD(l0, l1), l2), l3); merged = true; }
                const double s0 = DADD(sum, p0);
                sum = (k < len) ? s0 : sum;
                const double s1 = DADD(sum, p1);
                sum = (k + 1 < len) ? s1 : sum;
                const double s2 = DADD(sum, p2);
                sum = (k + 2 < len) ? s2 : sum;
                const double s3 = DADD(sum, p3);
                sum = (k + 3 < len) ? s3 : sum;
            }
        }
        if (ORDER == 1 && !merged) sum = DADD(DADD(DADD(l0, l1), l2), l3);
    }
    return sum;
}

// Epilogue of a live row.  e_t / e_d / e_x were fetched before the row walk; dself = the gathered
// vector's own entry (PUSH only).
// SL_EPI_PUSH (dense round of the thresholded push): x_i += delta_i (the frontier being consumed); r -= A delta; next
// frontier value delta'_i = r_i * dinv_i where |.| >= theta, else 0.  thr = the row's threshold, zc = the column's constant value
// (column-constant operators: a.zout) — fetched by the caller's functions WHERE THEY ARE USED: the one-slice-per-wave kernels load them
// there (no register held across the row walk), the pipelined band kernel hands over what it had in flight with the row's other vectors
template <class THR, class ZC>
__device__ __forceinline__ void sl_push_epilogue(const sl_row_args &a, uint64_t i, double sum, double e_t, double e_d, double e_x, double dself,
                                                 THR thr, ZC zc, double &part0, double &part1)
{
    if (dself != 0.0) a.x[i] = DADD(e_x, dself);
    const double rn = DSUB(e_t, sum);
    a.r[i] = rn;
    const double p = DMUL(rn, e_d);
    const bool f = fabs(p) >= thr();
    const double dn = f ? p : 0.0;
    a.out[i] = dn;
    if (a.zout) a.zout[i] = DMUL(zc(), dn);       // column-constant operator: the one product every entry of column i will contribute next round
    part0 = DADD(part0, DMUL(rn, rn));
    part1 += f ? 1.0 : 0.0;
}
template <int EPI>
__device__ __forceinline__ void sl_row_epilogue(const sl_row_args &a, uint64_t i, double sum, double e_t, double e_d,
                                                double e_x, double dself, double &part0, double &part1)
{
    if constexpr (EPI == SL_EPI_SPMV) {
        a.out[i] = sum;
    } else if constexpr (EPI == SL_EPI_NEUMANN) {
        // neumann.rs:289-296: tmp *= dinv ; term -= tmp ; :264-266: solution += term
        const double tmp = DMUL(sum, e_d);
        const double tn = DSUB(e_t, tmp);
        // streamed once per iteration: keep them out of the way of the LDS-window / gather lines in L2
        __builtin_nontemporal_store(tn, &a.out[i]);
        __builtin_nontemporal_store(DADD(e_x, tn), &a.x[i]);
        part0 = DADD(part0, DMUL(tn, tn));
    } else if constexpr (EPI == SL_EPI_RESIDUAL) {
        if (a.aux_dot) {            // (wave-uniform) CG: A p with p . Ap in the same launch, optimized_solver.rs:227-233 — e_t = p_i
            a.out[i] = sum;
            part0 = DADD(part0, DMUL(e_t, sum));
            return;
        }
        // neumann.rs:303-309: residual = A x - rhs
        const double rr = DSUB(sum, e_t);
        if (a.out) a.out[i] = rr;
        part0 = DADD(part0, DMUL(rr, rr));
    } else { // SL_EPI_PUSH
        sl_push_epilogue(a, i, sum, e_t, e_d, e_x, dself, [&] { return a.theta_rows ? a.theta_rows[i] : a.theta; }, [&] { return a.zcol[i]; }, part0, part1);
    }
}

template <int EPI, int NWB = SL_WAVES_PER_BLOCK>
__device__ __forceinline__ void sl_block_partials(const sl_row_args &a, double *red, uint32_t lane, uint32_t wave, uint32_t lb,
                                                  uint32_t nparts, double part0, double part1)
{
    if constexpr (EPI != SL_EPI_SPMV) {
        part0 = wave_sum(part0);
        if constexpr (EPI == SL_EPI_PUSH) part1 = wave_sum(part1);
        if (lane == 0) { red[wave] = part0; red[NWB + wave] = part1; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double p0 = red[0], p1 = red[NWB];
#pragma unroll
            for (int w = 1; w < NWB; ++w) { p0 += red[w]; p1 += red[NWB + w]; }
            a.partials[lb] = p0;
            if constexpr (EPI == SL_EPI_PUSH) a.partials[(uint64_t)nparts + lb] = p1;
        }
    }
}

// ---- general kernel: gathers served by L1/L2/Infinity Cache/HBM --------------------------------
// One wave = one 64-row slice; lane = row.
template <int ORDER, int EPI, int UW>
__global__ __launch_bounds__(SL_BLOCK) void sl_rows_kernel(sl_row_args a, uint32_t nb8)
{
    __shared__ double red[2 * SL_WAVES_PER_BLOCK];
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;           // speculative solve loop: the stop rule already fired
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // a scalar: slice numbers and stream bases in SGPRs
    // XCD-aware mapping: physical block b is dispatched to XCD b % 8; give every XCD a
    // contiguous range of row blocks so that its private L2 sees one band of the gathered
    // vector (speed only — correctness does not depend on placement).
    uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    if (a.blk_cnt) { if (lb >= a.blk_cnt) return; lb += a.blk_lo; }     // a range of the launch's blocks (block-uniform)
    const uint64_t s = (uint64_t)lb * SL_WAVES_PER_BLOCK + wave;
    const uint64_t i = s * SL_SLICE + lane;
    const bool live_slice = s < a.n_slices;
    bool live = live_slice && i < a.n_rows;
    if constexpr (UW == 0) { if (live && a.n_long && a.row_len[i] == SL_LONG_SENTINEL) live = false; }
    const double *__restrict__ g = a.gather;

    // epilogue operands are fetched up front so their latency hides under the row walk
    double e_t = 0.0, e_d = 0.0, e_x = 0.0, dself = 0.0;
    if (live) {
        if constexpr (EPI == SL_EPI_NEUMANN) { e_t = g[a.row_offset + i]; e_d = a.dinv[i]; e_x = a.x[i]; }
        else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t = a.aux[i]; }
        else if constexpr (EPI == SL_EPI_PUSH) { e_t = a.r[i]; e_d = a.dinv[i]; e_x = a.x[i]; dself = g[a.row_offset + i]; }
    }
    double sum = 0.0;
    if (live_slice) sum = sl_row_walk<ORDER, UW>(a, s, lane, i, [g](uint32_t c) { return g[c]; });
    double part0 = 0.0, part1 = 0.0;
    if (live) sl_row_epilogue<EPI>(a, i, sum, e_t, e_d, e_x, dself, part0, part1);
    sl_block_partials<EPI>(a, red, lane, wave, lb, a.part_stride, part0, part1);
}

// y += A x, the running sum of row i seeded with y_i: CSRStorage::multiply_vector_add (sparse.rs:192-203; Matrix::multiply_vector_add,
// matrix/mod.rs:441-465).  (y_i + p_0) + p_1 ... rounds differently from y_i + ((p_0 + p_1) + ...): a launch of its own over the
// row-slice layout (every matrix carries it), whatever other layouts the matrix has; hub rows by sl_long_rows_kernel<0, SPMV, true>.
__global__ __launch_bounds__(SL_BLOCK) void sl_rows_add_kernel(sl_row_args a, uint32_t nb8)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // a scalar: slice numbers and stream bases in SGPRs
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    const uint64_t s = (uint64_t)lb * SL_WAVES_PER_BLOCK + wave;
    if (s >= a.n_slices) return;                                       // whole waves; no block barrier below
    const uint64_t i = s * SL_SLICE + lane;
    bool live = i < a.n_rows;
    if (live && a.n_long && a.row_len[i] == SL_LONG_SENTINEL) live = false;
    const double *__restrict__ g = a.gather;
    const double seed = live ? a.out[i] : 0.0;
    const double sum = sl_row_walk<0, 0>(a, s, lane, i, [g](uint32_t c) { return g[c]; }, seed);
    if (live) a.out[i] = sum;
}

// ---- band kernel: gathers served by an LDS-staged window of the gathered vector -----------------
// For matrices whose entries all satisfy |col - row| <= w (measured at create time).  A block of
// 4 waves owns R = 4 * spw * 64 consecutive rows; every entry those rows can touch lies in the window
// [r0 - w, r0 + R + w) of the gathered vector, which is staged ONCE into LDS with coalesced 16-B
// loads.  The 16 irregular gathers per row then become ds_read_b64 (a few LDS cycles per wave)
// instead of 64 serialized L1 tag lookups each, and the kernel is a pure HBM stream.
// Uniform-width matrices (UW = 8 / 16):
//   PIPE : software pipelining — the matrix bytes of slice j+1 (and of slice 0 before the window
//          barrier) are in flight while slice j is reduced (pays when the LDS window limits occupancy);
//   C16  : column indices are read as 16-bit offsets col - row (stored next to the u32 columns when
//          w < 32768): 10 instead of 12 bytes per entry cross the HBM interface.
template <int UW, bool C16>
struct sl_slice_regs {
    u32x4 c[UW > 0 ? (C16 ? UW / 8 : UW / 4) : 1];
    f64x2 va[UW > 0 ? UW / 4 : 1], vb[UW > 0 ? UW / 4 : 1];
    double e_d, e_x, e_aux;
    double e_th, e_z;          // push epilogue: the row's threshold and column value (where the launch has them)
};

// A wave-uniform pointer pinned to an SGPR pair and opaque to the optimiser.  Without it the compiler hoists `stream + lane * 16` out of
// the slice loop as a 64-bit VGPR pair per stream and adds the slice's offset on the VALU — into registers that double as load
// destinations, so that the next slice's loads wait (s_waitcnt vmcnt(0)) for the loads still in flight: one slice of loads per wave
// instead of the two the pipeline is written for.  Through this the loads are `global_load vdst, v_lane16, s[base:base+1] offset:imm`.
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *sl_scalar_ptr(const T *p)
{
    auto g = (const __attribute__((address_space(1))) T *)p;      // (global, not flat: the asm hides where the pointer came from)
    asm volatile("" : "+s"(g));
    return g;
}

// `s` must be wave-uniform (the callers derive it from the block's and the wave's number)
template <typename T>
__device__ __forceinline__ T sl_stream_load(const __attribute__((address_space(1))) T *base, uint32_t lane, int k)      // base[k * 64 + lane]
{
    return __builtin_nontemporal_load(&base[k * 64 + lane]);
}
template <int EPI, int UW, bool C16>
__device__ __forceinline__ void sl_slice_load(const sl_row_args &a, uint64_t s, uint32_t lane, sl_slice_regs<UW, C16> &r)
{
    constexpr int NQ = UW / 4;
    const auto *vq = sl_scalar_ptr(reinterpret_cast<const f64x2 *>(a.vals) + s * (NQ * 2 * 64));
    if constexpr (C16) {
        const auto *c16 = sl_scalar_ptr(reinterpret_cast<const u32x4 *>(a.cols16) + s * ((UW / 8) * 64));
#pragma unroll
        for (int o = 0; o < UW / 8; ++o) r.c[o] = sl_stream_load(c16, lane, o);
    } else {
        const auto *cq = sl_scalar_ptr(reinterpret_cast<const u32x4 *>(a.cols) + s * (NQ * 64));
#pragma unroll
        for (int q = 0; q < NQ; ++q) r.c[q] = sl_stream_load(cq, lane, q);
    }
#pragma unroll
    for (int q = 0; q < NQ; q += 2) {                                 // a scalar base per 4 KiB: the instruction's immediate offset reaches 4095
        const auto *vh = q ? sl_scalar_ptr(vq + q * 2 * 64) : vq;
#pragma unroll
        for (int h = 0; h < 2 && q + h < NQ; ++h) {
            r.va[q + h] = sl_stream_load(vh, lane, h * 2);
            r.vb[q + h] = sl_stream_load(vh, lane, h * 2 + 1);
        }
    }
    // the row's epilogue operands, from EVERY lane: the lanes past the last row of the matrix's last slice read that row's (and never use
    // them) — a load under a lane mask is a branch, and a branch around loads leaves the compiler without the count its waits need
    const uint64_t rows_left = a.n_rows - s * SL_SLICE;                 // >= 1: s is a slice of the matrix
    const uint32_t li = lane < rows_left ? lane : (uint32_t)rows_left - 1u;
    r.e_d = 0.0; r.e_x = 0.0; r.e_aux = 0.0;
    if constexpr (EPI == SL_EPI_NEUMANN) { r.e_d = sl_scalar_ptr(a.dinv + s * SL_SLICE)[li]; r.e_x = sl_scalar_ptr(a.x + s * SL_SLICE)[li]; }
    else if constexpr (EPI == SL_EPI_RESIDUAL) { r.e_aux = sl_scalar_ptr(a.aux + s * SL_SLICE)[li]; }
    else if constexpr (EPI == SL_EPI_PUSH) {
        r.e_aux = sl_scalar_ptr(a.r + s * SL_SLICE)[li]; r.e_d = sl_scalar_ptr(a.dinv + s * SL_SLICE)[li]; r.e_x = sl_scalar_ptr(a.x + s * SL_SLICE)[li];
        // the optional per-row threshold / column value: always a load (of dinv again — the same line, a hit — where the launch has none), never
        // a load under a branch
        r.e_th = sl_scalar_ptr((a.theta_rows ? a.theta_rows : a.dinv) + s * SL_SLICE)[li];
        r.e_z = sl_scalar_ptr((a.zout ? a.zcol : a.dinv) + s * SL_SLICE)[li];
    }
}

// LDS index of entry e (0..3) of quad q: window position of the column
template <int UW, bool C16>
__device__ __forceinline__ uint32_t sl_lds_index(const sl_slice_regs<UW, C16> &r, int q, int e, uint32_t base, uint32_t rowpos)
{
    if constexpr (C16) {
        const u32x4 v = r.c[q >> 1];
        const uint32_t word = (q & 1) ? (e < 2 ? v.z : v.w) : (e < 2 ? v.x : v.y);
        const int delta = (e & 1) ? ((int)word >> 16) : (int)(short)(word & 0xffffu);
        return (uint32_t)((int)rowpos + delta);
    } else {
        const u32x4 v = r.c[q];
        const uint32_t c = e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
        return c - base;
    }
}

template <int EPI, int UW, bool C16>
__device__ __forceinline__ void sl_slice_finish(const sl_row_args &a, uint64_t s, uint32_t lane, const sl_slice_regs<UW, C16> &r,
                                                const double *lw, uint32_t base, double &part0, double &part1)
{
    constexpr int NQ = UW / 4;
    const uint64_t i = s * SL_SLICE + lane;
    const uint32_t rowpos = (uint32_t)(a.row_offset + i) - base;       // window position of this lane's own row
    // LDS gathers are cheap and short-latency: fetch one quad ahead of the dependent add chain
    double sum = 0.0;
    double t0 = lw[sl_lds_index<UW, C16>(r, 0, 0, base, rowpos)], t1 = lw[sl_lds_index<UW, C16>(r, 0, 1, base, rowpos)];
    double t2 = lw[sl_lds_index<UW, C16>(r, 0, 2, base, rowpos)], t3 = lw[sl_lds_index<UW, C16>(r, 0, 3, base, rowpos)];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double n0 = 0.0, n1 = 0.0, n2 = 0.0, n3 = 0.0;
        if (q + 1 < NQ) {
            n0 = lw[sl_lds_index<UW, C16>(r, q + 1, 0, base, rowpos)]; n1 = lw[sl_lds_index<UW, C16>(r, q + 1, 1, base, rowpos)];
            n2 = lw[sl_lds_index<UW, C16>(r, q + 1, 2, base, rowpos)]; n3 = lw[sl_lds_index<UW, C16>(r, q + 1, 3, base, rowpos)];
        }
        sum = DADD(sum, DMUL(r.va[q].x, t0));
        sum = DADD(sum, DMUL(r.va[q].y, t1));
        sum = DADD(sum, DMUL(r.vb[q].x, t2));
        sum = DADD(sum, DMUL(r.vb[q].y, t3));
        t0 = n0; t1 = n1; t2 = n2; t3 = n3;
    }
    if (i < a.n_rows) {
        const double own = lw[rowpos];                                   // the gathered vector's own entry
        if constexpr (EPI == SL_EPI_NEUMANN) sl_row_epilogue<EPI>(a, i, sum, own, r.e_d, r.e_x, 0.0, part0, part1);
        else if constexpr (EPI == SL_EPI_PUSH) sl_push_epilogue(a, i, sum, r.e_aux, r.e_d, r.e_x, own, [&] { return a.theta_rows ? r.e_th : a.theta; }, [&] { return r.e_z; }, part0, part1);
        else sl_row_epilogue<EPI>(a, i, sum, r.e_aux, 0.0, 0.0, 0.0, part0, part1);
    }
}

// ---- ragged rows in the band kernel: quad-granular batches, software pipelined ---------------------------
// A slice of width W is consumed in batches of up to 4 quads (16 entries per lane); the loads of the next
// batch — of the same slice or the first one of the wave's next slice, together with that slice's epilogue
// operands — are in flight while the current batch is reduced.  Everything that steers the loop (slice
// pointers, quad counts) is wave-uniform.  C16: columns come as 16-bit offsets col - row, layout
// [quad][lane][4] (8 B per lane per quad).
struct sl_batch { uint64_t s; uint32_t q, nq, kbase, q1; bool first, last, valid; };    // q, q1: pair blocks; nq: quads (the last may be half)
#ifndef SL_BATCH_QUADS_NW8
#define SL_BATCH_QUADS_NW8 3      /* measured: 3 quads per batch beat 2 and 4 (4 spills at 128 VGPRs), gpurun 2026-09 A/B in DESIGN.md §5 */
#endif
template <int BQ>
struct sl_batch_regs {
    u32x4 c[BQ];
    f64x2 va[BQ], vb[BQ];
    double e_d, e_x, e_aux;
    uint32_t len;
};
struct sl_batch_cursor { uint64_t s0, s; uint32_t j, spw, q, q0, q1; bool in_slice; uint32_t lane_q0, lane_q1, nw; };   // lane_q*: slice pointers of slice j held by lane j
struct sl_row_state { double sum, l0, l1, l2, l3, e_d, e_x, e_aux; uint32_t len, chunks; bool merged, live; };

template <int BQ>
__device__ __forceinline__ sl_batch sl_next_batch(const sl_row_args &a, sl_batch_cursor &c)
{
    sl_batch b{0, 0, 0, 0, 0, false, false, false};
    if (!c.in_slice) {
        if (c.j >= c.spw) return b;
        c.s = c.s0 + (uint64_t)c.j * c.nw;
        if (c.s >= a.n_slices) return b;
        c.q0 = __builtin_amdgcn_readlane(c.lane_q0, c.j);      // fetched for all of the wave's slices up front
        c.q1 = __builtin_amdgcn_readlane(c.lane_q1, c.j);
        c.q = c.q0;
        c.in_slice = true;
    }
    uint32_t nq = (c.q1 - c.q + 1u) >> 1;                      // quads left in the slice, an odd last pair block counting as one
    nq = nq > (uint32_t)BQ ? (uint32_t)BQ : nq;
    b.s = c.s; b.q = c.q; b.nq = nq; b.kbase = (c.q - c.q0) * 2u; b.q1 = c.q1;
    b.first = c.q == c.q0;
    c.q += 2u * nq;
    if (c.q >= c.q1) { c.q = c.q1; c.in_slice = false; ++c.j; b.last = true; }
    b.valid = true;
    return b;
}

template <int EPI, bool C16, int BQ>
__device__ __forceinline__ void sl_batch_load(const sl_row_args &a, const sl_batch &b, uint32_t lane, sl_batch_regs<BQ> &r)
{
    const f64x2 *__restrict__ vq = reinterpret_cast<const f64x2 *>(a.vals);
#pragma unroll
    for (int qq = 0; qq < BQ; ++qq) {
        if ((uint32_t)qq < b.nq) {
            const uint64_t h = (uint64_t)b.q + 2u * (uint32_t)qq;             // pair block of this quad's first half
            const bool half = h + 1 == b.q1;                                   // the slice's odd last pair block (wave-uniform)
            if constexpr (C16) {
                if (!half) {
                    const unsigned long long w2 = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long *>(a.cols16) + h * 32 + lane);
                    r.c[qq].x = (uint32_t)w2; r.c[qq].y = (uint32_t)(w2 >> 32);
                } else {
                    r.c[qq].x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(a.cols16) + h * 64 + lane);
                    r.c[qq].y = 0u;                                            // offsets 0: the row itself (never added)
                }
            } else {
                if (!half) {
                    r.c[qq] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.cols + h * 128) + lane);
                } else {
                    const u32x2 c2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.cols + h * 128) + lane);
                    r.c[qq].x = c2.x; r.c[qq].y = c2.y; r.c[qq].z = c2.x; r.c[qq].w = c2.x;
                }
            }
            r.va[qq] = __builtin_nontemporal_load(&vq[h * 64 + lane]);
            if (!half) r.vb[qq] = __builtin_nontemporal_load(&vq[(h + 1) * 64 + lane]);
            else { r.vb[qq].x = 0.0; r.vb[qq].y = 0.0; }
        }
    }
    if (b.first) {
        const uint64_t i = b.s * SL_SLICE + lane;
        r.len = a.row_len[i];
        r.e_d = 0.0; r.e_x = 0.0; r.e_aux = 0.0;
        if (i < a.n_rows && r.len != SL_LONG_SENTINEL) {
            if constexpr (EPI == SL_EPI_NEUMANN) { r.e_d = a.dinv[i]; r.e_x = a.x[i]; }
            else if constexpr (EPI == SL_EPI_RESIDUAL) { r.e_aux = a.aux[i]; }
            else if constexpr (EPI == SL_EPI_PUSH) { r.e_aux = a.r[i]; r.e_d = a.dinv[i]; r.e_x = a.x[i]; }
        }
    }
}

template <int ORDER, int EPI, bool C16, int BQ>
__device__ __forceinline__ void sl_batch_finish(const sl_row_args &a, const sl_batch &b, uint32_t lane, const sl_batch_regs<BQ> &r,
                                                sl_row_state &st, const double *lw, uint32_t base, double &part0, double &part1)
{
    const uint64_t i = b.s * SL_SLICE + lane;
    const uint32_t rowpos = (uint32_t)(a.row_offset + i) - base;
    if (b.first) {
        st.sum = 0.0; st.l0 = st.l1 = st.l2 = st.l3 = 0.0; st.merged = false;
        st.live = i < a.n_rows && r.len != SL_LONG_SENTINEL;
        st.len = r.len == SL_LONG_SENTINEL ? 0u : r.len;
        st.chunks = (st.len >= 8u) ? (st.len >> 2) : 0u;
        st.e_d = r.e_d; st.e_x = r.e_x; st.e_aux = r.e_aux;
    }
#pragma unroll
    for (int qq = 0; qq < BQ; ++qq) {
        if ((uint32_t)qq < b.nq) {
            uint32_t i0, i1, i2, i3;
            if constexpr (C16) {
                const uint32_t w0 = r.c[qq].x, w1 = r.c[qq].y;
                i0 = (uint32_t)((int)rowpos + (int)(short)(w0 & 0xffffu)); i1 = (uint32_t)((int)rowpos + ((int)w0 >> 16));
                i2 = (uint32_t)((int)rowpos + (int)(short)(w1 & 0xffffu)); i3 = (uint32_t)((int)rowpos + ((int)w1 >> 16));
            } else {
                i0 = r.c[qq].x - base; i1 = r.c[qq].y - base; i2 = r.c[qq].z - base; i3 = r.c[qq].w - base;
            }
            const double p0 = DMUL(r.va[qq].x, lw[i0]), p1 = DMUL(r.va[qq].y, lw[i1]);
            const double p2 = DMUL(r.vb[qq].x, lw[i2]), p3 = DMUL(r.vb[qq].y, lw[i3]);
            const uint32_t k = b.kbase + 4u * qq;
            if constexpr (ORDER == 0) {
                const double s0 = DADD(st.sum, p0); st.sum = (k < st.len) ? s0 : st.sum;
                const double s1 = DADD(st.sum, p1); st.sum = (k + 1 < st.len) ? s1 : st.sum;
                const double s2 = DADD(st.sum, p2); st.sum = (k + 2 < st.len) ? s2 : st.sum;
                const double s3 = DADD(st.sum, p3); st.sum = (k + 3 < st.len) ? s3 : st.sum;
            } else {
                if ((k >> 2) < st.chunks) {
                    st.l0 = DADD(st.l0, p0); st.l1 = DADD(st.l1, p1); st.l2 = DADD(st.l2, p2); st.l3 = DADD(st.l3, p3);
                } else {
                    if (!st.merged) { st.sum = DADD(DADD(DADD(st.l0, st.l1), st.l2), st.l3); st.merged = true; }
                    if (k < st.len) st.sum = DADD(st.sum, p0);
                    if (k + 1 < st.len) st.sum = DADD(st.sum, p1);
                    if (k + 2 < st.len) st.sum = DADD(st.sum, p2);
                    if (k + 3 < st.len) st.sum = DADD(st.sum, p3);
                }
            }
        }
    }
    if (b.last) {
        if constexpr (ORDER == 1) { if (!st.merged) st.sum = DADD(DADD(DADD(st.l0, st.l1), st.l2), st.l3); }
        if (st.live) {
            const double own = lw[rowpos];
            if constexpr (EPI == SL_EPI_NEUMANN) sl_row_epilogue<EPI>(a, i, st.sum, own, st.e_d, st.e_x, 0.0, part0, part1);
            else if constexpr (EPI == SL_EPI_PUSH) sl_row_epilogue<EPI>(a, i, st.sum, st.e_aux, st.e_d, st.e_x, own, part0, part1);
            else sl_row_epilogue<EPI>(a, i, st.sum, st.e_aux, 0.0, 0.0, 0.0, part0, part1);
        }
    }
}

// NW = waves per block.  4 by default; 8 for wide windows, where the LDS window (not registers) caps the CU at two
// blocks: 8-wave blocks then double the waves in flight per window (4 per SIMD — the second launch-bounds argument
// holds the kernel to 128 VGPRs for that).  Measured at w = 4096: +5..6 % (profiles/r01_ab_wave_blocks.txt).
template <int ORDER, int EPI, int UW, bool PIPE, bool C16, int NW>
__global__ __launch_bounds__(NW * 64, NW >= 8 ? 4 : 1) void sl_band_kernel(sl_row_args a, uint32_t nb8, uint32_t spw, uint32_t w)
{
    extern __shared__ __attribute__((aligned(16))) double win[];
    __shared__ double red[2 * NW];
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;           // speculative solve loop: the stop rule already fired
    const uint32_t lane = threadIdx.x & 63u;
    // the wave's number as a SCALAR: slice numbers and every stream address derived from them then live in SGPRs (loads take the
    // saddr + lane-offset form) instead of one 64-bit VGPR pair per stream — the 16-wave instantiation, held to 128 VGPRs, spilled
    // such a pair and reloaded it from scratch in front of every prefetch (s_waitcnt vmcnt(0): the pipeline drained each slice)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);
    if (a.blk_cnt) { if (lb >= a.blk_cnt) return; lb += a.blk_lo; }     // a range of the launch's blocks (block-uniform)
    const uint64_t R = (uint64_t)NW * spw * SL_SLICE;
    const uint64_t r0 = (uint64_t)lb * R;                       // first local row of the block
    double part0 = 0.0, part1 = 0.0;
    if (r0 < a.n_rows) {                                        // block-uniform
        const uint64_t s0 = (uint64_t)lb * spw * NW + wave;     // this wave's slices: s0, s0 + NW, ...
        [[maybe_unused]] sl_slice_regs<UW, C16> ra, rb;
        if constexpr (UW > 0 && PIPE) { if (s0 < a.n_slices) sl_slice_load<EPI, UW, C16>(a, s0, lane, ra); }
        [[maybe_unused]] uint32_t pre_q0 = 0, pre_q1 = 0;        // ragged path: lane j prefetches the pointers of slice j
        if constexpr (UW == 0 && PIPE) {
            const uint64_t sj = s0 + (uint64_t)lane * NW;
            if (lane < spw && sj < a.n_slices) { pre_q0 = a.slice_ptr[sj]; pre_q1 = a.slice_ptr[sj + 1]; }
        }

        const uint64_t g0 = a.row_offset + r0;                  // global index of the block's first row
        const uint64_t win_lo = (g0 > w ? g0 - w : 0) & ~1ull;  // even => 16-B aligned staging loads
        uint64_t win_hi = g0 + R + w;
        if (win_hi > a.n_cols) win_hi = a.n_cols;
        if (win_hi < win_lo) win_hi = win_lo;                    // rows past the last column (never selected for such matrices: sl_matrix.hip): empty window
        const uint32_t len = (uint32_t)(win_hi - win_lo);
        const double *__restrict__ src = a.gather + win_lo;
        const f64x2 *__restrict__ src2 = reinterpret_cast<const f64x2 *>(src);
        f64x2 *win2 = reinterpret_cast<f64x2 *>(win);
        const uint32_t pairs = len >> 1;
#ifndef SL_NO_LDS_DMA
        // the window is staged with direct global -> LDS loads (gfx950: 16 B per lane, no VGPR round trip): each wave moves 64 x 16 B
        // = 1 KiB per instruction into consecutive LDS; +2 % at w = 4096 over load + ds_write, neutral for narrow windows
        for (uint32_t p0 = wave * 64u; p0 < pairs; p0 += NW * 64) {
            if (p0 + lane < pairs)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src2 + p0 + lane),
                                                 (__attribute__((address_space(3))) void *)(win2 + p0), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
        for (uint32_t p = threadIdx.x; p < pairs; p += NW * 64) win2[p] = src2[p];
#endif
        if ((len & 1u) && threadIdx.x == 0) win[len - 1] = src[len - 1];
        __syncthreads();
        const double *lw = win;
        const uint32_t base = (uint32_t)win_lo;

        if constexpr (UW > 0 && PIPE) {
            // ra holds the wave's first slice.  Steady state: both successors exist, so every load of the body is issued unconditionally
            // and the compiler's wait counts are exact — slice j + 1's loads stay in flight while slice j is reduced, slice j + 2's are
            // issued before slice j + 1's are waited for.  (With the loads under `if (next slice exists)` the waits were placed for the
            // path that skips them: vmcnt(0) a few instructions into every reduction, the loads just issued included.)
            uint32_t nsl = 0;                                           // this wave's slices: s0, s0 + NW, ... (scalar)
            if (s0 < a.n_slices) { const uint64_t left = (a.n_slices - s0 + NW - 1) / NW; nsl = left < spw ? (uint32_t)left : spw; }
            uint32_t j = 0;
            for (; j + 2 < nsl; j += 2) {
                const uint64_t sa = s0 + (uint64_t)j * NW, sb = sa + NW, sc = sb + NW;
                sl_slice_load<EPI, UW, C16>(a, sb, lane, rb);
                sl_slice_finish<EPI, UW, C16>(a, sa, lane, ra, lw, base, part0, part1);
                sl_slice_load<EPI, UW, C16>(a, sc, lane, ra);
                sl_slice_finish<EPI, UW, C16>(a, sb, lane, rb, lw, base, part0, part1);
            }
            if (j < nsl) {                                              // one or two slices left, the first of them in ra
                const uint64_t sa = s0 + (uint64_t)j * NW;
                if (j + 1 < nsl) {
                    sl_slice_load<EPI, UW, C16>(a, sa + NW, lane, rb);
                    sl_slice_finish<EPI, UW, C16>(a, sa, lane, ra, lw, base, part0, part1);
                    sl_slice_finish<EPI, UW, C16>(a, sa + NW, lane, rb, lw, base, part0, part1);
                } else {
                    sl_slice_finish<EPI, UW, C16>(a, sa, lane, ra, lw, base, part0, part1);
                }
            }
        } else if constexpr (UW > 0) {
            for (uint32_t j = 0; j < spw; ++j) {
                const uint64_t s = s0 + (uint64_t)j * NW;
                if (s >= a.n_slices) break;
                sl_slice_load<EPI, UW, C16>(a, s, lane, ra);
                sl_slice_finish<EPI, UW, C16>(a, s, lane, ra, lw, base, part0, part1);
            }
        } else if constexpr (PIPE) {
            sl_batch_cursor cur{s0, 0, 0, spw, 0, 0, 0, false, pre_q0, pre_q1, (uint32_t)NW};
            constexpr int BQ = (NW >= 8) ? (ORDER == 1 ? 2 : SL_BATCH_QUADS_NW8) : 4;   // 8-wave blocks live in 128 VGPRs: shorter batches, no spills
            sl_batch_regs<BQ> ga, gb;
            sl_row_state st{};
            sl_batch ba = sl_next_batch<BQ>(a, cur);
            if (ba.valid) sl_batch_load<EPI, C16, BQ>(a, ba, lane, ga);
            while (ba.valid) {
                const sl_batch bb = sl_next_batch<BQ>(a, cur);
                if (bb.valid) sl_batch_load<EPI, C16, BQ>(a, bb, lane, gb);
                sl_batch_finish<ORDER, EPI, C16, BQ>(a, ba, lane, ga, st, lw, base, part0, part1);
                if (!bb.valid) break;
                ba = sl_next_batch<BQ>(a, cur);
                if (ba.valid) sl_batch_load<EPI, C16, BQ>(a, ba, lane, ga);
                sl_batch_finish<ORDER, EPI, C16, BQ>(a, bb, lane, gb, st, lw, base, part0, part1);
            }
        } else {
            for (uint32_t j = 0; j < spw; ++j) {
                const uint64_t s = s0 + (uint64_t)j * NW;
                if (s >= a.n_slices) break;
                const uint64_t i = s * SL_SLICE + lane;
                bool live = i < a.n_rows;
                if (live && a.n_long && a.row_len[i] == SL_LONG_SENTINEL) live = false;
                double e_t = 0.0, e_d = 0.0, e_x = 0.0, dself = 0.0;
                if (live) {
                    if constexpr (EPI == SL_EPI_NEUMANN) { e_t = lw[(uint32_t)(a.row_offset + i) - base]; e_d = a.dinv[i]; e_x = a.x[i]; }
                    else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t = a.aux[i]; }
                    else if constexpr (EPI == SL_EPI_PUSH) { e_t = a.r[i]; e_d = a.dinv[i]; e_x = a.x[i]; dself = lw[(uint32_t)(a.row_offset + i) - base]; }
                }
                const double sum = sl_row_walk<ORDER, 0>(a, s, lane, i, [lw, base](uint32_t c) { return lw[c - base]; });
                if (live) sl_row_epilogue<EPI>(a, i, sum, e_t, e_d, e_x, dself, part0, part1);
            }
        }
    }
    sl_block_partials<EPI, NW>(a, red, lane, wave, lb, a.part_stride, part0, part1);
}

// ---- multi-pass window kernel: bands too wide for ONE LDS window ------------------------------------------------------------
// Half-widths beyond ~9.5 K (window > 158 KiB) used to fall back to the general kernel, whose gathers go through L1 / L2 at one
// lane per cycle: w = 32768 43 %, the 7-point stencil on 215^3 69 % of the roofline.  Here a block of 8 waves keeps the matrix
// bytes of its rows — SPW slices per wave, every row at most 4 * MAXQ entries — in REGISTERS (one coalesced pass over HBM, as
// in the band kernel), and walks the window [r0 - w, r0 + R + w) segment by segment: a segment of SL_MP_SEG columns (64 KiB) is
// staged into one of two LDS buffers with direct global -> LDS loads while the other one is being consumed; every lane then adds
// those of its entries whose column lies in the segment.  Entries of a row ascend in column and segments ascend, so a row's
// products are still added left to right, one by one, by the lane that owns the row: the bits of the sequential reference loop
// (sparse.rs:187-203).  Segments none of the block's entries fall into are skipped (one 64-bit occupancy mask per block: the three
// clusters of a 3-D stencil stage 6 segments instead of 12), and inside a segment an entry position k is only visited when some
// lane of the wave has it in range.  Staged bytes: (R + 2 w) * 8 per R rows out of the L2 (neighbouring blocks stage overlapping
// windows; the union is read from HBM once), R = 8 * SPW * 64 rows — 2048 at 16 entries per row.
#define SL_MP_SEG 8192u                       // columns per segment: 64 KiB
#define SL_MP_WAVES 8
template <int EPI, int MAXQ, int SPW>
__global__ __launch_bounds__(SL_MP_WAVES * 64, 2) void sl_mpass_kernel(sl_row_args a, uint32_t nb8, uint32_t w)
{
    extern __shared__ __attribute__((aligned(16))) double mp_seg[];          // two segment buffers
    __shared__ double red[2 * SL_MP_WAVES];
    __shared__ unsigned long long occ;
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;
    const uint32_t lane = threadIdx.x & 63u, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t lb = (blockIdx.x & 7u) * nb8 + (blockIdx.x >> 3);        // XCD-aware: an XCD's blocks own one contiguous row range
    constexpr uint64_t R = (uint64_t)SL_MP_WAVES * SPW * SL_SLICE;
    const uint64_t r0 = (uint64_t)lb * R;
    double part0 = 0.0, part1 = 0.0;
    if (threadIdx.x == 0) occ = 0ull;
    __syncthreads();
    if (r0 < a.n_rows) {                                                     // block-uniform
        const uint64_t s0 = (uint64_t)lb * SPW * SL_MP_WAVES + wave;         // this wave's slices: s0, s0 + 8, ...
        const uint64_t g0 = a.row_offset + r0;
        const uint64_t win_lo = (g0 > w ? g0 - w : 0) & ~1ull;
        uint64_t win_hi = g0 + R + w;
        if (win_hi > a.n_cols) win_hi = a.n_cols;
        if (win_hi < win_lo) win_hi = win_lo;
        const uint32_t nseg = (uint32_t)((win_hi - win_lo + SL_MP_SEG - 1) / SL_MP_SEG);      // <= 64 (checked by the launcher)
        const u32x4 *__restrict__ cq = reinterpret_cast<const u32x4 *>(a.cols);
        const f64x2 *__restrict__ vq = reinterpret_cast<const f64x2 *>(a.vals);
        // ---- the block's matrix bytes into registers; column = position in the window (0 for slots that are never added) ----
        uint32_t col[SPW][MAXQ * 4], len[SPW];
        double val[SPW][MAXQ * 4], sum[SPW], e_d[SPW], e_x[SPW], e_aux[SPW], e_own[SPW];
        unsigned long long mymask = 0ull;
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const uint64_t s = s0 + (uint64_t)j * SL_MP_WAVES;
            const uint64_t i = s * SL_SLICE + lane;
            len[j] = 0u; sum[j] = 0.0; e_d[j] = 0.0; e_x[j] = 0.0; e_aux[j] = 0.0; e_own[j] = 0.0;
#pragma unroll
            for (int k = 0; k < MAXQ * 4; ++k) { col[j][k] = 0u; val[j][k] = 0.0; }
            if (s < a.n_slices) {                                            // wave-uniform
                const uint32_t h0 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s]), h1 = __builtin_amdgcn_readfirstlane(a.slice_ptr[s + 1]);
                const uint32_t lr = a.row_len[i];
                len[j] = (i < a.n_rows && lr != SL_LONG_SENTINEL) ? lr : 0u;
#pragma unroll
                for (int q = 0; q < MAXQ; ++q) {
                    const uint32_t h = h0 + 2u * (uint32_t)q;
                    if (h + 1 < h1) {
                        const u32x4 c = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.cols + (uint64_t)h * 128) + lane);
                        const f64x2 va = __builtin_nontemporal_load(&vq[(uint64_t)h * 64 + lane]), vb = __builtin_nontemporal_load(&vq[(uint64_t)(h + 1) * 64 + lane]);
                        col[j][4 * q] = c.x; col[j][4 * q + 1] = c.y; col[j][4 * q + 2] = c.z; col[j][4 * q + 3] = c.w;
                        val[j][4 * q] = va.x; val[j][4 * q + 1] = va.y; val[j][4 * q + 2] = vb.x; val[j][4 * q + 3] = vb.y;
                    } else if (h < h1) {                                     // the slice's odd last pair block: entries 4q, 4q + 1 only
                        const u32x2 c2 = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(a.cols + (uint64_t)h * 128) + lane);
                        const f64x2 va = __builtin_nontemporal_load(&vq[(uint64_t)h * 64 + lane]);
                        col[j][4 * q] = c2.x; col[j][4 * q + 1] = c2.y;
                        val[j][4 * q] = va.x; val[j][4 * q + 1] = va.y;
                    }
                }
                if (len[j]) {
                    if constexpr (EPI == SL_EPI_NEUMANN) { e_own[j] = a.gather[a.row_offset + i]; e_d[j] = a.dinv[i]; e_x[j] = a.x[i]; }
                    else if constexpr (EPI == SL_EPI_RESIDUAL) { e_aux[j] = a.aux[i]; }
                    else if constexpr (EPI == SL_EPI_PUSH) { e_aux[j] = a.r[i]; e_d[j] = a.dinv[i]; e_x[j] = a.x[i]; e_own[j] = a.gather[a.row_offset + i]; }
                }
#pragma unroll
                for (int k = 0; k < MAXQ * 4; ++k) {
                    const bool in = (uint32_t)k < len[j];
                    const uint32_t rel = in ? (uint32_t)(col[j][k] - (uint32_t)win_lo) : 0u;
                    col[j][k] = in ? rel : 0xffffffffu;                      // out of every segment
                    if (in) mymask |= 1ull << (rel / SL_MP_SEG);
                }
            }
        }
        (void)cq;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mymask |= __shfl_xor(mymask, o);
        if (lane == 0 && mymask) atomicOr(&occ, mymask);
        __syncthreads();
        unsigned long long todo = occ;
        if (nseg < 64u) todo &= (1ull << nseg) - 1ull;
        // ---- segments: stage the next occupied one while the current one is consumed ----
        auto stage = [&](uint32_t sg, uint32_t buf) {
            const uint64_t lo = win_lo + (uint64_t)sg * SL_MP_SEG;
            uint64_t hi = lo + SL_MP_SEG;
            if (hi > win_hi) hi = win_hi;
            const uint32_t cnt = (uint32_t)(hi - lo), pairs = cnt >> 1;
            const f64x2 *__restrict__ src2 = reinterpret_cast<const f64x2 *>(a.gather + lo);
            f64x2 *dst2 = reinterpret_cast<f64x2 *>(mp_seg + (size_t)buf * SL_MP_SEG);
            for (uint32_t p0 = wave * 64u; p0 < pairs; p0 += SL_MP_WAVES * 64)
                if (p0 + lane < pairs)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src2 + p0 + lane),
                                                     (__attribute__((address_space(3))) void *)(dst2 + p0), 16, 0, 0);
            if ((cnt & 1u) && threadIdx.x == 0) mp_seg[(size_t)buf * SL_MP_SEG + cnt - 1] = a.gather[lo + cnt - 1];
        };
        uint32_t buf = 0;
        if (todo) stage((uint32_t)__builtin_ctzll(todo), 0);
        while (todo) {
            const uint32_t sg = (uint32_t)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                 // segment sg is complete in `buf`; the other buffer is free again
            if (todo) stage((uint32_t)__builtin_ctzll(todo), buf ^ 1u);
            const double *lw = mp_seg + (size_t)buf * SL_MP_SEG;
            const uint32_t base = sg * SL_MP_SEG;
#pragma unroll
            for (int j = 0; j < SPW; ++j) {
#pragma unroll
                for (int k = 0; k < MAXQ * 4; ++k) {
                    const uint32_t off = col[j][k] - base;
                    const bool in = off < SL_MP_SEG;                         // padding slots (0xffffffff) are in no segment
                    if (__ballot(in)) {
                        const double tv = lw[in ? off : 0u];
                        const double sn = DADD(sum[j], DMUL(val[j][k], tv));
                        sum[j] = in ? sn : sum[j];
                    }
                }
            }
            buf ^= 1u;
        }
#pragma unroll
        for (int j = 0; j < SPW; ++j) {
            const uint64_t s = s0 + (uint64_t)j * SL_MP_WAVES;
            const uint64_t i = s * SL_SLICE + lane;
            if (s < a.n_slices && i < a.n_rows && (len[j] || !(a.n_long && a.row_len[i] == SL_LONG_SENTINEL))) {
                if constexpr (EPI == SL_EPI_NEUMANN) {
                    const double own = len[j] ? e_own[j] : a.gather[a.row_offset + i];
                    const double dd = len[j] ? e_d[j] : a.dinv[i], xx = len[j] ? e_x[j] : a.x[i];
                    sl_row_epilogue<EPI>(a, i, sum[j], own, dd, xx, 0.0, part0, part1);
                } else if constexpr (EPI == SL_EPI_PUSH) {
                    const double own = len[j] ? e_own[j] : a.gather[a.row_offset + i];
                    const double rr = len[j] ? e_aux[j] : a.r[i], dd = len[j] ? e_d[j] : a.dinv[i], xx = len[j] ? e_x[j] : a.x[i];
                    sl_row_epilogue<EPI>(a, i, sum[j], rr, dd, xx, own, part0, part1);
                } else if constexpr (EPI == SL_EPI_RESIDUAL) {
                    sl_row_epilogue<EPI>(a, i, sum[j], len[j] ? e_aux[j] : a.aux[i], 0.0, 0.0, 0.0, part0, part1);
                } else {
                    sl_row_epilogue<EPI>(a, i, sum[j], 0.0, 0.0, 0.0, 0.0, part0, part1);
                }
            }
        }
    }
    sl_block_partials<EPI, SL_MP_WAVES>(a, red, lane, wave, lb, a.part_stride, part0, part1);
}

// ---- cross-lane moves on the VALU (DPP) instead of the LDS pipe (ds_bpermute) ----------------------------------------------------
// gfx9 wave shifts (checked on gfx950, tools/dpp_check.hip): wave_shl:1 hands lane l the value of lane l + 1, wave_shr:1 that of
// lane l - 1; the lane without a source reads 0.
__device__ __forceinline__ uint32_t dpp_from_right(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_from_left(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ double dpp_from_right(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = dpp_from_right((uint32_t)b), hi = dpp_from_right((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// minimum over each row of 16 lanes, left in every lane of the row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ uint32_t dpp_row_min(uint32_t m)
{
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xf, 0xf, false));
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xf, 0xf, false));
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xf, 0xf, false));
    m = min(m, (uint32_t)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x140, 0xf, 0xf, false));
    return m;
}
// number of set bits of a wave mask at lanes below this one
__device__ __forceinline__ uint32_t mask_count_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }

// One group of 64 consecutive stream entries of a column-panel layout (sorted by (panel, row slot, column)) added into the running
// sums `acc` (LDS, one wave's tile) in the order of the sequential reference loop.  Entries of one row inside one panel sit in
// neighbouring lanes and form a RUN: the run's first lane adds its followers' products in lane order and makes ONE LDS update; a row
// that comes back behind a panel boundary inside the same 64 entries is a second run, and the runs of different panels (PARTS) are
// applied one panel after the other.  The cross-lane work runs on the VALU / SALU (DPP wave shifts, wave masks): the LDS pipe sees
// one read and one write per run — it carried 21 instructions per 64 entries when the neighbours, the run lengths and the follower
// products travelled by ds_bpermute.  LDS accesses of one wave execute in issue order, so a row that comes back in a later part
// (or a later group) reads what the earlier one wrote without a wait in between.  All 64 lanes must be active.
// ROWBITS: row slots < 2^ROWBITS; panels < 2^(32 - ROWBITS).
template <int ROWBITS>
__device__ __forceinline__ void sl_ordered_accumulate(double *acc, uint32_t lane, uint32_t row, double prod, uint32_t pan)
{
    const uint32_t key = (pan << ROWBITS) | row;
    const uint32_t pkey = dpp_from_left(key);
    const unsigned long long same = __ballot(pkey == key) & ~1ull;                       // continues its left neighbour's run
    const unsigned long long cut = __ballot(((pkey ^ key) >> ROWBITS) != 0u) & ~1ull;   // first entry of a panel
    if (!(same | cut)) {                                 // 64 distinct rows of one panel
        acc[row] = DADD(acc[row], prod);
        return;
    }
    // lane l leads a run when it does not continue one; its followers sit in lanes l + 1, l + 2, ...
    const unsigned long long s1 = same >> 1, s2 = s1 & (same >> 2), s3 = s2 & (same >> 3), s4 = s3 & (same >> 4);
    if (s4) {
        // a run longer than four entries (a row with five entries inside one panel: rare for uniform columns, the rule for the
        // heavy rows of a graph): the follower products by general shuffles, as many steps as the longest run
        const unsigned long long lead = ~same;
        const unsigned long long above = lead & ~((2ull << lane) - 1ull);                 // run leaders to my right
        const uint32_t runlen = (above ? (uint32_t)__builtin_ctzll(above) : 64u) - lane;  // meaningful on leaders
        const bool leader = (lead >> lane) & 1ull;
        uint32_t maxrun = 1u;                                                            // longest run = longest stretch of set bits in `same`, + 1
        for (unsigned long long m = same; m; m &= m >> 1) ++maxrun;                       // scalar: one step per entry of the longest run
        const uint32_t partno = mask_count_below(cut) + (__builtin_amdgcn_inverse_ballot_w64(cut) ? 1u : 0u);
        const uint32_t nparts = (uint32_t)__popcll(cut) + 1u;
        for (uint32_t f = 0; f < nparts; ++f) {
            const bool mine = leader && partno == f;
            double sacc = mine ? DADD(acc[row], prod) : 0.0;
            for (uint32_t st = 1; st < maxrun; ++st) {                // wave-uniform trip count; the shuffle runs on all lanes
                const double qq = __shfl_down(prod, st);
                if (mine && runlen > st) sacc = DADD(sacc, qq);
            }
            if (mine) acc[row] = sacc;
            asm volatile("" ::: "memory");                           // program order between the parts
        }
        return;
    }
    const bool leader = !__builtin_amdgcn_inverse_ballot_w64(same);
    const bool f1 = __builtin_amdgcn_inverse_ballot_w64(s1);
    const double q1 = dpp_from_right(prod);
    double q2 = 0.0, q3 = 0.0;
    bool f2 = false, f3 = false;
    if (s2) {                                            // some row has three entries in this panel
        q2 = dpp_from_right(q1); f2 = __builtin_amdgcn_inverse_ballot_w64(s2);
        if (s3) { q3 = dpp_from_right(q2); f3 = __builtin_amdgcn_inverse_ballot_w64(s3); }
    }
    // one panel (part) after the other, and a part's runs COMPLETELY before the next part begins: the row a panel ends with may be
    // the row the next panel starts with
    if (!cut) {
        if (leader) {
            double sacc = DADD(acc[row], prod);
            if (f1) sacc = DADD(sacc, q1);
            if (f2) sacc = DADD(sacc, q2);
            if (f3) sacc = DADD(sacc, q3);
            acc[row] = sacc;
        }
    } else {
        const uint32_t partno = mask_count_below(cut) + (__builtin_amdgcn_inverse_ballot_w64(cut) ? 1u : 0u);
        const uint32_t nparts = (uint32_t)__popcll(cut) + 1u;
        for (uint32_t f = 0; f < nparts; ++f) {
            if (leader && partno == f) {
                double sacc = DADD(acc[row], prod);
                if (f1) sacc = DADD(sacc, q1);
                if (f2) sacc = DADD(sacc, q2);
                if (f3) sacc = DADD(sacc, q3);
                acc[row] = sacc;
            }
            asm volatile("" ::: "memory");               // program order between the parts
        }
    }
}


// The epilogue of a tile whose rows' running sums sit in LDS (acc[slot], slots < nslots): a lane owns slots lane, lane + 64, ...; row_of(slot)
// = the slot's local row, or ~0 for none.  The vectors of SL_PW_EPI_ROWS of a lane's rows are in flight together and every load is
// unconditional (a slot without a row reads row 0's operands and uses none; the push epilogue's optional threshold / column value are read
// from dinv's line again where the launch has none).  As one row per trip — loads, wait, arithmetic, stores, and `s_waitcnt vmcnt(0)` at
// the head of the next trip (the compiler's wait for values loaded before the loop, which then also waits for the trip's own stores) — a
// paced round's ~20 trips were 20 serialised memory round trips per wave at the very moment the block's 16 paced waves all stop
// streaming: round 3 measured 0.875 ms without the epilogue vectors against 0.944 with them, for 17 % of the bytes.  The rows are finished in
// slot order: the partial sums add the same terms in the same sequence as the one-row loop.
#ifndef SL_PW_EPI_ROWS
#define SL_PW_EPI_ROWS 4          // (a paced round has rpw / 64 ~ 20 slots per lane, a dynamic tile 32)
#endif
template <int EPI, bool NO_VECTORS, class ROWOF>
__device__ __forceinline__ void sl_tile_epilogue(const sl_row_args &a, const double *__restrict__ g, const double *acc, uint32_t lane, uint32_t nslots, ROWOF row_of,
                                                 double &part0, double &part1)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // once per tile, not once per trip
    for (uint32_t r0 = lane; r0 < nslots; r0 += 64u * SL_PW_EPI_ROWS) {
        bool ok[SL_PW_EPI_ROWS];
        uint64_t ri[SL_PW_EPI_ROWS];
        double e_t[SL_PW_EPI_ROWS], e_d[SL_PW_EPI_ROWS], e_x[SL_PW_EPI_ROWS], dself[SL_PW_EPI_ROWS];
        [[maybe_unused]] double e_th[SL_PW_EPI_ROWS], e_z[SL_PW_EPI_ROWS];      // push: the row's threshold / column value where the launch has them
        [[maybe_unused]] const double *thp = a.theta_rows ? a.theta_rows : a.dinv, *zp = a.zout ? a.zcol : a.dinv;     // (else dinv's line again: never a load under a branch)
#pragma unroll
        for (int u = 0; u < SL_PW_EPI_ROWS; ++u) {
            const uint32_t r = r0 + 64u * (uint32_t)u;
            const uint64_t i = r < nslots ? row_of(r) : ~0ull;
            ok[u] = i < a.n_rows;
            ri[u] = ok[u] ? i : 0;
            e_t[u] = 0.0; e_d[u] = 0.0; e_x[u] = 0.0; dself[u] = 0.0;
            if constexpr (!NO_VECTORS) {
                if constexpr (EPI == SL_EPI_NEUMANN) { e_t[u] = g[a.row_offset + ri[u]]; e_d[u] = a.dinv[ri[u]]; e_x[u] = a.x[ri[u]]; }
                else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t[u] = a.aux[ri[u]]; }
                else if constexpr (EPI == SL_EPI_PUSH) {
                    e_t[u] = a.r[ri[u]]; e_d[u] = a.dinv[ri[u]]; e_x[u] = a.x[ri[u]]; dself[u] = g[a.row_offset + ri[u]];
                    e_th[u] = thp[ri[u]]; e_z[u] = zp[ri[u]];
                }
            }
        }
        if (a.n_long) {                                           // hub rows belong to the long-row kernel
            uint32_t rl[SL_PW_EPI_ROWS];
#pragma unroll
            for (int u = 0; u < SL_PW_EPI_ROWS; ++u) rl[u] = a.row_len[ri[u]];
#pragma unroll
            for (int u = 0; u < SL_PW_EPI_ROWS; ++u) ok[u] = ok[u] && rl[u] != SL_LONG_SENTINEL;
        }
#pragma unroll
        for (int u = 0; u < SL_PW_EPI_ROWS; ++u) {
            if (!ok[u]) continue;
            const uint32_t r = r0 + 64u * (uint32_t)u;
            if constexpr (NO_VECTORS) part0 += acc[r];
            else if constexpr (EPI == SL_EPI_PUSH)
                sl_push_epilogue(a, ri[u], acc[r], e_t[u], e_d[u], e_x[u], dself[u], [&] { return a.theta_rows ? e_th[u] : a.theta; }, [&] { return e_z[u]; }, part0, part1);
            else sl_row_epilogue<EPI>(a, ri[u], acc[r], e_t[u], e_d[u], e_x[u], dself[u], part0, part1);
        }
    }
}

// ---- column-panel kernel: gathers served by the L2 -----------------------------------------------------
// For matrices whose columns are spread over a vector far larger than the L2 (uniformly random columns — the reference
// generators' recipe): every gather of the general kernel misses L2, and misses are served at 58 G/s whatever the table size
// (2.7 ms for 1.6e8 of them; gathers that hit L2 run at 265 G/s).  Here the entries of a tile of SL_PANEL_TILE rows are one
// stream sorted by (panel of 2^16 columns, row, column); a wave owns a tile, keeps the running sum of each of its rows in
// LDS and walks the stream: all waves pass the panels in the same order at about the same pace, so the half megabyte of the
// vector they gather from stays in L2.  Within a row the products still arrive in ascending column order and are added one
// by one (entries of one row that sit in neighbouring lanes of a chunk are applied in lane order), so the sum has the bits
// of the sequential reference loop.  CSR order only; the 4-lane order keeps the general kernel.
template <int EPI>
__global__ __launch_bounds__(SL_PANEL_WAVES * 64) void sl_panel_kernel(sl_row_args a, uint32_t block0)
{
    extern __shared__ __attribute__((aligned(16))) double pan_acc[];
    __shared__ double red[2 * SL_PANEL_WAVES];
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;           // speculative solve loop: the stop rule already fired
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lb = block0 + blockIdx.x;
    const uint32_t tile = lb * SL_PANEL_WAVES + wave;
    double *acc = pan_acc + (size_t)wave * (SL_PANEL_TILE + 64);    // + the slot padding entries add their zeros to
    double part0 = 0.0, part1 = 0.0;
    if (tile < a.n_pan_tiles) {
        for (uint32_t r = lane; r < SL_PANEL_TILE + 64; r += 64) acc[r] = 0.0;
        const double *__restrict__ g = a.gather;
        const uint32_t s = a.pan_tile_ptr[tile], e = a.pan_tile_ptr[tile + 1];
        constexpr int U = SL_PANEL_CHUNK / 64;
        for (uint32_t c0 = s; c0 < e; c0 += SL_PANEL_CHUNK) {
            uint32_t rl[U], cl[U];
            double v[U], tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t k = c0 + (uint32_t)u * 64u + lane;
                rl[u] = __builtin_nontemporal_load(&a.pan_row[k]);
                cl[u] = __builtin_nontemporal_load(&a.pan_col[k]);
                v[u] = __builtin_nontemporal_load(&a.pan_val[k]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) tv[u] = g[cl[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t row = rl[u];
                const double prod = DMUL(v[u], tv[u]);
                // 64 lanes = 64 consecutive entries of the stream (row slots < 2048 + 64: 12 bits)
                sl_ordered_accumulate<12>(acc, lane, row, prod, cl[u] >> SL_PANEL_COL_BITS);
            }
        }
        sl_tile_epilogue<EPI, false>(a, g, acc, lane, SL_PANEL_TILE, [&](uint32_t r) -> uint64_t { return (uint64_t)tile * SL_PANEL_TILE + r; }, part0, part1);
    }
    sl_block_partials<EPI, SL_PANEL_WAVES>(a, red, lane, wave, lb, a.part_stride, part0, part1);
}

// ---- paced column-panel kernel: persistent blocks, gathers served by the L2 ------------------------------------------------
// Layout: sl_internal.hpp (sl_matrix::d_pw_*).  ONE 16-wave block per CU; wave w of block b owns, in round r, tile
// (r * blocks + b) * 16 + w: pw_rpw rows whose running sums live in LDS (the CU's whole LDS = its 16 tiles), and walks the tile's
// stream — sorted by (panel of 2^16 columns, row, column) — 256 entries at a time: stream loads two chunks ahead, gathers one
// chunk ahead of the LDS updates (three stream buffers that change ROLES; a copy would wait for everything in flight).
// What makes the gathers L2 hits is that the waves of an XCD sit on the same few panels (512 KiB of the vector each) at the same
// time.  Nothing but equal work kept round 1's kernel there (L2 hit rate 64 %, 1.39 ms at n = 10^7 x 16 where perfect locality
// gives 0.95).  Here the 16 waves of a block PACE each other through progress words in LDS: a wave about to gather from a panel
// more than one ahead of the slowest wave of its block sleeps.  The pace is a hint with a bounded wait that switches itself off,
// never a correctness condition; blocks of an XCD start together and carry equal work (the layout is only built for balanced
// matrices), which keeps them within the L2's reach of each other (tools/panel2_bench.hip: free-running 1.95-2.2 ms, paced inside
// the block 1.16 ms, pacing across the XCD through an L2-resident line of progress words as well: no further gain).
// Summation order: within a row the products arrive in ascending column order (panel order = column order) and are added one by
// one (sl_ordered_accumulate above).  Bits = the sequential reference loop (sparse.rs:187-203).  CSR order only; the 4-lane order
// keeps the general kernel.
// Two forms of the layout share the kernel (launch parameters pw_deal / pw_pbits): uniform columns — panels of 2^16 columns, row
// groups dealt among ALL tiles, gathers served by the L2 as described; wide bands — panels of 2^9..2^10 columns, row groups dealt
// among the 16 tiles of ONE block, so that the CU's waves gather from the same few KB of vector: hits in its own L1.
#ifndef SL_PW_SLEEP
#define SL_PW_SLEEP 4            // s_sleep argument of a paced wave that waits (x 64 cycles)
#endif
// ANY_ORDER / PWV: measurement builds only (SL_PW_ANY, SL_PW_VAR): relaxed accumulation inside this layout; PWV & 8 = no epilogue
// traffic, PWV & 32 = gathers folded into the first 2 MB of the vector (every gather an L2 hit, no first touches), PWV & 2 = no LDS update
// IDX (round 4): the index-only form for column-constant operators (sl_matrix::d_colval) — only the index words of the stream are
// loaded (4 of its 12 bytes per entry); an off-diagonal entry's product is gathered ready-made from a.zgather (= colval_u * gather_u,
// rounded once: the same double DMUL(value, gather) gives for every entry of column u), the diagonal entry's — value exactly 1 — is the
// gathered vector's own entry; which of the two an entry is follows from its row slot (the row's global index against its column).
template <int EPI, bool ANY_ORDER = false, int PWV = 0, bool IDX = false>
__global__ __launch_bounds__(SL_PW_WAVES * 64) void sl_pw_kernel(sl_row_args a)
{
    extern __shared__ __attribute__((aligned(16))) double pw_acc[];
    __shared__ double red[2 * SL_PW_WAVES];
    __shared__ uint32_t prog[SL_PW_WAVES + 1];                     // [16] = pacing alive
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;           // speculative solve loop: the stop rule already fired
    // the wave index as a SCALAR: tile, chunk counts, loop control and the stream's base addresses then live in SGPRs (the compiler
    // cannot see that threadIdx.x >> 6 is the same in all lanes and would keep all of them — and their arithmetic — in vector registers)
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rpw = a.pw_rpw, slack = a.pw_slack, pbits = a.pw_pbits, deal = a.pw_deal;
    double *acc = pw_acc + (size_t)wave * (rpw + 1);                // + the spare slot padding entries add their zeros to
    // progress words: relaxed workgroup-scope atomics = plain ds_read / ds_write.  NOT volatile: a volatile access makes the backend
    // wait for every load in flight (s_waitcnt vmcnt(0)), which would serialise the stream loads and gathers the pipeline keeps ahead
    auto prog_ld = [&](uint32_t k) { return __hip_atomic_load(&prog[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto prog_st = [&](uint32_t k, uint32_t v) { __hip_atomic_store(&prog[k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (threadIdx.x <= SL_PW_WAVES) prog[threadIdx.x] = threadIdx.x == SL_PW_WAVES ? 1u : 0u;
    __syncthreads();
    const double *__restrict__ g = a.gather;
    const uint32_t nblocks = gridDim.x;
    // XCD-local spans: the hardware deals workgroups to the XCDs round robin (block b on XCD b % G), the layout deals a span of rows among
    // nblocks / G consecutive LOGICAL blocks — so blocks b, b + G, b + 2 G, ... become logical neighbours and share their L2's panels
    const uint32_t lblock = a.pw_xcd ? (blockIdx.x % a.pw_xcd) * (nblocks / a.pw_xcd) + blockIdx.x / a.pw_xcd : blockIdx.x;
    const uint32_t rounds = (a.pw_tiles + nblocks * SL_PW_WAVES - 1) / (nblocks * SL_PW_WAVES);
    // a range of ROUNDS (blk_cnt != 0; layouts with edge-first rounds): the partitioned step runs the edge rounds first, the exchange beside the rest
    const uint32_t round_lo = a.blk_cnt ? a.blk_lo : 0u, round_hi = a.blk_cnt ? min(rounds, a.blk_lo + a.blk_cnt) : rounds;
    double part0 = 0.0, part1 = 0.0;
    for (uint32_t round = round_lo; round < round_hi; ++round) {
        const uint32_t tile = (round * nblocks + lblock) * SL_PW_WAVES + wave;
        if (tile >= a.pw_tiles) { if (lane == 0) prog_st(wave, 0xffffffffu); continue; }     // nothing to wait for
        for (uint32_t r = lane; r <= rpw; r += 64) acc[r] = 0.0;
        const uint32_t ch0 = a.pw_tile_ptr[tile], chunks = a.pw_tile_ptr[tile + 1] - ch0;
        // groups of SL_PW_GROUP rows are dealt among `deal` tiles: a span owns the row groups [g0, g0 + gcnt) — deal * rpw consecutive rows
        // in row order, or what the span table says (XCD-local spans, edge-first rounds)
        const uint32_t ps = tile / deal, tdeal = tile % deal;
        // (scalars: the tile is the wave's; left in VGPRs the compiler waits for these loads at their first use INSIDE the epilogue loop, which
        // then waits for every store of the trip before as well)
        const uint64_t g0v = a.pw_span_tab ? a.pw_span_tab[2 * ps] : (uint64_t)ps * deal * (rpw / SL_PW_GROUP);
        const uint64_t g0 = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(g0v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)g0v);
        const uint32_t gcnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a.pw_span_tab ? a.pw_span_tab[2 * ps + 1] : deal * (rpw / SL_PW_GROUP)));
        const u32x4 *__restrict__ idxq = reinterpret_cast<const u32x4 *>(a.pw_idx) + (uint64_t)ch0 * 64;
        const f64x2 *__restrict__ valq = reinterpret_cast<const f64x2 *>(a.pw_val) + (uint64_t)ch0 * 128;
        uint32_t sp_g = 0;                                          // super-panel at the gather stage (wave-uniform)
        uint32_t SI[3][4], GC[3][4];
        double SV[3][4], GG[3][4];
        auto load_stream = [&](uint32_t ch, uint32_t (&ii)[4], double (&vv)[4]) {
            const u32x4 q = __builtin_nontemporal_load(idxq + (uint64_t)ch * 64 + lane);
            ii[0] = q.x; ii[1] = q.y; ii[2] = q.z; ii[3] = q.w;
            if constexpr (IDX) { vv[0] = vv[1] = vv[2] = vv[3] = 1.0; return; }        // (never multiplied: the products come ready-made)
            const f64x2 va = __builtin_nontemporal_load(valq + (uint64_t)ch * 128 + lane);
            const f64x2 vb = __builtin_nontemporal_load(valq + (uint64_t)ch * 128 + 64 + lane);
            vv[0] = va.x; vv[1] = va.y; vv[2] = vb.x; vv[3] = vb.y;
        };
        auto pace = [&](uint32_t pan) {
            const uint32_t me = (round << 20) + pan + 1u;           // monotone over the launch (panels < 2^16)
            if (lane == 0) prog_st(wave, me);
            if (slack >= (1u << 20) || !prog_ld(SL_PW_WAVES)) return;
            for (uint32_t spins = 0;; ++spins) {
                uint32_t m = lane < SL_PW_WAVES ? prog_ld(lane) : 0xffffffffu;
                m = __builtin_amdgcn_readfirstlane(dpp_row_min(m));   // the 16 progress words sit in lanes 0..15 = one DPP row
                if (me <= m + slack) break;                         // at most `slack` panels ahead of the block's slowest wave
                if (spins > 2048u) { if (lane == 0) prog_st(SL_PW_WAVES, 0u); break; }     // the hint switches itself off
                __builtin_amdgcn_s_sleep(SL_PW_SLEEP);
            }
        };
        // real = false: the pipeline's filler behind the last chunk (the loads are issued all the same, so that the number of loads
        // in flight at every wait is a compile-time constant; a load under a branch would turn every wait into vmcnt(0))
        auto gather = [&](const uint32_t (&ii)[4], double (&gg)[4], uint32_t (&cc)[4], bool real) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t stepbit = (ii[u] >> SL_PW_SP_BITS) & 1u;
                const unsigned long long fl = __ballot(stepbit);
                uint32_t sp = sp_g;
                if (fl) sp += mask_count_below(fl) + stepbit;           // a handful of times per tile: the first entry of a super-panel
                sp_g = __builtin_amdgcn_readfirstlane(sp_g + (real ? (uint32_t)__popcll(fl) : 0u));
                cc[u] = real ? ((sp << SL_PW_SP_BITS) | (ii[u] & ((1u << SL_PW_SP_BITS) - 1u))) : 0u;
                if constexpr (IDX) {
                    // the row this entry belongs to (slot -> group of the span -> global row); its diagonal entry is the one at its own column
                    const uint32_t r = ii[u] >> SL_PW_ROW_SHIFT;
                    const uint64_t irow = (g0 + (uint64_t)(r / SL_PW_GROUP) * deal + tdeal) * SL_PW_GROUP + (r % SL_PW_GROUP);
                    const double *__restrict__ base = ((uint64_t)cc[u] == a.row_offset + irow && r < rpw) ? g : a.zgather;
                    gg[u] = base[cc[u]];
                } else
                gg[u] = (PWV & 32) ? g[cc[u] & 0x3ffffu] : g[cc[u]];
            }
        };
        auto accumulate = [&](const uint32_t (&ii)[4], const double (&vv)[4], const double (&gg)[4], const uint32_t (&cc)[4]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {                            // row slots < 2^SL_PW_ROW_BITS (SL_PW_MAX_ROWS + the spare slot)
                if constexpr (PWV & 2) { if (DMUL(vv[u], gg[u]) == 123.456) acc[0] = 1.0; }
                else if constexpr (ANY_ORDER) (void)__hip_atomic_fetch_add(&acc[ii[u] >> SL_PW_ROW_SHIFT], DMUL(vv[u], gg[u]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else sl_ordered_accumulate<SL_PW_ROW_BITS>(acc, lane, ii[u] >> SL_PW_ROW_SHIFT, IDX ? gg[u] : DMUL(vv[u], gg[u]), cc[u] >> pbits);
            }
        };
        if (chunks) {
            const uint32_t lastc = chunks - 1u;
            load_stream(0, SI[0], SV[0]);
            load_stream(min(1u, lastc), SI[1], SV[1]);
            pace(0);
            gather(SI[0], GG[0], GC[0], true);
#define SL_PW_STEP(r, r1, r2)                                                                                        \
            load_stream(min(ch + 2u, lastc), SI[r2], SV[r2]);                                                        \
            {                                                                                                        \
                const bool real1 = ch + 1u < chunks;                                                                 \
                if (real1) {                                                                                         \
                    const uint32_t nextcol = (sp_g << SL_PW_SP_BITS) | (__builtin_amdgcn_readfirstlane(SI[r1][0]) & ((1u << SL_PW_SP_BITS) - 1u)); \
                    pace(nextcol >> pbits);                                                              \
                }                                                                                                    \
                gather(SI[r1], GG[r1], GC[r1], real1);                                                               \
            }                                                                                                        \
            accumulate(SI[r], SV[r], GG[r], GC[r]);                                                                  \
            if (++ch >= chunks) break;
            for (uint32_t ch = 0;;) {
                SL_PW_STEP(0, 1, 2)
                SL_PW_STEP(1, 2, 0)
                SL_PW_STEP(2, 0, 1)
            }
#undef SL_PW_STEP
        }
        if (lane == 0) prog_st(wave, (round + 1u) << 20);            // as far along as the round's end while the vectors are written
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // the round's epilogue: slot r = group r / 16 of the tile, row r % 16 of the group
        sl_tile_epilogue<EPI, (PWV & 8) != 0>(a, g, acc, lane, rpw, [&](uint32_t r) -> uint64_t {
            const uint32_t gl = (r / SL_PW_GROUP) * deal + tdeal;
            return gl < gcnt ? (g0 + gl) * SL_PW_GROUP + (r % SL_PW_GROUP) : ~0ull;          // (gl >= gcnt: a span shorter than its tiles' slots)
        }, part0, part1);
    }
    // (a range of rounds that does not start at round 0 = the second of two launches: its own set of partial sums behind the first's)
    sl_block_partials<EPI, SL_PW_WAVES>(a, red, lane, wave, blockIdx.x + (round_lo ? nblocks : 0u), a.part_stride, part0, part1);
}


// ---- order-free column stream: persistent blocks, a CU's entries sorted by column, sums by LDS atomics (SL_ORDER_ANY) --------------
// Layout: sl_internal.hpp (sl_matrix::d_pwr_*).  ONE 16-wave block per CU; in round r block b owns block tile r * blocks + b: pwr_rpb
// rows whose running sums fill the CU's LDS, and ONE stream of their off-diagonal entries sorted by column, which the 16 waves take
// chunk by chunk in turn (wave w: chunks w, w + 16, ...).  What the ordered kernel above pays for the reference's summation order —
// runs detected per 64 entries, a stream per wave sorted by (panel, row), waves pacing each other — falls away: the caller asked for
// any order (results to rounding, BASELINE north_star's 1e-10), so a product goes to its row by ds_add_f64 whenever it arrives and the
// 64 gathers of a wave-instruction walk ascending lines of the vector (a CU-wide sort has ~0.5 entries per 128-byte line at n = 10^7).
// MEASURED (round 3, profiles/r03_order_any.txt; n = 10^7 x 16, uniform columns): this kernel is NOT faster than the ordered one —
// 1.00-1.09 ms against 0.95 ms.  The accumulation was never what bounded the step: with the LDS update removed altogether the launch
// takes the same time (1.04-1.09 ms), with plain read-add-write instead of atomics 0.98-1.08; gathers + stream alone, no sums and no
// epilogue, take 0.77-1.00 ms.  The sorted gathers do coalesce (L1 -> L2 read requests 1.40 * 10^8 against 1.63 * 10^8), but the L2
// misses go UP (4.0-4.2 * 10^7 against 2.79 * 10^7, memory-side reads 5.1 against 3.5 GB): every block of an XCD meets a line of the
// vector within the same few microseconds, and what the panels of the ordered layout buy — an XCD dwelling on 512 KB for the time of a
// panel, every line touched ~16 times at leisure — is lost.  Pacing the blocks of an XCD against each other through a line of progress
// words changed nothing about the misses (and cost a factor of five through the polling), 256-entry chunks (1.00 ms) beat 64-entry
// chunks (1.08 ms) by their cheaper stream loads only, touching the lines ahead with return-less atomics made it 1.35 ms.  Kept as the
// tested, opt-in implementation of SL_ORDER_ANY (hub rows need no kernel of their own here), not selected by default anywhere.
// Pipeline per wave: stream loads three chunks ahead, gathers one chunk ahead of the LDS updates, buffers that change roles.
// VAR (measurement builds only, SL_PWR_VAR): 1 = plain read-add-write instead of the LDS atomic (WRONG sums: timing of the atomics),
// 2 = no LDS update, 4 = no gathers, 8 = no epilogue traffic, 32 = gathers folded into the first 2 MB of the vector (no first touches).  (Tried and dropped: the XCD's waves touching the vector lines of the chunk
// positions ahead with return-less atomic-or 0, so that the L2 fetches them without a read slot of an L1 held: 0.99 -> 1.35 ms.)
template <int EPI, int VAR = 0>
__global__ __launch_bounds__(SL_PW_WAVES * 64) void sl_pwr_kernel(sl_row_args a)
{
    extern __shared__ __attribute__((aligned(16))) double pwr_acc[];
    __shared__ double red[2 * SL_PW_WAVES];
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;           // speculative solve loop: the stop rule already fired
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rpb = a.pwr_rpb, nblocks = gridDim.x, ntiles = a.pwr_tiles;
    const double *__restrict__ g = a.gather;
    const uint32_t rounds = (ntiles + nblocks - 1) / nblocks;
    for (uint32_t r = threadIdx.x; r <= rpb; r += SL_PW_WAVES * 64) pwr_acc[r] = 0.0;      // rpb rows + the spare slot of the padding entries
    __syncthreads();
    double part0 = 0.0, part1 = 0.0;
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t tile = round * nblocks + blockIdx.x;
        if (tile >= ntiles) break;                                  // block-uniform
        const uint32_t ch0 = a.pwr_tile_ptr[tile], chunks = a.pwr_tile_ptr[tile + 1] - ch0;
        const uint32_t *__restrict__ idxq = a.pwr_idx + (uint64_t)ch0 * SL_PWR_CHUNK;
        const double *__restrict__ valq = a.pwr_val + (uint64_t)ch0 * SL_PWR_CHUNK;
        const uint32_t *__restrict__ baseq = a.pwr_base + ch0;
        // this wave's chunks: wave, wave + 16, ... ; mine = how many
        const uint32_t mine = chunks > wave ? (chunks - wave + SL_PW_WAVES - 1) / SL_PW_WAVES : 0u;
        uint32_t SI[4], SB[4];
        double SV[4], GG[2];
        uint32_t vzero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));              // a zero the compiler cannot fold: keeps the base-column load on the vector memory path
        auto load_stream = [&](uint32_t j, uint32_t &ii, double &vv, uint32_t &bb) {
            const uint32_t ch = wave + j * SL_PW_WAVES;
            ii = __builtin_nontemporal_load(idxq + (uint64_t)ch * SL_PWR_CHUNK + lane);
            vv = __builtin_nontemporal_load(valq + (uint64_t)ch * SL_PWR_CHUNK + lane);
            bb = __builtin_nontemporal_load(baseq + ch + vzero);     // a VECTOR load of one address (one request): a scalar load would share its counter with the LDS atomics
        };
        auto gather = [&](uint32_t ii, double &gg, uint32_t bb) {
            asm volatile("" : "+v"(ii), "+v"(bb));                   // the address is formed HERE (hoisted to the top of the loop it would wait for the newest stream loads)
            gg = (VAR & 4) ? 1.0 : (VAR & 32) ? g[(bb + (ii & ((1u << SL_PWR_OFF_BITS) - 1u))) & 0x3ffffu] : g[bb + (ii & ((1u << SL_PWR_OFF_BITS) - 1u))];
        };
        auto accumulate = [&](uint32_t ii, double vv, double gg, bool real) {
            asm volatile("" : "+v"(gg));                             // the product is formed HERE, behind the loads issued above: hoisted over them it would wait for the newest gather
            const double prod = real ? DMUL(vv, gg) : 0.0;
            if constexpr (VAR & 2) { if (prod == 123.456) pwr_acc[0] = prod; }
            else if constexpr (VAR & 1) { double *q = &pwr_acc[ii >> SL_PWR_OFF_BITS]; *q = DADD(*q, prod); }
            else (void)__hip_atomic_fetch_add(&pwr_acc[ii >> SL_PWR_OFF_BITS], prod, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (mine) {
            const uint32_t last = mine - 1u;
            load_stream(0, SI[0], SV[0], SB[0]);
            load_stream(min(1u, last), SI[1], SV[1], SB[1]);
            load_stream(min(2u, last), SI[2], SV[2], SB[2]);
            asm volatile("" ::: "memory");
            gather(SI[0], GG[0], SB[0]);
            asm volatile("" ::: "memory");
            // Stream loads three chunks ahead, gathers one chunk ahead of the LDS update.  The steps behind the last chunk are fillers:
            // they repeat the last chunk's loads (the number of loads in flight at every wait is then a compile-time constant) and add
            // zeros; the loop runs whole quadruples — one back edge, no exit in the middle — so that the buffers change ROLES instead of
            // being copied (a copy would wait for everything in flight).  The empty asm statements keep the compiler from sinking the
            // loads of a stage towards their uses: issue order = program order.
#define SL_PWR_STEP(r, r1, r3, gq, gq1, jj)                                                                          \
            load_stream(min((jj) + 3u, last), SI[r3], SV[r3], SB[r3]);                                               \
            asm volatile("" ::: "memory");                                                                           \
            gather(SI[r1], GG[gq1], SB[r1]);                                                                         \
            asm volatile("" ::: "memory");                                                                           \
            accumulate(SI[r], SV[r], GG[gq], (jj) < mine);                                                           \
            asm volatile("" ::: "memory");
            for (uint32_t j = 0; j < mine; j += 4u) {
                SL_PWR_STEP(0, 1, 3, 0, 1, j)
                SL_PWR_STEP(1, 2, 0, 1, 0, j + 1u)
                SL_PWR_STEP(2, 3, 1, 0, 1, j + 2u)
                SL_PWR_STEP(3, 0, 2, 1, 0, j + 3u)
            }
#undef SL_PWR_STEP
        }
        __syncthreads();                                             // every wave's sums are in
        // epilogue: four rows per thread at a time, their sixteen loads in flight together (one row at a time, 20 dependent round trips
        // per thread and round stood beside the stream phase instead of under it: 0.23 ms of the launch at n = 10^7)
        constexpr uint32_t NT = SL_PW_WAVES * 64;
        for (uint32_t r0 = threadIdx.x; r0 < rpb; r0 += 4u * NT) {   // slot r = group r / 16 of the block tile, row r % 16 of the group
            uint64_t ri[4];
            bool live[4];
            double own[4], dg[4], e_t[4], e_d[4], e_x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r = r0 + (uint32_t)q * NT;
                const uint64_t i = ((uint64_t)(r / SL_PW_GROUP) * ntiles + tile) * SL_PW_GROUP + (r % SL_PW_GROUP);
                live[q] = r < rpb && i < a.n_rows;
                ri[q] = live[q] ? i : 0;                             // dead slots load row 0 (no branch around the loads)
                if constexpr (VAR & 8) continue;
                own[q] = g[a.row_offset + ri[q]];
                dg[q] = a.pwr_diag[ri[q]];
                e_t[q] = 0.0; e_d[q] = 0.0; e_x[q] = 0.0;
                if constexpr (EPI == SL_EPI_NEUMANN) { e_d[q] = a.dinv[ri[q]]; e_x[q] = a.x[ri[q]]; }
                else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t[q] = a.aux[ri[q]]; }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r = r0 + (uint32_t)q * NT;
                if (!live[q]) continue;
                if constexpr (VAR & 8) { part0 += pwr_acc[r]; pwr_acc[r] = 0.0; continue; }
                const double sum = DADD(pwr_acc[r], DMUL(dg[q], own[q]));
                pwr_acc[r] = 0.0;                                    // ready for the next round
                if constexpr (EPI == SL_EPI_NEUMANN) e_t[q] = own[q];
                sl_row_epilogue<EPI>(a, ri[q], sum, e_t[q], e_d[q], e_x[q], 0.0, part0, part1);
            }
        }
        __syncthreads();
    }
    sl_block_partials<EPI, SL_PW_WAVES>(a, red, lane, wave, blockIdx.x, a.part_stride, part0, part1);
}

// ---- long rows: one wave per row ---------------------------------------------------------------------
// Rows with more than the matrix's long_row entries (hubs of power-law graphs; 3.5 * 10^5 of them in the transposed PageRank
// graph at n = 10^7, 27 to 31 000 entries each).  The 64 lanes fetch the raw CSR entries coalesced and form the products in
// parallel; the additions stay sequential in the reference's order (lane 0 walks the products through the wave's LDS line), so
// the result is bit-identical to the slice path.  One WAVE per row, four rows per block, no block barrier: most long rows are one
// or two batches of 64 entries, i.e. a chain of load latencies (row pointers -> entries -> gather -> 64 adds -> store) — what
// counts is how many rows are in flight per CU (round 2: a 256-thread block per row took 1.27 ms per dense PageRank round, more
// than the panel kernel beside it).  Longer rows keep the next batch's loads in flight under the add chain of the current one.
template <int ORDER, int EPI, bool SEED = false>
__global__ __launch_bounds__(SL_BLOCK) void sl_long_rows_kernel(sl_row_args a, uint32_t slot0)
{
    __shared__ double prod_lds[SL_BLOCK];
    if (a.ctl && a.gate_it > a.ctl->stop_after) return;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t li = blockIdx.x * (SL_BLOCK / 64) + wave;           // position in the list of long rows = partial slot
    if (li >= a.n_long) return;                                       // whole waves; no block barrier below
    double *prod = prod_lds + wave * 64;
    const uint32_t i = a.long_rows[li];
    const double *__restrict__ g = a.gather;
    const uint32_t s = a.csr_ptr[i], e = a.csr_ptr[i + 1], len = e - s;
    // epilogue operands up front (lane 0): their latency hides under the row
    double e_t = 0.0, e_d = 0.0, e_x = 0.0, dself = 0.0;
    if (lane == 0) {
        if constexpr (EPI == SL_EPI_NEUMANN) { e_t = g[a.row_offset + i]; e_d = a.dinv[i]; e_x = a.x[i]; }
        else if constexpr (EPI == SL_EPI_RESIDUAL) { e_t = a.aux[i]; }
        else if constexpr (EPI == SL_EPI_PUSH) { e_t = a.r[i]; e_d = a.dinv[i]; e_x = a.x[i]; dself = g[a.row_offset + i]; }
    }
    const uint32_t chunks4 = (ORDER == 1) ? ((len >> 2) << 2) : 0u;     // entries covered by full simd chunks (len >= 8 here)
    double sum = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
    if constexpr (SEED) sum = a.out[i];                                 // y += A x: the chain starts from y_i (sparse.rs:192-203)
    bool merged = false;
    // pipeline: entries (value, column) one batch ahead of their gather, the gather one batch ahead of the add chain
    uint32_t k = s + lane;
    double v_cur = k < e ? a.csr_val[k] : 0.0;
    uint32_t c_cur = k < e ? a.csr_idx[k] : (uint32_t)(a.row_offset + i);
    double g_cur = g[c_cur];
    k += 64;
    double v_nxt = k < e ? a.csr_val[k] : 0.0;
    uint32_t c_nxt = k < e ? a.csr_idx[k] : (uint32_t)(a.row_offset + i);
    for (uint32_t base = s; base < e; base += 64) {
        const double g_nxt = g[c_nxt];                                  // gather of the next batch: in flight under this batch's chain
        const uint32_t k2 = base + 128 + lane;
        const double v_n2 = k2 < e ? a.csr_val[k2] : 0.0;               // entries of the batch after it
        const uint32_t c_n2 = k2 < e ? a.csr_idx[k2] : (uint32_t)(a.row_offset + i);
        prod[lane] = DMUL(v_cur, g_cur);                                // padding lanes: 0 * (own entry) — never added
        __builtin_amdgcn_wave_barrier();                                // same wave: LDS accesses execute in issue order
        if (lane == 0) {
            const uint32_t cnt = (e - base) < 64u ? (e - base) : 64u;
            if constexpr (ORDER == 0) {
#pragma unroll 8
                for (uint32_t j = 0; j < cnt; ++j) sum = DADD(sum, prod[j]);
            } else {
                for (uint32_t j = 0; j < cnt; ++j) {
                    const uint32_t q = base - s + j;
                    const double p = prod[j];
                    if (q < chunks4) {
                        const uint32_t ln = q & 3u;
                        if (ln == 0) l0 = DADD(l0, p); else if (ln == 1) l1 = DADD(l1, p); else if (ln == 2) l2 = DADD(l2, p); else l3 = DADD(l3, p);
                    } else {
                        if (!merged) { sum = DADD(DADD(DADD(l0, l1), l2), l3); merged = true; }
                        sum = DADD(sum, p);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        v_cur = v_nxt; g_cur = g_nxt;
        v_nxt = v_n2; c_nxt = c_n2;
    }
    if (lane == 0) {
        if constexpr (ORDER == 1) { if (!merged) sum = DADD(DADD(DADD(l0, l1), l2), l3); }
        double part0 = 0.0, part1 = 0.0;
        sl_row_epilogue<EPI>(a, i, sum, e_t, e_d, e_x, dself, part0, part1);
        if constexpr (EPI != SL_EPI_SPMV) {
            a.partials[slot0 + li] = part0;
            if constexpr (EPI == SL_EPI_PUSH) a.partials[(uint64_t)a.part_stride + slot0 + li] = part1;
        }
    }
}

// very many partials (one per long row of a 10^7-node graph: ~10^6): a first stage of SL_PRE_BLOCKS blocks brings each set down to
// SL_PRE_BLOCKS values (thread t of block b sums partials b*256+t, +SL_PRE_BLOCKS*256, ... in order; fixed butterfly), the
// one-block kernels below finish.  Same mapping every time: deterministic.
#define SL_PRE_BLOCKS 128
#define SL_PRE_THRESHOLD 65536u
__global__ __launch_bounds__(256) void sl_pre_reduce_kernel(const double *partials, uint32_t nparts, double *out, const sl_solve_ctl *ctl, uint32_t gate_it)
{
    __shared__ double red[4];
    if (ctl && gate_it > ctl->stop_after) return;
    const double *p = partials + (uint64_t)blockIdx.y * nparts;
    double acc = 0.0;
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < nparts; j += SL_PRE_BLOCKS * 256) acc += p[j];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.y * SL_PRE_BLOCKS + blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
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

// the same reduction inside a speculatively enqueued solve loop: gated, logs the sum, applies the stop rule.
// The thresholds are squared norms prepared by the host so that (sum < thr) == (sqrt(sum) < tolerance) exactly.
__global__ __launch_bounds__(1024) void sl_judge_reduce_kernel(const double *partials, uint32_t nparts, double *result,
                                                               sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot, int mode, double thr)
{
    __shared__ double red[16];
    if (gate_it > ctl->stop_after) return;
    double acc = 0.0;
    for (uint32_t j = threadIdx.x; j < nparts; j += 1024) acc += partials[j];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = red[0];
        for (int w = 1; w < 16; ++w) t += red[w];
        if (result) result[0] = t;
        if (mode == SL_JUDGE_LOCAL) return;                 // partitioned solve: the sum over all ranks is logged and judged by sl_comm_ticket_kernel
        ctl->log[slot] = t;
        ctl->n_done = slot + 1;
        bool stop = false;
        if (mode == SL_JUDGE_LT) stop = t < thr;
        else if (mode == SL_JUDGE_LE_OR_NONFINITE) stop = (t <= thr) || (t != t) || (fabs(t) == INFINITY);
        if (stop && gate_it < ctl->stop_after) ctl->stop_after = gate_it;
    }
}

__global__ void sl_ctl_reset_kernel(sl_solve_ctl *ctl)
{
    ctl->stop_after = 0xffffffffu;
    ctl->n_done = 0;
}

sl_status sl_launch_ctl_reset(sl_solve_ctl *ctl, hipStream_t s)
{
    hipLaunchKernelGGL(sl_ctl_reset_kernel, dim3(1), dim3(1), 0, s, ctl);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

uint32_t sl_row_grid(uint64_t n_slices)
{
    uint64_t nb = (n_slices + SL_WAVES_PER_BLOCK - 1) / SL_WAVES_PER_BLOCK;
    uint64_t nb8 = (nb + 7) / 8;
    if (nb8 == 0) nb8 = 1;
    return (uint32_t)(nb8 * 8);
}

// band-kernel geometry for half bandwidth w: slices per wave, dynamic LDS bytes, pipelining; spw = 0: not eligible
#define SL_BAND_MAX_LDS (80u * 1024u)      // two blocks per CU (160 KiB LDS)
#define SL_BAND_MAX_LDS_ONE (158u * 1024u)  // one 16-wave block per CU: windows up to w ~ 9500 (leaves room for the static arrays)
struct band_geom { uint32_t spw, lds, nw; bool pipe, c16; };
// matrices with at least this many slices run their long-row kernel beside the slice kernel (env SL_LONG_ROWS_BESIDE_MIN; tests use 0)
// Environment knobs of this file (A/B measurements and tests): read ONCE per process, in a thread-safe static
// initialisation — the library documents itself re-entrant (include/sublinear_hip.h), so nothing here is lazily written.
struct sl_kernel_knobs {
    uint64_t long_rows_beside_min = 16384;   // matrices with at least this many slices run their long-row kernel beside the slice kernel
    bool band_disabled = false, c16_off = false, wide_off = false;
    int forced_spw = 0, forced_pipe = -1, forced_nw = 0;
    long panel_round = -1;                   // SL_PANEL_ROUND: blocks per launch of the dynamic-tile panel kernel (-1: from the occupancy)
    uint32_t pw_slack = 0;                   // SL_PW_SLACK: panels of lead in the paced panel kernel (0: the matrix's own figure)
    bool mpass_force = false;                // SL_MPASS=1: use the multi-pass window kernel wherever it applies (measured slower: not selected by default)
    sl_kernel_knobs()
    {
        auto flag = [](const char *n, char v) { const char *e = getenv(n); return e && e[0] == v; };
        auto num = [](const char *n, long dflt) { const char *e = getenv(n); return e && *e ? atol(e) : dflt; };
        long_rows_beside_min = (uint64_t)num("SL_LONG_ROWS_BESIDE_MIN", 16384);
        if (flag("SL_LONG_ROWS_SERIAL", '1')) long_rows_beside_min = 1ull << 62;
        band_disabled = flag("SL_BAND_DISABLE", '1');
        forced_spw = (int)num("SL_BAND_SPW", 0);
        forced_pipe = (int)num("SL_BAND_PIPE", -1);
        c16_off = flag("SL_BAND_C16", '0');
        forced_nw = (int)num("SL_BAND_NW", 0);
        wide_off = flag("SL_BAND_NW16", '0');
        panel_round = num("SL_PANEL_ROUND", -1);
        pw_slack = (uint32_t)num("SL_PW_SLACK", 0);
        mpass_force = flag("SL_MPASS", '1');
    }
};
static const sl_kernel_knobs &knobs()
{
    static const sl_kernel_knobs k;
    return k;
}
static uint64_t long_rows_beside_min() { return knobs().long_rows_beside_min; }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device (one process may drive several devices from several
// threads); KFN is the kernel itself, so every instantiation has its own flags
#define SL_MAX_DEVICES 32
template <auto KFN>
static sl_status set_max_lds_once(int bytes)
{
    static std::once_flag once[SL_MAX_DEVICES];
    static hipError_t err[SL_MAX_DEVICES];
    int dev = 0;
    SL_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= SL_MAX_DEVICES) {
        SL_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(KFN), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        return SL_OK;
    }
    std::call_once(once[dev], [&] { err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(KFN), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); });
    SL_HIP(err[dev]);
    return SL_OK;
}

static band_geom band_geometry(const sl_row_args &a, bool offsets16_usable, bool nw8_pays, bool ragged)
{
    const sl_kernel_knobs &kn = knobs();
    const int disabled = kn.band_disabled ? 1 : 0, forced_spw = kn.forced_spw, forced_pipe = kn.forced_pipe, c16_off = kn.c16_off ? 1 : 0, forced_nw = kn.forced_nw;
    band_geom out{0, 0, 4, false, false};
    if (disabled || a.bandwidth == ~0ull || a.n_cols > 0xffffffffull) return out;
    // measured (profiles/r01_ab_wave_blocks.txt and earlier spw sweeps; +-5 % DVFS noise between repetitions): narrow windows
    // 4 waves x 4 slices per block; wide windows (w > 1024), where the window itself caps the CU at two blocks,
    // 8 waves x 3 slices (the window is re-staged nw * spw * 64 rows at a time); pipelining never hurts
    const bool pipe = forced_pipe >= 0 ? forced_pipe != 0 : true;
    const bool c16 = a.cols16 != nullptr && !c16_off && offsets16_usable;
    // ragged rows (the batched path) behind a narrow window: 8 waves x 8 slices — a slice of short rows is one or two loads per
    // array, and it takes that many of them in flight per CU to keep the stream going (tools/sweep_short_rows.sh, w = 512, ms per
    // step 4 x 4 -> 8 x 8: 5 entries per row 0.211 -> 0.183, 9 per row 0.280 -> 0.252; no gain behind wide windows or for uniform rows)
    const bool ragged_narrow = ragged && a.bandwidth <= 1024;
    uint32_t nw = (forced_nw == 4 || forced_nw == 8) ? (uint32_t)forced_nw : ((a.bandwidth > 1024 || ragged_narrow) ? 8u : 4u);
    if (!pipe || !c16 || !nw8_pays) nw = 4;                         // 8-wave blocks: pipelined 16-bit-offset variants that gain from them
    uint32_t spw = forced_spw > 0 ? (uint32_t)forced_spw : (a.bandwidth <= 1024 ? ((ragged_narrow && nw == 8) ? 8u : 4u) : (nw == 8 ? 3u : 6u));
    uint64_t entries = (uint64_t)nw * spw * SL_SLICE + 2 * a.bandwidth + 2;
    while (entries * 8 > SL_BAND_MAX_LDS && forced_spw <= 0 && spw > 1) {   // a shorter block may still fit
        spw = spw == 3 ? 2 : spw >> 1;
        entries = (uint64_t)nw * spw * SL_SLICE + 2 * a.bandwidth + 2;
    }
    if (entries * 8 > SL_BAND_MAX_LDS || forced_nw == 16) {
        // the window does not fit twice per CU: one 16-wave block per CU with (almost) the whole LDS, if that variant exists
        // (w = 6000..9400: 52-56 % with the general kernel -> 72-82 %)
        const bool wide_off = kn.wide_off;
        if (wide_off || !pipe || !c16 || !nw8_pays || forced_nw == 4 || forced_nw == 8) return out;
        nw = 16;
        spw = forced_spw > 0 ? (uint32_t)forced_spw : 4u;
        entries = (uint64_t)nw * spw * SL_SLICE + 2 * a.bandwidth + 2;
        while (entries * 8 > SL_BAND_MAX_LDS_ONE && forced_spw <= 0 && spw > 1) { --spw; entries = (uint64_t)nw * spw * SL_SLICE + 2 * a.bandwidth + 2; }
        if (entries * 8 > SL_BAND_MAX_LDS_ONE) return out;
    }
    out.spw = spw; out.lds = (uint32_t)(entries * 8); out.nw = nw;
    out.pipe = pipe;
    out.c16 = c16;
    return out;
}

template <int ORDER, int EPI, int UWV, bool PIPE, bool C16, int NW>
static sl_status launch_band_nw(const sl_row_args &a, const band_geom &g, uint32_t grid, uint32_t nb8, hipStream_t s)
{
    constexpr auto kfn = sl_band_kernel<(UWV) ? 0 : ORDER, EPI, UWV, PIPE, C16, NW>;
    SL_TRY(set_max_lds_once<kfn>(NW == 16 ? SL_BAND_MAX_LDS_ONE : SL_BAND_MAX_LDS));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(NW * 64), g.lds, s, a, nb8, g.spw, (uint32_t)a.bandwidth);
    return SL_OK;
}
template <int ORDER, int EPI, int UWV, bool PIPE, bool C16>
static sl_status launch_band(const sl_row_args &a, const band_geom &g, uint32_t grid, uint32_t nb8, hipStream_t s)
{
    constexpr bool nw8_built = PIPE && C16 && !(EPI == SL_EPI_PUSH && UWV == 16);   // = the cases band_geometry picks 8 for
    if constexpr (nw8_built) {
        if (g.nw == 8) return launch_band_nw<ORDER, EPI, UWV, PIPE, C16, 8>(a, g, grid, nb8, s);
        if (g.nw == 16) return launch_band_nw<ORDER, EPI, UWV, PIPE, C16, 16>(a, g, grid, nb8, s);
    }
    if (g.nw != 4) return sl_fail(SL_DEVICE_ERROR, "band geometry asks for %u waves per block, variant built for 4", g.nw);
    return launch_band_nw<ORDER, EPI, UWV, PIPE, C16, 4>(a, g, grid, nb8, s);
}

template <int ORDER, int EPI, int UWV>
static sl_status launch_band_u(const sl_row_args &a, const band_geom &g, uint32_t grid, uint32_t nb8, hipStream_t s)
{
    if (g.pipe) return g.c16 ? launch_band<ORDER, EPI, UWV, true, true>(a, g, grid, nb8, s) : launch_band<ORDER, EPI, UWV, true, false>(a, g, grid, nb8, s);
    return g.c16 ? launch_band<ORDER, EPI, UWV, false, true>(a, g, grid, nb8, s) : launch_band<ORDER, EPI, UWV, false, false>(a, g, grid, nb8, s);
}

// multi-pass window kernel: rows of at most 16 entries in the slice layout, a measured bandwidth beyond the LDS window, a window of
// at most 64 segments, and enough rows per staged window that the staging (out of the L2) stays below ~2.5 x the matrix bytes
static bool mpass_eligible(const sl_row_args &a)
{
    if (a.bandwidth == ~0ull || a.n_cols > 0xffffffffull || a.max_row_nnz == 0 || a.max_row_nnz > 16 || a.n_long) return false;
    const uint64_t R = (uint64_t)SL_MP_WAVES * (a.max_row_nnz <= 8 ? 6 : 4) * SL_SLICE;
    const uint64_t window = R + 2 * a.bandwidth + 2;
    if (window > 64ull * SL_MP_SEG - 2) return false;
    // Measured (tools/ab_wide.sh, n = 10^7 x 16, ms per step multi-pass / general kernel): w = 12000 0.588 / 0.598, w = 32768 0.812 / 0.683,
    // w = 100000 1.49 / 0.73, 7-point stencil on 215^3 0.358 / 0.244 — a block alternates between its HBM phase (matrix bytes into
    // registers) and its LDS phase (9+ segments behind barriers) with two waves per SIMD, and stages R + 2 w entries for R = 2048
    // rows; the general kernel's L2-served gathers win from w ~ 15 K on.  Kept as an opt-in (SL_MPASS=1, parity-tested), not selected.
    return knobs().mpass_force;
}

template <int ORDER, int EPI>
static band_geom pick_band(const sl_row_args &a)
{
    const bool uniform_unrolled = ORDER == 0 && (a.uniform_width == 16 || a.uniform_width == 8);
    // a uniform-width matrix carries its 16-bit offsets in the OCTET layout of the unrolled path; when it runs through
    // the batched path instead (simd4 order), that path — which reads the QUAD layout — uses the u32 columns
    const bool uniform_octets = a.uniform_width == 8 || a.uniform_width == 16;
    // 8-wave blocks (wide windows) are held to 128 VGPRs.  The unrolled uniform-width variants fit (except the push
    // epilogue at width 16, which would spill ~100 B per lane and lose 10 %: it stays at 4 waves); the batched path fits
    // with 3 quads per batch (2 in the 4-lane order) instead of 4
    const bool nw8_pays = !(EPI == SL_EPI_PUSH && uniform_unrolled && a.uniform_width == 16);
    return band_geometry(a, uniform_unrolled || !uniform_octets, nw8_pays, !uniform_unrolled);
}

// rows per block / blocks of the launch launch_rows_t would make; *rows = 0: a layout without range launches
template <int ORDER, int EPI>
static void rows_geom_t(const sl_row_args &a, uint32_t *rows, uint32_t *blocks)
{
    *rows = 0; *blocks = 0;
    if (a.n_long || a.n_slices == 0) return;
    if (ORDER == 0 && ((a.pw_idx && a.pw_tiles) || (a.pan_tile_ptr && a.n_pan_tiles))) return;
    const band_geom g = pick_band<ORDER, EPI>(a);
    if (g.spw) {
        const uint64_t per_block = (uint64_t)g.nw * g.spw;
        const uint64_t nb = (a.n_slices + per_block - 1) / per_block;
        *rows = (uint32_t)(per_block * SL_SLICE);
        *blocks = (uint32_t)((nb + 7) / 8) * 8;
    } else if (ORDER == 0 && mpass_eligible(a)) {
        return;
    } else {
        *rows = SL_WAVES_PER_BLOCK * SL_SLICE;
        *blocks = sl_row_grid(a.n_slices);
    }
}

template <int ORDER, int EPI>
static sl_status launch_rows_t(const sl_row_args &a_in, hipStream_t s, uint32_t *nparts)
{
    sl_row_args a = a_in;
    if (a.n_long && a.n_slices >= long_rows_beside_min()) {          // fork point for the long-row kernel (see the end of this function)
        sl_ctx &c = sl_context();
        if (sl_side_stream(c)) {
            SL_HIP(hipEventRecord(c.ev_fork, s));
            SL_HIP(hipStreamWaitEvent(c.side, c.ev_fork, 0));
        }
    }
    const band_geom g = pick_band<ORDER, EPI>(a);
    const uint32_t range_cnt = a.blk_cnt;                              // != 0: a range of the blocks (band and general kernel only)
    const bool pw_rounds = ORDER == 0 && a.pw_idx && a.pw_tiles && a.pw_span_tab;       // explicit spans of the paced layout (edge-first rounds): ranges are rounds
    if (range_cnt && (a.n_long || (ORDER == 0 && !pw_rounds && ((a.pw_idx && a.pw_tiles) || (a.pan_tile_ptr && a.n_pan_tiles) || (!g.spw && mpass_eligible(a))))))
        return sl_fail(SL_UNSUPPORTED_FORMAT, "this matrix layout has no range launches");
    if (ORDER == 0 && a.pw_idx && a.pw_tiles) {
        const uint32_t lds = (uint32_t)(SL_PW_WAVES * ((size_t)a.pw_rpw + 1) * sizeof(double));
        SL_TRY(set_max_lds_once<sl_pw_kernel<EPI>>((int)(SL_PW_WAVES * ((size_t)SL_PW_MAX_ROWS + 1) * sizeof(double))));
        if (knobs().pw_slack) a.pw_slack = knobs().pw_slack;               // A/B knob; 1048576 = no pacing
        *nparts = range_cnt ? 2 * a.pw_blocks : a.pw_blocks + a.n_long;        // two launches (edge rounds, the rest), a set of partial sums each
        a.part_stride = *nparts;
#ifdef SL_PWR_VARIANTS
        static const bool pw_any = [] { const char *e = getenv("SL_PW_ANY"); return e && *e == '1'; }();
        static const int pw_var = [] { const char *e = getenv("SL_PW_VAR"); return e ? atoi(e) : 0; }();
        if (pw_var && EPI == SL_EPI_NEUMANN) {
#define SL_PW_V(v) case v: SL_TRY((set_max_lds_once<sl_pw_kernel<SL_EPI_NEUMANN, false, v>>((int)(SL_PW_WAVES * ((size_t)SL_PW_MAX_ROWS + 1) * sizeof(double))))); \
                           hipLaunchKernelGGL((sl_pw_kernel<SL_EPI_NEUMANN, false, v>), dim3(a.pw_blocks), dim3(SL_PW_WAVES * 64), lds, s, a); break;
            switch (pw_var) { SL_PW_V(2) SL_PW_V(8) SL_PW_V(10) SL_PW_V(32) SL_PW_V(40) SL_PW_V(42) default: return sl_fail(SL_INVALID_INPUT, "SL_PW_VAR"); }
#undef SL_PW_V
        } else
        if (pw_any && EPI == SL_EPI_NEUMANN) {
            SL_TRY((set_max_lds_once<sl_pw_kernel<SL_EPI_NEUMANN, true>>((int)(SL_PW_WAVES * ((size_t)SL_PW_MAX_ROWS + 1) * sizeof(double)))));
            hipLaunchKernelGGL((sl_pw_kernel<SL_EPI_NEUMANN, true>), dim3(a.pw_blocks), dim3(SL_PW_WAVES * 64), lds, s, a);
        } else
#endif
        if (EPI == SL_EPI_PUSH && a.zgather) {                             // column-constant operator: the index words of the stream alone
            SL_TRY((set_max_lds_once<sl_pw_kernel<SL_EPI_PUSH, false, 0, true>>((int)(SL_PW_WAVES * ((size_t)SL_PW_MAX_ROWS + 1) * sizeof(double)))));
            hipLaunchKernelGGL((sl_pw_kernel<SL_EPI_PUSH, false, 0, true>), dim3(a.pw_blocks), dim3(SL_PW_WAVES * 64), lds, s, a);
        } else
        hipLaunchKernelGGL((sl_pw_kernel<EPI>), dim3(a.pw_blocks), dim3(SL_PW_WAVES * 64), lds, s, a);
    } else if (ORDER == 0 && a.pan_tile_ptr && a.n_pan_tiles) {
        const uint32_t grid = (a.n_pan_tiles + SL_PANEL_WAVES - 1) / SL_PANEL_WAVES;
        constexpr uint32_t lds = SL_PANEL_WAVES * (SL_PANEL_TILE + 64) * sizeof(double);
        SL_TRY(set_max_lds_once<sl_panel_kernel<EPI>>((int)lds));
        *nparts = grid + a.n_long;
        a.part_stride = *nparts;
        // one launch per round of resident blocks: blocks that start together pass the panels together, a second round that trickles
        // in behind the first does not (n = 10^7 x 16: 1.66 -> 1.50 ms; partial rounds lose it again, and so do tiles of unequal length — the PageRank solve
        // went 0.314 -> 0.345 s — so only matrices whose tiles are balanced are launched this way).  SL_PANEL_ROUND: blocks per
        // launch, 0 = a single launch
        static const uint32_t round_blocks = [] {                         // thread-safe static initialisation, per template instance
            if (knobs().panel_round >= 0) return (uint32_t)knobs().panel_round;
            int per_cu = 0, dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess
                && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sl_panel_kernel<EPI>, SL_PANEL_WAVES * 64, lds) == hipSuccess && per_cu > 0)
                return (uint32_t)per_cu * (uint32_t)prop.multiProcessorCount;
            return 0u;
        }();
        const uint32_t per = (round_blocks && a.pan_balanced) ? round_blocks : grid;     // unequal tiles: every round would wait for its longest
        for (uint32_t b0 = 0; b0 < grid; b0 += per)
            hipLaunchKernelGGL((sl_panel_kernel<EPI>), dim3(std::min(per, grid - b0)), dim3(SL_PANEL_WAVES * 64), lds, s, a, b0);
    } else if (g.spw) {
        const uint64_t per_block = (uint64_t)g.nw * g.spw;
        const uint64_t nb = (a.n_slices + per_block - 1) / per_block;
        uint32_t nb8 = (uint32_t)((nb + 7) / 8);
        uint32_t grid = nb8 * 8;
        *nparts = grid + a.n_long;
        a.part_stride = *nparts;
        if (range_cnt) { nb8 = (range_cnt + 7) / 8; grid = nb8 * 8; }   // the range's own XCD-contiguous mapping
        sl_status st;
        if (ORDER == 0 && a.uniform_width == 16) st = launch_band_u<ORDER, EPI, 16>(a, g, grid, nb8, s);
        else if (ORDER == 0 && a.uniform_width == 8) st = launch_band_u<ORDER, EPI, 8>(a, g, grid, nb8, s);
        else {
            st = launch_band_u<ORDER, EPI, 0>(a, g, grid, nb8, s);
        }
        if (st != SL_OK) return st;
    } else if (ORDER == 0 && mpass_eligible(a)) {
        // bands too wide for one LDS window: the window in segments, the rows' matrix bytes in registers (sl_mpass_kernel)
        const bool narrow = a.max_row_nnz <= 8;                          // rows of at most 8 entries: six slices per wave instead of four
        const uint64_t per_block = (uint64_t)SL_MP_WAVES * (narrow ? 6 : 4);
        const uint64_t nb = (a.n_slices + per_block - 1) / per_block;
        const uint32_t nb8 = (uint32_t)((nb + 7) / 8), grid = nb8 * 8;
        constexpr int lds = 2 * SL_MP_SEG * sizeof(double);
        *nparts = grid + a.n_long;
        a.part_stride = *nparts;
        if (narrow) {
            SL_TRY((set_max_lds_once<sl_mpass_kernel<EPI, 2, 6>>(lds)));
            hipLaunchKernelGGL((sl_mpass_kernel<EPI, 2, 6>), dim3(grid), dim3(SL_MP_WAVES * 64), lds, s, a, nb8, (uint32_t)a.bandwidth);
        } else {
            SL_TRY((set_max_lds_once<sl_mpass_kernel<EPI, 4, 4>>(lds)));
            hipLaunchKernelGGL((sl_mpass_kernel<EPI, 4, 4>), dim3(grid), dim3(SL_MP_WAVES * 64), lds, s, a, nb8, (uint32_t)a.bandwidth);
        }
    } else {
        uint32_t grid = sl_row_grid(a.n_slices), nb8 = grid / 8;
        *nparts = grid + a.n_long;
        a.part_stride = *nparts;
        if (range_cnt) { nb8 = (range_cnt + 7) / 8; grid = nb8 * 8; }
        if (ORDER == 0 && a.uniform_width == 16)
            hipLaunchKernelGGL((sl_rows_kernel<0, EPI, 16>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
        else if (ORDER == 0 && a.uniform_width == 8)
            hipLaunchKernelGGL((sl_rows_kernel<0, EPI, 8>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
        else
            hipLaunchKernelGGL((sl_rows_kernel<ORDER, EPI, 0>), dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
    }
    if (a.n_long) {
        // hub rows: one block per row with a sequential add chain (10^4..10^5 entries: 0.1 ms and more).  On large matrices it
        // runs BESIDE the slice kernel on a side stream — same inputs, disjoint rows and partial slots; the launches that follow
        // on `s` (final reduction, next step) are ordered behind both by the join event.
        sl_ctx &c = sl_context();
        const bool beside = a.n_slices >= long_rows_beside_min() && sl_side_stream(c);
        if (beside) {
            // fork point = everything enqueued on s BEFORE the slice kernel of this launch; it was recorded by the caller below
            hipLaunchKernelGGL((sl_long_rows_kernel<ORDER, EPI>), dim3((a.n_long + SL_BLOCK / 64 - 1) / (SL_BLOCK / 64)), dim3(SL_BLOCK), 0, c.side, a, *nparts - a.n_long);
            SL_HIP(hipEventRecord(c.ev_join, c.side));
            SL_HIP(hipStreamWaitEvent(s, c.ev_join, 0));
        } else {
            hipLaunchKernelGGL((sl_long_rows_kernel<ORDER, EPI>), dim3((a.n_long + SL_BLOCK / 64 - 1) / (SL_BLOCK / 64)), dim3(SL_BLOCK), 0, s, a, *nparts - a.n_long);
        }
    }
    SL_HIP(hipGetLastError());
    return SL_OK;
}

// SL_ORDER_ANY on a matrix that carries the order-free column stream (Neumann / SpMV / residual; whole launches only)
static bool order_free_launch(const sl_row_args &a, sl_order order, sl_epilogue epi)
{
    return order == SL_ORDER_ANY && a.pwr_idx && a.pwr_tiles && epi != SL_EPI_PUSH && !a.blk_cnt;
}

template <int EPI>
static sl_status launch_pwr(sl_row_args a, hipStream_t s, uint32_t *nparts)
{
    SL_TRY(set_max_lds_once<sl_pwr_kernel<EPI>>((int)(((size_t)SL_PWR_MAX_ROWS + 1) * sizeof(double))));
    *nparts = a.pwr_blocks;
    a.part_stride = *nparts;
    a.n_long = 0;                                                    // hub rows are in the stream like every other row
    const uint32_t lds = (uint32_t)(((size_t)a.pwr_rpb + 1) * sizeof(double));      // + the spare slot padding entries add their products to
#ifdef SL_PWR_VARIANTS
    static const int var = [] { const char *e = getenv("SL_PWR_VAR"); return e ? atoi(e) : 0; }();
    if (EPI == SL_EPI_NEUMANN && var) {
#define SL_PWR_V(v) case v: SL_TRY((set_max_lds_once<sl_pwr_kernel<SL_EPI_NEUMANN, v>>((int)(((size_t)SL_PWR_MAX_ROWS + 1) * sizeof(double))))); \
                            hipLaunchKernelGGL((sl_pwr_kernel<SL_EPI_NEUMANN, v>), dim3(a.pwr_blocks), dim3(SL_PW_WAVES * 64), lds, s, a); break;
        switch (var) { SL_PWR_V(1) SL_PWR_V(2) SL_PWR_V(4) SL_PWR_V(6) SL_PWR_V(10) SL_PWR_V(14) SL_PWR_V(32) SL_PWR_V(42) default: return sl_fail(SL_INVALID_INPUT, "SL_PWR_VAR"); }
#undef SL_PWR_V
        SL_HIP(hipGetLastError());
        return SL_OK;
    }
#endif
    hipLaunchKernelGGL((sl_pwr_kernel<EPI>), dim3(a.pwr_blocks), dim3(SL_PW_WAVES * 64), lds, s, a);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

sl_status sl_launch_rows(const sl_row_args &a, sl_order order, sl_epilogue epi, hipStream_t s, uint32_t *n_partials)
{
    if (n_partials) *n_partials = 0;
    if (a.n_slices == 0) {
        if (epi != SL_EPI_SPMV && a.ctl)
            hipLaunchKernelGGL(sl_judge_reduce_kernel, dim3(1), dim3(1024), 0, s, a.partials, 0u, a.result, a.ctl, a.gate_it,
                               a.ctl_slot, a.ctl_mode, a.ctl_threshold);
        else if (epi != SL_EPI_SPMV && a.result) SL_HIP(hipMemsetAsync(a.result, 0, 2 * sizeof(double), s));
        return SL_OK;
    }
    const bool simd4 = (order == SL_ORDER_SIMD4);
    uint32_t nparts = 0;
    sl_status st = SL_OK;
    if (order_free_launch(a, order, epi)) {
        st = epi == SL_EPI_SPMV ? launch_pwr<SL_EPI_SPMV>(a, s, &nparts) : epi == SL_EPI_NEUMANN ? launch_pwr<SL_EPI_NEUMANN>(a, s, &nparts) : launch_pwr<SL_EPI_RESIDUAL>(a, s, &nparts);
        if (st != SL_OK) return st;
        if (n_partials) *n_partials = nparts;
        sl_row_args b = a;
        b.n_long = 0;
        return sl_launch_rows_reduce(b, epi, nparts, s);
    }
    switch (epi) {
    case SL_EPI_SPMV: st = simd4 ? launch_rows_t<1, SL_EPI_SPMV>(a, s, &nparts) : launch_rows_t<0, SL_EPI_SPMV>(a, s, &nparts); break;
    case SL_EPI_NEUMANN: st = simd4 ? launch_rows_t<1, SL_EPI_NEUMANN>(a, s, &nparts) : launch_rows_t<0, SL_EPI_NEUMANN>(a, s, &nparts); break;
    case SL_EPI_RESIDUAL: st = simd4 ? launch_rows_t<1, SL_EPI_RESIDUAL>(a, s, &nparts) : launch_rows_t<0, SL_EPI_RESIDUAL>(a, s, &nparts); break;
    case SL_EPI_PUSH: st = simd4 ? launch_rows_t<1, SL_EPI_PUSH>(a, s, &nparts) : launch_rows_t<0, SL_EPI_PUSH>(a, s, &nparts); break;
    }
    if (st != SL_OK) return st;
    if (n_partials) *n_partials = nparts;
    if (a.blk_cnt) return SL_OK;                                        // a range: the caller closes with sl_launch_rows_reduce
    return sl_launch_rows_reduce(a, epi, nparts, s);
}

// y += A x in the CSR order with the running sums seeded by y (a.out = y in / out, a.gather = x)
sl_status sl_launch_rows_add(const sl_row_args &a, hipStream_t s)
{
    if (a.n_slices == 0) return SL_OK;
    const uint32_t grid = sl_row_grid(a.n_slices), nb8 = grid / 8;
    hipLaunchKernelGGL(sl_rows_add_kernel, dim3(grid), dim3(SL_BLOCK), 0, s, a, nb8);
    if (a.n_long)      // disjoint rows: order against the slice kernel does not matter, the same stream keeps it simple
        hipLaunchKernelGGL((sl_long_rows_kernel<0, SL_EPI_SPMV, true>), dim3((a.n_long + SL_BLOCK / 64 - 1) / (SL_BLOCK / 64)), dim3(SL_BLOCK), 0, s, a, 0u);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

sl_status sl_rows_geometry(const sl_row_args &a, sl_order order, sl_epilogue epi, uint32_t *rows_per_block, uint32_t *n_blocks)
{
    if (order_free_launch(a, order, epi)) { *rows_per_block = 0; *n_blocks = 0; return SL_OK; }      // persistent blocks: no range launches
    const bool simd4 = (order == SL_ORDER_SIMD4);
    switch (epi) {
    case SL_EPI_SPMV: simd4 ? rows_geom_t<1, SL_EPI_SPMV>(a, rows_per_block, n_blocks) : rows_geom_t<0, SL_EPI_SPMV>(a, rows_per_block, n_blocks); break;
    case SL_EPI_NEUMANN: simd4 ? rows_geom_t<1, SL_EPI_NEUMANN>(a, rows_per_block, n_blocks) : rows_geom_t<0, SL_EPI_NEUMANN>(a, rows_per_block, n_blocks); break;
    case SL_EPI_RESIDUAL: simd4 ? rows_geom_t<1, SL_EPI_RESIDUAL>(a, rows_per_block, n_blocks) : rows_geom_t<0, SL_EPI_RESIDUAL>(a, rows_per_block, n_blocks); break;
    case SL_EPI_PUSH: simd4 ? rows_geom_t<1, SL_EPI_PUSH>(a, rows_per_block, n_blocks) : rows_geom_t<0, SL_EPI_PUSH>(a, rows_per_block, n_blocks); break;
    }
    return SL_OK;
}

sl_status sl_launch_rows_reduce(const sl_row_args &a, sl_epilogue epi, uint32_t nparts, hipStream_t s)
{
    const double *parts = a.partials;
    if (epi != SL_EPI_SPMV && (a.ctl || a.result) && nparts > SL_PRE_THRESHOLD && a.partials_slack >= 2 * SL_PRE_BLOCKS) {
        const int nsets = epi == SL_EPI_PUSH ? 2 : 1;
        double *stage = a.partials + 2 * (uint64_t)nparts;             // the slack behind the partial sets
        hipLaunchKernelGGL(sl_pre_reduce_kernel, dim3(SL_PRE_BLOCKS, nsets), dim3(256), 0, s, a.partials, nparts, stage, a.ctl, a.gate_it);
        parts = stage;
        nparts = SL_PRE_BLOCKS;
    }
    if (epi != SL_EPI_SPMV && a.ctl) {
        hipLaunchKernelGGL(sl_judge_reduce_kernel, dim3(1), dim3(1024), 0, s, parts, nparts, a.result, a.ctl, a.gate_it,
                           a.ctl_slot, a.ctl_mode, a.ctl_threshold);
        SL_HIP(hipGetLastError());
    } else if (epi != SL_EPI_SPMV && a.result) {
        hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, parts, nparts, a.result,
                           epi == SL_EPI_PUSH ? 2 : 1);
        SL_HIP(hipGetLastError());
    }
    return SL_OK;
}

sl_status sl_launch_final_reduce(const double *partials, uint32_t n, double *result, hipStream_t s)
{
    hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, n, result, 1);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

// ---- vector primitives (a5) -----------------------------------------------------------------
#define SL_VEC_BLOCKS 2048

template <int MODE> // 0: sum x^2, 1: sum x*y, 2: sum |x|, 3: sum (x - y)^2
__global__ __launch_bounds__(256) void sl_reduce_kernel(uint64_t n, const double *__restrict__ x,
                                                        const double *__restrict__ y, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double v = x[i];
        if (MODE == 0) acc = DADD(acc, DMUL(v, v));
        else if (MODE == 1) acc = DADD(acc, DMUL(v, y[i]));
        else if (MODE == 3) { const double d = DSUB(v, y[i]); acc = DADD(acc, DMUL(d, d)); }
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
sl_status sl_launch_sumsq_judged(uint64_t n, const double *x, double *partials, sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot,
                                 int mode, double threshold, hipStream_t s, double *result)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL((sl_reduce_kernel<0>), dim3(g), dim3(256), 0, s, n, x, x, partials);
    hipLaunchKernelGGL(sl_judge_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, g, result, ctl, gate_it, slot, mode, threshold);
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

sl_status sl_launch_diff_sumsq(uint64_t n, const double *x, const double *y, double *partials, double *result, hipStream_t s)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL((sl_reduce_kernel<3>), dim3(g), dim3(256), 0, s, n, x, y, partials);
    hipLaunchKernelGGL(sl_final_reduce_kernel, dim3(1), dim3(1024), 0, s, partials, g, result, 1);
    SL_HIP(hipGetLastError());
    return SL_OK;
}
// solver::utils::linf_norm (solver/mod.rs:379-381): v.iter().map(|x| x.abs()).fold(0.0, f64::max) — f64::max returns the other
// operand when one is NaN, so NaN entries never enter; fmax has the same rule, and a maximum is exact in any order
__global__ __launch_bounds__(256) void sl_absmax_kernel(uint64_t n, const double *__restrict__ x, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc = fmax(acc, fabs(x[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = fmax(acc, __shfl_xor(acc, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
__global__ __launch_bounds__(256) void sl_absmax_final_kernel(uint32_t n, const double *partials, double *result)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 256) acc = fmax(acc, partials[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc = fmax(acc, __shfl_xor(acc, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) result[0] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}
sl_status sl_launch_abs_max(uint64_t n, const double *x, double *partials, double *result, hipStream_t s)
{
    const uint32_t g = vec_grid(n);
    hipLaunchKernelGGL(sl_absmax_kernel, dim3(g), dim3(256), 0, s, n, x, partials);
    hipLaunchKernelGGL(sl_absmax_final_kernel, dim3(1), dim3(256), 0, s, g, partials, result);
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
