// sl_matrix.hip — device-side matrix ingest: CSR validation, CSR -> row-slice layout,
// diagonal-dominance check, D^-1 extraction, column structure (pattern of A^T).
// One-off work per matrix (not in the per-iteration metric).
#include "sl_internal.hpp"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

// ---- validation ---------------------------------------------------------------------------
// err[0]: bit0 row_ptr not monotone / bad ends, bit1 column out of bounds
__global__ void sl_validate_csr_kernel(uint64_t n_rows, uint64_t n_cols, uint64_t nnz, const uint32_t *row_ptr,
                                       const uint32_t *col_idx, uint32_t *err)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride)
        if (row_ptr[i] > row_ptr[i + 1]) atomicOr(err, 1u);
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride)
        if ((uint64_t)col_idx[k] >= n_cols) atomicOr(err, 2u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (row_ptr[0] != 0u) atomicOr(err, 1u);
        if ((uint64_t)row_ptr[n_rows] != nnz) atomicOr(err, 1u);
    }
}

// row lengths + slice widths (in pair blocks) + min/max row length
__global__ __launch_bounds__(256) void sl_row_len_kernel(uint64_t n_rows, uint64_t n_slices, uint32_t long_row, const uint32_t *row_ptr,
                                                         uint32_t *row_len, uint32_t *slice_w, uint32_t *minmax)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t s = i >> 6;
    if (s >= n_slices) return;
    uint32_t len = 0;
    if (i < n_rows) {
        len = row_ptr[i + 1] - row_ptr[i];
        if (minmax) {
            atomicMin(&minmax[0], len);
            atomicMax(&minmax[1], len);
            if (len > long_row) atomicAdd(&minmax[2], 1u);
        }
    }
    const bool is_long = len > long_row;             // leaves the slice layout (sl_long_rows_kernel owns it)
    row_len[i] = is_long ? SL_LONG_SENTINEL : len;
    uint32_t mx = is_long ? 0u : len;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t other = __shfl_xor(mx, o);
        mx = other > mx ? other : mx;
    }
    if ((threadIdx.x & 63) == 0) slice_w[s] = (mx + 1u) >> 1;
}

// CSR -> row-slice layout.  One wave per slice, lane = row.  Padding entries carry
// value 0.0 and a valid column (the row's own index when it exists); the kernels never
// add them (guarded by row_len), they only keep every gather in bounds.
#define SL_PW_BAND_AUTO 0xffu
#define SL_FAR_COLUMN (1ull << 18)     // |col - row| beyond this (2 MB of vector, half an XCD's L2): the gather is far from what the row's neighbours keep cached
__global__ __launch_bounds__(256) void sl_fill_slices_kernel(uint64_t n_rows, uint64_t n_cols, uint64_t n_slices,
                                                             uint64_t row_offset, const uint32_t *row_ptr,
                                                             const uint32_t *col_idx, const double *values,
                                                             const uint32_t *slice_ptr, const uint32_t *row_len, uint32_t *cols,
                                                             double *vals, unsigned long long *band)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    const uint32_t q0 = slice_ptr[s], q1 = slice_ptr[s + 1];
    uint32_t start = 0, len = 0;
    // padding column: the row's own index; dead lanes of the last slice use the slice's first row
    // (always a live row) so that every padded gather stays inside the band window of its block
    uint64_t gi = row_offset + (i < n_rows ? i : s * 64);
    uint32_t padcol = gi < n_cols ? (uint32_t)gi : 0u;
    if (i < n_rows) {
        start = row_ptr[i];
        len = row_len[i] == SL_LONG_SENTINEL ? 0u : row_len[i];
    }
    unsigned long long bw = 0, far = 0, diag = 0;
    const uint32_t slots = (q1 - q0) * 2;                      // q0, q1: pair blocks
    if (len > slots) atomicAdd(band + 4, 1ull);                // the slice pointers do not match the row lengths: the host rebuilds them
    for (uint32_t k = 0; k < slots; ++k) {
        const bool in = k < len;
        const uint32_t c = in ? col_idx[start + k] : padcol;
        const double v = in ? values[start + k] : 0.0;
        if (in) { const unsigned long long d = c > gi ? c - gi : gi - c; bw = d > bw ? d : bw; far += d > SL_FAR_COLUMN ? 1u : 0u; }
        // the same slot of the row above holds the column to the left: a diagonal of a stencil — neighbouring lanes gather from one line
        const uint32_t c_up = __shfl_up(c, 1);
        const uint32_t in_up = __shfl_up((uint32_t)in, 1);
        if (in && lane > 0 && in_up && c == c_up + 1u) ++diag;
        cols[sl_col_slot(q0, q1, k, lane)] = c;
        vals[sl_val_slot(q0, k, lane)] = v;
    }
    if (bw) atomicMax(band, bw);
    unsigned long long held = len;                             // entries of the slice layout (the long rows' entries are not in it)
    for (int off = 32; off > 0; off >>= 1) { far += __shfl_xor(far, off); held += __shfl_xor(held, off); diag += __shfl_xor(diag, off); }
    if (lane == 0 && far) atomicAdd(band + 1, far);            // entries more than 2 MB of vector away from their row
    if (lane == 0 && held) atomicAdd(band + 2, held);
    if (lane == 0 && diag) atomicAdd(band + 3, diag);          // entries that continue a diagonal from the row above
}

// 16-bit column offsets for uniform-width band matrices: [slice][octet][lane][8] int16 = col - row,
// one 16-B load per lane per 8 entries.  Padding entries and dead lanes carry offset 0.
__global__ __launch_bounds__(256) void sl_fill_cols16_kernel(uint64_t n_rows, uint64_t n_slices, uint64_t row_offset, uint32_t uw,
                                                             const uint32_t *cols, uint16_t *cols16)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    const uint64_t gi = row_offset + i;
    const uint32_t nq = uw / 4;
    for (uint32_t q = 0; q < nq; ++q)
        for (uint32_t e = 0; e < 4; ++e) {
            const uint32_t c = cols[((s * nq + q) * 64 + lane) * 4 + e];
            const int delta = (i < n_rows) ? (int)((long long)c - (long long)gi) : 0;
            const uint32_t k = q * 4 + e;
            cols16[((s * (uw / 8) + (k >> 3)) * 64 + lane) * 8 + (k & 7u)] = (uint16_t)(int16_t)delta;
        }
}

// 16-bit column offsets for ragged band matrices, in the slots of `cols` (8 B per lane per quad, 4 B in an odd last pair block)
__global__ __launch_bounds__(256) void sl_fill_cols16_quads_kernel(uint64_t n_rows, uint64_t n_slices, uint64_t row_offset,
                                                                   const uint32_t *slice_ptr, const uint32_t *row_len,
                                                                   const uint32_t *cols, uint16_t *cols16)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    const uint64_t gi = row_offset + i;
    uint32_t len = i < n_rows ? row_len[i] : 0u;
    if (len == SL_LONG_SENTINEL) len = 0u;
    const uint32_t q0 = slice_ptr[s], q1 = slice_ptr[s + 1];
    for (uint32_t k = 0; k < (q1 - q0) * 2; ++k) {
        const uint64_t at = sl_col_slot(q0, q1, k, lane);
        const int delta = k < len ? (int)((long long)cols[at] - (long long)gi) : 0;
        cols16[at] = (uint16_t)(int16_t)delta;
    }
}

// a6 + a7: one pass over the slice layout.  Per row, in stored order:
//   diag = |a_ii| of the LAST diagonal entry seen (0 if none), off += |a_ij|   (matrix/mod.rs:467-485)
//   d    = that diagonal entry; missing or |d| < 1e-14 is an error            (neumann.rs:172-188)
// status[0] bits: 1 = some row not dominant, 2 = missing diagonal, 4 = near-zero diagonal
// status[1..3] = smallest offending row index per class (for the message).
__global__ __launch_bounds__(256) void sl_diag_kernel(uint64_t n_rows, uint64_t n_slices, uint64_t row_offset,
                                                      const uint32_t *slice_ptr, const uint32_t *row_len,
                                                      const uint32_t *cols, const double *vals, double *dinv,
                                                      unsigned long long *status)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    const uint32_t q0 = slice_ptr[s], q1 = slice_ptr[s + 1];
    if (row_len[i] == SL_LONG_SENTINEL) return;     // sl_long_diag_kernel handles it
    const uint32_t len = row_len[i];
    const uint32_t gi = (uint32_t)(row_offset + i);
    double diag_abs = 0.0, off = 0.0, d = 0.0;
    bool found = false;
    uint32_t ndiag = 0;
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t c = cols[sl_col_slot(q0, q1, k, lane)];
        const double v = vals[sl_val_slot(q0, k, lane)];
        if (c == gi) { diag_abs = fabs(v); d = v; found = true; ++ndiag; }
        else off = __dadd_rn(off, fabs(v));
    }
    if (ndiag > 1) {        // duplicated diagonal: SparseMatrix::get is a binary search (sparse.rs:142-155) — take the entry IT lands on
        uint32_t lo = 0, hi = len;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2;
            const uint32_t c = cols[sl_col_slot(q0, q1, mid, lane)];
            if (c == gi) { d = vals[sl_val_slot(q0, mid, lane)]; break; }
            if (c < gi) lo = mid + 1; else hi = mid;
        }
    }
    if (i >= n_rows) return;
    if (diag_abs < off) { atomicOr(&status[0], 1ull); atomicMin(&status[1], (unsigned long long)i); }
    if (!found) { atomicOr(&status[0], 2ull); atomicMin(&status[2], (unsigned long long)i); }
    else if (fabs(d) < 1e-14) { atomicOr(&status[0], 4ull); atomicMin(&status[3], (unsigned long long)i); }
    if (dinv) dinv[i] = (found && fabs(d) >= 1e-14) ? 1.0 / d : 0.0;
}

// duplicated diagonal entries: the value SparseMatrix::get's binary search lands on (sparse.rs:142-155)
__device__ __forceinline__ void sl_bsearch_diag(uint32_t lo, uint32_t hi, const uint32_t *col_idx, const double *values, uint32_t gi, double *d)
{
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (col_idx[mid] == gi) { *d = values[mid]; return; }
        if (col_idx[mid] < gi) lo = mid + 1; else hi = mid;
    }
}

// the same a6 + a7 rules for the long rows, over their raw CSR entries (thread per row; one-off)
__global__ void sl_long_diag_kernel(uint32_t n_long, const uint32_t *long_rows, uint64_t row_offset, const uint32_t *row_ptr,
                                    const uint32_t *col_idx, const double *values, double *dinv, unsigned long long *status)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_long) return;
    const uint32_t i = long_rows[t];
    const uint32_t gi = (uint32_t)(row_offset + i);
    double diag_abs = 0.0, off = 0.0, d = 0.0;
    bool found = false;
    uint32_t ndiag = 0;
    for (uint32_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const double v = values[k];
        if (col_idx[k] == gi) { diag_abs = fabs(v); d = v; found = true; ++ndiag; }
        else off = __dadd_rn(off, fabs(v));
    }
    if (ndiag > 1) sl_bsearch_diag(row_ptr[i], row_ptr[i + 1], col_idx, values, gi, &d);
    if (diag_abs < off) { atomicOr(&status[0], 1ull); atomicMin(&status[1], (unsigned long long)i); }
    if (!found) { atomicOr(&status[0], 2ull); atomicMin(&status[2], (unsigned long long)i); }
    else if (fabs(d) < 1e-14) { atomicOr(&status[0], 4ull); atomicMin(&status[3], (unsigned long long)i); }
    if (dinv) dinv[i] = (found && fabs(d) >= 1e-14) ? 1.0 / d : 0.0;
}
// ConditioningInfo's two row statistics (matrix/mod.rs:548-556) in one pass over the slice layout, per row in stored order as a6:
//   factor = min over rows with off > 0 of |a_ii| / off      Matrix::diagonal_dominance_factor, matrix/mod.rs:487-514
//   radius = max over rows of |a_ii| + off                    Matrix::spectral_radius_estimate (Gershgorin), matrix/mod.rs:83-100
// Both are folds with f64::min / f64::max (a NaN operand loses) over values >= +0.0, whose IEEE bit patterns order like the numbers:
// integer atomicMin / atomicMax on the bits give the same result whatever the order the rows arrive in — exact, not "to rounding".
// out[0] starts as the bits of +inf, out[1] as the bits of 0.0.
__device__ __forceinline__ void sl_cond_fold(double diag_abs, double off, unsigned long long *out)
{
    if (off > 0.0) {
        const double f = diag_abs / off;
        if (f == f) atomicMin(&out[0], (unsigned long long)__double_as_longlong(f));
    }
    const double r = __dadd_rn(diag_abs, off);
    if (r == r) atomicMax(&out[1], (unsigned long long)__double_as_longlong(r));
}
__global__ __launch_bounds__(256) void sl_cond_kernel(uint64_t n_rows, uint64_t n_slices, uint64_t row_offset, const uint32_t *slice_ptr,
                                                      const uint32_t *row_len, const uint32_t *cols, const double *vals, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    if (i >= n_rows) return;
    const uint32_t len = row_len[i];
    if (len == SL_LONG_SENTINEL) return;            // sl_long_cond_kernel
    const uint32_t q0 = slice_ptr[s], q1 = slice_ptr[s + 1];
    const uint32_t gi = (uint32_t)(row_offset + i);
    double diag_abs = 0.0, off = 0.0;
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t c = cols[sl_col_slot(q0, q1, k, lane)];
        const double v = vals[sl_val_slot(q0, k, lane)];
        if (c == gi) diag_abs = fabs(v); else off = __dadd_rn(off, fabs(v));
    }
    sl_cond_fold(diag_abs, off, out);
}
__global__ void sl_long_cond_kernel(uint32_t n_long, const uint32_t *long_rows, uint64_t row_offset, const uint32_t *row_ptr,
                                    const uint32_t *col_idx, const double *values, unsigned long long *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_long) return;
    const uint32_t i = long_rows[t];
    const uint32_t gi = (uint32_t)(row_offset + i);
    double diag_abs = 0.0, off = 0.0;
    for (uint32_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
        const double v = values[k];
        if (col_idx[k] == gi) diag_abs = fabs(v); else off = __dadd_rn(off, fabs(v));
    }
    sl_cond_fold(diag_abs, off, out);
}
__global__ void sl_long_collect_kernel(uint64_t n_rows, const uint32_t *row_len, uint32_t *list, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += stride)
        if (row_len[i] == SL_LONG_SENTINEL) list[atomicAdd(count, 1u)] = (uint32_t)i;
}
sl_status sl_sort_keys_u32(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n, hipStream_t s);

// transpose (CSR of A^T, values included, rows of each column ascending): histogram of
// columns -> tptr; stable radix sort of (col, entry id) pairs -> entry order; gather.
__global__ void sl_col_count_kernel(uint64_t nnz, const uint32_t *col_idx, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += stride)
        atomicAdd(&count[col_idx[k] + 1], 1u);
}
__global__ void sl_iota_kernel(uint64_t n, uint32_t *out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) out[k] = (uint32_t)k;
}
__global__ void sl_transpose_gather_kernel(uint64_t nnz, uint64_t n_rows, const uint32_t *row_ptr, const double *values,
                                           const uint32_t *entry, uint32_t *trow, double *tval)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < nnz; p += stride) {
        const uint32_t k = entry[p];
        // row of entry k: largest i with row_ptr[i] <= k
        uint64_t lo = 0, hi = n_rows;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (row_ptr[mid] <= k) lo = mid; else hi = mid;
        }
        trow[p] = (uint32_t)lo;
        tval[p] = values[k];
    }
}

sl_status sl_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                            uint64_t n, int end_bit, hipStream_t s);

// ---- column-panel layout ------------------------------------------------------------------------------
// sort key of every CSR entry: (tile of its row) * n_panels + (panel of its column); entries of long rows (served by the
// long-row kernel) get the largest key and fall off the end.  Also the row inside the tile, for the gather pass below.
__global__ __launch_bounds__(256) void sl_panel_keys_kernel(uint64_t n_rows, uint32_t n_panels, uint32_t tile_rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                                                            const uint32_t *row_len, uint32_t *key, uint16_t *rowl)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t i = wave; i < n_rows; i += nwaves) {                        // a wave per row: coalesced over the row's entries
        const uint32_t s = row_ptr[i], e = row_ptr[i + 1];
        const bool is_long = row_len[i] == SL_LONG_SENTINEL;
        const uint32_t base = (uint32_t)(i / tile_rows) * n_panels;
        const uint16_t rl = (uint16_t)(i % tile_rows);
        for (uint32_t k = s + lane; k < e; k += 64u) {
            key[k] = is_long ? 0xffffffffu : base + (col_idx[k] >> SL_PANEL_COL_BITS);
            rowl[k] = rl;
        }
    }
}
// entries per tile (long rows excluded)
__global__ __launch_bounds__(256) void sl_panel_tile_count_kernel(uint64_t n_rows, uint64_t n_tiles, uint32_t tile_rows, const uint32_t *row_ptr, const uint32_t *row_len,
                                                                  uint32_t *count)
{
    const uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    uint32_t acc = 0;
    for (uint32_t r = threadIdx.x; r < tile_rows; r += 256) {
        const uint64_t i = t * tile_rows + r;
        if (i < n_rows && row_len[i] != SL_LONG_SENTINEL) acc += row_ptr[i + 1] - row_ptr[i];
    }
    __shared__ uint32_t red[4];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) count[t] = red[0] + red[1] + red[2] + red[3];
}
// the streams: tile t takes sorted positions [src[t], src[t] + count) and lands at [dst[t], dst[t + 1]), the tail padded with entries
// that add 0 * t[0] to a slot of the sums nobody reads
__global__ __launch_bounds__(256) void sl_panel_fill_kernel(uint64_t n_tiles, const uint32_t *src, const uint32_t *dst, const uint32_t *perm,
                                                            const uint16_t *rowl, const uint32_t *col_idx, const double *values,
                                                            uint16_t *pan_row, uint32_t *pan_col, double *pan_val)
{
    const uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    const uint32_t s0 = src[t], cnt = src[t + 1] - s0, d0 = dst[t], padded = dst[t + 1] - d0;
    for (uint32_t e = threadIdx.x; e < padded; e += 256) {
        if (e < cnt) {
            const uint32_t k = perm[s0 + e];
            pan_row[d0 + e] = rowl[k]; pan_col[d0 + e] = col_idx[k]; pan_val[d0 + e] = values[k];
        } else {
            pan_row[d0 + e] = (uint16_t)(SL_PANEL_TILE + (e & 63u)); pan_col[d0 + e] = 0u; pan_val[d0 + e] = 0.0;   // 64 distinct unread slots
        }
    }
}

static uint32_t grid_for(uint64_t n, uint32_t block)
{
    uint64_t g = (n + block - 1) / block;
    if (g > 65535 * 16) g = 65535 * 16;
    if (g == 0) g = 1;
    return (uint32_t)g;
}

// One stable radix sort of (key, entry index) regroups the CSR entries by (tile, panel) and keeps them in (row, column) order
// inside a group — the order the running sums need.  Temporary memory: 18 bytes per entry.
static sl_status sl_build_column_panels(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_values,
                                        uint64_t n_tiles, uint32_t n_panels, hipStream_t st)
{
    const uint64_t n = m->n_rows, nnz = m->nnz;
    DevBuf key, key_out, ent_in, perm, rowl, cnt;
    SL_TRY(key.alloc_owned(nnz * 4)); SL_TRY(key_out.alloc_owned(nnz * 4)); SL_TRY(ent_in.alloc_owned(nnz * 4)); SL_TRY(perm.alloc_owned(nnz * 4));
    SL_TRY(rowl.alloc_owned(nnz * 2)); SL_TRY(cnt.alloc_owned((n_tiles + 1) * 4));
    const uint32_t g = grid_for(nnz, 256) > 8192 ? 8192 : grid_for(nnz, 256);
    hipLaunchKernelGGL(sl_panel_keys_kernel, dim3(g), dim3(256), 0, st, n, n_panels, (uint32_t)SL_PANEL_TILE, d_row_ptr, d_col_idx, m->d_row_len, key.as<uint32_t>(), rowl.as<uint16_t>());
    hipLaunchKernelGGL(sl_iota_kernel, dim3(g), dim3(256), 0, st, nnz, ent_in.as<uint32_t>());
    hipLaunchKernelGGL(sl_panel_tile_count_kernel, dim3((uint32_t)n_tiles), dim3(256), 0, st, n, n_tiles, (uint32_t)SL_PANEL_TILE, d_row_ptr, m->d_row_len, cnt.as<uint32_t>());
    std::vector<uint32_t> count(n_tiles + 1), src(n_tiles + 1), dst(n_tiles + 1);
    SL_TRY(sl_read_back(count.data(), cnt.p, n_tiles * 4, st));
    int bits = 32;                                                             // long rows carry the key 0xffffffff
    if (!m->n_long) { bits = 1; while (bits < 32 && (1ull << bits) < n_tiles * n_panels) ++bits; }
    SL_TRY(sl_sort_pairs_u32(key.as<uint32_t>(), key_out.as<uint32_t>(), ent_in.as<uint32_t>(), perm.as<uint32_t>(), nnz, bits, st));   // synchronises
    uint64_t run_s = 0, run_d = 0;
    for (uint64_t t = 0; t < n_tiles; ++t) {
        src[t] = (uint32_t)run_s; dst[t] = (uint32_t)run_d;
        run_s += count[t];
        run_d += (count[t] + SL_PANEL_CHUNK - 1) / SL_PANEL_CHUNK * SL_PANEL_CHUNK;
    }
    if (run_s > nnz || (!m->n_long && run_s != nnz))                          // the tiles' counts are the entries of the slice layout: all of them, or all but the long rows'
        return sl_fail(SL_DEVICE_ERROR, "column panels: the tiles' entry counts as read back sum to %llu, the matrix holds %llu entries", (unsigned long long)run_s, (unsigned long long)nnz);
    if (run_d > 0xfffffff0ull) return SL_OK;                                   // would not fit 32-bit stream offsets: stay without panels
    src[n_tiles] = (uint32_t)run_s; dst[n_tiles] = (uint32_t)run_d;
    DevBuf dsrc;
    SL_TRY(dsrc.alloc_owned((n_tiles + 1) * 4));
    SL_HIP(sl_malloc(&m->d_pan_tile_ptr, (n_tiles + 1) * 4));
    SL_HIP(sl_malloc(&m->d_pan_row, (run_d ? run_d : 1) * 2));
    SL_HIP(sl_malloc(&m->d_pan_col, (run_d ? run_d : 1) * 4));
    SL_HIP(sl_malloc(&m->d_pan_val, (run_d ? run_d : 1) * 8));
    SL_TRY(sl_upload(dsrc.p, src.data(), (n_tiles + 1) * 4, st));
    SL_TRY(sl_upload(m->d_pan_tile_ptr, dst.data(), (n_tiles + 1) * 4, st));
    hipLaunchKernelGGL(sl_panel_fill_kernel, dim3((uint32_t)n_tiles), dim3(256), 0, st, n_tiles, dsrc.as<uint32_t>(), m->d_pan_tile_ptr, perm.as<uint32_t>(),
                       rowl.as<uint16_t>(), d_col_idx, d_values, m->d_pan_row, m->d_pan_col, m->d_pan_val);
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(st));
    m->n_pan_tiles = n_tiles; m->pan_entries = run_d;
    uint32_t longest = 0;
    for (uint64_t t = 0; t < n_tiles; ++t) longest = std::max(longest, count[t]);
    m->pan_balanced = n_tiles > 0 && (double)longest * (double)n_tiles <= 1.1 * (double)run_s;
    m->device_bytes += run_d * 14 + (n_tiles + 1) * 4;
    return SL_OK;
}

// keys and per-tile entry counts for the paced layout: groups of SL_PW_GROUP rows dealt round robin to n_tiles tiles
__global__ __launch_bounds__(256) void sl_pw_keys_kernel(uint64_t n_rows, uint32_t n_panels, uint32_t n_tiles, const uint32_t *row_ptr, const uint32_t *col_idx,
                                                         const uint32_t *row_len, uint32_t *key, uint16_t *rowl, uint32_t *count, uint32_t deal, uint32_t gpt,
                                                         uint32_t pbits, const uint32_t *span_g0, const uint32_t *span_phys, uint32_t n_spans)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t i = wave; i < n_rows; i += nwaves) {                        // a wave per row: coalesced over the row's entries
        const uint32_t s = row_ptr[i], e = row_ptr[i + 1];
        const bool is_long = row_len[i] == SL_LONG_SENTINEL;
        // group q of SL_PW_GROUP rows: span q / (deal * gpt) owns `deal` tiles (gpt = groups per tile), its groups go round them
        const uint64_t q = i / SL_PW_GROUP;
        uint64_t span = q / ((uint64_t)deal * gpt), ql = q % ((uint64_t)deal * gpt);
        if (span_g0) {                                                       // explicit spans (a few dozen): span_g0[s] <= q < span_g0[s + 1]
            uint32_t sp = 0;
            while (sp + 1 < n_spans && q >= span_g0[sp + 1]) ++sp;
            span = span_phys[sp];                                            // its place in the launch (edge-first rounds)
            ql = q - span_g0[sp];
        }
        const uint32_t tile = (uint32_t)(span * deal + ql % deal);
        const uint16_t rl = (uint16_t)((ql / deal) * SL_PW_GROUP + i % SL_PW_GROUP);
        for (uint32_t k = s + lane; k < e; k += 64u) {
            key[k] = is_long ? 0xffffffffu : tile * n_panels + (col_idx[k] >> pbits);
            rowl[k] = rl;
        }
        if (lane == 0 && !is_long && e > s) atomicAdd(&count[tile], e - s);
    }
}

// ---- paced column-panel layout (sl_internal.hpp, sl_pw_kernel) --------------------------------------------------------------
// One wave per tile walks the tile's sorted entries (perm[src[t] ..]) 64 at a time.  An entry whose super-panel (col >> 20) lies d
// steps beyond its predecessor's needs d - 1 bridging entries in front of it (the step bit moves one super-panel at a time).
// WRITE = false: pads[t] = number of bridging entries of tile t.  WRITE = true: the stream is written, chunk-transposed, and the
// tail of the last chunk filled with padding entries (value 0, spare row slot, no step).
__device__ __forceinline__ void sl_pw_store(uint32_t *idx, double *val, uint64_t pos, uint32_t word, double v)
{
    const uint64_t chunk0 = pos & ~255ull;
    const uint32_t e = (uint32_t)(pos & 255u), u = e >> 6, l = e & 63u;
    idx[chunk0 + l * 4 + u] = word;
    val[chunk0 + (u >> 1) * 128 + l * 2 + (u & 1u)] = v;
}
template <bool WRITE>
__global__ __launch_bounds__(256) void sl_pw_fill_kernel(uint64_t n_tiles, uint32_t rpw, const uint32_t *src, const uint32_t *dst_chunks, const uint32_t *perm,
                                                         const uint16_t *rowl, const uint32_t *col_idx, const double *values, uint32_t *pads,
                                                         uint32_t *idx, double *val)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t t = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const uint32_t s0 = src[t], cnt = src[t + 1] - s0;
    uint32_t prev_sp = 0, pad_run = 0;                         // super-panel of the entry before this group of 64; bridging entries so far
    const uint64_t out0 = WRITE ? (uint64_t)dst_chunks[t] * 256 : 0;
    for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool in = e < cnt;
        uint32_t k = 0, sp = prev_sp, c = 0;
        if (in) { k = perm[s0 + e]; c = col_idx[k]; sp = c >> SL_PW_SP_BITS; }
        uint32_t left = __shfl_up(sp, 1);
        if (lane == 0) left = prev_sp;
        const uint32_t step = in ? sp - left : 0u;             // entries are sorted by panel: never negative
        const uint32_t bridge = step > 1u ? step - 1u : 0u;
        uint32_t incl = bridge;                                // inclusive scan of the bridging entries inside the group
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (lane >= (uint32_t)o) incl += v; }
        if (WRITE && in) {
            const uint64_t pos = out0 + e + pad_run + incl;    // behind its own bridging entries
            for (uint32_t b = 0; b < bridge; ++b)
                sl_pw_store(idx, val, pos - bridge + b, (rpw << SL_PW_ROW_SHIFT) | (1u << SL_PW_SP_BITS), 0.0);
            sl_pw_store(idx, val, pos, ((uint32_t)rowl[k] << SL_PW_ROW_SHIFT) | ((step ? 1u : 0u) << SL_PW_SP_BITS) | (c & ((1u << SL_PW_SP_BITS) - 1u)), values[k]);
        }
        pad_run += __shfl(incl, 63);
        const uint32_t last = cnt - e0 < 64u ? cnt - e0 - 1u : 63u;
        prev_sp = __shfl(sp, last);
    }
    if (!WRITE) { if (lane == 0) pads[t] = pad_run; return; }
    const uint64_t end = out0 + cnt + pad_run, stop = (uint64_t)dst_chunks[t + 1] * 256;
    for (uint64_t pos = end + lane; pos < stop; pos += 64) sl_pw_store(idx, val, pos, rpw << SL_PW_ROW_SHIFT, 0.0);
}

// returns SL_OK with m->d_pw_idx == nullptr when the matrix does not qualify (unbalanced tiles, too few rows per wave ...)
// band_pbits = 0: uniform columns (panels of 2^16 columns, row groups dealt among all tiles).  band_pbits > 0: a wide band — panels of
// 2^band_pbits columns, row groups dealt among the 16 tiles of one block (sl_internal.hpp)
static sl_status sl_build_paced_panels(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_values, hipStream_t st,
                                       uint32_t band_pbits = 0)
{
    const uint64_t n = m->n_rows, nnz = m->nnz;
    int dev = 0;
    hipDeviceProp_t prop;
    SL_HIP(hipGetDevice(&dev));
    SL_HIP(hipGetDeviceProperties(&prop, dev));
    uint64_t cus = prop.multiProcessorCount > 0 ? (uint64_t)prop.multiProcessorCount : 256;
    // test knobs (read per call): SL_PW_CUS = pretend the device has this many CUs (several rounds on small systems),
    // SL_PW_FORCE=1 = build the paced layout whatever the tile size and balance
    if (const char *e = getenv("SL_PW_CUS")) { const long v = atol(e); if (v > 0) cus = std::min<uint64_t>(cus, (uint64_t)v); }
    const bool force = getenv("SL_PW_FORCE") && getenv("SL_PW_FORCE")[0] == '1';
    const uint64_t waves = cus * SL_PW_WAVES;
    const uint64_t n_groups = (n + SL_PW_GROUP - 1) / SL_PW_GROUP, max_groups = SL_PW_MAX_ROWS / SL_PW_GROUP;
    const uint64_t rounds = (n_groups + waves * max_groups - 1) / (waves * max_groups);
    uint64_t n_tiles = std::min<uint64_t>(rounds * waves, n_groups);                        // every tile of every round carries work
    if (band_pbits) n_tiles = (n_tiles + SL_PW_WAVES - 1) / SL_PW_WAVES * SL_PW_WAVES;       // whole blocks: a block's 16 tiles share its rows
    // XCD-local spans (sl_matrix::pw_xcd): columns bounded by a bandwidth far beyond the L2 but well inside the matrix — a span of rows is
    // dealt among the tiles of ONE XCD's blocks of one round, and that L2 sees the columns of those rows +- w only.  What it buys is entries
    // per (tile, panel), not fewer first touches (those get dearer as they get fewer: no other XCD has pulled the line into the memory-side
    // cache a moment earlier).  Measured at 10^7 x 16, ms per step without / with: ONE RANK'S ROWS of the 8 * 10^7 system at w = 2.5 * 10^6
    // (15 million columns walked = 229 panels, 85 entries per tile and panel) 1.159 / 1.046; the whole 10^7 system (153 panels, 127 entries)
    // w = 6 * 10^5 0.944 / 0.962, 10^6 0.946 / 0.959, 2.5 * 10^6 0.953 / 0.936 — so: only where the tiles would meet fewer than ~110 entries
    // per panel and the span cuts the panels by 40 % or more.  SL_PW_XCD = G: the number of L2s to assume (forced builds: tests on pretended
    // devices; 0 = never).
    uint32_t xcd = 0;
    if (!band_pbits && m->bandwidth != ~0ull && rounds * waves <= n_groups && nnz) {
        uint32_t G = 8;
        if (const char *e = getenv("SL_PW_XCD")) G = (uint32_t)atoi(e);
        if (G >= 2 && cus % G == 0) {
            const uint64_t span_cols = (waves / G) * (uint64_t)SL_PW_MAX_ROWS + 2 * m->bandwidth, walked = std::min<uint64_t>(m->n_cols, n + 2 * m->bandwidth);
            const double per_panel = (double)nnz / (double)(rounds * waves) / (double)((walked >> SL_PANEL_COL_BITS) + 1);
            if (force ? getenv("SL_PW_XCD") != nullptr : (per_panel < 110.0 && 5 * span_cols <= 3 * walked)) xcd = G;
        }
    }
    if (xcd) n_tiles = (n_tiles + waves / xcd - 1) / (waves / xcd) * (waves / xcd);         // whole spans
    const uint32_t deal = band_pbits ? (uint32_t)SL_PW_WAVES : xcd ? (uint32_t)(waves / xcd) : (uint32_t)n_tiles;
    uint32_t gpt = (uint32_t)((n_groups + n_tiles - 1) / n_tiles);                          // groups per tile
    // XCD-local spans are explicit (d_pw_span_tab).  EDGE FIRST where the matrix leaves an interior: E spans at either end sized to cover
    // the rows within the bandwidth of the range's ends (+ a group: the ends need not fall on group boundaries), the rest shared by the
    // S - 2 E interior spans; launch order = low edge, high edge, interior, so that the edge takes the first ceil(2 E / G) rounds (the
    // first interior spans ride along when 2 E is no multiple of G).  Only if the interior spans grow by no more than 5 % for it (a round
    // is as long as its longest span: edge spans shorter than their share leave the interior more rows) and at least one round has no
    // edge spans.
    std::vector<uint32_t> span_g0, phys_of_log, span_tab;
    uint32_t edge_rounds = 0;
    uint64_t edge_rows = 0;
    if (xcd) {
        const uint64_t S = n_tiles / deal, cap = (uint64_t)deal * max_groups, nominal = (n_groups + S - 1) / S;
        const uint64_t bwg = (m->bandwidth + SL_PW_GROUP - 1) / SL_PW_GROUP + 1;
        const uint64_t E = (bwg + nominal - 1) / nominal, edge_g = (bwg + E - 1) / E;
        uint64_t interior_g = 0;
        bool edge_first = 2 * E < S && (2 * E + xcd - 1) / xcd < S / xcd && n_groups > 2 * E * edge_g;
        if (edge_first) {
            interior_g = (n_groups - 2 * E * edge_g + (S - 2 * E) - 1) / (S - 2 * E);
            edge_first = interior_g <= cap && edge_g <= cap && 20 * interior_g <= 21 * nominal;      // (edge_g <= nominal by construction)
        }
        span_g0.resize(S + 1); phys_of_log.resize(S); span_tab.resize(2 * S);
        if (edge_first) {
            for (uint64_t q = 0; q <= E; ++q) span_g0[q] = (uint32_t)(q * edge_g);
            for (uint64_t q = 1; q <= S - 2 * E; ++q) span_g0[E + q] = (uint32_t)std::min<uint64_t>(E * edge_g + q * interior_g, n_groups - E * edge_g);
            for (uint64_t q = 1; q <= E; ++q) span_g0[S - E + q] = (uint32_t)(n_groups - E * edge_g + q * edge_g);
            for (uint64_t q = 0; q < E; ++q) { phys_of_log[q] = (uint32_t)q; phys_of_log[S - E + q] = (uint32_t)(E + q); }
            for (uint64_t q = E; q < S - E; ++q) phys_of_log[q] = (uint32_t)(E + q);
            edge_rounds = (uint32_t)((2 * E + xcd - 1) / xcd);
            edge_rows = E * edge_g * SL_PW_GROUP - SL_PW_GROUP;             // from either end, whatever the alignment of the last group
            gpt = (uint32_t)((std::max(edge_g, interior_g) + deal - 1) / deal);
        } else {
            for (uint64_t q = 0; q <= S; ++q) span_g0[q] = (uint32_t)std::min<uint64_t>(q * nominal, n_groups);
            for (uint64_t q = 0; q < S; ++q) phys_of_log[q] = (uint32_t)q;
            gpt = (uint32_t)((nominal + deal - 1) / deal);
        }
        for (uint64_t q = 0; q < S; ++q) { span_tab[2 * phys_of_log[q]] = span_g0[q]; span_tab[2 * phys_of_log[q] + 1] = span_g0[q + 1] - span_g0[q]; }
    }
    const uint32_t rpw = gpt * SL_PW_GROUP;
    // wide band: a block of 16 tiles owns deal * rpw consecutive rows and walks the columns of those rows +- the bandwidth.  The panel
    // width is the smallest (from 2^9 = 4 KB of vector) that gives a tile ~90 entries per panel (tools/ab_band_paced*.sh, n = 10^7 x 16,
    // ms per step 2^9 / 2^10 / 2^11: w = 12 000 0.473 / 0.536 / 0.677, 32 768 0.499 / 0.532 / 0.675, 100 000 0.634 / 0.592 / 0.731);
    // SL_PW_BAND_AUTO (0xff): that rule, any other value: the width as given (experiments)
    uint32_t pbits = band_pbits ? band_pbits : (uint32_t)SL_PANEL_COL_BITS;
    if (!band_pbits) if (const char *e = getenv("SL_PW_PBITS")) { const int v = atoi(e); if (v >= 12 && v <= 20) pbits = (uint32_t)v; }      // experiments: log2 of the panel width
    const uint64_t span_cols = (uint64_t)deal * rpw + 2 * (m->bandwidth == ~0ull ? 0 : m->bandwidth) + 1;
    const double tile_entries = (double)nnz / (double)n_tiles;
    if (band_pbits == SL_PW_BAND_AUTO) {
        pbits = 9;
        while (pbits < 10 && tile_entries * (double)(1u << pbits) / (double)span_cols < 90.0) ++pbits;
        if (tile_entries * (double)(1u << pbits) / (double)span_cols < 30.0) return SL_OK;      // too thin a band for the CU's L1 to see reuse: general kernel
    }
    const size_t lds_avail = std::max({(size_t)prop.sharedMemPerBlock, (size_t)prop.sharedMemPerBlockOptin, (size_t)prop.maxSharedMemoryPerMultiProcessor});
    if (n == 0 || lds_avail < (size_t)SL_PW_WAVES * (SL_PW_MAX_ROWS + 1) * 8 + 512) return SL_OK;
    if (rpw < 256 && !force) return SL_OK;                                       // small systems keep the dynamic tiles
    const uint64_t n_panels = (m->n_cols + (1ull << pbits) - 1) >> pbits;
    if (n_tiles * n_panels >= 0xfffffff0ull || n_panels >= (1ull << 20)) return SL_OK;
    DevBuf key, key_out, ent_in, perm, rowl, cnt, pads;
    SL_TRY(key.alloc_owned(nnz * 4)); SL_TRY(key_out.alloc_owned(nnz * 4)); SL_TRY(ent_in.alloc_owned(nnz * 4)); SL_TRY(perm.alloc_owned(nnz * 4));
    SL_TRY(rowl.alloc_owned(nnz * 2)); SL_TRY(cnt.alloc_owned((n_tiles + 1) * 4)); SL_TRY(pads.alloc_owned((n_tiles + 1) * 4));
    const uint32_t g = grid_for(nnz, 256) > 8192 ? 8192 : grid_for(nnz, 256);
    SL_HIP(hipMemsetAsync(cnt.p, 0, (n_tiles + 1) * 4, st));
    DevBuf d_span_g0, d_span_phys;
    if (xcd) {
        SL_TRY(d_span_g0.alloc_owned(span_g0.size() * 4)); SL_TRY(d_span_phys.alloc_owned(phys_of_log.size() * 4));
        SL_TRY(sl_upload(d_span_g0.p, span_g0.data(), span_g0.size() * 4, st));
        SL_TRY(sl_upload(d_span_phys.p, phys_of_log.data(), phys_of_log.size() * 4, st));
    }
    hipLaunchKernelGGL(sl_pw_keys_kernel, dim3(g), dim3(256), 0, st, n, (uint32_t)n_panels, (uint32_t)n_tiles, d_row_ptr, d_col_idx, m->d_row_len,
                       key.as<uint32_t>(), rowl.as<uint16_t>(), cnt.as<uint32_t>(), deal, gpt, pbits, d_span_g0.as<uint32_t>(), d_span_phys.as<uint32_t>(),
                       (uint32_t)phys_of_log.size());
    std::vector<uint32_t> count(n_tiles + 1), src(n_tiles + 1), dst(n_tiles + 1), hpads(n_tiles + 1);
    SL_TRY(sl_read_back(count.data(), cnt.p, n_tiles * 4, st));
    uint64_t total = 0; uint32_t longest = 0;
    for (uint64_t t = 0; t < n_tiles; ++t) { src[t] = (uint32_t)total; total += count[t]; longest = std::max(longest, count[t]); }
    src[n_tiles] = (uint32_t)total;
    if (total > nnz || (!m->n_long && total != nnz))
        return sl_fail(SL_DEVICE_ERROR, "paced panels: the tiles' entry counts as read back sum to %llu, the matrix holds %llu entries", (unsigned long long)total, (unsigned long long)nnz);
    // persistent blocks with a fixed deal of tiles: only for matrices whose tiles carry (nearly) equal work
    if (total == 0 || (!force && (double)longest * (double)n_tiles > 1.1 * (double)total)) return SL_OK;
    hipLaunchKernelGGL(sl_iota_kernel, dim3(g), dim3(256), 0, st, nnz, ent_in.as<uint32_t>());
    int bits = 32;
    if (!m->n_long) { bits = 1; while (bits < 32 && (1ull << bits) < n_tiles * n_panels) ++bits; }
    // stable: inside a (tile, panel) group the entries keep their CSR order = (row, column) order, and a tile's slots grow with its rows
    SL_TRY(sl_sort_pairs_u32(key.as<uint32_t>(), key_out.as<uint32_t>(), ent_in.as<uint32_t>(), perm.as<uint32_t>(), nnz, bits, st));   // synchronises
    DevBuf dsrc, ddst;
    SL_TRY(dsrc.alloc_owned((n_tiles + 1) * 4));
    SL_TRY(sl_upload(dsrc.p, src.data(), (n_tiles + 1) * 4, st));
    const uint32_t fg = (uint32_t)((n_tiles + 3) / 4);
    hipLaunchKernelGGL((sl_pw_fill_kernel<false>), dim3(fg), dim3(256), 0, st, n_tiles, rpw, dsrc.as<uint32_t>(), (const uint32_t *)nullptr, perm.as<uint32_t>(),
                       rowl.as<uint16_t>(), d_col_idx, d_values, pads.as<uint32_t>(), (uint32_t *)nullptr, (double *)nullptr);
    SL_TRY(sl_read_back(hpads.data(), pads.p, n_tiles * 4, st));
    uint64_t chunks = 0;
    for (uint64_t t = 0; t < n_tiles; ++t) { dst[t] = (uint32_t)chunks; chunks += ((uint64_t)count[t] + hpads[t] + SL_PANEL_CHUNK - 1) / SL_PANEL_CHUNK; }
    dst[n_tiles] = (uint32_t)chunks;
    if (chunks * 256 > 0xfffffff0ull) return SL_OK;
    SL_HIP(sl_malloc(&m->d_pw_tile_ptr, (n_tiles + 1) * 4));
    SL_HIP(sl_malloc(&m->d_pw_idx, (chunks ? chunks : 1) * 256 * 4));
    SL_HIP(sl_malloc(&m->d_pw_val, (chunks ? chunks : 1) * 256 * 8));
    SL_TRY(sl_upload(m->d_pw_tile_ptr, dst.data(), (n_tiles + 1) * 4, st));
    hipLaunchKernelGGL((sl_pw_fill_kernel<true>), dim3(fg), dim3(256), 0, st, n_tiles, rpw, dsrc.as<uint32_t>(), m->d_pw_tile_ptr, perm.as<uint32_t>(),
                       rowl.as<uint16_t>(), d_col_idx, d_values, (uint32_t *)nullptr, m->d_pw_idx, m->d_pw_val);
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(st));
    m->n_pw_tiles = n_tiles; m->pw_chunks = chunks; m->pw_rpw = rpw;
    // a wave moves through about n_panels * 256 / (entries per tile) panels per chunk of its stream.  The lead it is allowed over the
    // slowest wave of its block: two thirds of that, at least one panel — since the accumulation got cheaper, waves that stay closer
    // together win (tools/ab_slack.sh, n = 10^7 x 16, lead 1 / 2 / 3 / 4 / 6 panels: 0.950 / 0.966 / 0.966 / 0.985 / 1.000 ms; the same
    // order at 10^6 x 8 .. 5 * 10^6 x 16; at two panels per chunk (10^7 x 8, 2 * 10^7 x 16) leads 1..3 lie within 1 %)
    // (the panels a tile's entries are spread over: all of them for uniform columns; for a bandwidth well inside the matrix — a rank's rows
    // of a larger system above all — only those of its rows +- w: taking all n_cols there let the waves drift 5 panels apart where 1 was
    // meant, n = 10^7 rows of 8 * 10^7, w = 2.5 * 10^6: 1.045 -> see profiles/r03_c5_locality.txt)
    const uint64_t tile_cols = m->bandwidth == ~0ull ? m->n_cols : std::min<uint64_t>(m->n_cols, (xcd ? (uint64_t)deal * rpw : n) + 2 * m->bandwidth);
    const uint64_t tile_panels = (tile_cols >> pbits) + 1;
    m->pw_slack = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (2 * 256 * tile_panels * n_tiles + total * 3 / 2) / (total * 3)));
    if (band_pbits) {
        // a tile walks only the panels of its block's window; the lead = the panels it crosses per chunk of 256 entries, held to
        // what the L1 keeps beside the stream (~16 KB of vector: w = 32 768 at 2^9 lead 1 / 2 / 4 0.533 / 0.499 / 0.513 ms, w = 100 000 at
        // 2^10 lead 1 / 2 / 3 0.592 / 0.619 / 0.630)
        const double per_chunk = 256.0 * (double)(span_cols >> pbits) * (double)n_tiles / (double)total;
        const uint32_t cap = std::max<uint32_t>(1u, (2048u >> pbits) ? (2048u >> pbits) - 1u : 1u);
        m->pw_slack = std::min<uint32_t>(cap, std::max<uint32_t>(1u, (uint32_t)(per_chunk + 0.5)));
    }
    m->pw_deal = deal; m->pw_pbits = pbits; m->pw_band = band_pbits != 0; m->pw_xcd = xcd;
    if (xcd) {
        SL_HIP(sl_malloc(&m->d_pw_span_tab, span_tab.size() * 4));
        SL_HIP(hipMemcpy(m->d_pw_span_tab, span_tab.data(), span_tab.size() * 4, hipMemcpyHostToDevice));
        m->pw_edge_rounds = edge_rounds; m->pw_edge_rows = edge_rows;
    }
    m->pw_blocks = (uint32_t)std::min<uint64_t>(cus, (n_tiles + SL_PW_WAVES - 1) / SL_PW_WAVES);       // (xcd: n_tiles >= cus * 16, whole spans: = cus)
    m->device_bytes += chunks * 256 * 12 + (n_tiles + 1) * 4;
    return SL_OK;
}


// ---- order-free column stream (sl_internal.hpp, sl_pwr_kernel; SL_MATRIX_ORDER_ANY) ---------------------------------------------------
sl_status sl_sort_pairs_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, int end_bit, hipStream_t s);

// a wave per row: key = block tile << 32 | column for the off-diagonal entries, block tile = n_btiles (sorts behind every stream) for
// the diagonal, whose values are summed into diag[row]; slot = the row's place among its block tile's running sums
__global__ __launch_bounds__(256) void sl_pwr_keys_kernel(uint64_t n_rows, uint32_t n_btiles, uint64_t row_offset, const uint32_t *row_ptr, const uint32_t *col_idx,
                                                          const double *values, unsigned long long *key, uint16_t *slot, uint32_t *count, double *diag)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
    for (uint64_t i = wave; i < n_rows; i += nwaves) {
        const uint32_t s = row_ptr[i], e = row_ptr[i + 1];
        const uint64_t q = i / SL_PW_GROUP;
        const uint32_t bt = (uint32_t)(q % n_btiles);
        const uint16_t sl = (uint16_t)((q / n_btiles) * SL_PW_GROUP + i % SL_PW_GROUP);
        uint32_t off = 0;
        for (uint32_t k0 = s; k0 < e; k0 += 64u) {
            const uint32_t k = k0 + lane;
            bool offd = false;
            if (k < e) {
                const uint32_t c = col_idx[k];
                offd = (uint64_t)c != row_offset + i;
                key[k] = ((unsigned long long)(offd ? bt : n_btiles) << 32) | c;
                slot[k] = sl;
                if (!offd) atomicAdd(&diag[i], values[k]);              // one entry per row unless the matrix carries duplicates
            }
            off += (uint32_t)__popcll(__ballot(offd));
        }
        if (lane == 0 && off) atomicAdd(&count[bt], off);
    }
}

// One wave per block tile walks the tile's column-sorted entries 64 at a time and cuts them into chunks of at most 64 entries whose
// columns lie within 2^SL_PWR_OFF_BITS of the chunk's first.  WRITE = false: nchunks[t].  WRITE = true: the chunks are written,
// entry l of a chunk in lane l's place, each padded to 64 entries with {slot 0, the base column, 0.0}.
template <bool WRITE>
__global__ __launch_bounds__(256) void sl_pwr_fill_kernel(uint64_t n_btiles, const uint32_t *src, const uint32_t *dst_chunks, const uint32_t *perm, const uint16_t *slot,
                                                          const uint32_t *col_idx, const double *values, uint32_t *nchunks, uint32_t *idx, double *val, uint32_t *base,
                                                          uint32_t spare_slot)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t t = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= n_btiles) return;
    const uint32_t s0 = src[t], cnt = src[t + 1] - s0;
    const uint64_t c0 = WRITE ? dst_chunks[t] : 0;
    uint32_t chunk = 0, fill = 0, B = 0;
    auto close_chunk = [&]() {
        // padding: value 0, column = the chunk's base, and the tile's SPARE slot (one double behind its rows) — its product 0 * g[base] is
        // NaN when the vector holds an Inf there (a diverging iterate), and must not reach a row that never referenced that column
        if (WRITE) for (uint32_t p = fill + lane; p < SL_PWR_CHUNK; p += 64u) { idx[(c0 + chunk) * SL_PWR_CHUNK + p] = spare_slot << SL_PWR_OFF_BITS; val[(c0 + chunk) * SL_PWR_CHUNK + p] = 0.0; }
        ++chunk; fill = 0;
    };
    for (uint32_t e0 = 0; e0 < cnt; e0 += 64) {
        const uint32_t nvalid = cnt - e0 < 64u ? cnt - e0 : 64u;
        uint32_t k = 0, c = 0xffffffffu;
        if (lane < nvalid) { k = perm[s0 + e0 + lane]; c = col_idx[k]; }
        uint32_t cur = 0;
        while (cur < nvalid) {
            if (fill == 0) { B = __shfl(c, (int)cur); if (WRITE && lane == 0) base[c0 + chunk] = B; }
            const bool fits = lane >= cur && lane < nvalid && (c - B) < (1u << SL_PWR_OFF_BITS) && fill + (lane - cur) < SL_PWR_CHUNK;
            const unsigned long long m = __ballot(fits) >> cur;                    // sorted columns: the lanes that fit are a prefix
            const uint32_t take = (~m) ? (uint32_t)__builtin_ctzll(~m) : 64u - cur;
            if (WRITE && fits && lane < cur + take) {
                const uint64_t pos = (c0 + chunk) * SL_PWR_CHUNK + fill + (lane - cur);
                idx[pos] = ((uint32_t)slot[k] << SL_PWR_OFF_BITS) | (c - B);
                val[pos] = values[k];
            }
            fill += take; cur += take;
            if (cur < nvalid || fill == SL_PWR_CHUNK) close_chunk();
        }
    }
    if (fill) close_chunk();
    if (!WRITE && lane == 0) nchunks[t] = chunk;
}

// returns SL_OK with m->d_pwr_idx == nullptr when the matrix does not qualify (block tiles of unequal work, columns too sparse for the
// 17-bit offsets ...): the caller then builds the ordered layouts as usual
static sl_status sl_build_order_free_stream(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_values, hipStream_t st)
{
    const uint64_t n = m->n_rows, nnz = m->nnz;
    int dev = 0;
    hipDeviceProp_t prop;
    SL_HIP(hipGetDevice(&dev));
    SL_HIP(hipGetDeviceProperties(&prop, dev));
    uint64_t cus = prop.multiProcessorCount > 0 ? (uint64_t)prop.multiProcessorCount : 256;
    if (const char *e = getenv("SL_PW_CUS")) { const long v = atol(e); if (v > 0) cus = std::min<uint64_t>(cus, (uint64_t)v); }      // tests: several rounds on small systems
    const bool force = getenv("SL_PW_FORCE") && getenv("SL_PW_FORCE")[0] == '1';
    uint32_t max_rows = SL_PWR_MAX_ROWS;
    if (const char *e = getenv("SL_PWR_ROWS")) { const long v = atol(e); if (v >= 16) max_rows = std::min<uint32_t>(SL_PWR_MAX_ROWS, (uint32_t)v / SL_PW_GROUP * SL_PW_GROUP); }   // tests / A-B
    const size_t lds_avail = std::max({(size_t)prop.sharedMemPerBlock, (size_t)prop.sharedMemPerBlockOptin, (size_t)prop.maxSharedMemoryPerMultiProcessor});
    if (n == 0 || nnz == 0 || lds_avail < (size_t)SL_PWR_MAX_ROWS * 8 + 2048 || m->row_offset + n > m->n_cols) return SL_OK;
    const uint64_t n_groups = (n + SL_PW_GROUP - 1) / SL_PW_GROUP, max_groups = max_rows / SL_PW_GROUP;
    const uint64_t rounds = (n_groups + cus * max_groups - 1) / (cus * max_groups);
    const uint64_t n_btiles = std::min<uint64_t>(rounds * cus, n_groups);
    const uint32_t gpt = (uint32_t)((n_groups + n_btiles - 1) / n_btiles), rpb = gpt * SL_PW_GROUP;
    if (n_btiles >= 0xffffull) return SL_OK;
    if (rpb < 1024 && !force) return SL_OK;
    DevBuf key, key_out, ent_in, perm, slot, cnt, nch, diag;
    SL_TRY(key.alloc_owned(nnz * 8)); SL_TRY(key_out.alloc_owned(nnz * 8)); SL_TRY(ent_in.alloc_owned(nnz * 4)); SL_TRY(perm.alloc_owned(nnz * 4));
    SL_TRY(slot.alloc_owned(nnz * 2)); SL_TRY(cnt.alloc_owned((n_btiles + 1) * 4)); SL_TRY(nch.alloc_owned((n_btiles + 1) * 4)); SL_TRY(diag.alloc_owned(n * 8));
    const uint32_t g = grid_for(nnz, 256) > 8192 ? 8192 : grid_for(nnz, 256);
    SL_HIP(hipMemsetAsync(cnt.p, 0, (n_btiles + 1) * 4, st));
    SL_HIP(hipMemsetAsync(diag.p, 0, n * 8, st));
    hipLaunchKernelGGL(sl_pwr_keys_kernel, dim3(g), dim3(256), 0, st, n, (uint32_t)n_btiles, m->row_offset, d_row_ptr, d_col_idx, d_values,
                       key.as<unsigned long long>(), slot.as<uint16_t>(), cnt.as<uint32_t>(), diag.as<double>());
    std::vector<uint32_t> count(n_btiles + 1), src(n_btiles + 1), dst(n_btiles + 1), hch(n_btiles + 1);
    SL_TRY(sl_read_back(count.data(), cnt.p, n_btiles * 4, st));
    uint64_t total = 0; uint32_t longest = 0;
    for (uint64_t t = 0; t < n_btiles; ++t) { src[t] = (uint32_t)total; total += count[t]; longest = std::max(longest, count[t]); }
    src[n_btiles] = (uint32_t)total;
    if (total > nnz) return sl_fail(SL_DEVICE_ERROR, "column stream: the tiles' entry counts as read back sum to %llu, the matrix holds %llu entries", (unsigned long long)total, (unsigned long long)nnz);
    // persistent blocks with a fixed deal of block tiles: only for matrices whose tiles carry (nearly) equal work
    if (total == 0 || (!force && (double)longest * (double)n_btiles > 1.1 * (double)total)) return SL_OK;
    hipLaunchKernelGGL(sl_iota_kernel, dim3(g), dim3(256), 0, st, nnz, ent_in.as<uint32_t>());
    int bits = 33;
    while (bits < 64 && (1ull << (bits - 32)) <= n_btiles) ++bits;                            // tile ids 0 .. n_btiles (the diagonal's)
    SL_TRY(sl_sort_pairs_u64(key.as<uint64_t>(), key_out.as<uint64_t>(), ent_in.as<uint32_t>(), perm.as<uint32_t>(), nnz, bits, st));   // synchronises
    key.reset(); key_out.reset(); ent_in.reset();
    DevBuf dsrc;
    SL_TRY(dsrc.alloc_owned((n_btiles + 1) * 4));
    SL_TRY(sl_upload(dsrc.p, src.data(), (n_btiles + 1) * 4, st));
    const uint32_t fg = (uint32_t)((n_btiles + 3) / 4);
    hipLaunchKernelGGL((sl_pwr_fill_kernel<false>), dim3(fg), dim3(256), 0, st, n_btiles, dsrc.as<uint32_t>(), (const uint32_t *)nullptr, perm.as<uint32_t>(), slot.as<uint16_t>(),
                       d_col_idx, d_values, nch.as<uint32_t>(), (uint32_t *)nullptr, (double *)nullptr, (uint32_t *)nullptr, rpb);
    SL_TRY(sl_read_back(hch.data(), nch.p, n_btiles * 4, st));
    uint64_t chunks = 0;
    for (uint64_t t = 0; t < n_btiles; ++t) { dst[t] = (uint32_t)chunks; chunks += hch[t]; }
    dst[n_btiles] = (uint32_t)chunks;
    // columns too sparse for the offsets (chunks cut early) or streams too short: more than 12 % padding — not a matrix for this layout
    if (chunks * SL_PWR_CHUNK > 0xfffffff0ull || (!force && (double)chunks * SL_PWR_CHUNK > 1.12 * (double)total + (double)SL_PWR_CHUNK * (double)n_btiles)) return SL_OK;
    SL_HIP(sl_malloc(&m->d_pwr_tile_ptr, (n_btiles + 1) * 4));
    SL_HIP(sl_malloc(&m->d_pwr_idx, (chunks ? chunks : 1) * SL_PWR_CHUNK * 4));
    SL_HIP(sl_malloc(&m->d_pwr_val, (chunks ? chunks : 1) * SL_PWR_CHUNK * 8));
    SL_HIP(sl_malloc(&m->d_pwr_base, (chunks ? chunks : 1) * 4));
    SL_TRY(sl_upload(m->d_pwr_tile_ptr, dst.data(), (n_btiles + 1) * 4, st));
    hipLaunchKernelGGL((sl_pwr_fill_kernel<true>), dim3(fg), dim3(256), 0, st, n_btiles, dsrc.as<uint32_t>(), m->d_pwr_tile_ptr, perm.as<uint32_t>(), slot.as<uint16_t>(),
                       d_col_idx, d_values, (uint32_t *)nullptr, m->d_pwr_idx, m->d_pwr_val, m->d_pwr_base, rpb);
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(st));
    m->d_pwr_diag = static_cast<double *>(diag.release());

    m->n_pwr_tiles = n_btiles; m->pwr_chunks = chunks; m->pwr_rpb = rpb;
    m->pwr_blocks = (uint32_t)std::min<uint64_t>(cus, n_btiles);
    m->device_bytes += chunks * (SL_PWR_CHUNK * 12 + 4) + (n_btiles + 1) * 4 + n * 8;
    return SL_OK;
}

// ---- column-constant operators (sl_matrix::d_colval) ----------------------------------------------------------------------------------
// One pass over the rows: the first off-diagonal entry seen of a column leaves its value (bit pattern) in colbits[column], every other
// one compares with it; a diagonal entry must be exactly 1.  *bad != 0: not such an operator (threads leave as soon as they see it).
#define SL_COLVAL_UNSET 0xfff8c01dc01dc01dull              // a NaN payload no stored value carries
__global__ __launch_bounds__(256) void sl_fill_u64_kernel(uint64_t n, unsigned long long *p, unsigned long long v)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n) p[j] = v;
}
__global__ __launch_bounds__(256) void sl_colval_detect_kernel(uint64_t n, uint64_t row_offset, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                                                               unsigned long long *colbits, uint32_t *bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    const uint32_t s = row_ptr[i], e = row_ptr[i + 1];
    for (uint32_t k = s; k < e; ++k) {
        const uint32_t c = col_idx[k];
        const unsigned long long vb = (unsigned long long)__double_as_longlong(values[k]);
        if ((uint64_t)c == row_offset + i) { if (values[k] != 1.0) { *bad = 1u; return; } continue; }
        const unsigned long long old = atomicCAS(&colbits[c], SL_COLVAL_UNSET, vb);
        if (old != SL_COLVAL_UNSET && old != vb) { *bad = 1u; return; }
        if ((k & 63u) == 63u && __hip_atomic_load(bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    }
}
__global__ __launch_bounds__(256) void sl_colval_finish_kernel(uint64_t n_cols, unsigned long long *colbits)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n_cols && colbits[j] == SL_COLVAL_UNSET) colbits[j] = 0ull;          // a column nobody references: +0.0
}
static sl_status sl_detect_column_constant(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_values, hipStream_t st)
{
    const uint64_t n = m->n_rows;
    if (!n || !m->nnz || m->n_rows != m->n_cols || m->row_offset) return SL_OK;
    DevBuf bits, bad;
    SL_TRY(bits.alloc_owned(m->n_cols * 8)); SL_TRY(bad.alloc(16));
    SL_HIP(hipMemsetAsync(bad.p, 0, 16, st));
    // fill with the sentinel: its eight bytes are not all equal, so no memset — a tiny kernel would do; reuse the finish kernel's shape
    hipLaunchKernelGGL(sl_fill_u64_kernel, dim3((uint32_t)((m->n_cols + 255) / 256)), dim3(256), 0, st, m->n_cols, bits.as<unsigned long long>(), SL_COLVAL_UNSET);
    hipLaunchKernelGGL(sl_colval_detect_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, m->row_offset, d_row_ptr, d_col_idx, d_values,
                       bits.as<unsigned long long>(), bad.as<uint32_t>());
    uint32_t h_bad = 0;
    SL_TRY(sl_read_back(&h_bad, bad.p, 4, st));
    if (h_bad) return SL_OK;
    hipLaunchKernelGGL(sl_colval_finish_kernel, dim3((uint32_t)((m->n_cols + 255) / 256)), dim3(256), 0, st, m->n_cols, bits.as<unsigned long long>());
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(st));
    m->d_colval = static_cast<double *>(bits.release());
    m->device_bytes += m->n_cols * 8;
    sl_log(1, "matrix layout: column-constant operator with a unit diagonal (one value per column: %llu columns) — dense push rounds may run on the index words of the paced stream alone",
           (unsigned long long)m->n_cols);
    return SL_OK;
}

// 2b / 2c of the build, as a step of its own: the sorted column streams (dynamic column panels, paced panels, order-free stream) and
// the column-constant table, chosen from the statistics the slice fill left in the matrix.  The in-place mutators re-run it on the
// updated rows (sl_matrix_rebuild_column_streams) — the same choices, the same geometry, new values.
static sl_status sl_build_column_streams(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx, const double *d_values, hipStream_t st)
{
    const uint64_t n = m->n_rows, nnz = m->nnz;
    const uint64_t far_entries = m->far_entries, slice_entries = m->slice_entries, diagonal_entries = m->diagonal_entries;
    const uint64_t bytes_before = m->device_bytes;
    struct account { sl_matrix *m; uint64_t before; ~account() { m->stream_bytes = m->device_bytes - before; } } account_{m, bytes_before};
    // 2b. column panels: where neither the LDS window (bandwidth) nor the L2 (vector of a few MB) can serve the gathers
    {
        static const int env_panels = [] { const char *e = getenv("SL_COLUMN_PANELS"); return e && *e ? atoi(e) : -1; }();   // 0 never, 1 always
        const bool forced = (m->flags & SL_MATRIX_COLUMN_PANELS) || env_panels == 1 || env_panels == 3;
        const bool refused = (m->flags & SL_MATRIX_NO_COLUMN_PANELS) || env_panels == 0;
        // pays where most entries sit megabytes of vector away from their row (uniformly random columns) and the vector is far larger
        // than the L2 — measured at n = 10^7 x 16: 2.73 -> 1.71 ms; band structures, however wide, are served better by the general
        // kernel (their gathers hit L2; here the entries of a row would crowd one panel and their sums serialise): w = 10^6: 1.86 vs 2.47 ms
        // Round 2 (tools/ab_c2.sh, paced layout against the general kernel, ms per step): n = 10^6 x 8 0.087 / 0.087, 10^6 x 16
        // 0.108 / 0.146, 2 * 10^6 x 8 0.135 / 0.223, 3 * 10^6 x 16 0.290 / 0.660 — the paced layout pays from a vector of ~8 MB on;
        // the DYNAMIC tiles (fallback for unbalanced matrices) only from ~24 MB on (10^6 x 8: 0.168, worse than no panels).
        // of the off-diagonal entries of the rows the count covers (the diagonal is never far; the long rows' entries are not counted:
        // a graph whose hubs hold most of the entries — PageRank's out-link side — is judged by its other rows)
        const bool spread = 2 * far_entries > slice_entries - std::min<uint64_t>(n, slice_entries);
        const bool pays = m->n_cols >= (3ull << 20) && spread;
        // (tools/ab_small_paced.sh after the accumulation moved to the VALU, paced / general: 7 * 10^5 x 8 0.058 / 0.048, 7 * 10^5 x 16
        // 0.074 / 0.088, 10^6 x 8 0.070 / 0.086, 10^6 x 16 0.099 / 0.145, 1.5 * 10^6 x 8 0.094 / 0.151: from 9 * 10^5 columns on)
        const bool pays_paced = m->n_cols >= 900000ull && spread;
        const uint64_t n_tiles = (n + SL_PANEL_TILE - 1) / SL_PANEL_TILE;
        const uint64_t n_panels = (m->n_cols + (1ull << SL_PANEL_COL_BITS) - 1) >> SL_PANEL_COL_BITS;
        // wide bands (experiment knob SL_PW_BAND = log2 of the panel width): the paced layout with block-local rows and narrow panels
        // Wide bands — a measured bandwidth beyond the LDS window of the band kernel (w > ~9 500), gathers that the general kernel
        // serves from the L2 one request each: the paced layout with block-local rows and narrow panels makes them L1 hits
        // (w = 12 000 .. 100 000: 0.60-0.74 -> 0.47-0.59 ms at n = 10^7 x 16; from w ~ 3 * 10^5 on the band is too thin for that and
        // the general kernel — beyond ~10^6 the uniform-column layout above — keeps it).  SL_PW_BAND: 0 off, n = force panels of 2^n.
        static const int env_band = [] { const char *e = getenv("SL_PW_BAND"); return e && *e ? atoi(e) : -1; }();
        // Not for stencils: where most entries continue a diagonal from the row above, neighbouring lanes of the general kernel gather
        // from the same lines already (7-point stencil on 215^3, w = 46 225: 0.234 ms there, 0.364 ms on this layout).
        const bool band_wide = m->bandwidth != ~0ull && m->bandwidth >= 9500 && !spread && !m->n_long && nnz && nnz < 0x7fffffffull
                               && m->row_offset + n <= m->n_cols;
        const bool band_pays = n >= 1500000ull                       // n = 2^20 x 16, w = 32 768: 0.090 against 0.077 ms; 1.5 * 10^6: 0.092 / 0.098
                               && 2 * diagonal_entries < slice_entries;
        // SL_MATRIX_ORDER_ANY: where column panels pay, the order-free column stream takes their place (SL_ORDER_ANY solves run on it; exact
        // orders on such a matrix take the row-slice kernels).  Hub rows need no separate kernel there: LDS atomics do not care how many
        // entries of a row arrive at once.
        if ((m->flags & SL_MATRIX_ORDER_ANY) && !refused && (forced || pays_paced) && nnz && nnz < 0x7fffffffull) {
            const sl_status ps = sl_build_order_free_stream(m, d_row_ptr, d_col_idx, d_values, st);
            if (ps != SL_OK) return ps;
        }
        if (m->d_pwr_idx) {
        } else
        if (env_band != 0 && !refused && band_wide && (env_band > 0 || band_pays)) {
            sl_status ps = sl_build_paced_panels(m, d_row_ptr, d_col_idx, d_values, st, env_band > 0 ? (uint32_t)env_band : SL_PW_BAND_AUTO);
            if (ps != SL_OK) return ps;
        } else
        if (!refused && (forced || pays || pays_paced) && nnz && nnz < 0x7fffffffull && n_tiles * n_panels < 0xfffffff0ull) {
            // balanced matrices of some size: persistent paced blocks (SL_COLUMN_PANELS=3 forces the dynamic tiles instead)
            sl_status ps = env_panels == 3 ? SL_OK : sl_build_paced_panels(m, d_row_ptr, d_col_idx, d_values, st);
            if (ps == SL_OK && !m->d_pw_idx && (forced || pays)) ps = sl_build_column_panels(m, d_row_ptr, d_col_idx, d_values, n_tiles, (uint32_t)n_panels, st);
            if (ps != SL_OK) return ps;
        }
    }

    // 2c. column-constant operators with a unit diagonal (PageRank / PPR systems of unweighted graphs) on the uniform-column paced layout:
    //     opt-in for now (SL_PW_INDEX_ONLY=1, read when the matrix is built and when a push runs on it)
    {
        static const bool idx_only = [] { const char *e = getenv("SL_PW_INDEX_ONLY"); return e && *e == '1'; }();
        if (idx_only && m->d_pw_idx && !m->pw_band) SL_TRY(sl_detect_column_constant(m, d_row_ptr, d_col_idx, d_values, st));
    }

    return SL_OK;
}

sl_status sl_build_from_device_csr(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx,
                                   const double *d_values, bool keep_csr_copy)
{
    hipStream_t st = sl_context().stream;
    const uint64_t n = m->n_rows, nnz = m->nnz;
    m->n_slices = (n + SL_SLICE - 1) / SL_SLICE;
    sl_range trace_range("matrix layout build");

    // 1. validate
    uint32_t *d_err = nullptr;
    SL_HIP(sl_malloc(&d_err, 4 * sizeof(uint32_t)));
    SL_HIP(hipMemsetAsync(d_err, 0, 4 * sizeof(uint32_t), st));
    hipLaunchKernelGGL(sl_validate_csr_kernel, dim3(grid_for(nnz > n ? nnz : n, 256) > 4096 ? 4096 : grid_for(nnz > n ? nnz : n, 256)),
                       dim3(256), 0, st, n, m->n_cols, nnz, d_row_ptr, d_col_idx, d_err);
    uint32_t h_err = 0;
    SL_TRY(sl_read_back(&h_err, d_err, sizeof(uint32_t), st));
    if (h_err & 1u) { hipFree(d_err); return sl_fail(SL_INVALID_SPARSE_MATRIX, "row_ptr is not a monotone 0..nnz prefix array"); }
    if (h_err & 2u) { hipFree(d_err); return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "column index >= n_cols (%llu)", (unsigned long long)m->n_cols); }

    // rows far longer than the typical row leave the slice layout (one of them would stretch its whole 64-row slice): 2.5 x the
    // mean length, within [24, 256].  Measured on the 10^7-node power-law graph (mean 10.9) — full PageRank solve by threshold:
    // 256: 0.73 s, 44: 0.62, 32: 0.59, 28 / 24: 0.58, 16: 0.76 (too many one-block rows).  Uniform systems are unaffected.
    {
        const uint64_t scaled = n ? (5 * nnz + 2 * n - 1) / (2 * n) : 0;
        m->long_row = (uint32_t)(scaled < 24 ? 24 : (scaled > SL_LONG_ROW ? SL_LONG_ROW : scaled));
        if (const char *e = getenv("SL_LONG_ROW_MIN")) m->long_row = (uint32_t)atoi(e);      // experiments
    }
    // 2. row lengths, slice widths
    const uint64_t padded_rows = m->n_slices * SL_SLICE;
    uint32_t *d_slice_w = nullptr;
    SL_HIP(sl_malloc(&m->d_row_len, (padded_rows ? padded_rows : 1) * sizeof(uint32_t)));
    SL_HIP(sl_malloc(&d_slice_w, (m->n_slices ? m->n_slices : 1) * sizeof(uint32_t)));
    const uint32_t mm_init[4] = {0xffffffffu, 0u, 0u, 0u};
    SL_TRY(sl_upload(d_err, mm_init, sizeof(mm_init), st));
    if (m->n_slices)
        hipLaunchKernelGGL(sl_row_len_kernel, dim3((uint32_t)((padded_rows + 255) / 256)), dim3(256), 0, st, n, m->n_slices, m->long_row,
                           d_row_ptr, m->d_row_len, d_slice_w, d_err);
    std::vector<uint32_t> slice_w(m->n_slices), slice_ptr(m->n_slices + 1);
    uint32_t mm[4];
    SL_TRY(sl_read_back(mm, d_err, sizeof(mm), st));
    if (m->n_slices) SL_TRY(sl_read_back(slice_w.data(), d_slice_w, m->n_slices * sizeof(uint32_t), st));
    DevBuf slice_w_keep;                                          // (freed at the end of the build: the fill below may want the widths again)
    slice_w_keep.p = d_slice_w; slice_w_keep.pooled = false;
    m->min_row_nnz = n ? mm[0] : 0;
    m->max_row_nnz = n ? mm[1] : 0;
    m->n_long = n ? mm[2] : 0;
    if (m->n_long) {                                 // ascending list of the long rows (deterministic partial slots)
        uint32_t *d_tmp = nullptr;
        SL_HIP(sl_malloc(&d_tmp, m->n_long * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_long_rows, m->n_long * sizeof(uint32_t)));
        SL_HIP(hipMemsetAsync(d_err, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(sl_long_collect_kernel, dim3(1024), dim3(256), 0, st, n, m->d_row_len, d_tmp, d_err);
        sl_status ss = sl_sort_keys_u32(d_tmp, m->d_long_rows, m->n_long, st);
        hipFree(d_tmp);
        if (ss != SL_OK) { hipFree(d_err); return ss; }
    }
    hipFree(d_err);
#ifdef SL_DEBUG_HOOKS                                            // test builds only (make EXTRA=-DSL_DEBUG_HOOKS): what a stale read-back would look like
    if (getenv("SL_DEBUG_STALE_SLICE_WIDTHS"))                    // every third width one pair block short
        for (uint64_t s = 0; s < m->n_slices; s += 3) if (slice_w[s]) --slice_w[s];
#endif
    uint64_t acc = 0;
    for (uint64_t s = 0; s < m->n_slices; ++s) { slice_ptr[s] = (uint32_t)acc; acc += slice_w[s]; }
    if (acc > 0xffffffffull) return sl_fail(SL_ALLOCATION, "matrix too large for 32-bit slice pointers");
    slice_ptr[m->n_slices] = (uint32_t)acc;
    m->padded_nnz = acc * 2 * SL_SLICE;                         // acc counts pair blocks
    m->uniform_width = (n && m->min_row_nnz == m->max_row_nnz && (m->max_row_nnz % 4u) == 0u) ? m->max_row_nnz : 0u;
    if (getenv("SL_NO_UNROLLED")) m->uniform_width = 0;          // experiments: send uniform-width matrices through the batched path

    SL_HIP(sl_malloc(&m->d_slice_ptr, (m->n_slices + 1) * sizeof(uint32_t)));
    SL_TRY(sl_upload(m->d_slice_ptr, slice_ptr.data(), (m->n_slices + 1) * sizeof(uint32_t), st));
#ifdef SL_DEBUG_HOOKS
    if (const char *e = getenv("SL_DEBUG_STALE_SLICE_PTRS")) if (*e == '1' && m->n_slices > 4) {     // what a stale host-to-device copy would look like
        std::vector<uint32_t> bad(slice_ptr);
        for (uint64_t q = 3; q <= m->n_slices; ++q) bad[q] -= 1;                                       // slice 2 one pair block short, the rest shifted
        SL_TRY(sl_upload(m->d_slice_ptr, bad.data(), (m->n_slices + 1) * sizeof(uint32_t), st));
    }
#endif
    SL_HIP(sl_malloc(&m->d_cols, (m->padded_nnz ? m->padded_nnz : 4) * sizeof(uint32_t)));
    SL_HIP(sl_malloc(&m->d_vals, (m->padded_nnz ? m->padded_nnz : 4) * sizeof(double)));
    unsigned long long *d_band = nullptr;
    SL_HIP(sl_malloc(&d_band, 5 * sizeof(unsigned long long)));
    SL_HIP(hipMemsetAsync(d_band, 0, 5 * sizeof(unsigned long long), st));
    if (m->n_slices)
        hipLaunchKernelGGL(sl_fill_slices_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, n, m->n_cols,
                           m->n_slices, m->row_offset, d_row_ptr, d_col_idx, d_values, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, d_band);
    SL_HIP(hipGetLastError());
    unsigned long long h_band[5] = {0, 0, 0, 0, 0};
    SL_TRY(sl_read_back(h_band, d_band, sizeof(h_band), st));
    hipFree(d_band);
    if (h_band[4]) {
        // Rows that do not fit the slice the host's pointers give them: somewhere between sl_row_len_kernel and sl_fill_slices_kernel the
        // widths changed hands wrongly.  Seen twice in ~2200 many-process runs on one GPU in round 3 (a row lost its diagonal, the dominance
        // check refused a dominant matrix) and never with the control-plane transfers on the library's own pinned staging (round 4).  The
        // repair is also the diagnosis — every link of the chain is checked apart and the counts are logged (level 0: always):
        //   (a) the widths the host summed against a second, blocking read of the SAME device buffer    -> a stale device-to-host copy
        //   (b) that buffer against a re-run of sl_row_len_kernel on the same inputs                    -> inputs not complete when the kernel ran
        //   (c) the slice pointers ON THE DEVICE against the ones the host uploaded                     -> a stale host-to-device copy
        std::vector<uint32_t> again(m->n_slices), rerun(m->n_slices), ptr_dev(m->n_slices + 1);
        SL_HIP(hipDeviceSynchronize());
        SL_HIP(hipMemcpy(again.data(), slice_w_keep.p, m->n_slices * sizeof(uint32_t), hipMemcpyDeviceToHost));
        SL_HIP(hipMemcpy(ptr_dev.data(), m->d_slice_ptr, (m->n_slices + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost));
        uint64_t differ_a = 0, differ_b = 0, differ_c = 0;
        {
            DevBuf w2, len2, mm2;
            SL_TRY(w2.alloc_owned(m->n_slices * 4)); SL_TRY(len2.alloc_owned((padded_rows ? padded_rows : 1) * 4)); SL_TRY(mm2.alloc_owned(16));
            SL_HIP(hipMemcpy(mm2.p, mm_init, sizeof(mm_init), hipMemcpyHostToDevice));
            hipLaunchKernelGGL(sl_row_len_kernel, dim3((uint32_t)((padded_rows + 255) / 256)), dim3(256), 0, st, n, m->n_slices, m->long_row,
                               d_row_ptr, len2.as<uint32_t>(), w2.as<uint32_t>(), mm2.as<uint32_t>());
            SL_HIP(hipDeviceSynchronize());
            SL_HIP(hipMemcpy(rerun.data(), w2.p, m->n_slices * sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
        for (uint64_t q = 0; q < m->n_slices; ++q) { differ_a += again[q] != slice_w[q]; differ_b += rerun[q] != again[q]; }
        for (uint64_t q = 0; q <= m->n_slices; ++q) differ_c += ptr_dev[q] != slice_ptr[q];
        sl_log(0, "matrix layout: %llu rows did not fit their slices (%llu slices).  (a) host copy of the widths vs a blocking re-read of the same buffer: %llu differ; "
                  "(b) that buffer vs a re-run of the row-length kernel: %llu differ; (c) slice pointers on the device vs the ones uploaded: %llu differ.  "
                  "Library stream %p, staging %s, caller's CSR arrays %s — rebuilding from the re-run",
               h_band[4], (unsigned long long)m->n_slices, (unsigned long long)differ_a, (unsigned long long)differ_b, (unsigned long long)differ_c, (void *)st,
               getenv("SL_STAGING") ? getenv("SL_STAGING") : "pinned", m->caller_device_arrays ? "in device memory (SL_MEM_DEVICE)" : "uploaded by the library (SL_MEM_HOST)");
        acc = 0;
        for (uint64_t q = 0; q < m->n_slices; ++q) { slice_ptr[q] = (uint32_t)acc; acc += rerun[q]; }
        if (acc > 0xffffffffull) return sl_fail(SL_ALLOCATION, "matrix too large for 32-bit slice pointers");
        slice_ptr[m->n_slices] = (uint32_t)acc;
        m->padded_nnz = acc * 2 * SL_SLICE;
        hipFree(m->d_cols); hipFree(m->d_vals); m->d_cols = nullptr; m->d_vals = nullptr;
        SL_HIP(sl_malloc(&m->d_cols, (m->padded_nnz ? m->padded_nnz : 4) * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_vals, (m->padded_nnz ? m->padded_nnz : 4) * sizeof(double)));
        SL_HIP(hipMemcpy(m->d_slice_ptr, slice_ptr.data(), (m->n_slices + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
        // the row lengths the fill walks must be the re-run's too (case b)
        hipLaunchKernelGGL(sl_row_len_kernel, dim3((uint32_t)((padded_rows + 255) / 256)), dim3(256), 0, st, n, m->n_slices, m->long_row,
                           d_row_ptr, m->d_row_len, static_cast<uint32_t *>(slice_w_keep.p), static_cast<uint32_t *>(nullptr));
        SL_HIP(sl_malloc(&d_band, 5 * sizeof(unsigned long long)));
        SL_HIP(hipMemsetAsync(d_band, 0, 5 * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(sl_fill_slices_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, n, m->n_cols,
                           m->n_slices, m->row_offset, d_row_ptr, d_col_idx, d_values, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, d_band);
        SL_HIP(hipStreamSynchronize(st));
        SL_HIP(hipMemcpy(h_band, d_band, sizeof(h_band), hipMemcpyDeviceToHost));
        hipFree(d_band);
        if (h_band[4])
            return sl_fail(SL_DEVICE_ERROR, "matrix layout: %llu rows do not fit their slices after a rebuild of the slice pointers", h_band[4]);
    }
    m->bandwidth = h_band[0];
    // rows beyond the last column (tall matrix / row slice reaching past n_cols): their own index is not a column, so neither the
    // band window [row - w, row + w] nor the padding column "the row itself" exists for them — such matrices keep the general kernel
    if (m->row_offset + n > m->n_cols) m->bandwidth = ~0ull;
    const uint64_t far_entries = h_band[1], slice_entries = h_band[2], diagonal_entries = h_band[3];
    if (m->bandwidth < 32768 && m->n_slices && m->padded_nnz) {
        SL_HIP(sl_malloc(&m->d_cols16, m->padded_nnz * sizeof(uint16_t)));
        if (m->uniform_width == 8 || m->uniform_width == 16)      // octet layout of the unrolled uniform path
            hipLaunchKernelGGL(sl_fill_cols16_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, n, m->n_slices, m->row_offset,
                               m->uniform_width, m->d_cols, m->d_cols16);
        else                                                      // quad layout of the batched ragged path
            hipLaunchKernelGGL(sl_fill_cols16_quads_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, n, m->n_slices,
                               m->row_offset, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_cols16);
        SL_HIP(hipGetLastError());
    }
    m->device_bytes = (m->n_slices + 1 + padded_rows) * sizeof(uint32_t) + m->padded_nnz * 12 + (m->d_cols16 ? m->padded_nnz * 2 : 0);

    m->far_entries = far_entries; m->slice_entries = slice_entries; m->diagonal_entries = diagonal_entries;
    SL_TRY(sl_build_column_streams(m, d_row_ptr, d_col_idx, d_values, st));

    // 3. transpose
    if (m->flags & SL_MATRIX_WITH_TRANSPOSE) {
        SL_HIP(sl_malloc(&m->d_tptr, (m->n_cols + 1) * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_trow, (nnz ? nnz : 1) * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_tval, (nnz ? nnz : 1) * sizeof(double)));
        SL_HIP(hipMemsetAsync(m->d_tptr, 0, (m->n_cols + 1) * sizeof(uint32_t), st));
        const uint32_t g = grid_for(nnz, 256) > 8192 ? 8192 : grid_for(nnz, 256);
        if (nnz) hipLaunchKernelGGL(sl_col_count_kernel, dim3(g), dim3(256), 0, st, nnz, d_col_idx, m->d_tptr);
        std::vector<uint32_t> tptr(m->n_cols + 1);
        SL_TRY(sl_read_back(tptr.data(), m->d_tptr, (m->n_cols + 1) * sizeof(uint32_t), st));
        uint64_t run = 0;
        for (uint64_t j = 0; j <= m->n_cols; ++j) { run += tptr[j]; tptr[j] = (uint32_t)run; }
        if (run != nnz) return sl_fail(SL_DEVICE_ERROR, "transpose: the column histogram as read back sums to %llu, the matrix holds %llu entries", (unsigned long long)run, (unsigned long long)nnz);
        SL_TRY(sl_upload(m->d_tptr, tptr.data(), (m->n_cols + 1) * sizeof(uint32_t), st));
        if (nnz) {
            uint32_t *d_keys = nullptr, *d_ent_in = nullptr, *d_ent = nullptr;
            SL_HIP(sl_malloc(&d_keys, nnz * sizeof(uint32_t)));
            SL_HIP(sl_malloc(&d_ent_in, nnz * sizeof(uint32_t)));
            SL_HIP(sl_malloc(&d_ent, nnz * sizeof(uint32_t)));
            hipLaunchKernelGGL(sl_iota_kernel, dim3(g), dim3(256), 0, st, nnz, d_ent_in);
            int bits = 1;
            while (bits < 32 && (1ull << bits) < m->n_cols) ++bits;
            sl_status ss = sl_sort_pairs_u32(d_col_idx, d_keys, d_ent_in, d_ent, nnz, bits, st);
            if (ss == SL_OK) {
                hipLaunchKernelGGL(sl_transpose_gather_kernel, dim3(g), dim3(256), 0, st, nnz, n, d_row_ptr, d_values, d_ent,
                                   m->d_trow, m->d_tval);
                hipStreamSynchronize(st);
            }
            hipFree(d_keys); hipFree(d_ent_in);
            if (ss != SL_OK) { hipFree(d_ent); return ss; }
            m->d_tent = d_ent;
        }
        m->device_bytes += (m->n_cols + 1) * sizeof(uint32_t) + nnz * 16;
    }

    // 4. raw CSR copy (also needed by the long-row kernel)
    if (keep_csr_copy || (m->n_long && !m->d_row_ptr)) {
        SL_HIP(sl_malloc(&m->d_row_ptr, (n + 1) * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_col_idx, (nnz ? nnz : 1) * sizeof(uint32_t)));
        SL_HIP(sl_malloc(&m->d_values, (nnz ? nnz : 1) * sizeof(double)));
        SL_HIP(hipMemcpyAsync(m->d_row_ptr, d_row_ptr, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        if (nnz) {
            SL_HIP(hipMemcpyAsync(m->d_col_idx, d_col_idx, nnz * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            SL_HIP(hipMemcpyAsync(m->d_values, d_values, nnz * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        m->device_bytes += (n + 1) * sizeof(uint32_t) + nnz * 12;
    }
    SL_HIP(hipStreamSynchronize(st));
    sl_log(1, "matrix %llu x %llu, %llu entries (rows %u..%u, %llu long): row slices%s%s, bandwidth %llu, %s%s, %.1f MB on the device",
           (unsigned long long)n, (unsigned long long)m->n_cols, (unsigned long long)nnz, m->min_row_nnz, m->max_row_nnz, (unsigned long long)m->n_long,
           m->uniform_width ? " (uniform width)" : "", m->d_cols16 ? " + 16-bit offsets" : "", (unsigned long long)m->bandwidth,
           m->d_pwr_idx ? "order-free column stream" : m->d_pw_idx ? "paced column panels" : (m->d_pan_tile_ptr ? "column panels (dynamic tiles)" : "no column panels"),
           m->d_tptr ? ", transpose" : "", (double)m->device_bytes / 1e6);
    return SL_OK;
}

// run the a6/a7 pass; dinv may be null (dominance check only)
sl_status sl_matrix_diag_pass(const sl_matrix *m, double *d_dinv, unsigned long long h_status[4])
{
    hipStream_t st = sl_context().stream;
    DevBuf status_buf;
    SL_TRY(status_buf.alloc(4 * sizeof(unsigned long long)));
    unsigned long long *d_status = status_buf.as<unsigned long long>();
    const unsigned long long init[4] = {0ull, ~0ull, ~0ull, ~0ull};
    SL_HIP(hipMemcpyAsync(d_status, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (m->n_slices)
        hipLaunchKernelGGL(sl_diag_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, m->n_rows, m->n_slices,
                           m->row_offset, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, d_dinv, d_status);
    if (m->n_long)
        hipLaunchKernelGGL(sl_long_diag_kernel, dim3((uint32_t)((m->n_long + 63) / 64)), dim3(64), 0, st, (uint32_t)m->n_long, m->d_long_rows,
                           m->row_offset, m->d_row_ptr, m->d_col_idx, m->d_values, d_dinv, d_status);
    SL_HIP(hipMemcpyAsync(h_status, d_status, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SL_HIP(hipStreamSynchronize(st));
    return SL_OK;
}

// the conditioning pass: h_out[0] = diagonal dominance factor (+inf: no row has off-diagonal weight), h_out[1] = Gershgorin radius
sl_status sl_matrix_cond_pass(const sl_matrix *m, double h_out[2])
{
    hipStream_t st = sl_context().stream;
    DevBuf buf;
    SL_TRY(buf.alloc(2 * sizeof(unsigned long long)));
    unsigned long long *d_out = buf.as<unsigned long long>();
    const unsigned long long init[2] = {0x7ff0000000000000ull, 0ull};
    unsigned long long bits[2];
    SL_HIP(hipMemcpyAsync(d_out, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (m->n_slices)
        hipLaunchKernelGGL(sl_cond_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, m->n_rows, m->n_slices, m->row_offset,
                           m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, d_out);
    if (m->n_long)
        hipLaunchKernelGGL(sl_long_cond_kernel, dim3((uint32_t)((m->n_long + 63) / 64)), dim3(64), 0, st, (uint32_t)m->n_long, m->d_long_rows,
                           m->row_offset, m->d_row_ptr, m->d_col_idx, m->d_values, d_out);
    SL_HIP(hipGetLastError());
    SL_HIP(hipMemcpyAsync(bits, d_out, sizeof(bits), hipMemcpyDeviceToHost, st));
    SL_HIP(hipStreamSynchronize(st));
    memcpy(h_out, bits, sizeof(bits));
    return SL_OK;
}

// |a_ii| and the sum of |a_ij|, j != i, of ONE row of the slice layout, recomputed for the error message of a failed dominance check
// (a caller who sees "row 114483: |a_ii| = 10.83 < 11.02" knows what to look at; long rows are not re-read: NaN)
__global__ void sl_row_dominance_kernel(uint64_t i, uint64_t row_offset, const uint32_t *slice_ptr, const uint32_t *row_len, const uint32_t *cols,
                                        const double *vals, double *out)
{
    const uint64_t s = i / 64;
    const uint32_t lane = (uint32_t)(i % 64), q0 = slice_ptr[s], q1 = slice_ptr[s + 1], len = row_len[i], gi = (uint32_t)(row_offset + i);
    double diag_abs = 0.0, off = 0.0;
    if (len == SL_LONG_SENTINEL) { out[0] = out[1] = __builtin_nan(""); return; }
    for (uint32_t k = 0; k < len; ++k) {
        const uint32_t c = cols[sl_col_slot(q0, q1, k, lane)];
        const double v = vals[sl_val_slot(q0, k, lane)];
        if (c == gi) diag_abs = fabs(v); else off = __dadd_rn(off, fabs(v));
    }
    out[0] = diag_abs; out[1] = off;
}

void sl_matrix_row_dominance(const sl_matrix *m, uint64_t row, double out[2])
{
    out[0] = out[1] = __builtin_nan("");
    if (!m->n_slices || row >= m->n_rows) return;
    hipStream_t st = sl_context().stream;
    DevBuf buf;
    if (buf.alloc(2 * sizeof(double)) != SL_OK) return;
    hipLaunchKernelGGL(sl_row_dominance_kernel, dim3(1), dim3(1), 0, st, row, m->row_offset, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, buf.as<double>());
    if (hipMemcpyAsync(out, buf.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        out[0] = out[1] = __builtin_nan("");
}

// ---- the element / iterator / norm side of trait Matrix (matrix/mod.rs:33-41, 74-82), served from the layouts the matrix already
// carries: rows of the slice layout by their slots, hub rows by their raw CSR entries.  Nothing here is on a hot path; it is what a
// caller holding a `&dyn Matrix` may ask of the device-resident matrix without keeping a host copy beside it.
struct sl_rows_view {
    uint64_t n_rows;
    const uint32_t *slice_ptr, *row_len, *cols;
    const double *vals;
    const uint32_t *row_ptr, *col_idx;      // raw CSR: only the hub rows (row_len == SL_LONG_SENTINEL) are read from it
    const double *values;
};
static sl_rows_view rows_view(const sl_matrix *m)
{
    return sl_rows_view{m->n_rows, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, m->d_row_ptr, m->d_col_idx, m->d_values};
}
// CSRStorage::get (sparse.rs:142-155): `col_indices[start..end].binary_search(&col)` — the halving search over the row's column
// slice; with duplicate columns the entry IT lands on (the same walk as sl_bsearch_diag / orc_csr_get; positions relative to the
// row's first entry, so the slice layout and the raw CSR entries of a hub row take the same steps)
__device__ __forceinline__ bool sl_view_get(const sl_rows_view &v, uint64_t i, uint32_t col, double *out)
{
    const uint32_t len = v.row_len[i];
    if (len == SL_LONG_SENTINEL) {
        uint32_t lo = v.row_ptr[i], hi = v.row_ptr[i + 1];
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2, c = v.col_idx[mid];
            if (c == col) { *out = v.values[mid]; return true; }
            if (c < col) lo = mid + 1; else hi = mid;
        }
        return false;
    }
    const uint64_t s = i >> 6;
    const uint32_t lane = (uint32_t)(i & 63u), q0 = v.slice_ptr[s], q1 = v.slice_ptr[s + 1];
    uint32_t lo = 0, hi = len;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2, c = v.cols[sl_col_slot(q0, q1, mid, lane)];
        if (c == col) { *out = v.vals[sl_val_slot(q0, mid, lane)]; return true; }
        if (c < col) lo = mid + 1; else hi = mid;
    }
    return false;
}
__global__ void sl_get_kernel(sl_rows_view v, uint64_t i, uint32_t col, double *out)      // out[0] = found (1.0 / 0.0), out[1] = value
{
    double val = 0.0;
    const bool f = sl_view_get(v, i, col, &val);
    out[0] = f ? 1.0 : 0.0; out[1] = val;
}
// CSRStorage::row_iter (sparse.rs:158-176): the row's (column, value) pairs in stored order; count[0] = the row's length
__global__ __launch_bounds__(64) void sl_row_fetch_kernel(sl_rows_view v, uint64_t i, uint32_t cap, uint32_t *cols_out, double *vals_out, uint32_t *count)
{
    const uint32_t len = v.row_len[i];
    if (len == SL_LONG_SENTINEL) {
        const uint32_t k0 = v.row_ptr[i], n = v.row_ptr[i + 1] - k0;
        for (uint32_t k = threadIdx.x; k < n && k < cap; k += 64) { cols_out[k] = v.col_idx[k0 + k]; vals_out[k] = v.values[k0 + k]; }
        if (threadIdx.x == 0) count[0] = n;
        return;
    }
    const uint64_t s = i >> 6;
    const uint32_t lane = (uint32_t)(i & 63u), q0 = v.slice_ptr[s], q1 = v.slice_ptr[s + 1];
    for (uint32_t k = threadIdx.x; k < len && k < cap; k += 64) { cols_out[k] = v.cols[sl_col_slot(q0, q1, k, lane)]; vals_out[k] = v.vals[sl_val_slot(q0, k, lane)]; }
    if (threadIdx.x == 0) count[0] = len;
}
// CSRColIter (sparse.rs:273-298): row after row, the entry get(row, col) lands on — at most ONE pair per row even where a row holds
// the column twice.  Matches are appended in arrival order (the host sorts the pairs by row: rows are unique); count[0] = all matches
__global__ __launch_bounds__(256) void sl_col_fetch_kernel(sl_rows_view v, uint32_t col, uint32_t cap, uint32_t *rows_out, double *vals_out, uint32_t *count)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < v.n_rows; i += stride) {
        double val;
        if (!sl_view_get(v, i, col, &val)) continue;
        const uint32_t at = atomicAdd(count, 1u);
        if (at < cap) { rows_out[at] = (uint32_t)i; vals_out[at] = val; }
    }
}
// Matrix::frobenius_norm (matrix/mod.rs:74-82): sum of value * value over all stored entries.  The reference adds them one after the
// other in row-major order; here every row is summed in its stored order, the rows of a block in a fixed tree, the blocks by
// sl_frob_final_kernel in a fixed order — deterministic, and equal to the reference's sum to rounding (tests: 1e-12 relative)
__global__ __launch_bounds__(256) void sl_frob_kernel(sl_rows_view v, uint64_t n_slices, double *partials)
{
    __shared__ double red[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + wave;
    double sum = 0.0;
    if (s < n_slices) {
        const uint64_t i = s * 64 + lane;
        const uint32_t len = i < v.n_rows ? v.row_len[i] : 0u;
        if (len && len != SL_LONG_SENTINEL) {
            const uint32_t q0 = v.slice_ptr[s];
            for (uint32_t k = 0; k < len; ++k) { const double x = v.vals[sl_val_slot(q0, k, lane)]; sum = __dadd_rn(sum, __dmul_rn(x, x)); }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum = __dadd_rn(sum, __shfl_xor(sum, o));
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = __dadd_rn(__dadd_rn(red[0], red[1]), __dadd_rn(red[2], red[3]));
}
__global__ void sl_long_frob_kernel(uint32_t n_long, const uint32_t *long_rows, const uint32_t *row_ptr, const double *values, double *partials)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_long) return;
    const uint32_t i = long_rows[t];
    double sum = 0.0;
    for (uint32_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) sum = __dadd_rn(sum, __dmul_rn(values[k], values[k]));
    partials[t] = sum;
}
__global__ __launch_bounds__(256) void sl_frob_final_kernel(uint64_t n, const double *partials, double *out)
{
    __shared__ double red[256];
    double sum = 0.0;
    for (uint64_t k = threadIdx.x; k < n; k += 256) sum = __dadd_rn(sum, partials[k]);
    red[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = __dadd_rn(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// SparsityInfo::bandwidth (matrix/mod.rs:535-541): max |r - c| over ALL stored entries.  sl_matrix::bandwidth is the layout's own figure
// (slice rows only — the hub rows never enter the band kernel — and "none" for a row range reaching past the last column), so this
// one is measured on request: integer max, exact in any order
__global__ __launch_bounds__(256) void sl_bandw_kernel(sl_rows_view v, uint64_t n_slices, uint64_t row_offset, unsigned long long *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    if (i >= v.n_rows) return;
    const uint32_t len = v.row_len[i];
    const uint64_t gi = row_offset + i;
    uint64_t bw = 0;
    if (len == SL_LONG_SENTINEL) {
        for (uint32_t k = v.row_ptr[i]; k < v.row_ptr[i + 1]; ++k) { const uint64_t c = v.col_idx[k], d = gi > c ? gi - c : c - gi; bw = d > bw ? d : bw; }
    } else {
        const uint32_t q0 = v.slice_ptr[s], q1 = v.slice_ptr[s + 1];
        for (uint32_t k = 0; k < len; ++k) { const uint64_t c = v.cols[sl_col_slot(q0, q1, k, lane)], d = gi > c ? gi - c : c - gi; bw = d > bw ? d : bw; }
    }
    if (bw) atomicMax(out, (unsigned long long)bw);
}
sl_status sl_matrix_entry_bandwidth(const sl_matrix *m, uint64_t *bandwidth)
{
    *bandwidth = 0;
    if (!m->n_slices) return SL_OK;
    hipStream_t st = sl_context().stream;
    DevBuf buf;
    SL_TRY(buf.alloc(sizeof(unsigned long long)));
    SL_HIP(hipMemsetAsync(buf.p, 0, sizeof(unsigned long long), st));
    hipLaunchKernelGGL(sl_bandw_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, rows_view(m), m->n_slices, m->row_offset, buf.as<unsigned long long>());
    SL_HIP(hipGetLastError());
    unsigned long long h = 0;
    SL_TRY(sl_read_back(&h, buf.p, sizeof(h), st));
    *bandwidth = h;
    return SL_OK;
}

sl_status sl_matrix_get_entry(const sl_matrix *m, uint64_t row, uint64_t col, int *found, double *value)
{
    *found = 0; *value = 0.0;
    if (row >= m->n_rows || col >= m->n_cols || !m->n_slices) return SL_OK;               // SparseMatrix::get, matrix/mod.rs:383-386
    hipStream_t st = sl_context().stream;
    DevBuf buf;
    SL_TRY(buf.alloc(2 * sizeof(double)));
    hipLaunchKernelGGL(sl_get_kernel, dim3(1), dim3(1), 0, st, rows_view(m), row, (uint32_t)col, buf.as<double>());
    SL_HIP(hipGetLastError());
    double h[2];
    SL_TRY(sl_read_back(h, buf.p, sizeof(h), st));
    *found = h[0] != 0.0; *value = h[1];
    return SL_OK;
}

sl_status sl_matrix_fetch_row(const sl_matrix *m, uint64_t row, uint64_t capacity, uint32_t *cols, double *values, uint64_t *count)
{
    *count = 0;
    if (row >= m->n_rows || !m->n_slices) return SL_OK;                                     // CSRStorage::row_iter: an empty iterator
    hipStream_t st = sl_context().stream;
    const uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, m->max_row_nnz);
    DevBuf dc, dv, dn;
    SL_TRY(dc.alloc((size_t)cap * sizeof(uint32_t))); SL_TRY(dv.alloc((size_t)cap * sizeof(double))); SL_TRY(dn.alloc(sizeof(uint32_t)));
    hipLaunchKernelGGL(sl_row_fetch_kernel, dim3(1), dim3(64), 0, st, rows_view(m), row, cap, dc.as<uint32_t>(), dv.as<double>(), dn.as<uint32_t>());
    SL_HIP(hipGetLastError());
    uint32_t n = 0;
    SL_TRY(sl_read_back(&n, dn.p, sizeof(n), st));
    const uint32_t take = std::min(n, cap);
    if (take) { SL_TRY(sl_read_back(cols, dc.p, (size_t)take * sizeof(uint32_t), st)); SL_TRY(sl_read_back(values, dv.p, (size_t)take * sizeof(double), st)); }
    *count = n;
    return SL_OK;
}

sl_status sl_matrix_fetch_col(const sl_matrix *m, uint64_t col, uint64_t capacity, uint32_t *rows, double *values, uint64_t *count)
{
    *count = 0;
    if (col >= m->n_cols || !m->n_slices || !m->n_rows) return SL_OK;
    hipStream_t st = sl_context().stream;
    const uint32_t grid = (uint32_t)std::min<uint64_t>((m->n_rows + 255) / 256, 4096);
    uint32_t cap = (uint32_t)std::min<uint64_t>(capacity, m->n_rows), n = 0;
    std::vector<uint32_t> hr;
    std::vector<double> hv;
    for (int pass = 0; pass < 2; ++pass) {      // a second pass only when the column holds more entries than the caller's buffers: the FIRST `capacity` rows are wanted
        DevBuf dr, dv, dn;
        SL_TRY(dr.alloc((size_t)cap * sizeof(uint32_t))); SL_TRY(dv.alloc((size_t)cap * sizeof(double))); SL_TRY(dn.alloc(sizeof(uint32_t)));
        SL_HIP(hipMemsetAsync(dn.p, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(sl_col_fetch_kernel, dim3(grid), dim3(256), 0, st, rows_view(m), (uint32_t)col, cap, dr.as<uint32_t>(), dv.as<double>(), dn.as<uint32_t>());
        SL_HIP(hipGetLastError());
        SL_TRY(sl_read_back(&n, dn.p, sizeof(n), st));
        if (n > cap) { cap = n; continue; }
        hr.resize(n); hv.resize(n);
        if (n) { SL_TRY(sl_read_back(hr.data(), dr.p, (size_t)n * sizeof(uint32_t), st)); SL_TRY(sl_read_back(hv.data(), dv.p, (size_t)n * sizeof(double), st)); }
        break;
    }
    std::vector<uint32_t> order(hr.size());
    for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return hr[a] < hr[b]; });          // rows are unique: a total order
    const uint64_t take = std::min<uint64_t>(capacity, order.size());
    for (uint64_t k = 0; k < take; ++k) { rows[k] = hr[order[k]]; values[k] = hv[order[k]]; }
    *count = n;
    return SL_OK;
}

sl_status sl_matrix_frobenius_sq(const sl_matrix *m, double *sum_sq)
{
    *sum_sq = 0.0;
    if (!m->n_slices) return SL_OK;
    hipStream_t st = sl_context().stream;
    const uint64_t nb = (m->n_slices + 3) / 4, np = nb + m->n_long;
    DevBuf part, out;
    SL_TRY(part.alloc(np * sizeof(double))); SL_TRY(out.alloc(sizeof(double)));
    hipLaunchKernelGGL(sl_frob_kernel, dim3((uint32_t)nb), dim3(256), 0, st, rows_view(m), m->n_slices, part.as<double>());
    if (m->n_long)
        hipLaunchKernelGGL(sl_long_frob_kernel, dim3((uint32_t)((m->n_long + 63) / 64)), dim3(64), 0, st, (uint32_t)m->n_long, m->d_long_rows, m->d_row_ptr,
                           m->d_values, part.as<double>() + nb);
    hipLaunchKernelGGL(sl_frob_final_kernel, dim3(1), dim3(256), 0, st, np, part.as<double>(), out.as<double>());
    SL_HIP(hipGetLastError());
    return sl_read_back(sum_sq, out.p, sizeof(double), st);
}

// diag of a CSR operator (A^T has the same diagonal as A; used when only CSR arrays exist)
__global__ __launch_bounds__(256) void sl_csr_dinv_kernel(uint64_t n, const uint32_t *ptr, const uint32_t *idx, const double *val,
                                                          double *dinv, unsigned long long *status)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool found = false; double d = 0.0;
    uint32_t ndiag = 0;
    for (uint32_t k = ptr[i]; k < ptr[i + 1]; ++k) if (idx[k] == (uint32_t)i) { d = val[k]; found = true; ++ndiag; }
    if (ndiag > 1) sl_bsearch_diag(ptr[i], ptr[i + 1], idx, val, (uint32_t)i, &d);
    if (!found) { atomicOr(&status[0], 2ull); atomicMin(&status[2], (unsigned long long)i); }
    else if (fabs(d) < 1e-14) { atomicOr(&status[0], 4ull); atomicMin(&status[3], (unsigned long long)i); }
    dinv[i] = (found && fabs(d) >= 1e-14) ? 1.0 / d : 0.0;
}


sl_status sl_csr_diag_pass(uint64_t n, const uint32_t *ptr, const uint32_t *idx, const double *val, double *d_dinv,
                           unsigned long long h_status[4])
{
    hipStream_t st = sl_context().stream;
    DevBuf status_buf;
    SL_TRY(status_buf.alloc(4 * sizeof(unsigned long long)));
    unsigned long long *d_status = status_buf.as<unsigned long long>();
    const unsigned long long init[4] = {0ull, ~0ull, ~0ull, ~0ull};
    SL_HIP(hipMemcpyAsync(d_status, init, sizeof(init), hipMemcpyHostToDevice, st));
    if (n) hipLaunchKernelGGL(sl_csr_dinv_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, ptr, idx, val, d_dinv, d_status);
    SL_HIP(hipMemcpyAsync(h_status, d_status, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    SL_HIP(hipStreamSynchronize(st));
    return SL_OK;
}


// ---- in-place mutators: SparseMatrix::scale / add_diagonal (matrix/mod.rs:346-372) over CSRStorage::scale / add_diagonal
// (sparse.rs:229-248).  Every value array the matrix carries is brought up to date: the row slices entry by entry (padding slots stay
// untouched), the raw CSR and the transpose in place; the sorted column streams either in place (scale by a finite factor: one
// rounded product per stored value, padding 0 stays 0) or rebuilt from the updated rows.
__global__ __launch_bounds__(256) void sl_scale_slices_kernel(uint64_t n_rows, uint64_t n_slices, const uint32_t *slice_ptr, const uint32_t *row_len,
                                                              double *vals, double factor)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t s = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_slices) return;
    const uint64_t i = s * 64 + lane;
    const uint32_t len = i < n_rows ? row_len[i] : 0u;
    if (!len || len == SL_LONG_SENTINEL) return;
    const uint32_t q0 = slice_ptr[s];
    for (uint32_t k = 0; k < len; ++k) { double *p = vals + sl_val_slot(q0, k, lane); *p = __dmul_rn(*p, factor); }
}
__global__ __launch_bounds__(256) void sl_scale_array_kernel(uint64_t n, double *p, double factor)
{
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256) p[k] = __dmul_rn(p[k], factor);
}
// CSRStorage::add_diagonal (sparse.rs:236-248): `col_indices[start..end].binary_search(&row)` — the halving search of sl_view_get —
// and `values[start + pos] += alpha`; rows where it finds nothing are skipped.  One thread per row updates the row's slot in the slice
// layout, the same entry of the raw CSR (csr_* null: not kept) and of the transpose (t* null: none; tent = the CSR entry behind every
// transposed entry, so the ONE duplicate the search landed on is the one that changes there too).
__global__ __launch_bounds__(256) void sl_add_diag_kernel(uint64_t n_rows, uint64_t row_offset, uint64_t n_cols, const uint32_t *slice_ptr,
                                                          const uint32_t *row_len, const uint32_t *cols, double *vals, const uint32_t *csr_ptr,
                                                          const uint32_t *csr_idx, double *csr_val, const uint32_t *tptr, const uint32_t *tent,
                                                          double *tval, double alpha)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const uint64_t gi = row_offset + i;
    if (gi >= n_cols) return;                                   // (rows below the last column of a tall slice have no own column)
    const uint32_t col = (uint32_t)gi, len = row_len[i];
    uint32_t pos = 0xffffffffu;                                 // CSR entry index of the diagonal entry
    if (len == SL_LONG_SENTINEL) {
        uint32_t lo = csr_ptr[i], hi = csr_ptr[i + 1];
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2, c = csr_idx[mid];
            if (c == col) { pos = mid; break; }
            if (c < col) lo = mid + 1; else hi = mid;
        }
        if (pos == 0xffffffffu) return;
        csr_val[pos] = __dadd_rn(csr_val[pos], alpha);
    } else {
        const uint64_t s = i >> 6;
        const uint32_t lane = (uint32_t)(i & 63u), q0 = slice_ptr[s], q1 = slice_ptr[s + 1];
        uint32_t lo = 0, hi = len, at = 0xffffffffu;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2, c = cols[sl_col_slot(q0, q1, mid, lane)];
            if (c == col) { at = mid; break; }
            if (c < col) lo = mid + 1; else hi = mid;
        }
        if (at == 0xffffffffu) return;
        double *p = vals + sl_val_slot(q0, at, lane);
        *p = __dadd_rn(*p, alpha);
        if (csr_ptr) { pos = csr_ptr[i] + at; csr_val[pos] = __dadd_rn(csr_val[pos], alpha); }
    }
    if (tptr && pos != 0xffffffffu)
        for (uint32_t e = tptr[col]; e < tptr[col + 1]; ++e)
            if (tent[e] == pos) { tval[e] = __dadd_rn(tval[e], alpha); break; }
}
// the rows of the slice layout back as a plain CSR (a matrix built without SL_MATRIX_KEEP_CSR holds no other copy of its rows, and
// has no hub rows: those make the build keep the raw arrays)
__global__ __launch_bounds__(256) void sl_slices_to_csr_kernel(uint64_t n_rows, const uint32_t *slice_ptr, const uint32_t *row_len, const uint32_t *cols,
                                                               const double *vals, const uint32_t *row_ptr, uint32_t *col_idx, double *values)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_rows) return;
    const uint64_t s = i >> 6;
    const uint32_t lane = (uint32_t)(i & 63u), q0 = slice_ptr[s], q1 = slice_ptr[s + 1], len = row_len[i], k0 = row_ptr[i];
    for (uint32_t k = 0; k < len; ++k) { col_idx[k0 + k] = cols[sl_col_slot(q0, q1, k, lane)]; values[k0 + k] = vals[sl_val_slot(q0, k, lane)]; }
}

static void free_column_streams(sl_matrix *m)
{
    hipFree(m->d_pan_tile_ptr); hipFree(m->d_pan_row); hipFree(m->d_pan_col); hipFree(m->d_pan_val);
    hipFree(m->d_pw_idx); hipFree(m->d_pw_val); hipFree(m->d_pw_tile_ptr); hipFree(m->d_pw_span_tab);
    hipFree(m->d_pwr_idx); hipFree(m->d_pwr_val); hipFree(m->d_pwr_base); hipFree(m->d_pwr_tile_ptr); hipFree(m->d_pwr_diag);
    hipFree(m->d_colval);
    m->d_pan_tile_ptr = nullptr; m->d_pan_row = nullptr; m->d_pan_col = nullptr; m->d_pan_val = nullptr;
    m->d_pw_idx = nullptr; m->d_pw_val = nullptr; m->d_pw_tile_ptr = nullptr; m->d_pw_span_tab = nullptr;
    m->d_pwr_idx = nullptr; m->d_pwr_val = nullptr; m->d_pwr_base = nullptr; m->d_pwr_tile_ptr = nullptr; m->d_pwr_diag = nullptr;
    m->d_colval = nullptr;
    m->n_pan_tiles = m->pan_entries = 0; m->pan_balanced = false;
    m->n_pw_tiles = m->pw_chunks = 0; m->pw_rpw = m->pw_blocks = 0; m->pw_deal = 0; m->pw_pbits = 16; m->pw_xcd = 0; m->pw_edge_rounds = 0;
    m->pw_edge_rows = 0; m->pw_band = false; m->pw_slack = 4;
    m->n_pwr_tiles = m->pwr_chunks = 0; m->pwr_rpb = m->pwr_blocks = 0;
    m->device_bytes -= m->stream_bytes; m->stream_bytes = 0;
}
static bool has_column_streams(const sl_matrix *m) { return m->d_pan_tile_ptr || m->d_pw_idx || m->d_pwr_idx || m->d_colval; }

// the rows of a matrix that keeps no raw CSR, written back from the slice layout into device arrays of the caller's (row_ptr: n + 1,
// col_idx / values: nnz) — such a matrix has no hub rows (those make the build keep the raw arrays)
sl_status sl_matrix_slices_to_csr(const sl_matrix *m, uint32_t *d_rp, uint32_t *d_ci, double *d_va)
{
    hipStream_t st = sl_context().stream;
    const uint64_t n = m->n_rows;
    std::vector<uint32_t> len(n), rp(n + 1, 0);
    if (n) SL_TRY(sl_read_back(len.data(), m->d_row_len, n * sizeof(uint32_t), st));
    for (uint64_t i = 0; i < n; ++i) {
        if (len[i] == SL_LONG_SENTINEL) return sl_fail(SL_ALGORITHM_ERROR, "internal: a hub row in a matrix without raw CSR arrays");
        rp[i + 1] = rp[i] + len[i];
    }
    SL_TRY(sl_upload(d_rp, rp.data(), (n + 1) * sizeof(uint32_t), st));
    if (n) hipLaunchKernelGGL(sl_slices_to_csr_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n, m->d_slice_ptr, m->d_row_len, m->d_cols, m->d_vals, d_rp, d_ci, d_va);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

// the sorted column streams again, from the rows as they are now (the raw CSR where the matrix keeps it, else the slices written back)
static sl_status sl_matrix_rebuild_column_streams(sl_matrix *m)
{
    hipStream_t st = sl_context().stream;
    free_column_streams(m);
    if (m->d_row_ptr) return sl_build_column_streams(m, m->d_row_ptr, m->d_col_idx, m->d_values, st);
    DevBuf d_rp, d_ci, d_va;
    SL_TRY(d_rp.alloc_owned((m->n_rows + 1) * sizeof(uint32_t))); SL_TRY(d_ci.alloc_owned((m->nnz ? m->nnz : 1) * sizeof(uint32_t)));
    SL_TRY(d_va.alloc_owned((m->nnz ? m->nnz : 1) * sizeof(double)));
    SL_TRY(sl_matrix_slices_to_csr(m, d_rp.as<uint32_t>(), d_ci.as<uint32_t>(), d_va.as<double>()));
    const sl_status r = sl_build_column_streams(m, d_rp.as<uint32_t>(), d_ci.as<uint32_t>(), d_va.as<double>(), st);
    SL_HIP(hipStreamSynchronize(st));
    return r;
}

sl_status sl_matrix_scale_values(sl_matrix *m, double factor)
{
    hipStream_t st = sl_context().stream;
    sl_range trace_range("matrix scale");
    auto scale = [&](double *p, uint64_t count) {
        if (p && count) hipLaunchKernelGGL(sl_scale_array_kernel, dim3((uint32_t)std::min<uint64_t>((count + 255) / 256, 8192)), dim3(256), 0, st, count, p, factor);
    };
    if (m->n_slices)
        hipLaunchKernelGGL(sl_scale_slices_kernel, dim3((uint32_t)((m->n_slices + 3) / 4)), dim3(256), 0, st, m->n_rows, m->n_slices, m->d_slice_ptr, m->d_row_len,
                           m->d_vals, factor);
    scale(m->d_values, m->nnz);
    scale(m->d_tval, m->nnz);
    SL_HIP(hipGetLastError());
    if (has_column_streams(m)) {
        // the column-constant table (unit diagonal by definition) does not survive a scale; the streams take the product in place when
        // the factor is finite (their padding entries hold 0, and 0 * finite = 0), else they are rebuilt (0 * inf would poison sums)
        const bool in_place = std::isfinite(factor) && !m->d_colval;
        if (in_place) {
            scale(m->d_pan_val, m->pan_entries);
            scale(m->d_pw_val, m->pw_chunks * 256);
            scale(m->d_pwr_val, m->pwr_chunks * SL_PWR_CHUNK);
            scale(m->d_pwr_diag, m->d_pwr_diag ? m->n_rows : 0);
            SL_HIP(hipGetLastError());
        } else {
            SL_TRY(sl_matrix_rebuild_column_streams(m));
        }
    }
    SL_HIP(hipStreamSynchronize(st));
    return SL_OK;
}

sl_status sl_matrix_shift_diagonal(sl_matrix *m, double alpha)
{
    hipStream_t st = sl_context().stream;
    sl_range trace_range("matrix add_diagonal");
    if (m->n_rows)
        hipLaunchKernelGGL(sl_add_diag_kernel, dim3((uint32_t)((m->n_rows + 255) / 256)), dim3(256), 0, st, m->n_rows, m->row_offset, m->n_cols, m->d_slice_ptr,
                           m->d_row_len, m->d_cols, m->d_vals, m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_tent, m->d_tval, alpha);
    SL_HIP(hipGetLastError());
    if (has_column_streams(m)) SL_TRY(sl_matrix_rebuild_column_streams(m));
    SL_HIP(hipStreamSynchronize(st));
    return SL_OK;
}
