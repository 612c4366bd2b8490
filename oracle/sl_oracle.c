/*
 * sl_oracle.c — CPU restatement of the reference's push / Neumann hot path.
 * TEST INFRASTRUCTURE ONLY — see sl_oracle.h.  Build: make -C oracle
 * (gcc -O2 -ffp-contract=off -fno-fast-math; no FMA contraction, no reassociation).
 *
 * Reference = ruvnet/sublinear-time-solver @ 2025-09-19; citations are paths
 * relative to the reference root.
 */
#include "sl_oracle.h"
#include <math.h>
#define _GNU_SOURCE
#include <pthread.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ a1 -- */

typedef struct { uint64_t r; uint32_t c; double v; } trip_t;

/* stable merge sort by (row, col): Vec::sort_by is a stable sort, sparse.rs:96 */
static void trip_msort(trip_t *a, trip_t *tmp, uint64_t n)
{
    if (n < 2) return;
    uint64_t h = n / 2;
    trip_msort(a, tmp, h);
    trip_msort(a + h, tmp, n - h);
    uint64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take right only if strictly smaller => stable */
        int right_less = (a[j].r < a[i].r) || (a[j].r == a[i].r && a[j].c < a[i].c);
        tmp[k++] = right_less ? a[j++] : a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(trip_t));
}

/* SparseMatrix::from_triplets (matrix/mod.rs:160-199): validate every triplet in
 * input order (row bound, col bound, finiteness); COOStorage::from_triplets
 * (sparse.rs:530-548) drops v == 0.0; CSRStorage::from_coo (sparse.rs:80-132)
 * stable-sorts by (row, col) and emits; duplicates stay separate entries. */
int orc_csr_from_triplets(uint64_t ntrip, const uint64_t *tr, const uint64_t *tc, const double *tv,
                          uint64_t rows, uint64_t cols,
                          uint32_t *row_ptr, uint32_t *col_idx, double *values, uint64_t *nnz_out)
{
    for (uint64_t k = 0; k < ntrip; ++k) {
        if (tr[k] >= rows) return ORC_INDEX_OUT_OF_BOUNDS;
        if (tc[k] >= cols) return ORC_INDEX_OUT_OF_BOUNDS;
        if (!isfinite(tv[k])) return ORC_INVALID_INPUT;
    }
    trip_t *a = (trip_t *)malloc((ntrip ? ntrip : 1) * sizeof(trip_t));
    trip_t *tmp = (trip_t *)malloc((ntrip ? ntrip : 1) * sizeof(trip_t));
    if (!a || !tmp) { free(a); free(tmp); return ORC_ALLOCATION; }
    uint64_t m = 0;
    for (uint64_t k = 0; k < ntrip; ++k) {
        if (tv[k] != 0.0) { a[m].r = tr[k]; a[m].c = (uint32_t)tc[k]; a[m].v = tv[k]; ++m; }
    }
    trip_msort(a, tmp, m);
    for (uint64_t i = 0; i <= rows; ++i) row_ptr[i] = 0;
    uint64_t cur = 0, cnt = 0;
    for (uint64_t k = 0; k < m; ++k) {
        while (cur < a[k].r) { ++cur; row_ptr[cur] = (uint32_t)cnt; }
        values[cnt] = a[k].v;
        col_idx[cnt] = a[k].c;
        ++cnt;
    }
    while (cur < rows) { ++cur; row_ptr[cur] = (uint32_t)cnt; }
    *nnz_out = cnt;
    free(a); free(tmp);
    return ORC_OK;
}

/* CSRStorage::get, sparse.rs:142-155 — binary search of the row's column slice.
 * (With duplicate columns the reference returns whichever entry the search
 * lands on; so does this.)  PINNED ALGORITHM: the classical halving search with an early return on Equal — core's
 * binary_search_by up to Rust 1.81.  slice::binary_search documents "if there are multiple matches, then any one of the matches
 * could be returned", and std >= 1.82 (a branchless loop without the early return) can pick another of the equal keys: with
 * duplicates in a row this restatement is bit-exact against a reference built with Rust <= 1.81 and returns one of the stored
 * matches otherwise; without duplicates there is nothing to choose. */
int orc_csr_get(const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                uint64_t rows, uint64_t r, uint64_t c, double *out)
{
    if (r >= rows) return 0;
    uint64_t lo = row_ptr[r], hi = row_ptr[r + 1];
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (col_idx[mid] == c) { *out = values[mid]; return 1; }
        if (col_idx[mid] < c) lo = mid + 1; else hi = mid;
    }
    return 0;
}

/* CSRStorage::scale (sparse.rs:229-233): `for value in &mut self.values { *value *= factor; }` */
void orc_csr_scale(uint64_t nnz, double *values, double factor)
{
    for (uint64_t k = 0; k < nnz; ++k) values[k] = values[k] * factor;
}

/* CSRStorage::add_diagonal (sparse.rs:236-248): per row `col_indices[start..end].binary_search(&row)` (the halving search of
 * orc_csr_get: with the diagonal stored twice, the entry IT lands on) and `values[start + pos] += alpha`; a row without a stored
 * diagonal entry is skipped.  row_offset: the rows are rows [row_offset, row_offset + rows) of a larger square system (their own
 * column is row_offset + row; 0 for a whole matrix).  Returns the number of rows changed. */
uint64_t orc_csr_add_diagonal(uint64_t rows, uint64_t row_offset, const uint32_t *row_ptr, const uint32_t *col_idx, double *values, double alpha)
{
    uint64_t changed = 0;
    for (uint64_t r = 0; r < rows; ++r) {
        const uint64_t c = row_offset + r;
        uint64_t lo = row_ptr[r], hi = row_ptr[r + 1];
        while (lo < hi) {
            uint64_t mid = lo + (hi - lo) / 2;
            if (col_idx[mid] == c) { values[mid] = values[mid] + alpha; ++changed; break; }
            if (col_idx[mid] < c) lo = mid + 1; else hi = mid;
        }
    }
    return changed;
}

/* ------------------------------------------------------------- a2/a3/a4 -- */

/* CSRStorage::multiply_vector (sparse.rs:187-203): result.fill(0.0) then
 * `*row_sum += values[i] * x[col]` left to right. */
void orc_spmv_csr_sequential(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                             const double *values, const double *x, double *y)
{
    for (uint64_t i = 0; i < rows; ++i) {
        double s = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            double p = values[k] * x[col_idx[k]];
            s = s + p;
        }
        y[i] = s;
    }
}

/* CSRStorage::multiply_vector_add (sparse.rs:192-203; trait method matrix/mod.rs:47, dispatch :441-465):
 * `*row_sum += values[i] * x[col]` directly into result[row] — the running sum STARTS FROM y_i, so the
 * rounding differs from multiply_vector followed by an add. */
void orc_spmv_add_csr_sequential(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                                 const double *values, const double *x, double *y)
{
    for (uint64_t i = 0; i < rows; ++i) {
        double s = y[i];
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            double p = values[k] * x[col_idx[k]];
            s = s + p;
        }
        y[i] = s;
    }
}

/* simd_ops::matrix_vector_multiply_simd with feature "simd" (simd_ops.rs:20-88):
 * rows with nnz >= 8 use four lane accumulators over chunks of 4 (wide::f64x4 mul
 * then add, lane-wise IEEE), horizontal ((l0+l1)+l2)+l3, tail sequential;
 * shorter rows accumulate sequentially from 0.0. */
static inline double row_simd4(const double *v, const uint32_t *c, uint64_t nnz, const double *x)
{
    if (nnz >= 8) {
        uint64_t chunks = nnz / 4;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        for (uint64_t q = 0; q < chunks; ++q) {
            uint64_t i = q * 4;
            double p0 = v[i] * x[c[i]], p1 = v[i + 1] * x[c[i + 1]];
            double p2 = v[i + 2] * x[c[i + 2]], p3 = v[i + 3] * x[c[i + 3]];
            l0 = l0 + p0; l1 = l1 + p1; l2 = l2 + p2; l3 = l3 + p3;
        }
        double y = l0 + l1; y = y + l2; y = y + l3;
        for (uint64_t i = chunks * 4; i < nnz; ++i) { double p = v[i] * x[c[i]]; y = y + p; }
        return y;
    }
    double s = 0.0;
    for (uint64_t i = 0; i < nnz; ++i) { double p = v[i] * x[c[i]]; s = s + p; }
    return s;
}

void orc_spmv_simd4(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                    const double *values, const double *x, double *y)
{
    for (uint64_t i = 0; i < rows; ++i) {
        uint64_t s = row_ptr[i], e = row_ptr[i + 1];
        y[i] = (e <= s) ? 0.0 : row_simd4(values + s, col_idx + s, e - s, x);
    }
}

/* simd_ops::parallel_matrix_vector_multiply (simd_ops.rs:201-239):
 * chunk_size = ceil(rows / num_threads); each chunk sequential rows, scalar sums. */
typedef struct {
    uint64_t lo, hi; const uint32_t *rp, *ci; const double *v, *x; double *y;
    /* mode 1 (NOT in the reference — its vector passes are serial, neumann.rs:289-296,264-266): the a8 / a9 vector passes of the
     * chunk's rows: tmp *= dinv, t -= tmp, x += t, partial sum of t^2 */
    int mode; const double *dinv; double *t, *xs, *tmp; double acc;
} par_arg;
static void *par_worker(void *p)
{
    par_arg *a = (par_arg *)p;
    if (a->mode == 1) {
        double s = 0.0;
        for (uint64_t i = a->lo; i < a->hi; ++i) {
            const double q = a->tmp[i] * a->dinv[i];
            const double tn = a->t[i] - q;
            a->t[i] = tn;
            a->xs[i] = a->xs[i] + tn;
            s = s + tn * tn;
        }
        a->acc = s;
        return 0;
    }
    for (uint64_t i = a->lo; i < a->hi; ++i) {
        double s = 0.0;
        for (uint64_t k = a->rp[i]; k < a->rp[i + 1]; ++k) { double q = a->v[k] * a->x[a->ci[k]]; s = s + q; }
        a->y[i] = s;
    }
    return 0;
}
/* persistent worker pool — rayon keeps its threads too (simd_ops.rs:227 par_chunks_mut); the pool is
 * rebuilt only when the requested thread count changes.  The calling thread runs chunk 0. */
static struct {
    int n;                       /* workers besides the caller */
    int quit;
    pthread_barrier_t start, done;
    pthread_t tid[256];
    par_arg args[257];
    int has_work[257];
} g_pool;
static void *pool_worker(void *p)
{
    const int id = (int)(intptr_t)p;
    for (;;) {
        pthread_barrier_wait(&g_pool.start);
        if (g_pool.quit) break;
        if (g_pool.has_work[id]) par_worker(&g_pool.args[id]);
        pthread_barrier_wait(&g_pool.done);
    }
    return 0;
}
static void pool_resize(int workers)
{
    if (g_pool.n == workers) return;
    if (g_pool.n > 0) {
        g_pool.quit = 1;
        pthread_barrier_wait(&g_pool.start);
        for (int t = 0; t < g_pool.n; ++t) pthread_join(g_pool.tid[t], 0);
        pthread_barrier_destroy(&g_pool.start); pthread_barrier_destroy(&g_pool.done);
        g_pool.quit = 0;
    }
    g_pool.n = workers;
    if (workers > 0) {
        pthread_barrier_init(&g_pool.start, 0, (unsigned)workers + 1);
        pthread_barrier_init(&g_pool.done, 0, (unsigned)workers + 1);
        for (int t = 0; t < workers; ++t) pthread_create(&g_pool.tid[t], 0, pool_worker, (void *)(intptr_t)(t + 1));
    }
}
void orc_spmv_parallel(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                       const double *values, const double *x, double *y, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    const uint64_t chunk = (rows + (uint64_t)threads - 1) / (uint64_t)threads;   /* simd_ops.rs:219 */
    pool_resize(threads - 1);
    for (int t = 0; t < threads; ++t) {
        uint64_t lo = (uint64_t)t * chunk, hi = lo + chunk;
        if (hi > rows) hi = rows;
        g_pool.has_work[t] = lo < rows;
        g_pool.args[t] = (par_arg){lo, hi, row_ptr, col_idx, values, x, y, 0, 0, 0, 0, 0, 0.0};
    }
    if (threads > 1) pthread_barrier_wait(&g_pool.start);
    if (g_pool.has_work[0]) par_worker(&g_pool.args[0]);
    if (threads > 1) pthread_barrier_wait(&g_pool.done);
}

/* ------------------------------------------------------------------ a5 -- */

/* simd_ops::dot_product_simd, simd_ops.rs:116-147 */
double orc_dot_simd4(uint64_t n, const double *x, const double *y)
{
    uint64_t chunks = n / 4;
    double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
    for (uint64_t q = 0; q < chunks; ++q) {
        uint64_t i = q * 4;
        double p0 = x[i] * y[i], p1 = x[i + 1] * y[i + 1], p2 = x[i + 2] * y[i + 2], p3 = x[i + 3] * y[i + 3];
        l0 = l0 + p0; l1 = l1 + p1; l2 = l2 + p2; l3 = l3 + p3;
    }
    double r = l0 + l1; r = r + l2; r = r + l3;
    for (uint64_t i = chunks * 4; i < n; ++i) { double p = x[i] * y[i]; r = r + p; }
    return r;
}
/* fallback dot (simd_ops.rs:150-154) and fast_solver.rs:182-203's scalar twin */
double orc_dot_sequential(uint64_t n, const double *x, const double *y)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) { double p = x[i] * y[i]; s = s + p; }
    return s;
}
/* simd_ops::axpy_simd, simd_ops.rs:158-189: y = (alpha*x) + y */
void orc_axpy(uint64_t n, double alpha, const double *x, double *y)
{
    for (uint64_t i = 0; i < n; ++i) { double p = alpha * x[i]; y[i] = p + y[i]; }
}
/* solver::utils::{l2,l1,linf}_norm, solver/mod.rs:369-381 */
double orc_l2_norm(uint64_t n, const double *v)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) { double p = v[i] * v[i]; s = s + p; }
    return sqrt(s);
}
double orc_l1_norm(uint64_t n, const double *v)
{
    double s = 0.0;
    for (uint64_t i = 0; i < n; ++i) s = s + fabs(v[i]);
    return s;
}
double orc_linf_norm(uint64_t n, const double *v)
{
    double m = 0.0;                      /* fold(0.0, f64::max): max ignores NaN, as fmax does */
    for (uint64_t i = 0; i < n; ++i) m = fmax(m, fabs(v[i]));
    return m;
}

/* ---- the other storages' multiply loops (matrix/sparse.rs) ------------------------------------------------------------
 * SparseMatrix::from_triplets always builds CSR (matrix/mod.rs:160-199); COO / CSC / Graph storage exist only as
 * convert_to_format() of it (matrix/mod.rs:244-296), filled from to_triplets() of the storage before.  These restate the
 * three other multiply loops over entries in THEIR storage order, so that a test can show what the device relies on: every
 * storage reachable through SparseMatrix adds a row's products in the same sequence as the CSR loop (ascending column,
 * duplicates in insertion order), i.e. the same bits.
 * COOStorage::multiply_vector (sparse.rs:584-597): result.fill(0), then entry after entry result[row] += value * x[col]. */
void orc_spmv_coo(uint64_t rows, uint64_t nnz, const uint32_t *row_idx, const uint32_t *col_idx, const double *values, const double *x, double *y)
{
    for (uint64_t i = 0; i < rows; ++i) y[i] = 0.0;
    for (uint64_t k = 0; k < nnz; ++k) { const double p = values[k] * x[col_idx[k]]; y[row_idx[k]] = y[row_idx[k]] + p; }
}
/* CSCStorage::multiply_vector (sparse.rs:409-430): column after column, a column whose x is exactly 0.0 is skipped. */
void orc_spmv_csc(uint64_t rows, uint64_t cols, const uint32_t *col_ptr, const uint32_t *row_idx, const double *values, const double *x, double *y)
{
    for (uint64_t i = 0; i < rows; ++i) y[i] = 0.0;
    for (uint64_t c = 0; c < cols; ++c) {
        const double xc = x[c];
        if (xc == 0.0) continue;
        for (uint64_t k = col_ptr[c]; k < col_ptr[c + 1]; ++k) { const double p = values[k] * xc; y[row_idx[k]] = y[row_idx[k]] + p; }
    }
}
/* CSCStorage::from_coo (sparse.rs:303-356): stable sort by (column, row).  In: COO entries in any order; out: col_ptr[cols + 1], row_idx, values. */
void orc_coo_to_csc(uint64_t cols, uint64_t nnz, const uint32_t *row_in, const uint32_t *col_in, const double *val_in, uint32_t *col_ptr, uint32_t *row_out, double *val_out)
{
    uint64_t *perm = (uint64_t *)malloc((nnz ? nnz : 1) * sizeof(uint64_t));
    for (uint64_t k = 0; k < nnz; ++k) perm[k] = k;
    /* insertion-stable merge sort by (col, row) */
    uint64_t *tmp = (uint64_t *)malloc((nnz ? nnz : 1) * sizeof(uint64_t));
    for (uint64_t width = 1; width < nnz; width *= 2)
        for (uint64_t lo = 0; lo < nnz; lo += 2 * width) {
            uint64_t mid = lo + width < nnz ? lo + width : nnz, hi = lo + 2 * width < nnz ? lo + 2 * width : nnz, a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                const uint64_t pa = perm[a], pb = perm[b];
                const int b_less = col_in[pb] < col_in[pa] || (col_in[pb] == col_in[pa] && row_in[pb] < row_in[pa]);
                tmp[o++] = b_less ? perm[b++] : perm[a++];
            }
            while (a < mid) tmp[o++] = perm[a++];
            while (b < hi) tmp[o++] = perm[b++];
            for (uint64_t q = lo; q < hi; ++q) perm[q] = tmp[q];
        }
    for (uint64_t c = 0; c <= cols; ++c) col_ptr[c] = 0;
    for (uint64_t k = 0; k < nnz; ++k) { col_ptr[col_in[k] + 1] += 1; row_out[k] = row_in[perm[k]]; val_out[k] = val_in[perm[k]]; }
    for (uint64_t c = 0; c < cols; ++c) col_ptr[c + 1] += col_ptr[c];
    free(perm); free(tmp);
}
/* GraphStorage::from_triplets + multiply_vector (sparse.rs:655-690, 763-773): out_edges[row] in triplet order (entries with weight 0
 * or an index >= nodes are dropped), then row after row, edge after edge result[row] += weight * x[target] (targets >= len(x) skipped). */
void orc_spmv_graph(uint64_t nodes, uint64_t nnz, const uint32_t *row_idx, const uint32_t *col_idx, const double *values, uint64_t x_len, const double *x, double *y)
{
    /* a row's edges in triplet order = the triplets of that row in the order given: walk the triplets once per row block */
    for (uint64_t i = 0; i < nodes; ++i) y[i] = 0.0;
    uint64_t *count = (uint64_t *)calloc(nodes + 1, sizeof(uint64_t));
    for (uint64_t k = 0; k < nnz; ++k) if (values[k] != 0.0 && row_idx[k] < nodes && col_idx[k] < nodes) count[row_idx[k] + 1] += 1;
    for (uint64_t i = 0; i < nodes; ++i) count[i + 1] += count[i];
    uint64_t *at = (uint64_t *)malloc((nodes ? nodes : 1) * sizeof(uint64_t));
    uint64_t *edge = (uint64_t *)malloc((nnz ? nnz : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < nodes; ++i) at[i] = count[i];
    for (uint64_t k = 0; k < nnz; ++k) if (values[k] != 0.0 && row_idx[k] < nodes && col_idx[k] < nodes) edge[at[row_idx[k]]++] = k;
    for (uint64_t i = 0; i < nodes; ++i)
        for (uint64_t e = count[i]; e < count[i + 1]; ++e) {
            const uint64_t k = edge[e];
            if (col_idx[k] < x_len) { const double p = values[k] * x[col_idx[k]]; y[i] = y[i] + p; }
        }
    free(count); free(at); free(edge);
}

/* ------------------------------------------------------------------ a6 -- */

/* SparseMatrix::is_diagonally_dominant, matrix/mod.rs:467-485 */
int orc_is_diagonally_dominant(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                               const double *values)
{
    for (uint64_t i = 0; i < rows; ++i) {
        double diag = 0.0, off = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if ((uint64_t)col_idx[k] == i) diag = fabs(values[k]);
            else off = off + fabs(values[k]);
        }
        if (diag < off) return 0;
    }
    return 1;
}

/* Matrix::diagonal_dominance_factor (matrix/mod.rs:487-514): min over rows WITH off-diagonal weight of
 * |a_ii| / sum |a_ij|; f64::min ignores a NaN operand; returns 1 and *factor when the minimum is finite
 * (Some), 0 for None (no row has off-diagonal entries, or every ratio was +inf / NaN). */
int orc_diagonal_dominance_factor(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx,
                                  const double *values, double *factor)
{
    double min_factor = INFINITY;
    for (uint64_t i = 0; i < rows; ++i) {
        double diag = 0.0, off = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if ((uint64_t)col_idx[k] == i) diag = fabs(values[k]);
            else off = off + fabs(values[k]);
        }
        if (off > 0.0) {
            double f = diag / off;
            if (!(f != f) && f < min_factor) min_factor = f;          /* f64::min: a NaN operand loses */
        }
    }
    if (isfinite(min_factor)) { *factor = min_factor; return 1; }
    return 0;
}

/* Matrix::spectral_radius_estimate (matrix/mod.rs:83-100), Gershgorin: max over rows of |a_ii| + sum |a_ij|,
 * folded with f64::max from 0.0 (a NaN row sum loses). */
double orc_spectral_radius_estimate(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values)
{
    double max_radius = 0.0;
    for (uint64_t i = 0; i < rows; ++i) {
        double diag = 0.0, off = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            if ((uint64_t)col_idx[k] == i) diag = fabs(values[k]);
            else off = off + fabs(values[k]);
        }
        double r = diag + off;
        if (!(r != r) && r > max_radius) max_radius = r;
    }
    return max_radius;
}

/* ---- the element / iterator / norm side of trait Matrix (matrix/mod.rs:33-41, 74-82, 523-545) ---- */

/* SparseMatrix::get (matrix/mod.rs:383-395): out of bounds -> None, else CSRStorage::get (orc_csr_get above) */
int orc_matrix_get(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                   uint64_t r, uint64_t c, double *out)
{
    if (r >= rows || c >= cols) return 0;
    return orc_csr_get(row_ptr, col_idx, values, rows, r, c, out);
}

/* CSRStorage::row_iter (sparse.rs:158-176): a row out of bounds is an empty iterator; otherwise the row's pairs in stored order.
 * Returns the row's length; writes the first min(length, cap) pairs. */
uint64_t orc_csr_row(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, uint64_t r,
                     uint64_t cap, uint32_t *cols_out, double *vals_out)
{
    if (r >= rows) return 0;
    const uint64_t start = row_ptr[r], end = row_ptr[r + 1];
    for (uint64_t k = start; k < end && k - start < cap; ++k) { cols_out[k - start] = col_idx[k]; vals_out[k - start] = values[k]; }
    return end - start;
}

/* CSRColIter::next (sparse.rs:279-297): `while self.row < rows { if let Ok(pos) = col_indices[start..end].binary_search(&col)
 * { yield (row, values[start + pos]) } row += 1 }` — one search per row, so a row that holds the column twice yields ONE pair,
 * the one the search lands on.  Returns the number of pairs; writes the first min(count, cap). */
uint64_t orc_csr_col(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, uint64_t c,
                     uint64_t cap, uint32_t *rows_out, double *vals_out)
{
    uint64_t n = 0;
    for (uint64_t r = 0; r < rows; ++r) {
        double v;
        if (c > 0xffffffffull) break;                       /* `col as IndexType` (sparse.rs:181) would wrap; the callers stay in u32 */
        if (!orc_csr_get(row_ptr, col_idx, values, rows, r, c, &v)) continue;
        if (n < cap) { rows_out[n] = (uint32_t)r; vals_out[n] = v; }
        ++n;
    }
    return n;
}

/* Matrix::frobenius_norm (matrix/mod.rs:74-82): `norm_sq += value * value` over row_iter(row), rows ascending; sqrt. */
double orc_frobenius_norm(uint64_t rows, const uint32_t *row_ptr, const double *values)
{
    double norm_sq = 0.0;
    for (uint64_t r = 0; r < rows; ++r)
        for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            double sq = values[k] * values[k];
            norm_sq = norm_sq + sq;
        }
    return sqrt(norm_sq);
}

/* Matrix::sparsity_info (matrix/mod.rs:523-545) over SparsityInfo::new (types.rs:344-369):
 * out_u[0] = max_nnz_per_row, out_u[1] = bandwidth (always Some: 0 without entries), out_u[2] = is_banded (bandwidth < rows / 4);
 * out_f[0] = sparsity_ratio (nnz / (rows * cols), 0 for an empty shape), out_f[1] = avg_nnz_per_row (0 without rows) */
void orc_sparsity_info(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx, uint64_t out_u[3], double out_f[2])
{
    const uint64_t nnz = rows ? row_ptr[rows] : 0, total = rows * cols;
    out_f[0] = total > 0 ? (double)nnz / (double)total : 0.0;
    out_f[1] = rows > 0 ? (double)nnz / (double)rows : 0.0;
    uint64_t max_row = 0, max_bw = 0;
    for (uint64_t r = 0; r < rows; ++r) {
        const uint64_t len = row_ptr[r + 1] - row_ptr[r];
        if (len > max_row) max_row = len;
        for (uint64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
            const uint64_t c = col_idx[k], bw = r > c ? r - c : c - r;
            if (bw > max_bw) max_bw = bw;
        }
    }
    out_u[0] = max_row; out_u[1] = max_bw; out_u[2] = max_bw < rows / 4 ? 1 : 0;
}

/* f64::powi (neumann.rs:336): rustc lowers it to llvm.powi.f64, which for a run-time exponent calls compiler-rt's
 * __powidf2 — square and multiply, NOT libm pow (third-party arithmetic outside /root/reference: LLVM compiler-rt
 * lib/builtins/powidf2.c, unchanged across the LLVM versions rustc 1.7x ships; restated from its published algorithm). */
double orc_powi(double a, int b)
{
    const int recip = b < 0;
    double r = 1.0;
    while (1) {
        if (b & 1) r = r * a;
        b /= 2;
        if (b == 0) break;
        a = a * a;
    }
    return recip ? 1.0 / r : r;
}

/* NeumannState::estimate_error_bounds (neumann.rs:321-347) on the state a solve ends in: returns 1 and *bound for
 * Some(ErrorBounds::upper_bound_only(bound)), 0 when the state keeps error_bounds = None.
 * terms_computed == 1 leaves matrix_norm_estimate = 0.0 => Some(0^1 / (1 - 0) * ||rhs||) = Some(0.0). */
int orc_neumann_error_bound(uint64_t n, const double *current_term, const double *rhs, uint64_t terms_computed,
                            int series_converged, double *bound)
{
    if (!series_converged || terms_computed == 0) return 0;                          /* :322-324 */
    double est = 0.0;
    if (terms_computed > 1) {                                                        /* :328-332 */
        double ratio = orc_l2_norm(n, current_term) / orc_l2_norm(n, rhs);
        est = pow(ratio, 1.0 / (double)(terms_computed - 1));
    }
    if (est < 1.0) {                                                                 /* :334-344 */
        double remaining = orc_powi(est, (int)terms_computed) / (1.0 - est);
        *bound = remaining * orc_l2_norm(n, rhs);
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------- a7..a11 -- */

/* NeumannState::new, neumann.rs:139-249 — order of checks preserved */
int orc_neumann_init(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx,
                     const double *values, uint64_t b_len, const double *b,
                     double *dinv, double *rhs)
{
    if (rows != cols) return ORC_INVALID_INPUT;                       /* :147-152 */
    if (b_len != rows) return ORC_DIMENSION_MISMATCH;                 /* :154-160 */
    if (!orc_is_diagonally_dominant(rows, row_ptr, col_idx, values))  /* :163-169 */
        return ORC_NOT_DIAGONALLY_DOMINANT;
    for (uint64_t i = 0; i < rows; ++i) {                             /* :172-188 */
        double d;
        if (!orc_csr_get(row_ptr, col_idx, values, rows, i, i, &d)) return ORC_INVALID_SPARSE_MATRIX;
        if (fabs(d) < 1e-14) return ORC_INVALID_SPARSE_MATRIX;
        dinv[i] = 1.0 / d;
    }
    for (uint64_t i = 0; i < rows; ++i) rhs[i] = b[i] * dinv[i];      /* :191-194 */
    return ORC_OK;
}

static void spmv_by_opts(const orc_neumann_opts *o, uint64_t rows, const uint32_t *rp, const uint32_t *ci,
                         const double *v, const double *x, double *y)
{
    if (o->order == ORC_ORDER_SIMD4) orc_spmv_simd4(rows, rp, ci, v, x, y);
    else if (o->threads > 1) orc_spmv_parallel(rows, rp, ci, v, x, y, o->threads);
    else orc_spmv_csr_sequential(rows, rp, ci, v, x, y);
}

/* NeumannSolver::solve (neumann.rs:469-555) over compute_next_term (:252-277),
 * apply_iteration_matrix (:280-299), update_residual (:302-318), is_converged
 * (:422-430).  On CONVERGENCE_FAILURE the outputs are still filled (the
 * reference drops them, :523-530). */
int orc_neumann_solve(uint64_t rows, uint64_t cols, const uint32_t *row_ptr, const uint32_t *col_idx,
                      const double *values, uint64_t b_len, const double *b,
                      const double *initial_guess, const orc_neumann_opts *o,
                      double *x, double *term_out, double *term_norms, orc_neumann_result *res)
{
    memset(res, 0, sizeof(*res));
    res->residual_norm = INFINITY;
    uint64_t n = rows;
    double *dinv = (double *)malloc((n ? n : 1) * sizeof(double));
    double *rhs = (double *)malloc((n ? n : 1) * sizeof(double));
    double *term = (double *)malloc((n ? n : 1) * sizeof(double));
    double *tmp = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!dinv || !rhs || !term || !tmp) { free(dinv); free(rhs); free(term); free(tmp); return ORC_ALLOCATION; }
    int st = orc_neumann_init(rows, cols, row_ptr, col_idx, values, b_len, b, dinv, rhs);
    if (st != ORC_OK) goto done;

    if (o->start == ORC_START_INITIAL_GUESS && initial_guess) memcpy(x, initial_guess, n * sizeof(double));
    else if (o->start == ORC_START_REFERENCE_DEFAULT) memcpy(x, rhs, n * sizeof(double));   /* :197-208 */
    else memset(x, 0, n * sizeof(double));
    memcpy(term, rhs, n * sizeof(double));                                                    /* :211 */

    double resn = INFINITY;
    int series_conv = 0;
    uint64_t terms = 0, it = 0, matvec = 0;
    const double *res_rhs = (o->residual == ORC_RESIDUAL_REFERENCE_SCALED) ? rhs : b;

#define IS_CONVERGED() ((resn <= o->tolerance) || (series_conv && !(terms >= o->max_terms)))
#define UPDATE_RESIDUAL() do { \
        spmv_by_opts(o, n, row_ptr, col_idx, values, x, tmp); ++matvec; \
        for (uint64_t i_ = 0; i_ < n; ++i_) tmp[i_] = tmp[i_] - res_rhs[i_]; \
        resn = orc_l2_norm(n, tmp); } while (0)

    while (!IS_CONVERGED() && it < o->max_iterations) {
        if (terms < o->max_terms) {                                   /* compute_next_term :253-255 */
            if (terms > 0) {                                          /* apply_iteration_matrix */
                spmv_by_opts(o, n, row_ptr, col_idx, values, term, tmp); ++matvec;
                for (uint64_t i = 0; i < n; ++i) tmp[i] = tmp[i] * dinv[i];
                for (uint64_t i = 0; i < n; ++i) term[i] = term[i] - tmp[i];
            }
            for (uint64_t i = 0; i < n; ++i) x[i] = x[i] + term[i];   /* :264-266 */
            double tn = orc_l2_norm(n, term);
            if (term_norms) term_norms[terms] = tn;
            ++terms;
            if (tn < o->series_tolerance) series_conv = 1;            /* :270-274 */
        }
        if (it % 5 == 0) UPDATE_RESIDUAL();                           /* :489-491 */
        ++it;
        if (!isfinite(resn)) { st = ORC_NUMERICAL_INSTABILITY; goto fill; } /* :501-507 */
        if (series_conv) break;                                       /* :510-512 */
    }
    UPDATE_RESIDUAL();                                                /* :516 */
    {
        int conv = IS_CONVERGED();
        res->converged = conv;
        if (!conv && it >= o->max_iterations) st = ORC_CONVERGENCE_FAILURE; /* :523-530 */
    }
fill:
    res->iterations = it; res->terms_computed = terms; res->matvec_count = matvec;
    res->residual_norm = resn; res->series_converged = series_conv;
    if (term_out) memcpy(term_out, term, n * sizeof(double));
done:
    free(dinv); free(rhs); free(term); free(tmp);
    return st;
#undef IS_CONVERGED
#undef UPDATE_RESIDUAL
}

/* cpu_baseline leg of bench.py: exactly `steps` passes of a8 + a9 (apply_iteration_matrix,
 * solution += term, l2 norm of the term) on a row block [0, rows) of a larger system whose
 * gathered vector has n_cols entries — the reference's CPU hot loop with its own SpMV
 * variants: order SIMD4 = simd_ops.rs:20-88 (1 thread), threads > 1 = simd_ops.rs:201-239.
 * t (n_cols) in/out: rows beyond `rows` are left untouched.  Returns the last term norm. */
double orc_neumann_steps(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values,
                         const double *dinv, double *t, double *x, double *tmp, uint64_t steps, int order, int threads)
{
    orc_neumann_opts o; memset(&o, 0, sizeof(o));
    o.order = order; o.threads = threads;
    double tn = 0.0;
    for (uint64_t s = 0; s < steps; ++s) {
        spmv_by_opts(&o, rows, row_ptr, col_idx, values, t, tmp);
        for (uint64_t i = 0; i < rows; ++i) tmp[i] = tmp[i] * dinv[i];
        for (uint64_t i = 0; i < rows; ++i) t[i] = t[i] - tmp[i];
        for (uint64_t i = 0; i < rows; ++i) x[i] = x[i] + t[i];
        tn = orc_l2_norm(rows, t);
    }
    return tn;
}

/* The same loop with its time split: seconds in the SpMV and in the vector passes + norm.  parallel_passes = 1: the vector passes run
 * row-chunk threaded like the SpMV (simd_ops.rs:219 chunks) — NOT what the reference does (its passes are serial loops), reported by
 * bench.py as a clearly labelled second figure ("port + parallel passes"); the norm is then a sum of per-chunk sums. */
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
double orc_neumann_steps_split(uint64_t rows, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *dinv, double *t,
                               double *x, double *tmp, uint64_t steps, int order, int threads, int parallel_passes, double *sec_spmv, double *sec_vec)
{
    orc_neumann_opts o; memset(&o, 0, sizeof(o));
    o.order = order; o.threads = threads;
    double tn = 0.0, a_spmv = 0.0, a_vec = 0.0;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    for (uint64_t s = 0; s < steps; ++s) {
        const double t0 = now_s();
        spmv_by_opts(&o, rows, row_ptr, col_idx, values, t, tmp);
        const double t1 = now_s();
        if (parallel_passes && threads > 1) {
            const uint64_t chunk = (rows + (uint64_t)threads - 1) / (uint64_t)threads;
            pool_resize(threads - 1);
            for (int k = 0; k < threads; ++k) {
                uint64_t lo = (uint64_t)k * chunk, hi = lo + chunk;
                if (hi > rows) hi = rows;
                g_pool.has_work[k] = lo < rows;
                g_pool.args[k] = (par_arg){lo, hi, 0, 0, 0, 0, 0, 1, dinv, t, x, tmp, 0.0};
            }
            pthread_barrier_wait(&g_pool.start);
            if (g_pool.has_work[0]) par_worker(&g_pool.args[0]);
            pthread_barrier_wait(&g_pool.done);
            double sum = 0.0;
            for (int k = 0; k < threads; ++k) if (g_pool.has_work[k]) sum = sum + g_pool.args[k].acc;
            tn = sqrt(sum);
        } else {
            for (uint64_t i = 0; i < rows; ++i) tmp[i] = tmp[i] * dinv[i];
            for (uint64_t i = 0; i < rows; ++i) t[i] = t[i] - tmp[i];
            for (uint64_t i = 0; i < rows; ++i) x[i] = x[i] + t[i];
            tn = orc_l2_norm(rows, t);
        }
        const double t2 = now_s();
        a_spmv += t1 - t0; a_vec += t2 - t1;
    }
    if (sec_spmv) *sec_spmv = a_spmv;
    if (sec_vec) *sec_vec = a_vec;
    return tn;
}

/* ------------------------------------------------------------------ a-P -- */

/* Synchronous thresholded push, SURVEY.md §8 (a-P): the data-parallel member of
 * the family {NeumannState::apply_iteration_matrix (neumann.rs:280-299),
 * ForwardPushSolver::push_node (forward_push.rs:179-216), TS solveForwardPush
 * (core/solver.ts:437-522)} — invariant r = b - A x, every round pushes ALL
 * rows whose scaled residual |r_i * dinv_i| >= theta (ascending index order):
 *     delta_i = r_i * dinv_i (i in F, else 0);  x_i += delta_i;
 *     r_i -= sum_k a_ik * delta_{c_k}   (row-wise, column order, product rounded
 *                                        then added — same order as a2)
 * over the candidate rows {i : exists k, c_k in F}.  With |F| = 1 = argmax this is
 * one step of a14. */
int orc_push_sync_solve(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                        const double *values, const double *b, const orc_push_opts *o,
                        double *x, double *r,
                        uint32_t *flog, uint64_t fcap, uint64_t *fwords, orc_push_result *res)
{
    memset(res, 0, sizeof(*res));
    if (fwords) *fwords = 0;
    double *dinv = (double *)malloc((n ? n : 1) * sizeof(double));
    double *delta = (double *)calloc((n ? n : 1), sizeof(double));
    double *ax = (double *)malloc((n ? n : 1) * sizeof(double));
    uint8_t *inF = (uint8_t *)calloc((n ? n : 1), 1);
    if (!dinv || !delta || !ax || !inF) { free(dinv); free(delta); free(ax); free(inF); return ORC_ALLOCATION; }
    /* D^-1 with the rejection rules of neumann.rs:172-188 (missing / near-zero diagonal); no
     * row-dominance requirement: the push family never checks it (forward_push.rs:67-122) and
     * PageRank systems (solver.ts:664-722) are column- not row-dominant. */
    int st = ORC_OK;
    for (uint64_t i = 0; i < n; ++i) {
        double d;
        if (!orc_csr_get(row_ptr, col_idx, values, n, i, i, &d) || fabs(d) < 1e-14) { st = ORC_INVALID_SPARSE_MATRIX; goto done; }
        dinv[i] = 1.0 / d;
    }
    /* r = b - A x0 */
    if (o->order == ORC_ORDER_SIMD4) orc_spmv_simd4(n, row_ptr, col_idx, values, x, ax);
    else orc_spmv_csr_sequential(n, row_ptr, col_idx, values, x, ax);
    for (uint64_t i = 0; i < n; ++i) r[i] = b[i] - ax[i];

    uint64_t w = 0;
    while (res->rounds < o->max_rounds) {
        uint64_t nf = 0;
        uint64_t hdr = w;
        if (flog && w < fcap) ++w;
        for (uint64_t i = 0; i < n; ++i) {
            double p = r[i] * dinv[i];
            if (fabs(p) >= (o->theta_rows ? o->theta_rows[i] : o->theta)) {
                inF[i] = 1; delta[i] = p; ++nf;
                if (flog && w < fcap) flog[w++] = (uint32_t)i;
            } else { inF[i] = 0; delta[i] = 0.0; }
        }
        if (flog && hdr < fcap) flog[hdr] = (uint32_t)nf;
        if (nf == 0) { res->converged = 1; break; }
        for (uint64_t i = 0; i < n; ++i) if (inF[i]) x[i] = x[i] + delta[i];
        uint64_t touched = 0;
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t s = row_ptr[i], e = row_ptr[i + 1];
            int cand = 0;
            for (uint64_t k = s; k < e; ++k) if (inF[col_idx[k]]) { cand = 1; break; }
            if (!cand) continue;
            ++touched;
            double acc;
            if (o->order == ORC_ORDER_SIMD4) acc = row_simd4(values + s, col_idx + s, e - s, delta);
            else { acc = 0.0; for (uint64_t k = s; k < e; ++k) { double p = values[k] * delta[col_idx[k]]; acc = acc + p; } }
            r[i] = r[i] - acc;
        }
        res->rounds += 1; res->pushes += nf; res->rows_touched += touched;
    }
    res->residual_norm = orc_l2_norm(n, r);
    if (fwords) *fwords = w;
done:
    free(dinv); free(delta); free(ax); free(inF);
    return st;
}

/* ------------------------------------------------------------------ a13 -- */

/* CompressedSparseRow::transpose, graph/mod.rs:92-130 (counting sort; entries of
 * each transposed row end up in increasing original-row order). */
void orc_csr_transpose(uint64_t nrows, uint64_t ncols, const uint32_t *row_ptr, const uint32_t *col_idx,
                       const double *values, uint32_t *t_row_ptr, uint32_t *t_col_idx, double *t_values)
{
    uint64_t nnz = row_ptr[nrows];
    for (uint64_t j = 0; j <= ncols; ++j) t_row_ptr[j] = 0;
    for (uint64_t k = 0; k < nnz; ++k) t_row_ptr[col_idx[k] + 1] += 1;
    for (uint64_t j = 0; j < ncols; ++j) t_row_ptr[j + 1] += t_row_ptr[j];
    uint32_t *pos = (uint32_t *)malloc((ncols ? ncols : 1) * sizeof(uint32_t));
    memcpy(pos, t_row_ptr, ncols * sizeof(uint32_t));
    for (uint64_t i = 0; i < nrows; ++i)
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
            uint32_t p = pos[col_idx[k]]++;
            t_col_idx[p] = (uint32_t)i; t_values[p] = values[k];
        }
    free(pos);
}

/* WorkQueue (graph/mod.rs:132-213): BinaryHeap<WorkItem> where WorkItem derives
 * PartialOrd over (priority, node_id) — lexicographic, max-heap — plus an
 * in-queue BitSet.  Items are distinct (a node is queued at most once), so the
 * pop sequence of any correct max-heap is the same. */
typedef struct { double pr; uint64_t node; } witem;
typedef struct { witem *h; uint64_t len, cap; uint8_t *inq; double threshold; } wqueue;
static int wless(const witem *a, const witem *b)
{
    if (a->pr < b->pr) return 1;
    if (a->pr > b->pr) return 0;
    return a->node < b->node;
}
static void wq_push_if_threshold(wqueue *q, uint64_t node, double residual, double degree)
{
    double pr = (degree > 0.0) ? residual / degree : residual;          /* graph/mod.rs:172 */
    if (pr >= q->threshold && !q->inq[node]) {
        if (q->len == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->h = (witem *)realloc(q->h, q->cap * sizeof(witem)); }
        uint64_t i = q->len++;
        q->h[i].pr = pr; q->h[i].node = node;
        while (i > 0) {
            uint64_t p = (i - 1) / 2;
            if (!wless(&q->h[p], &q->h[i])) break;
            witem t = q->h[p]; q->h[p] = q->h[i]; q->h[i] = t; i = p;
        }
        q->inq[node] = 1;
    }
}
static int wq_pop(wqueue *q, uint64_t *node)
{
    if (!q->len) return 0;
    *node = q->h[0].node;
    q->inq[*node] = 0;
    q->h[0] = q->h[--q->len];
    uint64_t i = 0;
    for (;;) {
        uint64_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < q->len && wless(&q->h[m], &q->h[l])) m = l;
        if (r < q->len && wless(&q->h[m], &q->h[r])) m = r;
        if (m == i) break;
        witem t = q->h[m]; q->h[m] = q->h[i]; q->h[i] = t; i = m;
    }
    return 1;
}
static void wq_adaptive(wqueue *q, uint64_t maxq, uint64_t minq)              /* graph/mod.rs:204-212 */
{
    if (q->len > maxq) q->threshold *= 1.1;
    else if (q->len < minq && q->threshold > 1e-12) q->threshold *= 0.9;
}

static void row_sums(uint64_t n, const uint32_t *rp, const double *w, double *out)  /* graph/mod.rs:81-89 */
{
    for (uint64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (uint64_t k = rp[i]; k < rp[i + 1]; ++k) s = s + w[k];
        out[i] = s;
    }
}

/* direction 0 = forward (forward_push.rs:67-216), 1 = backward (backward_push.rs:67-220) */
/* target != NULL: ForwardPushSolver::solve_with_target (forward_push.rs:233-290) or, backward, BackwardPushSolver::solve_with_source
 * (backward_push.rs:238-293; `src` is then the target node and `target` the source whose precision ends the loop) — both in range or an empty
 * result, and the loop ends as soon as estimate[target] > precision and residual[target] < 0.1 precision (checked before every pop).
 * push_log (may be NULL): the nodes pushed, in order, up to log_cap. */
static int acl_push(uint64_t n, const uint32_t *rp, const uint32_t *ci, const double *w,
                    uint64_t nsrc, const uint64_t *src, const orc_acl_opts *o,
                    double *est, double *res, orc_acl_result *out, int backward,
                    const uint64_t *target, double target_precision, uint32_t *push_log, uint64_t log_cap)
{
    memset(out, 0, sizeof(*out));
    for (uint64_t i = 0; i < n; ++i) { est[i] = 0.0; res[i] = 0.0; }
    if (nsrc == 1 && src[0] >= n) return ORC_OK;                       /* forward_push.rs:75-83 */
    if (target && *target >= n) return ORC_OK;                         /* :238-246 */
    uint64_t nnz = n ? rp[n] : 0;
    double *outdeg = (double *)malloc((n ? n : 1) * sizeof(double));
    double *indeg = 0;
    uint32_t *trp = 0, *tci = 0; double *tw = 0;
    uint8_t *visited = (uint8_t *)calloc((n ? n : 1), 1);
    wqueue q = {0, 0, 0, (uint8_t *)calloc((n ? n : 1), 1), o->queue_threshold};
    row_sums(n, rp, w, outdeg);
    const uint32_t *arp = rp, *aci = ci; const double *aw = w; const double *qdeg = outdeg;
    if (backward) {
        trp = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
        tci = (uint32_t *)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
        tw = (double *)malloc((nnz ? nnz : 1) * sizeof(double));
        indeg = (double *)malloc((n ? n : 1) * sizeof(double));
        orc_csr_transpose(n, n, rp, ci, w, trp, tci, tw);
        row_sums(n, trp, tw, indeg);                                   /* adjacency.rs:212-224 */
        arp = trp; aci = tci; aw = tw; qdeg = indeg;
    }
    double mass = 1.0 / (double)nsrc;                                  /* forward_push.rs:131 */
    if (nsrc == 1) res[src[0]] = 1.0;
    else for (uint64_t s = 0; s < nsrc; ++s) if (src[s] < n) res[src[s]] += mass;
    for (uint64_t s = 0; s < nsrc; ++s)
        if (src[s] < n) wq_push_if_threshold(&q, src[s], res[src[s]], fmax(qdeg[src[s]], 1.0));

    uint64_t pushes = 0, nvis = 0, u;
    while (q.len && pushes < o->max_pushes) {
        if (target && est[*target] > target_precision && res[*target] < target_precision * 0.1) break;   /* :262-265 */
        if (!wq_pop(&q, &u)) break;
        double du = fmax(qdeg[u], 1.0);
        if (res[u] < o->epsilon * du) continue;                        /* :96-99 */
        /* push_node :179-216 / backward_push_node :179-220 */
        if (!(res[u] <= 0.0)) {
            double push_amount = o->alpha * res[u];
            est[u] = est[u] + push_amount;
            double remaining = (1.0 - o->alpha) * res[u];
            res[u] = 0.0;
            double deg = qdeg[u];
            if (deg > 0.0) {
                for (uint64_t k = arp[u]; k < arp[u + 1]; ++k) {
                    uint64_t v = aci[k];
                    double m;
                    if (!backward) m = remaining * aw[k] / deg;        /* (rem*w)/deg */
                    else { double tp = aw[k] / fmax(outdeg[v], 1.0); m = remaining * tp; }
                    res[v] = res[v] + m;
                    wq_push_if_threshold(&q, v, res[v], fmax(qdeg[v], 1.0));
                }
            } else {
                res[u] = res[u] + remaining;
                wq_push_if_threshold(&q, u, res[u], 1.0);
            }
        }
        if (!visited[u]) { visited[u] = 1; ++nvis; }
        if (push_log && pushes < log_cap) push_log[pushes] = (uint32_t)u;
        ++pushes;
        if (o->adaptive_threshold && pushes % 1000 == 0) wq_adaptive(&q, 10000, 100);
    }
    out->push_count = pushes; out->nodes_visited = nvis; out->residual_norm = orc_l2_norm(n, res);
    free(outdeg); free(indeg); free(trp); free(tci); free(tw); free(visited); free(q.h); free(q.inq);
    return ORC_OK;
}

int orc_acl_forward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                         const double *weights, uint64_t nsrc, const uint64_t *sources,
                         const orc_acl_opts *opts, double *estimate, double *residual, orc_acl_result *res)
{
    return acl_push(n, row_ptr, col_idx, weights, nsrc, sources, opts, estimate, residual, res, 0, 0, 0.0, 0, 0);
}
int orc_acl_forward_push_logged(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t nsrc,
                                const uint64_t *sources, const orc_acl_opts *opts, double *estimate, double *residual, orc_acl_result *res,
                                int backward, uint32_t *push_log, uint64_t log_cap)
{
    return acl_push(n, row_ptr, col_idx, weights, nsrc, sources, opts, estimate, residual, res, backward, 0, 0.0, push_log, log_cap);
}
int orc_acl_forward_push_with_target(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t source,
                                     uint64_t target, double target_precision, const orc_acl_opts *opts, double *estimate, double *residual,
                                     orc_acl_result *res, uint32_t *push_log, uint64_t log_cap)
{
    return acl_push(n, row_ptr, col_idx, weights, 1, &source, opts, estimate, residual, res, 0, &target, target_precision, push_log, log_cap);
}
int orc_acl_backward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                          const double *weights, uint64_t ntgt, const uint64_t *targets,
                          const orc_acl_opts *opts, double *estimate, double *residual, orc_acl_result *res)
{
    return acl_push(n, row_ptr, col_idx, weights, ntgt, targets, opts, estimate, residual, res, 1, 0, 0.0, 0, 0);
}
/* BackwardPushSolver::solve_with_source, backward_push.rs:238-293: unit mass at `target`, pushed over the reverse adjacency; source or
 * target out of range: empty result (:243-251); the loop ends as soon as estimate[source] > precision and residual[source] < 0.1
 * precision (:262-264, checked before every pop) */
int orc_acl_backward_push_with_source(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, uint64_t source,
                                      uint64_t target, double source_precision, const orc_acl_opts *opts, double *estimate, double *residual,
                                      orc_acl_result *res, uint32_t *push_log, uint64_t log_cap)
{
    return acl_push(n, row_ptr, col_idx, weights, 1, &target, opts, estimate, residual, res, 1, &source, source_precision, push_log, log_cap);
}
/* ForwardPushSolver::extrapolated_solution (forward_push.rs:292-301) = BackwardPushSolver::extrapolated_solution (backward_push.rs:302-311):
 * solution = estimate.clone(); solution[i] += alpha * residual[i] */
void orc_acl_extrapolated_solution(uint64_t n, double alpha, const double *estimate, const double *residual, double *solution)
{
    for (uint64_t i = 0; i < n; ++i) solution[i] = estimate[i];
    for (uint64_t i = 0; i < n; ++i) solution[i] = solution[i] + alpha * residual[i];
}

/* ------------------------------------------------------------------ a14 -- */

/* TS solveForwardPush (src/core/solver.ts:437-522), Gauss-Southwell on r = b - A x.
 * The dense column sweep `r_j -= A[j][i] * p` for every j != i only changes r_j
 * where A[j][i] != 0 (r - (+-0) == r), so the sweep runs over column i of A
 * (row i of the transpose). */
int orc_ts_forward_push(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                        const double *values, const double *b, double epsilon, uint64_t max_iterations,
                        double *x, double *r, orc_ts_push_result *res)
{
    memset(res, 0, sizeof(*res));
    res->residual = INFINITY;
    uint64_t nnz = n ? row_ptr[n] : 0;
    uint32_t *trp = (uint32_t *)malloc((n + 1) * sizeof(uint32_t));
    uint32_t *tci = (uint32_t *)malloc((nnz ? nnz : 1) * sizeof(uint32_t));
    double *tv = (double *)malloc((nnz ? nnz : 1) * sizeof(double));
    orc_csr_transpose(n, n, row_ptr, col_idx, values, trp, tci, tv);
    for (uint64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; }
    int st = ORC_OK;
    for (uint64_t iter = 0; iter < max_iterations; ++iter) {
        double maxr = 0.0; int64_t node = -1;
        for (uint64_t i = 0; i < n; ++i) if (fabs(r[i]) > maxr) { maxr = fabs(r[i]); node = (int64_t)i; }
        if (maxr < epsilon) { res->converged = 1; break; }
        double d = 0.0;
        orc_csr_get(row_ptr, col_idx, values, n, (uint64_t)node, (uint64_t)node, &d);
        if (fabs(d) < 1e-15) { st = ORC_NUMERICAL_INSTABILITY; break; }
        double p = r[node] / d;
        x[node] = x[node] + p;
        r[node] = 0.0;
        for (uint64_t k = trp[node]; k < trp[node + 1]; ++k) {
            uint64_t j = tci[k];
            if (j != (uint64_t)node) { double q = tv[k] * p; r[j] = r[j] - q; }
        }
        res->iterations = iter + 1;
        res->residual = orc_l2_norm(n, r);
    }
    if (st == ORC_OK && !res->converged) st = ORC_CONVERGENCE_FAILURE;
    free(trp); free(tci); free(tv);
    return st;
}

/* BackwardPushSolver::combine_with_forward (backward_push.rs:314-333): over i in 0..min(|backward estimate|, |forward estimate|)
 * three adds per node, one after the other.  (`a * b * alpha` is (a * b) * alpha.) */
double orc_acl_combine_with_forward(uint64_t n_backward, uint64_t n_forward, double alpha, const double *b_est, const double *b_res,
                                    const double *f_est, const double *f_res)
{
    double total = 0.0;
    const uint64_t k = n_backward < n_forward ? n_backward : n_forward;
    for (uint64_t i = 0; i < k; ++i) {
        double t = b_est[i] * f_est[i];
        total = total + t;
        t = b_res[i] * f_est[i]; t = t * alpha;
        total = total + t;
        t = b_est[i] * f_res[i]; t = t * alpha;
        total = total + t;
    }
    return total;
}

/* ------------------------------------------------------------------ a15 -- */

/* createSeededRandom, src/core/utils.ts:161-168.  state*1664525 < 2^53, so the JS
 * double arithmetic is exact integer arithmetic mod 2^32. */
static inline double lcg_next(uint64_t *state)
{
    *state = (*state * 1664525ull + 1013904223ull) % 0x100000000ull;
    return (double)*state / 4294967296.0;
}
/* the generator's state after k draws from `state`: x -> 1664525 x + 1013904223 (mod 2^32) composed k times by squaring.  The
 * device cuts the reference's ONE stream into blocks of ORC_WALK_STRIDE draws, one block per walk (a walk of at most 1000 steps
 * uses at most 2000 draws): walk number s starts at orc_ts_lcg_jump(seed, s * ORC_WALK_STRIDE). */
#define ORC_WALK_STRIDE 2048ull
/* draws between the starting points of consecutive walks of a call with `total` walks: 2048 while total * 2048 fits the generator's
 * period of 2^32 draws (2^21 walks), then the largest power of two <= 2^32 / total, never below 16 (the device refuses more than 2^28
 * walks in one call) — so that no two walks of a call start at the same position of the stream */
uint64_t orc_walk_stride(uint64_t total)
{
    uint64_t stride = ORC_WALK_STRIDE;
    while (stride > 16 && total * stride > (1ull << 32)) stride >>= 1;
    return stride;
}
uint32_t orc_ts_lcg_jump(uint32_t state, uint64_t k)
{
    uint32_t cur_a = 1664525u, cur_c = 1013904223u, acc_a = 1u, acc_c = 0u;
    for (; k; k >>= 1) {
        if (k & 1ull) { acc_a = acc_a * cur_a; acc_c = acc_c * cur_a + cur_c; }
        cur_c = (cur_a + 1u) * cur_c;
        cur_a = cur_a * cur_a;
    }
    return acc_a * state + acc_c;
}
void orc_ts_lcg(uint32_t seed, uint64_t count, double *out)
{
    uint64_t s = seed;
    for (uint64_t i = 0; i < count; ++i) out[i] = lcg_next(&s);
}

/* estimateEntry random-walk branch (solver.ts:585-601,630-648) over
 * createTransitionMatrix (:359-385) and performRandomWalk (:390-432).  The dense
 * cumulative scan over j = 0..n-1 only grows at stored off-diagonal entries, so it
 * is walked over the CSR row; `rand <= cum[j]` picks the first j, which is j = 0
 * when rand == 0. */
int orc_ts_random_walk_estimate(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx,
                                const double *values, const double *b, uint64_t start_row,
                                double epsilon, uint32_t seed, double *mean, double *variance,
                                uint64_t *num_samples)
{
    double *absorb = (double *)malloc((n ? n : 1) * sizeof(double));
    double *diag = (double *)malloc((n ? n : 1) * sizeof(double));
    for (uint64_t i = 0; i < n; ++i) {
        double d = 0.0;
        orc_csr_get(row_ptr, col_idx, values, n, i, i, &d);
        if (fabs(d) < 1e-15) { free(absorb); free(diag); return ORC_NUMERICAL_INSTABILITY; }
        diag[i] = d; absorb[i] = 1.0 / d;
    }
    double ns = ceil(1.0 / (epsilon * epsilon));
    uint64_t N = (ns > 100.0) ? (uint64_t)ns : 100;
    double *est = (double *)malloc(N * sizeof(double));
    uint64_t state = seed;
    for (uint64_t s = 0; s < N; ++s) {
        uint64_t cur = start_row; double value = 0.0;
        for (int step = 0; step < 1000; ++step) {
            if (lcg_next(&state) < fabs(absorb[cur])) { value = value + b[cur] * absorb[cur]; break; }
            double sum = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k)
                if (col_idx[k] != cur) sum = sum + fabs(-values[k] / diag[cur]);
            if (sum == 0.0) { value = value + b[cur] * absorb[cur]; break; }
            double rnd = lcg_next(&state) * sum;
            if (rnd <= 0.0) { cur = 0; continue; }
            double cum = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k) {
                if (col_idx[k] == cur) continue;
                cum = cum + fabs(-values[k] / diag[cur]);
                if (rnd <= cum) { cur = col_idx[k]; break; }
            }
        }
        est[s] = value;
    }
    double m = 0.0;
    for (uint64_t s = 0; s < N; ++s) m = m + est[s];
    m = m / (double)N;
    double var = 0.0;
    if (N > 1) { for (uint64_t s = 0; s < N; ++s) { double d = est[s] - m; var = var + d * d; } var = var / (double)(N - 1); }
    *mean = m; *variance = var; *num_samples = N;
    free(absorb); free(diag); free(est);
    return ORC_OK;
}

/* The same branch with the walk count given and every walk's value returned: the reference as written — ONE stream, walk s starts
 * where walk s - 1 stopped (solver.ts:589-592) — for SL_WALK_STREAM_SERIAL, which must equal it bit for bit: values[] (num_samples),
 * mean and variance added in walk order as Array.reduce adds them (:630-633).  Diagonal = the last a_ii stored in the row (the dense
 * table of createTransitionMatrix, as orc_ts_random_walk_solve reads it). */
int orc_ts_random_walk_serial(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values_a, const double *b,
                              uint64_t start_row, uint64_t num_samples, uint32_t seed, double *values, double *mean, double *variance)
{
    for (uint64_t i = 0; i < n; ++i) {
        double d = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) if (col_idx[k] == i) d = values_a[k];
        if (fabs(d) < 1e-15) return ORC_NUMERICAL_INSTABILITY;
    }
    uint64_t state = seed;
    for (uint64_t s = 0; s < num_samples; ++s) {
        uint64_t cur = start_row; double value = 0.0;
        for (int step = 0; step < 1000; ++step) {
            double d = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k) if (col_idx[k] == cur) d = values_a[k];
            const double absorb = 1.0 / d;
            if (lcg_next(&state) < fabs(absorb)) { value = value + b[cur] * absorb; break; }
            double sum = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k)
                if (col_idx[k] != cur) sum = sum + fabs(-values_a[k] / d);
            if (sum == 0.0) { value = value + b[cur] * absorb; break; }
            const double rnd = lcg_next(&state) * sum;
            if (rnd <= 0.0) { cur = 0; continue; }
            double cum = 0.0;
            const uint64_t row = cur;
            for (uint64_t k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {
                if (col_idx[k] == row) continue;
                cum = cum + fabs(-values_a[k] / d);
                if (rnd <= cum) { cur = col_idx[k]; break; }
            }
        }
        values[s] = value;
    }
    double m = 0.0;
    for (uint64_t s = 0; s < num_samples; ++s) m = m + values[s];
    m = m / (double)num_samples;
    double var = 0.0;
    if (num_samples > 1) { for (uint64_t s = 0; s < num_samples; ++s) { double q = values[s] - m; var = var + q * q; } var = var / (double)(num_samples - 1); }
    *mean = m; *variance = var;
    return ORC_OK;
}

/* ------------------------------------------------------------ (f-1) CG -- */

/* OptimizedConjugateGradientSolver::solve (src/optimized_solver.rs:182-295) == FastConjugateGradient::solve
 * (src/fast_solver.rs:126-178): x0 = 0, r = p = b, sequential dot products, stop when rsold <= tol^2,
 * break when |p.Ap| < 1e-16.  order selects the SpMV (simd feature on: simd_ops.rs:20-88). */
int orc_cg_solve(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *b,
                 double tolerance, uint64_t max_iterations, int order, double *x, uint64_t *iterations,
                 double *residual_norm, int *converged, uint64_t *matvec_count)
{
    double *r = (double *)malloc((n ? n : 1) * sizeof(double));
    double *p = (double *)malloc((n ? n : 1) * sizeof(double));
    double *ap = (double *)malloc((n ? n : 1) * sizeof(double));
    if (!r || !p || !ap) { free(r); free(p); free(ap); return ORC_ALLOCATION; }
    for (uint64_t i = 0; i < n; ++i) { x[i] = 0.0; r[i] = b[i]; p[i] = b[i]; }
    const double tol_sq = tolerance * tolerance;
    double rsold = 0.0;
    for (uint64_t i = 0; i < n; ++i) { double q = r[i] * r[i]; rsold = rsold + q; }
    uint64_t it = 0, mv = 0;
    int conv = 0;
    while (it < max_iterations) {
        if (rsold <= tol_sq) { conv = 1; break; }
        if (order == ORC_ORDER_SIMD4) orc_spmv_simd4(n, row_ptr, col_idx, values, p, ap);
        else orc_spmv_csr_sequential(n, row_ptr, col_idx, values, p, ap);
        ++mv;
        double pap = 0.0;
        for (uint64_t i = 0; i < n; ++i) { double q = p[i] * ap[i]; pap = pap + q; }
        if (fabs(pap) < 1e-16) break;
        const double alpha = rsold / pap;
        for (uint64_t i = 0; i < n; ++i) { double q = alpha * p[i]; x[i] = x[i] + q; }
        for (uint64_t i = 0; i < n; ++i) { double q = alpha * ap[i]; r[i] = r[i] - q; }
        double rsnew = 0.0;
        for (uint64_t i = 0; i < n; ++i) { double q = r[i] * r[i]; rsnew = rsnew + q; }
        const double beta = rsnew / rsold;
        for (uint64_t i = 0; i < n; ++i) { double q = beta * p[i]; p[i] = r[i] + q; }
        rsold = rsnew;
        ++it;
    }
    *iterations = it; *residual_norm = sqrt(rsold); *converged = conv; *matvec_count = mv;
    free(r); free(p); free(ap);
    return ORC_OK;
}

/* ---------------------------------------------- (f-3) per-walk-stream MC -- */

/* ---- solveRandomWalk: the `random-walk` method of SublinearSolver.solve (solver.ts:278-357) ----
 * for i in 0..n: numWalks = max(100, ceil(1 / eps^2)) walks of performRandomWalk (:390-432) from i; solution[i] = their mean,
 * totalVariance += their sample variance (N - 1); then residual = ||A solution - b||_2 (multiplyMatrixVector, norm2) and
 * converged = residual < eps — otherwise the reference throws CONVERGENCE_FAILED (ORC_CONVERGENCE_FAILURE; x stays filled).
 * one_walk: performRandomWalk over the CSR row (see orc_ts_random_walk_estimate for the cumulative-scan argument). */
static double one_walk(const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *diag, const double *b,
                       uint64_t start, uint64_t *state)
{
    uint64_t cur = start; double value = 0.0;
    for (int step = 0; step < 1000; ++step) {
        const double absorb = 1.0 / diag[cur];
        if (lcg_next(state) < fabs(absorb)) { value = value + b[cur] * absorb; break; }
        double sum = 0.0;
        for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k)
            if (col_idx[k] != cur) sum = sum + fabs(-values[k] / diag[cur]);
        if (sum == 0.0) { value = value + b[cur] * absorb; break; }
        const double rnd = lcg_next(state) * sum;
        if (rnd <= 0.0) { cur = 0; continue; }
        double cum = 0.0;
        const uint64_t row = cur;
        for (uint64_t k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {
            if (col_idx[k] == row) continue;
            cum = cum + fabs(-values[k] / diag[row]);
            if (rnd <= cum) { cur = col_idx[k]; break; }
        }
    }
    return value;
}
/* per_walk_streams = 0: the reference as written — ONE stream createSeededRandom(seed) shared by every walk of every coordinate
 * (serial by construction: where a walk starts in the stream depends on the length of all walks before it).
 * per_walk_streams = 1: what the device computes — walk w of coordinate i is walk number i * num_walks + w and reads the SAME stream
 * from position number * ORC_WALK_STRIDE (orc_ts_lcg_jump): the rule of orc_ts_random_walk_streams coordinate after coordinate.
 * Walk 0 of coordinate 0 is the reference's first walk draw for draw; same estimator, same walk rule, same generator. */
int orc_ts_random_walk_solve(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values, const double *b,
                             double epsilon, uint32_t seed, uint64_t num_walks, int per_walk_streams,
                             double *x, double *variances /* [n] or NULL */, double *residual, double *total_variance)
{
    double *diag = (double *)malloc((n ? n : 1) * sizeof(double));
    for (uint64_t i = 0; i < n; ++i) {
        double d = 0.0;
        for (uint64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) if (col_idx[k] == i) d = values[k];     /* the dense table's a_ii: the last one stored */
        if (fabs(d) < 1e-15) { free(diag); return ORC_NUMERICAL_INSTABILITY; }                          /* solver.ts:368-371 */
        diag[i] = d;
    }
    if (num_walks == 0) { const double ns = ceil(1.0 / (epsilon * epsilon)); num_walks = ns > 100.0 ? (uint64_t)ns : 100; }
    double *est = (double *)malloc(num_walks * sizeof(double));
    uint64_t state = seed;
    double tv = 0.0;
    const uint64_t stride = orc_walk_stride(n * num_walks);
    for (uint64_t i = 0; i < n; ++i) {
        for (uint64_t w = 0; w < num_walks; ++w) {
            if (per_walk_streams) state = orc_ts_lcg_jump(seed, (i * num_walks + w) * stride);
            est[w] = one_walk(row_ptr, col_idx, values, diag, b, i, &state);
        }
        double m = 0.0;
        for (uint64_t w = 0; w < num_walks; ++w) m = m + est[w];
        m = m / (double)num_walks;
        double var = 0.0;
        for (uint64_t w = 0; w < num_walks; ++w) { const double q = est[w] - m; var = var + q * q; }
        var = var / (double)(num_walks - 1);
        x[i] = m; tv = tv + var;
        if (variances) variances[i] = var;
    }
    double *ax = (double *)malloc((n ? n : 1) * sizeof(double));
    orc_spmv_csr_sequential(n, row_ptr, col_idx, values, x, ax);
    for (uint64_t i = 0; i < n; ++i) ax[i] = ax[i] - b[i];
    *residual = orc_l2_norm(n, ax);
    *total_variance = tv;
    free(diag); free(est); free(ax);
    return *residual < epsilon ? ORC_OK : ORC_CONVERGENCE_FAILURE;
}

/* The data-parallel form of estimateEntry's random-walk branch (src/core/solver.ts:585-601,630-648 over
 * performRandomWalk :390-432): identical walk rule and estimator and the SAME TS LCG stream createSeededRandom(seed)
 * (core/utils.ts:161-168), cut into blocks: walk s reads it from position s * ORC_WALK_STRIDE (orc_ts_lcg_jump) instead of
 * where walk s - 1 happened to stop — a position that depends on how many numbers every earlier walk consumed, i.e. is
 * inherently serial.  Walk 0 is the reference's first walk.
 * values[] (num_samples) receives the per-walk estimates; mean / variance as in the reference (:630-634). */
int orc_ts_random_walk_streams(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *values_a,
                               const double *b, uint64_t start_row, uint64_t num_samples, uint32_t seed,
                               double *values, double *mean, double *variance)
{
    for (uint64_t i = 0; i < n; ++i) {
        double d = 0.0;
        orc_csr_get(row_ptr, col_idx, values_a, n, i, i, &d);
        if (fabs(d) < 1e-15) return ORC_NUMERICAL_INSTABILITY;
    }
    const uint64_t stride = orc_walk_stride(num_samples);
    for (uint64_t s = 0; s < num_samples; ++s) {
        uint64_t state = orc_ts_lcg_jump(seed, s * stride);
        uint64_t cur = start_row; double value = 0.0;
        for (int step = 0; step < 1000; ++step) {
            double d = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k) if (col_idx[k] == cur) d = values_a[k];
            const double absorb = 1.0 / d;
            if (lcg_next(&state) < fabs(absorb)) { value = value + b[cur] * absorb; break; }
            double sum = 0.0;
            for (uint64_t k = row_ptr[cur]; k < row_ptr[cur + 1]; ++k)
                if (col_idx[k] != cur) sum = sum + fabs(-values_a[k] / d);
            if (sum == 0.0) { value = value + b[cur] * absorb; break; }
            const double rnd = lcg_next(&state) * sum;
            if (rnd <= 0.0) { cur = 0; continue; }
            double cum = 0.0;
            const uint64_t row = cur;
            for (uint64_t k = row_ptr[row]; k < row_ptr[row + 1]; ++k) {
                if (col_idx[k] == row) continue;
                cum = cum + fabs(-values_a[k] / d);
                if (rnd <= cum) { cur = col_idx[k]; break; }
            }
        }
        values[s] = value;
    }
    double m = 0.0;
    for (uint64_t s = 0; s < num_samples; ++s) m = m + values[s];
    m = m / (double)num_samples;
    double var = 0.0;
    if (num_samples > 1) { for (uint64_t s = 0; s < num_samples; ++s) { double q = values[s] - m; var = var + q * q; } var = var / (double)(num_samples - 1); }
    *mean = m; *variance = var;
    return ORC_OK;
}
