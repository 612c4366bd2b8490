/* asan_check.c — every entry point of the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (make -C oracle asan):
 * small, ragged and degenerate inputs (empty matrix, empty rows, duplicates, hubs, 0 / 1 / many threads).  TEST INFRASTRUCTURE:
 * the oracle is the checker of the GPU parity tests, so its own memory safety is checked here (the reference ships
 * no sanitizer job either; its unsafe-free Rust gets this from the compiler).  Exit code 0 = clean; the sanitizers abort otherwise. */
#include "sl_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "asan_check: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

static uint64_t lcg(uint64_t *s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return *s >> 33; }

static int one_system(uint64_t n, uint64_t seed, int hubs)
{
    /* row dominant system with ragged rows, duplicates and (optionally) a hub row; triplets in shuffled order */
    uint64_t cap = n * 40 + 16, nt = 0, s = seed;
    uint64_t *tr = malloc(cap * sizeof *tr), *tc = malloc(cap * sizeof *tc);
    double *tv = malloc(cap * sizeof *tv);
    CHECK(tr && tc && tv);
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t m = lcg(&s) % 7;
        if (hubs && i == n / 2) m = n > 30 ? 30 : n;
        double off = 0.0;
        for (uint64_t k = 0; k < m; ++k) {
            uint64_t j = lcg(&s) % n;
            if (j == i) continue;
            double v = (double)(lcg(&s) % 2001) / 1000.0 - 1.0;
            tr[nt] = i; tc[nt] = j; tv[nt] = v; ++nt; off += fabs(v);
            if (k == 0 && (i % 5) == 0) { tr[nt] = i; tc[nt] = j; tv[nt] = 0.5 * v; ++nt; off += fabs(0.5 * v); }   /* duplicate entry */
        }
        tr[nt] = i; tc[nt] = i; tv[nt] = 2.0 * off + 1.0; ++nt;
        if ((i % 11) == 0) { tr[nt] = i; tc[nt] = (i + 1) % n; tv[nt] = 0.0; ++nt; }                                   /* explicit zero: dropped */
    }
    uint32_t *rp = calloc(n + 1, sizeof *rp), *ci = malloc((nt + 1) * sizeof *ci);
    double *va = malloc((nt + 1) * sizeof *va);
    uint64_t nnz = 0;
    CHECK(rp && ci && va);
    CHECK(orc_csr_from_triplets(nt, tr, tc, tv, n, n, rp, ci, va, &nnz) == ORC_OK);
    CHECK(nnz <= nt && rp[n] == nnz);
    double d = 0.0;
    if (n) CHECK(orc_csr_get(rp, ci, va, n, 0, 0, &d) && d > 0.0);
    CHECK(orc_is_diagonally_dominant(n, rp, ci, va));

    double *x = calloc(n + 1, sizeof *x), *y = calloc(n + 1, sizeof *y), *z = calloc(n + 1, sizeof *z), *b = calloc(n + 1, sizeof *b);
    CHECK(x && y && z && b);
    for (uint64_t i = 0; i < n; ++i) { x[i] = sin((double)i); b[i] = 1.0 + 0.001 * (double)(i % 1000); }
    orc_spmv_csr_sequential(n, rp, ci, va, x, y);
    orc_spmv_simd4(n, rp, ci, va, x, z);
    for (int t = 0; t <= 5; t += 5) { orc_spmv_parallel(n, rp, ci, va, x, z, t); CHECK(n == 0 || memcmp(y, z, n * sizeof *y) == 0); }
    orc_spmv_add_csr_sequential(n, rp, ci, va, x, z);                      /* z = y + A x, the seeded chain */
    { double f = -1.0; int has = orc_diagonal_dominance_factor(n, rp, ci, va, &f); CHECK(!has || f >= 1.0); }
    CHECK(orc_spectral_radius_estimate(n, rp, ci, va) >= 0.0 && orc_powi(0.5, 3) == 0.125);
    {   /* the element / iterator / norm side of trait Matrix: rows and columns at both ends and out of bounds, zero capacity */
        uint32_t idx[8]; double val[8]; uint64_t ou[3]; double of[2];
        for (uint64_t q = 0; q < 3; ++q) {
            const uint64_t at = q == 0 ? 0 : q == 1 ? (n ? n - 1 : 0) : n + 3;
            (void)orc_matrix_get(n, n, rp, ci, va, at, at, &d);
            CHECK(orc_csr_row(n, rp, ci, va, at, 8, idx, val) <= nnz && orc_csr_row(n, rp, ci, va, at, 0, idx, val) <= nnz);
            CHECK(orc_csr_col(n, rp, ci, va, at, 8, idx, val) <= n && orc_csr_col(n, rp, ci, va, at, 0, idx, val) <= n);
        }
        CHECK(orc_frobenius_norm(n, rp, va) >= 0.0);
        orc_sparsity_info(n, n, rp, ci, ou, of);
        CHECK(ou[0] <= nnz && (n == 0 || ou[1] < n) && of[0] >= 0.0 && of[1] >= 0.0);
    }
    (void)orc_dot_simd4(n, x, y); (void)orc_dot_sequential(n, x, y); orc_axpy(n, 0.5, x, y);
    (void)orc_l2_norm(n, y); (void)orc_l1_norm(n, y); (void)orc_linf_norm(n, y);

    double *dinv = calloc(n + 1, sizeof *dinv), *rhs = calloc(n + 1, sizeof *rhs), *term = calloc(n + 1, sizeof *term);
    double tn[64];
    CHECK(dinv && rhs && term);
    CHECK(orc_neumann_init(n, n, rp, ci, va, n, b, dinv, rhs) == ORC_OK);
    CHECK(orc_neumann_init(n, n, rp, ci, va, n + 1, b, dinv, rhs) == ORC_DIMENSION_MISMATCH);
    for (int order = 0; order < 2; ++order)
        for (int threads = 1; threads <= 3; threads += 2) {
            orc_neumann_opts o = {1e-10, 1000, 50, 1e-12, order, ORC_START_ZERO, ORC_RESIDUAL_TRUE, threads};
            orc_neumann_result res;
            int st = orc_neumann_solve(n, n, rp, ci, va, n, b, NULL, &o, x, term, tn, &res);
            CHECK(st == ORC_OK || st == ORC_CONVERGENCE_FAILURE);
            o.start = ORC_START_INITIAL_GUESS;
            st = orc_neumann_solve(n, n, rp, ci, va, n, b, x, &o, y, term, NULL, &res);
            CHECK(st == ORC_OK || st == ORC_CONVERGENCE_FAILURE);
        }
    { double eb = -1.0; (void)orc_neumann_error_bound(n, term, rhs, 7, 1, &eb); (void)orc_neumann_error_bound(n, term, rhs, 1, 1, &eb); CHECK(eb == 0.0); }
    for (uint64_t i = 0; i < n; ++i) { x[i] = 0.0; y[i] = rhs[i]; }
    (void)orc_neumann_steps(n, rp, ci, va, dinv, y, x, z, 3, 0, 2);

    /* pushes */
    uint32_t *flog = malloc(4096 * sizeof *flog);
    uint64_t fw = 0;
    CHECK(flog);
    double *th = malloc((n + 1) * sizeof *th);
    CHECK(th);
    for (uint64_t i = 0; i < n; ++i) { x[i] = 0.0; th[i] = 1e-7 * (double)(1 + i % 9); }
    orc_push_opts po = {1e-8, 100000, 0, 0, NULL};
    orc_push_result pr;
    CHECK(orc_push_sync_solve(n, rp, ci, va, b, &po, x, y, flog, 4096, &fw, &pr) == ORC_OK && fw <= 4096);
    for (uint64_t i = 0; i < n; ++i) x[i] = 0.0;
    po.theta_rows = th;
    CHECK(orc_push_sync_solve(n, rp, ci, va, b, &po, x, y, NULL, 0, NULL, &pr) == ORC_OK);
    orc_ts_push_result tr_;
    int st = orc_ts_forward_push(n, rp, ci, va, b, 1e-6, 200, x, y, &tr_);
    CHECK(st == ORC_OK || st == ORC_CONVERGENCE_FAILURE);
    uint32_t *trp = calloc(n + 2, sizeof *trp), *tci = malloc((nnz + 1) * sizeof *tci);
    double *tva = malloc((nnz + 1) * sizeof *tva);
    CHECK(trp && tci && tva);
    orc_csr_transpose(n, n, rp, ci, va, trp, tci, tva);
    CHECK(trp[n] == nnz);

    /* ACL pushes on the |weights| graph of the same pattern */
    double *w = malloc((nnz + 1) * sizeof *w);
    CHECK(w);
    for (uint64_t k = 0; k < nnz; ++k) w[k] = fabs(va[k]) + 0.1;
    uint64_t src[2] = {0, n ? n - 1 : 0};
    for (int ad = 0; ad < 2; ++ad) {
        orc_acl_opts ao = {0.15, 1e-6, 100000, 1e-8, ad, 0};
        orc_acl_result ar;
        if (n) {
            CHECK(orc_acl_forward_push(n, rp, ci, w, 2, src, &ao, x, y, &ar) == ORC_OK);
            CHECK(orc_acl_backward_push(n, rp, ci, w, 1, src, &ao, x, y, &ar) == ORC_OK);
            orc_acl_extrapolated_solution(n, ao.alpha, x, y, z);
            CHECK(orc_acl_backward_push_with_source(n, rp, ci, w, src[1], 0, 1e-3, &ao, x, y, &ar, 0, 0) == ORC_OK);
            CHECK(orc_acl_backward_push_with_source(n, rp, ci, w, n + 3, 0, 1e-3, &ao, x, y, &ar, 0, 0) == ORC_OK && ar.push_count == 0);
        }
    }
    double mean = 0.0, var = 0.0;
    uint64_t ns = 0;
    if (n) CHECK(orc_ts_random_walk_estimate(n, rp, ci, va, b, n / 3, 0.1, 42, &mean, &var, &ns) == ORC_OK && ns >= 100);
    orc_ts_lcg(42, 8, tn);

    free(tr); free(tc); free(tv); free(rp); free(ci); free(va); free(x); free(y); free(z); free(b); free(dinv); free(rhs); free(term);
    free(flog); free(th); free(trp); free(tci); free(tva); free(w);
    return 0;
}

int main(void)
{
    static const uint64_t sizes[] = {0, 1, 2, 3, 7, 63, 64, 65, 257, 1000};
    for (unsigned k = 0; k < sizeof sizes / sizeof sizes[0]; ++k)
        for (int hubs = 0; hubs < 2; ++hubs)
            if (one_system(sizes[k], 17 + k, hubs)) { fprintf(stderr, "asan_check: n = %llu failed\n", (unsigned long long)sizes[k]); return 1; }
    /* error paths of the builder */
    uint64_t r1[1] = {5}, c1[1] = {0};
    double v1[1] = {1.0};
    uint32_t rp[4], ci[1];
    double va[1];
    uint64_t nnz = 0;
    if (orc_csr_from_triplets(1, r1, c1, v1, 3, 3, rp, ci, va, &nnz) != ORC_INDEX_OUT_OF_BOUNDS) return 2;
    v1[0] = INFINITY; r1[0] = 0;
    if (orc_csr_from_triplets(1, r1, c1, v1, 3, 3, rp, ci, va, &nnz) != ORC_INVALID_INPUT) return 3;
    printf("asan_check ok\n");
    return 0;
}
