/* Two / three processes sharing the box's GPUs solve ONE system through the C ABI alone — no Python, no torch, no MPI: what a Rust
 * (or any other) host that starts one process per GPU does.  gcc -std=c99, linked against libsublinear_hip.so.
 *
 *   dist_smoke <world> <n> <half_bandwidth> [uneven]
 *
 * The parent forks `world` ranks BEFORE it touches the HIP runtime.  Rank r takes rows [lo_r, hi_r) of a seeded, strictly row
 * dominant system (columns within half_bandwidth of the row; half_bandwidth >= n: all over the matrix), creates its row slice with
 * GLOBAL column ids, joins the communicator, builds the partitioned NeumannState, changes the right-hand side (update_rhs,
 * neumann.rs:436-462), resets and solves, and writes its rows of the solution.  The parent then solves the whole system on one
 * GPU through the same ABI and compares: iteration count, convergence flag, every solution entry BIT FOR BIT. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>
#include "sublinear_hip.h"

static uint64_t mix(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }

/* rows [lo, hi): 2..9 off-diagonal entries per row in [i - w, i + w], ascending, duplicates merged away; diagonal = 2 * sum|off| + 1 */
static void build_rows(uint64_t n, uint64_t w, uint64_t lo, uint64_t hi, uint32_t **rp_o, uint32_t **ci_o, double **va_o, double **b_o)
{
    uint64_t rows = hi - lo, cap = rows * 10 + 1, nnz = 0, i, k;
    uint32_t *rp = malloc((rows + 1) * sizeof *rp), *ci = malloc(cap * sizeof *ci);
    double *va = malloc(cap * sizeof *va), *b = malloc((rows + 1) * sizeof *b);
    rp[0] = 0;
    for (i = lo; i < hi; ++i) {
        uint64_t m = 2 + mix(i * 3 + 1) % 8, cols[10], cnt = 0, a, c, j, dpos;
        double vals[10], off = 0.0;
        uint64_t wlo = i > w ? i - w : 0, whi = i + w + 1 < n ? i + w + 1 : n;
        for (k = 0; k < m; ++k) {
            c = wlo + mix(i * 131 + k * 7 + 5) % (whi - wlo);
            if (c == i) continue;
            for (a = 0; a < cnt && cols[a] != c; ++a) {}
            if (a < cnt) continue;
            cols[cnt] = c; vals[cnt] = (double)(mix(i * 977 + k) % 2001) / 1000.0 - 1.0; off += fabs(vals[cnt]); ++cnt;
        }
        cols[cnt] = i; vals[cnt] = 2.0 * off + 1.0; dpos = cnt; ++cnt; (void)dpos;
        for (a = 1; a < cnt; ++a) {                       /* insertion sort by column */
            uint64_t cc = cols[a]; double vv = vals[a];
            for (j = a; j > 0 && cols[j - 1] > cc; --j) { cols[j] = cols[j - 1]; vals[j] = vals[j - 1]; }
            cols[j] = cc; vals[j] = vv;
        }
        for (a = 0; a < cnt; ++a) { ci[nnz] = (uint32_t)cols[a]; va[nnz] = vals[a]; ++nnz; }
        rp[i - lo + 1] = (uint32_t)nnz;
        b[i - lo] = 1.0 + 0.001 * (double)(i % 1000);
    }
    *rp_o = rp; *ci_o = ci; *va_o = va; *b_o = b;
}

static void options(sl_neumann_options *o)
{
    sl_neumann_options_default(o);
    o->tolerance = 1e-11; o->series_tolerance = 1e-13; o->max_terms = 200; o->max_iterations = 500;
}

#define DIE(code, ...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, " [%s]\n", sl_last_error_message()); exit(code); } while (0)

static void bounds_of(uint64_t n, int world, int uneven, uint64_t *bnd)
{
    int r;
    bnd[0] = 0;
    for (r = 0; r < world; ++r) {
        uint64_t share = n / (uint64_t)world;
        if (uneven) share = share / 2 + (uint64_t)r * (n / (uint64_t)world) / (uint64_t)world;     /* growing shares */
        bnd[r + 1] = r == world - 1 ? n : bnd[r] + share;
    }
}

static int rank_main(int rank, int world, uint64_t n, uint64_t w, int uneven, const char *name)
{
    uint64_t bnd[17], lo, hi, counts[16], upd_i[3];
    double upd_v[3];
    uint32_t *rp, *ci; double *va, *b, *x;
    sl_comm *c = NULL; sl_matrix *m = NULL; sl_neumann_state *st = NULL;
    sl_neumann_options o; sl_neumann_result res;
    int ndev = 0, r2 = -1, w2 = -1;
    char path[256];
    FILE *f;
    sl_device_count(&ndev);
    if (ndev <= 0) DIE(20, "no device");
    if (sl_set_device(rank % ndev) != SL_OK) DIE(21, "set_device");
    bounds_of(n, world, uneven, bnd);
    lo = bnd[rank]; hi = bnd[rank + 1];
    build_rows(n, w, lo, hi, &rp, &ci, &va, &b);
    if (sl_comm_create(rank, world, name, &c) != SL_OK) DIE(22, "rank %d: comm_create", rank);
    if (sl_comm_rank(c, &r2, &w2) != SL_OK || r2 != rank || w2 != world) DIE(23, "comm_rank");
    if (sl_comm_allgather_u64(c, hi - lo, counts) != SL_OK || counts[world - 1] != bnd[world] - bnd[world - 1]) DIE(24, "allgather");
    if (sl_matrix_create_csr(hi - lo, n, rp[hi - lo], rp, ci, va, SL_MEM_HOST, lo, 0, &m) != SL_OK) DIE(25, "rank %d: create_csr", rank);
    options(&o);
    if (sl_neumann_state_create_partitioned(c, m, b, NULL, &o, &st) != SL_OK) DIE(26, "rank %d: create_partitioned", rank);
    /* b[7] += 0.5, b[n - 3] -= 0.25, b[7] += 0.125 (global rows), then a fresh series: SolverState::reset + the loop of solve() */
    upd_i[0] = 7; upd_v[0] = 0.5; upd_i[1] = n - 3; upd_v[1] = -0.25; upd_i[2] = 7; upd_v[2] = 0.125;
    if (sl_neumann_state_update_rhs(st, 3, upd_i, upd_v) != SL_OK) DIE(27, "rank %d: update_rhs", rank);
    if (sl_neumann_state_reset(st) != SL_OK) DIE(28, "rank %d: reset", rank);
    if (sl_neumann_state_run(st, NULL, &res) != SL_OK) DIE(29, "rank %d: run", rank);
    x = malloc((hi - lo + 1) * sizeof *x);
    if (sl_neumann_state_solution(st, x, SL_MEM_HOST) != SL_OK) DIE(30, "solution");
    snprintf(path, sizeof path, "/tmp/%s.rank%d", name, rank);
    f = fopen(path, "wb");
    fwrite(&res.iterations, sizeof res.iterations, 1, f); fwrite(&res.converged, sizeof res.converged, 1, f);
    fwrite(&res.residual_norm, sizeof res.residual_norm, 1, f);
    fwrite(x, sizeof *x, hi - lo, f);
    fclose(f);
    {   /* a few measured steps through the same state (bench.py's loop) */
        double nrm = -1.0; float ms = 0.f;
        uint64_t bad = 99; sl_comm_info_t ci2;
        if (sl_neumann_state_reset(st) != SL_OK || sl_neumann_state_run_steps(st, 5, &nrm, &ms) != SL_OK || !(nrm >= 0.0)) DIE(31, "run_steps");
        /* what this rank holds of its peers' rows of the current term = what the owners wrote (checksums, collective) */
        if (sl_neumann_state_verify_exchange(st, &bad) != SL_OK || bad != 0) DIE(33, "rank %d: verify_exchange, %llu pieces differ", rank, (unsigned long long)bad);
        if (sl_comm_info(c, &ci2) != SL_OK || ci2.world != world || ci2.rank != rank || ci2.ranks_joined != world || ci2.failed) DIE(34, "comm_info");
    }
    if (sl_comm_barrier(c) != SL_OK) DIE(32, "barrier");
    sl_neumann_state_destroy(st); sl_matrix_destroy(m); sl_comm_destroy(c);
    free(rp); free(ci); free(va); free(b); free(x);
    return 0;
}

int main(int argc, char **argv)
{
    int world = argc > 1 ? atoi(argv[1]) : 2, uneven = argc > 4 && strcmp(argv[4], "uneven") == 0, r, status, bad = 0;
    uint64_t n = argc > 2 ? strtoull(argv[2], NULL, 10) : 20000, w = argc > 3 ? strtoull(argv[3], NULL, 10) : 300, bnd[17];
    char name[64], path[256];
    pid_t pid[16];
    if (world < 1 || world > 16 || n < 64) { fprintf(stderr, "usage: dist_smoke <world 1..16> <n >= 64> <half_bandwidth> [uneven]\n"); return 2; }
    snprintf(name, sizeof name, "dsmoke_%ld", (long)getpid());
    for (r = 0; r < world; ++r) {
        pid[r] = fork();
        if (pid[r] < 0) { perror("fork"); return 3; }
        if (pid[r] == 0) return rank_main(r, world, n, w, uneven, name);
    }
    for (r = 0; r < world; ++r) { waitpid(pid[r], &status, 0); if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { fprintf(stderr, "rank %d exited with %d\n", r, status); bad = 1; } }
    if (bad) return 4;
    {   /* the same system, the same updates, one GPU, the same ABI */
        uint32_t *rp, *ci; double *va, *b, *x, *xr, resn = 0.0;
        sl_matrix *m = NULL; sl_neumann_state *st = NULL; sl_neumann_options o; sl_neumann_result res;
        uint64_t upd_i[3], it = 0, i, diff = 0; double upd_v[3]; int conv = 0;
        FILE *f;
        build_rows(n, w, 0, n, &rp, &ci, &va, &b);
        if (sl_matrix_create_csr(n, n, rp[n], rp, ci, va, SL_MEM_HOST, 0, 0, &m) != SL_OK) DIE(40, "whole matrix");
        options(&o);
        if (sl_neumann_state_create(m, b, NULL, &o, &st) != SL_OK) DIE(41, "state_create");
        upd_i[0] = 7; upd_v[0] = 0.5; upd_i[1] = n - 3; upd_v[1] = -0.25; upd_i[2] = 7; upd_v[2] = 0.125;
        if (sl_neumann_state_update_rhs(st, 3, upd_i, upd_v) != SL_OK || sl_neumann_state_reset(st) != SL_OK) DIE(42, "update");
        if (sl_neumann_state_run(st, NULL, &res) != SL_OK) DIE(43, "run");
        x = malloc(n * sizeof *x); xr = malloc(n * sizeof *xr);
        if (sl_neumann_state_solution(st, x, SL_MEM_HOST) != SL_OK) DIE(44, "solution");
        bounds_of(n, world, uneven, bnd);
        for (r = 0; r < world; ++r) {
            snprintf(path, sizeof path, "/tmp/%s.rank%d", name, r);
            f = fopen(path, "rb");
            if (!f || fread(&it, sizeof it, 1, f) != 1 || fread(&conv, sizeof conv, 1, f) != 1 || fread(&resn, sizeof resn, 1, f) != 1
                || fread(xr + bnd[r], sizeof *xr, bnd[r + 1] - bnd[r], f) != bnd[r + 1] - bnd[r]) { fprintf(stderr, "cannot read %s\n", path); return 5; }
            fclose(f); remove(path);
            if (it != res.iterations || conv != res.converged) { fprintf(stderr, "rank %d: %llu iterations (converged %d), one GPU: %llu (%d)\n", r,
                (unsigned long long)it, conv, (unsigned long long)res.iterations, res.converged); return 6; }
            if (fabs(resn - res.residual_norm) > 1e-12 * (1.0 + res.residual_norm)) { fprintf(stderr, "residual norms differ: %.17g vs %.17g\n", resn, res.residual_norm); return 7; }
        }
        for (i = 0; i < n; ++i) if (memcmp(&x[i], &xr[i], sizeof(double)) != 0) ++diff;
        if (diff) { fprintf(stderr, "%llu of %llu solution entries differ from the one-GPU solve\n", (unsigned long long)diff, (unsigned long long)n); return 8; }
        printf("dist_smoke ok: %d ranks, n = %llu, half bandwidth %llu%s: %llu iterations, residual %.3e, solution bit-identical to one GPU\n", world,
               (unsigned long long)n, (unsigned long long)w, uneven ? ", uneven ranges" : "", (unsigned long long)res.iterations, res.residual_norm);
        sl_neumann_state_destroy(st); sl_matrix_destroy(m);
    }
    return 0;
}
