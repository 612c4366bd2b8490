/* layout_stress.c — the layout build alone, in many processes at once on one GPU: what the many-rank campaigns do for a second per case,
 * a few thousand times a minute.  Round 3 saw, twice in ~2200 many-process runs, a row lose its diagonal because the slice pointers on the
 * device did not match the row lengths there; the build now counts such rows, says which link of the chain was off (the read-back of the
 * widths, the row-length kernel's result, the uploaded pointers) on stderr — "did not fit their slices" — and repairs it.  This tool puts
 * that build under the same contention without the rest of a partitioned solve:
 *
 *   layout_stress <processes> <builds per process> <rows> [pageable-every-k]
 *
 * forks <processes> children BEFORE touching the HIP runtime; each builds the ragged rows of dist_smoke (2..9 off-diagonals, diagonal
 * last inserted) for its own row range from host arrays (SL_MEM_HOST: the path of the cases that failed), asks the dominance check (a
 * row that lost its diagonal fails it) and destroys the matrix, <builds> times, alternating between a slice, a transpose-carrying build
 * and plain ones.  Children with index % k == 0 run with SL_STAGING=pageable (the round-3 copies).  Exit status: number of builds whose
 * dominance check failed (0 = none); the library's own misfit reports go to stderr, grep for "did not fit".
 * build: gcc -std=c99 -O2 -Iinclude tools/layout_stress.c -o tools/layout_stress -Lsublinear_time_solver_amd -lsublinear_hip -lm -Wl,-rpath,$PWD/sublinear_time_solver_amd */
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

static void build_rows(uint64_t n, uint64_t w, uint64_t lo, uint64_t hi, uint64_t salt, uint32_t **rp_o, uint32_t **ci_o, double **va_o)
{
    uint64_t rows = hi - lo, cap = rows * 10 + 1, nnz = 0, i, k;
    uint32_t *rp = malloc((rows + 1) * sizeof *rp), *ci = malloc(cap * sizeof *ci);
    double *va = malloc(cap * sizeof *va);
    rp[0] = 0;
    for (i = lo; i < hi; ++i) {
        uint64_t m = 2 + mix(i * 3 + 1 + salt) % 8, cols[10], cnt = 0, a, c, j;
        double vals[10], off = 0.0;
        uint64_t wlo = i > w ? i - w : 0, whi = i + w + 1 < n ? i + w + 1 : n;
        for (k = 0; k < m; ++k) {
            c = wlo + mix(i * 131 + k * 7 + 5 + salt) % (whi - wlo);
            if (c == i) continue;
            for (a = 0; a < cnt && cols[a] != c; ++a) {}
            if (a < cnt) continue;
            cols[cnt] = c; vals[cnt] = (double)(mix(i * 977 + k) % 2001) / 1000.0 - 1.0; off += fabs(vals[cnt]); ++cnt;
        }
        cols[cnt] = i; vals[cnt] = 2.0 * off + 1.0; ++cnt;
        for (a = 1; a < cnt; ++a) {
            uint64_t cc = cols[a]; double vv = vals[a];
            for (j = a; j > 0 && cols[j - 1] > cc; --j) { cols[j] = cols[j - 1]; vals[j] = vals[j - 1]; }
            cols[j] = cc; vals[j] = vv;
        }
        for (a = 0; a < cnt; ++a) { ci[nnz] = (uint32_t)cols[a]; va[nnz] = vals[a]; ++nnz; }
        rp[i - lo + 1] = (uint32_t)nnz;
    }
    *rp_o = rp; *ci_o = ci; *va_o = va;
}

static int child(int idx, int procs, long builds, uint64_t n)
{
    int ndev = 0, bad = 0;
    long b;
    if (sl_device_count(&ndev) != SL_OK || ndev <= 0) { fprintf(stderr, "no device\n"); return 99; }
    if (sl_set_device(idx % ndev) != SL_OK) return 98;
    for (b = 0; b < builds; ++b) {
        /* a different slice and bandwidth every few builds, so that allocations of different sizes come and go */
        const uint64_t share = n / (uint64_t)procs, lo = (uint64_t)idx * share, hi = idx == procs - 1 ? n : lo + share - (uint64_t)(b % 7) * 64;
        const uint64_t w = (b % 3 == 0) ? 40 : (b % 3 == 1) ? 300 : 5000;
        uint32_t *rp, *ci; double *va;
        sl_matrix *m = NULL;
        int dd = 0;
        build_rows(n, w, lo, hi, (uint64_t)(b / 16), &rp, &ci, &va);
        if (sl_matrix_create_csr(hi - lo, n, rp[hi - lo], rp, ci, va, SL_MEM_HOST, lo, (b % 5 == 0 && lo == 0 && hi == n) ? SL_MATRIX_WITH_TRANSPOSE : 0, &m) != SL_OK) {
            fprintf(stderr, "child %d build %ld: create_csr failed [%s]\n", idx, b, sl_last_error_message());
            ++bad;
        } else {
            if (sl_matrix_is_diagonally_dominant(m, &dd) != SL_OK || !dd) {
                fprintf(stderr, "child %d build %ld: a dominant matrix FAILED the dominance check (rows [%llu, %llu), w = %llu) [%s]\n", idx, b,
                        (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)w, sl_last_error_message());
                ++bad;
            }
            sl_matrix_destroy(m);
        }
        free(rp); free(ci); free(va);
    }
    return bad > 90 ? 90 : bad;
}

int main(int argc, char **argv)
{
    int procs, r, total = 0, every = 2;
    long builds;
    uint64_t n;
    pid_t pid[64];
    if (argc < 4) { fprintf(stderr, "usage: layout_stress <processes> <builds per process> <rows> [pageable-every-k]\n"); return 2; }
    procs = atoi(argv[1]); builds = atol(argv[2]); n = strtoull(argv[3], NULL, 10);
    if (argc > 4) every = atoi(argv[4]);
    if (procs < 1 || procs > 64 || n < 64ull * (uint64_t)procs * 8) return 2;
    for (r = 0; r < procs; ++r) {
        pid[r] = fork();
        if (pid[r] < 0) { perror("fork"); return 3; }
        if (pid[r] == 0) {
            if (every > 0 && r % every == 0) setenv("SL_STAGING", "pageable", 1);
            exit(child(r, procs, builds, n));
        }
    }
    for (r = 0; r < procs; ++r) {
        int st = 0;
        waitpid(pid[r], &st, 0);
        if (!WIFEXITED(st)) { fprintf(stderr, "child %d died\n", r); total += 1; }
        else total += WEXITSTATUS(st);
    }
    printf("{\"processes\": %d, \"builds_per_process\": %ld, \"rows\": %llu, \"pageable_children\": \"%s\", \"builds\": %ld, \"failed_dominance_checks_or_builds\": %d}\n",
           procs, builds, (unsigned long long)n, every > 0 ? "index % k == 0" : "none", builds * procs, total);
    return total ? 1 : 0;
}
