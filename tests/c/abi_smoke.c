/* The drop-in boundary is a plain C ABI: this file is compiled with gcc -std=c99 -pedantic (no C++, no HIP headers) and
 * linked against libsublinear_hip.so.  Without a GPU it checks that the library loads, reports its version and fails
 * loudly (SL_DEVICE_ERROR, no CPU fallback); with a GPU (argv[1] = "gpu") it solves a 3 x 3 system through the ABI. */
#include <stdio.h>
#include <string.h>
#include <math.h>
#include "sublinear_hip.h"

int main(int argc, char **argv)
{
    const uint64_t rows[] = {0, 0, 1, 1, 1, 2, 2}, cols[] = {0, 1, 0, 1, 2, 1, 2};
    const double vals[] = {4.0, 1.0, -1.0, 5.0, 2.0, 0.5, 3.0}, b[] = {1.0, 2.0, -1.0};
    sl_matrix *m = NULL;
    sl_status st;
    if (sl_abi_version() != SL_ABI_VERSION) { printf("version mismatch\n"); return 2; }
    st = sl_matrix_create_from_triplets(7, rows, cols, vals, 3, 3, SL_MATRIX_WITH_TRANSPOSE, &m);
    if (argc < 2 || strcmp(argv[1], "gpu") != 0) {
        int n = -1;
        sl_device_count(&n);
        if (n > 0) { printf("device present, run with 'gpu'\n"); if (m) sl_matrix_destroy(m); return 0; }
        if (st != SL_DEVICE_ERROR || m != NULL) { printf("expected SL_DEVICE_ERROR without a GPU, got %d\n", (int)st); return 3; }
        printf("no device: %s (%s)\n", sl_status_string(st), sl_last_error_message());
        return 0;
    }
    if (st != SL_OK) { printf("create failed: %s\n", sl_last_error_message()); return 4; }
    {
        sl_neumann_options o;
        sl_neumann_result r;
        sl_estimate_result e;
        double x[3], tn[64];
        sl_neumann_options_default(&o);
        o.tolerance = 1e-12;
        o.series_tolerance = 1e-14;     /* the series is cut at this term norm (neumann.rs:270-274); default 1e-8 */
        o.max_terms = 64;
        st = sl_neumann_solve(m, b, NULL, &o, x, tn, &r);
        if (st != SL_OK || !r.converged) { printf("solve failed: %s\n", sl_last_error_message()); return 5; }
        /* residual of the returned x, on the host */
        {
            const double r0 = b[0] - (4.0 * x[0] + 1.0 * x[1]), r1 = b[1] - (-1.0 * x[0] + 5.0 * x[1] + 2.0 * x[2]),
                         r2 = b[2] - (0.5 * x[1] + 3.0 * x[2]);
            if (sqrt(r0 * r0 + r1 * r1 + r2 * r2) > 1e-10) { printf("residual too large\n"); return 6; }
        }
        st = sl_estimate_entry(m, b, SL_MEM_HOST, 1, 1e-14, 100000, &e);
        if (st != SL_OK || fabs(e.estimate - x[1]) > 1e-10) { printf("estimate_entry mismatch\n"); return 7; }
        printf("ok: x = %.15g %.15g %.15g, %llu iterations, estimate[1] = %.15g\n", x[0], x[1], x[2], (unsigned long long)r.iterations, e.estimate);
    }
    sl_matrix_destroy(m);
    return 0;
}
