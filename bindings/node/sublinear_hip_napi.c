/* N-API addon over the C ABI of include/sublinear_hip.h — the binding a maintainer of the reference's shipped
 * TypeScript surface (src/core/solver.ts, compiled to dist/core/solver.js) would load in place of its JS loops and
 * WASM module.  Plain C against node_api.h; built by `make -C bindings/node` (gcc, no node-gyp).  Typed arrays cross the
 * boundary without copies on the way in; results come back as new Float64Array / plain objects.
 *
 * exported (all synchronous, they throw an Error whose .status is the sl_status and .kind its name):
 *   createMatrix(rows, cols, rowIdx:Float64Array|Uint32Array, colIdx, values:Float64Array, withTranspose:bool) -> handle
 *   destroyMatrix(handle)            matrixInfo(handle) -> {...}       isDiagonallyDominant(handle) -> bool
 *   neumannSolve(handle, b:Float64Array, {tolerance,maxIterations,maxTerms,seriesTolerance}) -> {...}
 *   pushSolve(handle, b, {theta,maxRounds}) -> {...}
 *   estimateEntry(handle, b, row, theta, maxRounds) -> {...}
 *   estimateEntryRandomWalk(handle, b, row, epsilon, seed) -> {...}
 *   cgSolve(handle, b, {tolerance,maxIterations}) -> {...}
 *   deviceCount() -> number
 */
#include <node_api.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "sublinear_hip.h"

#define NAPI_OK(call)                                                                   \
    do {                                                                                \
        if ((call) != napi_ok) {                                                        \
            napi_throw_error(env, NULL, "N-API call failed: " #call);                   \
            return NULL;                                                                \
        }                                                                               \
    } while (0)

static napi_value throw_status(napi_env env, sl_status st)
{
    napi_value msg, err, v;
    char buf[640];
    snprintf(buf, sizeof(buf), "%s: %s", sl_status_string(st), sl_last_error_message());
    napi_create_string_utf8(env, buf, NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, NULL, msg, &err);
    napi_create_int32(env, (int32_t)st, &v);
    napi_set_named_property(env, err, "status", v);
    napi_create_string_utf8(env, sl_status_string(st), NAPI_AUTO_LENGTH, &v);
    napi_set_named_property(env, err, "kind", v);
    napi_throw(env, err);
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv)
{
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

/* Float64Array -> pointer + length (no copy) */
static int f64_view(napi_env env, napi_value v, const double **p, size_t *n)
{
    napi_typedarray_type t;
    void *data;
    bool is_ta = false;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta ||
        napi_get_typedarray_info(env, v, &t, n, &data, NULL, NULL) != napi_ok || t != napi_float64_array) {
        napi_throw_type_error(env, NULL, "expected a Float64Array");
        return 0;
    }
    *p = (const double *)data;
    return 1;
}

/* index array (Uint32Array, Int32Array or Float64Array) -> freshly allocated uint64_t[] */
static uint64_t *index_copy(napi_env env, napi_value v, size_t *n)
{
    napi_typedarray_type t;
    void *data;
    bool is_ta = false;
    uint64_t *out;
    size_t i;
    if (napi_is_typedarray(env, v, &is_ta) != napi_ok || !is_ta || napi_get_typedarray_info(env, v, &t, n, &data, NULL, NULL) != napi_ok) {
        napi_throw_type_error(env, NULL, "expected a typed index array");
        return NULL;
    }
    out = (uint64_t *)malloc((*n ? *n : 1) * sizeof(uint64_t));
    if (!out) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    for (i = 0; i < *n; ++i) {
        double d;
        if (t == napi_uint32_array) d = (double)((const uint32_t *)data)[i];
        else if (t == napi_int32_array) d = (double)((const int32_t *)data)[i];
        else if (t == napi_float64_array) d = ((const double *)data)[i];
        else { free(out); napi_throw_type_error(env, NULL, "index arrays must be Uint32Array, Int32Array or Float64Array"); return NULL; }
        if (!(d >= 0.0) || d != (double)(uint64_t)d) { free(out); napi_throw_range_error(env, NULL, "negative or non-integer index"); return NULL; }
        out[i] = (uint64_t)d;
    }
    return out;
}

static double opt_number(napi_env env, napi_value obj, const char *key, double dflt)
{
    napi_value v;
    napi_valuetype t;
    double d;
    bool has = false;
    if (napi_typeof(env, obj, &t) != napi_ok || t != napi_object) return dflt;
    if (napi_has_named_property(env, obj, key, &has) != napi_ok || !has) return dflt;
    if (napi_get_named_property(env, obj, key, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok || t != napi_number) return dflt;
    if (napi_get_value_double(env, v, &d) != napi_ok) return dflt;
    return d;
}

static void set_num(napi_env env, napi_value obj, const char *key, double d)
{
    napi_value v;
    napi_create_double(env, d, &v);
    napi_set_named_property(env, obj, key, v);
}
static void set_bool(napi_env env, napi_value obj, const char *key, int b)
{
    napi_value v;
    napi_get_boolean(env, b != 0, &v);
    napi_set_named_property(env, obj, key, v);
}

/* The JS handle is an external holding a box, not the sl_matrix itself: destroyMatrix() empties the box, so a second destroy
 * or any later use throws instead of touching freed memory; a handle that is garbage collected releases what is left. */
typedef struct { sl_matrix *m; } matrix_box;

static void matrix_box_finalize(napi_env env, void *data, void *hint)
{
    matrix_box *b = (matrix_box *)data;
    (void)env; (void)hint;
    if (b) { if (b->m) sl_matrix_destroy(b->m); free(b); }
}

static matrix_box *box_of(napi_env env, napi_value v)
{
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected a matrix handle");
        return NULL;
    }
    return (matrix_box *)p;
}

static sl_matrix *matrix_of(napi_env env, napi_value v)
{
    matrix_box *b = box_of(env, v);
    if (!b) return NULL;
    if (!b->m) { napi_throw_error(env, NULL, "matrix handle was destroyed"); return NULL; }
    return b->m;
}

/* new Float64Array(n) backed by its own ArrayBuffer; *data receives the storage */
static napi_value new_f64(napi_env env, size_t n, double **data)
{
    napi_value ab, ta;
    void *p = NULL;
    if (napi_create_arraybuffer(env, n * sizeof(double), &p, &ab) != napi_ok ||
        napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta) != napi_ok) {
        napi_throw_error(env, NULL, "could not allocate the result array");
        return NULL;
    }
    *data = (double *)p;
    return ta;
}

static napi_value CreateMatrix(napi_env env, napi_callback_info info)
{
    napi_value argv[6], out;
    double rows, cols;
    const double *vals;
    size_t nr = 0, nc = 0, nv = 0;
    uint64_t *ri, *ci;
    bool with_t = false;
    sl_matrix *m = NULL;
    matrix_box *box;
    sl_status st;
    if (!get_args(env, info, 6, argv)) return NULL;
    NAPI_OK(napi_get_value_double(env, argv[0], &rows));
    NAPI_OK(napi_get_value_double(env, argv[1], &cols));
    if (!f64_view(env, argv[4], &vals, &nv)) return NULL;
    napi_get_value_bool(env, argv[5], &with_t);
    ri = index_copy(env, argv[2], &nr);
    if (!ri) return NULL;
    ci = index_copy(env, argv[3], &nc);
    if (!ci) { free(ri); return NULL; }
    if (nr != nv || nc != nv) { free(ri); free(ci); napi_throw_range_error(env, NULL, "COO matrix arrays must have same length"); return NULL; }
    st = sl_matrix_create_from_triplets(nv, ri, ci, vals, (uint64_t)rows, (uint64_t)cols,
                                        (with_t ? SL_MATRIX_WITH_TRANSPOSE : 0u) | SL_MATRIX_KEEP_CSR, &m);
    free(ri);
    free(ci);
    if (st != SL_OK) return throw_status(env, st);
    box = (matrix_box *)malloc(sizeof(*box));
    if (!box) { sl_matrix_destroy(m); napi_throw_error(env, NULL, "out of memory"); return NULL; }
    box->m = m;
    if (napi_create_external(env, box, matrix_box_finalize, NULL, &out) != napi_ok) { matrix_box_finalize(env, box, NULL); napi_throw_error(env, NULL, "could not create the matrix handle"); return NULL; }
    return out;
}

static napi_value DestroyMatrix(napi_env env, napi_callback_info info)
{
    napi_value argv[1];
    matrix_box *b;
    if (!get_args(env, info, 1, argv)) return NULL;
    b = box_of(env, argv[0]);
    if (b && b->m) { sl_matrix_destroy(b->m); b->m = NULL; }      /* idempotent */
    return NULL;
}

static napi_value MatrixInfo(napi_env env, napi_callback_info info)
{
    napi_value argv[1], o;
    sl_matrix *m;
    sl_matrix_info mi;
    sl_status st;
    if (!get_args(env, info, 1, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0]))) return NULL;
    st = sl_matrix_get_info(m, &mi);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "rows", (double)mi.n_rows);
    set_num(env, o, "cols", (double)mi.n_cols);
    set_num(env, o, "nnz", (double)mi.nnz);
    set_num(env, o, "maxRowNnz", (double)mi.max_row_nnz);
    set_num(env, o, "deviceBytes", (double)mi.device_bytes);
    return o;
}

static napi_value IsDiagonallyDominant(napi_env env, napi_callback_info info)
{
    napi_value argv[1], v;
    sl_matrix *m;
    int32_t f = 0;
    sl_status st;
    if (!get_args(env, info, 1, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0]))) return NULL;
    st = sl_matrix_is_diagonally_dominant(m, &f);
    if (st != SL_OK) return throw_status(env, st);
    napi_get_boolean(env, f != 0, &v);
    return v;
}

static napi_value NeumannSolve(napi_env env, napi_callback_info info)
{
    napi_value argv[3], o, sol;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    sl_neumann_options opt;
    sl_neumann_result r;
    sl_matrix_info mi;
    double *x, *tn;
    sl_status st;
    if (!get_args(env, info, 3, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    if (sl_matrix_get_info(m, &mi) != SL_OK || nb != mi.n_rows) { napi_throw_range_error(env, NULL, "vector length does not match the matrix"); return NULL; }
    sl_neumann_options_default(&opt);
    opt.tolerance = opt_number(env, argv[2], "tolerance", opt.tolerance);
    opt.max_iterations = (uint64_t)opt_number(env, argv[2], "maxIterations", (double)opt.max_iterations);
    opt.max_terms = (uint64_t)opt_number(env, argv[2], "maxTerms", (double)opt.max_terms);
    opt.series_tolerance = opt_number(env, argv[2], "seriesTolerance", opt.series_tolerance);
    opt.mem = SL_MEM_HOST;
    if (!(sol = new_f64(env, nb, &x))) return NULL;
    tn = (double *)calloc(opt.max_terms ? opt.max_terms : 1, sizeof(double));
    if (!tn) { napi_throw_error(env, NULL, "out of memory"); return NULL; }
    st = sl_neumann_solve(m, b, NULL, &opt, x, tn, &r);
    free(tn);
    if (st != SL_OK && st != SL_CONVERGENCE_FAILURE) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "solution", sol);
    set_num(env, o, "iterations", (double)r.iterations);
    set_num(env, o, "residualNorm", r.residual_norm);
    set_bool(env, o, "converged", r.converged);
    set_num(env, o, "termsComputed", (double)r.terms_computed);
    set_num(env, o, "deviceTimeMs", r.device_time_ms);
    set_num(env, o, "deviceBytes", (double)mi.device_bytes);
    return o;
}

static napi_value PushSolve(napi_env env, napi_callback_info info)
{
    napi_value argv[3], o, sol;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    sl_push_options opt;
    sl_push_result r;
    sl_matrix_info mi;
    double *x;
    sl_status st;
    if (!get_args(env, info, 3, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    if (sl_matrix_get_info(m, &mi) != SL_OK || nb != mi.n_rows) { napi_throw_range_error(env, NULL, "vector length does not match the matrix"); return NULL; }
    sl_push_options_default(&opt);
    opt.theta = opt_number(env, argv[2], "theta", opt.theta);
    opt.max_rounds = (uint64_t)opt_number(env, argv[2], "maxRounds", (double)opt.max_rounds);
    opt.mem = SL_MEM_HOST;
    if (!(sol = new_f64(env, nb, &x))) return NULL;
    memset(x, 0, nb * sizeof(double));
    st = sl_push_solve(m, b, &opt, x, NULL, NULL, 0, NULL, &r);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "solution", sol);
    set_num(env, o, "rounds", (double)r.rounds);
    set_num(env, o, "pushes", (double)r.pushes);
    set_num(env, o, "residualNorm", r.residual_norm);
    set_bool(env, o, "converged", r.converged);
    set_num(env, o, "deviceTimeMs", r.device_time_ms);
    set_num(env, o, "deviceBytes", (double)mi.device_bytes);
    return o;
}

/* solveForwardPush in the reference's own visiting order (src/core/solver.ts:437-522): one Gauss-Southwell push per iteration */
static napi_value ForwardPushSouthwell(napi_env env, napi_callback_info info)
{
    napi_value argv[3], o, sol;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    sl_southwell_options opt;
    sl_southwell_result r;
    sl_matrix_info mi;
    double *x;
    sl_status st;
    if (!get_args(env, info, 3, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    if (sl_matrix_get_info(m, &mi) != SL_OK || nb != mi.n_rows) { napi_throw_range_error(env, NULL, "vector length does not match the matrix"); return NULL; }
    sl_southwell_options_default(&opt);
    opt.epsilon = opt_number(env, argv[2], "epsilon", opt.epsilon);
    opt.max_iterations = (uint64_t)opt_number(env, argv[2], "maxIterations", (double)opt.max_iterations);
    opt.mem = SL_MEM_HOST;
    if (!(sol = new_f64(env, nb, &x))) return NULL;
    st = sl_forward_push_southwell(m, b, &opt, x, NULL, NULL, 0, &r);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "solution", sol);
    set_num(env, o, "iterations", (double)r.iterations);
    set_num(env, o, "residualNorm", r.residual_norm);
    set_bool(env, o, "converged", r.converged);
    set_num(env, o, "deviceTimeMs", r.device_time_ms);
    set_num(env, o, "deviceBytes", (double)mi.device_bytes);
    return o;
}

static napi_value EstimateEntry(napi_env env, napi_callback_info info)
{
    napi_value argv[5], o;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    double row, theta, max_rounds;
    sl_estimate_result r;
    sl_status st;
    if (!get_args(env, info, 5, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    NAPI_OK(napi_get_value_double(env, argv[2], &row));
    NAPI_OK(napi_get_value_double(env, argv[3], &theta));
    NAPI_OK(napi_get_value_double(env, argv[4], &max_rounds));
    st = sl_estimate_entry(m, b, SL_MEM_HOST, (uint64_t)row, theta, (uint64_t)max_rounds, &r);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "estimate", r.estimate);
    set_num(env, o, "residualL1", r.residual_l1);
    set_num(env, o, "rounds", (double)r.rounds);
    set_num(env, o, "pushes", (double)r.pushes);
    set_num(env, o, "rowsTouched", (double)r.rows_touched);
    set_bool(env, o, "converged", r.converged);
    return o;
}

static napi_value EstimateEntryRandomWalk(napi_env env, napi_callback_info info)
{
    napi_value argv[6], o;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    double row, eps, seed, stream;
    sl_walk_result r;
    sl_status st;
    if (!get_args(env, info, 6, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    NAPI_OK(napi_get_value_double(env, argv[2], &row));
    NAPI_OK(napi_get_value_double(env, argv[3], &eps));
    NAPI_OK(napi_get_value_double(env, argv[4], &seed));
    NAPI_OK(napi_get_value_double(env, argv[5], &stream));      /* sl_walk_stream: 0 blocks, 1 the reference's serial stream */
    st = sl_estimate_entry_random_walk(m, b, SL_MEM_HOST, (uint64_t)row, eps, (uint32_t)seed, (sl_walk_stream)(int)stream, 0, NULL, &r);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    set_num(env, o, "estimate", r.estimate);
    set_num(env, o, "variance", r.variance);
    set_num(env, o, "numSamples", (double)r.num_samples);
    return o;
}

/* solveRandomWalk (core/solver.ts:278-357): randomWalkSolve(matrix, b, epsilon, seed, stream) -> { solution, iterations, residualNorm,
 * converged, totalVariance, numWalks, deviceBytes }; a residual that misses epsilon is reported, the JS class throws as the reference does */
static napi_value RandomWalkSolve(napi_env env, napi_callback_info info)
{
    napi_value argv[5], o, sol;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    double eps, seed, stream, *x;
    sl_random_walk_result r;
    sl_matrix_info mi;
    sl_status st;
    if (!get_args(env, info, 5, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    if (sl_matrix_get_info(m, &mi) != SL_OK || nb != mi.n_rows) { napi_throw_range_error(env, NULL, "vector length does not match the matrix"); return NULL; }
    NAPI_OK(napi_get_value_double(env, argv[2], &eps));
    NAPI_OK(napi_get_value_double(env, argv[3], &seed));
    NAPI_OK(napi_get_value_double(env, argv[4], &stream));
    if (!(sol = new_f64(env, nb, &x))) return NULL;
    st = sl_solve_random_walk(m, b, SL_MEM_HOST, eps, (uint32_t)seed, (sl_walk_stream)(int)stream, 0, x, NULL, &r);
    if (st != SL_OK && st != SL_CONVERGENCE_FAILURE) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "solution", sol);
    set_num(env, o, "iterations", (double)r.iterations);
    set_num(env, o, "residualNorm", r.residual);
    set_bool(env, o, "converged", r.converged);
    set_num(env, o, "totalVariance", r.total_variance);
    set_num(env, o, "numWalks", (double)r.num_walks);
    set_num(env, o, "deviceBytes", (double)mi.device_bytes);
    return o;
}

static napi_value CgSolve(napi_env env, napi_callback_info info)
{
    napi_value argv[3], o, sol;
    sl_matrix *m;
    const double *b;
    size_t nb = 0;
    sl_cg_options opt;
    sl_cg_result r;
    sl_matrix_info mi;
    double *x;
    sl_status st;
    if (!get_args(env, info, 3, argv)) return NULL;
    if (!(m = matrix_of(env, argv[0])) || !f64_view(env, argv[1], &b, &nb)) return NULL;
    if (sl_matrix_get_info(m, &mi) != SL_OK || nb != mi.n_rows) { napi_throw_range_error(env, NULL, "vector length does not match the matrix"); return NULL; }
    sl_cg_options_default(&opt);
    opt.tolerance = opt_number(env, argv[2], "tolerance", opt.tolerance);
    opt.max_iterations = (uint64_t)opt_number(env, argv[2], "maxIterations", (double)opt.max_iterations);
    opt.mem = SL_MEM_HOST;
    if (!(sol = new_f64(env, nb, &x))) return NULL;
    st = sl_cg_solve(m, b, &opt, x, &r);
    if (st != SL_OK) return throw_status(env, st);
    NAPI_OK(napi_create_object(env, &o));
    napi_set_named_property(env, o, "solution", sol);
    set_num(env, o, "iterations", (double)r.iterations);
    set_num(env, o, "residualNorm", r.residual_norm);
    set_bool(env, o, "converged", r.converged);
    return o;
}

static napi_value DeviceCount(napi_env env, napi_callback_info info)
{
    napi_value v;
    int n = 0;
    (void)info;
    sl_device_count(&n);
    napi_create_int32(env, n, &v);
    return v;
}

static napi_value Init(napi_env env, napi_value exports)
{
    const napi_property_descriptor props[] = {
        {"createMatrix", NULL, CreateMatrix, NULL, NULL, NULL, napi_default, NULL},
        {"destroyMatrix", NULL, DestroyMatrix, NULL, NULL, NULL, napi_default, NULL},
        {"matrixInfo", NULL, MatrixInfo, NULL, NULL, NULL, napi_default, NULL},
        {"isDiagonallyDominant", NULL, IsDiagonallyDominant, NULL, NULL, NULL, napi_default, NULL},
        {"neumannSolve", NULL, NeumannSolve, NULL, NULL, NULL, napi_default, NULL},
        {"pushSolve", NULL, PushSolve, NULL, NULL, NULL, napi_default, NULL},
        {"forwardPushSouthwell", NULL, ForwardPushSouthwell, NULL, NULL, NULL, napi_default, NULL},
        {"estimateEntry", NULL, EstimateEntry, NULL, NULL, NULL, napi_default, NULL},
        {"estimateEntryRandomWalk", NULL, EstimateEntryRandomWalk, NULL, NULL, NULL, napi_default, NULL},
        {"randomWalkSolve", NULL, RandomWalkSolve, NULL, NULL, NULL, napi_default, NULL},
        {"cgSolve", NULL, CgSolve, NULL, NULL, NULL, napi_default, NULL},
        {"deviceCount", NULL, DeviceCount, NULL, NULL, NULL, napi_default, NULL},
    };
    if (sl_abi_version() != SL_ABI_VERSION) {       /* a stale libsublinear_hip.so: refuse at load, not at the first missing symbol */
        napi_throw_error(env, NULL, "libsublinear_hip.so ABI version differs from the header this addon was built against");
        return NULL;
    }
    if (napi_define_properties(env, exports, sizeof(props) / sizeof(props[0]), props) != napi_ok) return NULL;
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
