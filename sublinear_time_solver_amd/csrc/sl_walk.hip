// sl_walk.hip — Monte-Carlo branch of estimateEntry on the GPU.
// Spec: TS estimateEntry random-walk branch (src/core/solver.ts:585-601, 630-648) over performRandomWalk
// (:390-432) and createTransitionMatrix (:359-385): absorb[i] = 1/a_ii, T[i][j] = -a_ij/a_ii; a walk from
// `row` absorbs with probability |absorb[cur]| (value += b[cur] * absorb[cur]) or moves to the first j whose
// cumulative |T[cur][.]| reaches rand; at most 1000 steps; numSamples = max(100, ceil(1/eps^2)).
// The reference draws every walk from ONE LCG stream createSeededRandom(seed) (src/core/utils.ts:161-168), serial by construction:
// where a walk starts in the stream depends on the length of every walk before it.  Here the SAME stream is cut into blocks: walk s
// reads it from position s * SL_WALK_STRIDE (2048 draws: a walk of at most 1000 steps uses at most 2000), reached by the generator's
// own jump-ahead (an affine map composed by squaring, 32 steps).  Walk 0 is the reference's first walk draw for draw; the others
// read draws the reference would have reached later or skipped — the same generator, no second source of randomness.  (Round 3's
// rule, a stream createSeededRandom(seed + s) per walk, put the FIRST draws of consecutive walks 1664525 / 2^32 = 3.9e-4 apart: the
// first draws of 400 walks covered 15 % of [0, 1) — found by comparing against the serial form, tests/test_gpu_walk.py.)  One lane
// per walk, CSR rows instead of the dense n x n transition table.  Per-walk values are bit-identical to the CPU restatement of the same rule
// (tests/test_gpu_walk.py); mean / variance are tree reductions (1e-12 relative).
// (The estimator is mirrored as behaviour; SURVEY.md Appendix A10 explains why it is not unbiased in general.)
// The block stride follows the number of walks of the call (walk_stride below): the generator's period is 2^32 draws, so 2048-draw
// blocks give 2^21 distinct starting points; a call with more walks takes narrower blocks instead of reading the same block twice.
// SL_WALK_STREAM_SERIAL is the reference's ONE stream in the reference's order, every number bit-identical to solver.ts for the same seed:
// by default as a data-parallel pipeline (a walk simulated from every position of a window of the stream, the chain position ->
// position + draws followed through them: "the serial stream, data-parallel" below), with SL_WALK_SERIAL_PLAIN=1 as written — ONE lane
// walking the stream walk after walk (sl_walk_serial_kernel, the tests' cross-check).  Mean and variance are added in walk order.
// Calls that expect to visit at least as many rows as the matrix has tabulate the rows' diagonals and weight sums first (walk_table).
#include "sl_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#define SL_WALK_STRIDE 2048ull
#define SL_WALK_MAX_TOTAL (1ull << 28)      // block form: more walks than this in one call would need blocks of fewer than 16 draws
// draws between the starting points of consecutive walks of a call with `total` walks (include/sublinear_hip.h, sl_walk_stream)
static inline uint64_t walk_stride(uint64_t total)
{
    uint64_t stride = SL_WALK_STRIDE;
    while (stride > 16 && total * stride > (1ull << 32)) stride >>= 1;
    return stride;
}
// state after k draws of createSeededRandom from `state`: x -> A x + C (mod 2^32) composed k times by squaring
__host__ __device__ inline uint32_t walk_lcg_jump(uint32_t state, uint64_t k)
{
    uint32_t cur_a = 1664525u, cur_c = 1013904223u, acc_a = 1u, acc_c = 0u;
    for (; k; k >>= 1) {
        if (k & 1ull) { acc_a = acc_a * cur_a; acc_c = acc_c * cur_a + cur_c; }
        cur_c = (cur_a + 1u) * cur_c;
        cur_a = cur_a * cur_a;
    }
    return acc_a * state + acc_c;
}
__device__ __forceinline__ double walk_lcg(uint64_t &state, uint32_t &draws)
{
    ++draws;
    state = (state * 1664525ull + 1013904223ull) & 0xffffffffull;
    return __dmul_rn((double)state, 2.3283064365386963e-10);          // / 2^32
}

// per_row = 0: all walks start at start_row, walk s reads the stream from position s * SL_WALK_STRIDE (estimateEntry).
// per_row = W > 0 (solveRandomWalk, solver.ts:300-326): walk s of this launch belongs to coordinate start_row + s / W and is that
// coordinate's walk s % W = walk number coordinate * W + s % W of the solve, whatever batch of coordinates the launch holds
// performRandomWalk (solver.ts:390-432) over the CSR row of the current state; `state` is the generator's state, advanced in place
// `draws` counts the generator's draws the walk consumed (1 per absorption test, 1 per transition: at most 2000)
// rd / rs (both or neither): the call's per-row table (sl_walk_table_kernel below) — the row's diagonal and the sum of its transition
// weights, the two of the three passes over a row that do not depend on the draw; the same operations in the same order, done once per
// row of the matrix instead of once per visit
__device__ __forceinline__ double walk_one(uint32_t start, uint64_t &state, const uint32_t *row_ptr, const uint32_t *col_idx, const double *val,
                                           const double *b, uint32_t &draws, const double *rd, const double *rs)
{
    uint32_t cur = start;
    double value = 0.0;
    for (int step = 0; step < 1000; ++step) {
        const uint32_t k0 = row_ptr[cur], k1 = row_ptr[cur + 1];
        double d = 0.0;
        if (rd) d = rd[cur];
        else for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] == cur) d = val[k];
        const double absorb = 1.0 / d;
        if (walk_lcg(state, draws) < fabs(absorb)) { value = __dadd_rn(value, __dmul_rn(b[cur], absorb)); break; }
        double sum = 0.0;
        if (rs) sum = rs[cur];
        else for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] != cur) sum = __dadd_rn(sum, fabs(-val[k] / d));
        if (sum == 0.0) { value = __dadd_rn(value, __dmul_rn(b[cur], absorb)); break; }
        const double rnd = __dmul_rn(walk_lcg(state, draws), sum);
        if (rnd <= 0.0) { cur = 0; continue; }
        double cum = 0.0;
        const uint32_t row = cur;
        for (uint32_t k = k0; k < k1; ++k) {
            if (col_idx[k] == row) continue;
            cum = __dadd_rn(cum, fabs(-val[k] / d));
            if (rnd <= cum) { cur = col_idx[k]; break; }
        }
    }
    return value;
}
// the table: one lane per row, the scans of walk_one as written there (the LAST stored match is the diagonal; the weights added in storage order)
__global__ __launch_bounds__(256) void sl_walk_table_kernel(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *val, double *rd, double *rs)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k0 = row_ptr[i], k1 = row_ptr[i + 1];
    double d = 0.0, sum = 0.0;
    for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] == (uint32_t)i) d = val[k];
    for (uint32_t k = k0; k < k1; ++k) if (col_idx[k] != (uint32_t)i) sum = __dadd_rn(sum, fabs(-val[k] / d));
    rd[i] = d; rs[i] = sum;
}
__global__ __launch_bounds__(256) void sl_walk_kernel(uint64_t n_walks, uint32_t seed, uint32_t start_row, uint64_t per_row, uint64_t stride,
                                                      const uint32_t *row_ptr, const uint32_t *col_idx, const double *val, const double *b, double *values,
                                                      const double *rd, const double *rs)
{
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n_walks) return;
    const uint64_t coord = per_row ? start_row + s / per_row : start_row;
    uint64_t state = walk_lcg_jump(seed, (per_row ? coord * per_row + s % per_row : s) * stride);
    uint32_t draws = 0;
    values[s] = walk_one((uint32_t)coord, state, row_ptr, col_idx, val, b, draws, rd, rs);
}
// SL_WALK_STREAM_SERIAL — the reference as written (solver.ts:585-601 for one row, :300-326 coordinate after coordinate): ONE lane, ONE
// stream; walk s starts where walk s - 1 stopped.  n_coords coordinates from start_row, W walks each; the walk values of coordinate c
// go to values + (keep_all ? c * W : 0) (a solve reuses one buffer of W), its mean and sample variance — added in walk order as
// Array.reduce adds them (:630-633, :313-316) — to x[c] / var[c].
__global__ void sl_walk_serial_kernel(uint64_t n_coords, uint64_t W, uint32_t seed, uint32_t start_row, int keep_all, const uint32_t *row_ptr,
                                      const uint32_t *col_idx, const double *val, const double *b, double *values, double *x, double *var)
{
    if (blockIdx.x || threadIdx.x) return;
    uint64_t state = seed;
    for (uint64_t c = 0; c < n_coords; ++c) {
        double *v = values + (keep_all ? c * W : 0);
        uint32_t draws = 0;
        for (uint64_t w = 0; w < W; ++w) v[w] = walk_one((uint32_t)(start_row + c), state, row_ptr, col_idx, val, b, draws, nullptr, nullptr);
        double m = 0.0;
        for (uint64_t w = 0; w < W; ++w) m = __dadd_rn(m, v[w]);
        m = m / (double)W;
        double q2 = 0.0;
        for (uint64_t w = 0; w < W; ++w) { const double q = __dadd_rn(v[w], -m); q2 = __dadd_rn(q2, __dmul_rn(q, q)); }
        x[c] = m;
        var[c] = W > 1 ? q2 / (double)(W - 1) : 0.0;
    }
}

// ---- SL_WALK_STREAM_SERIAL at speed: the serial stream, data-parallel ------------------------------------------------------------------
// Where walk s starts in the ONE stream depends on how many draws walks 0 .. s - 1 used — a serial dependency.  But WHAT a walk does depends
// only on the stream position it starts at (and its start row).  So, for a window of M consecutive stream positions:
//   A  (sl_walk_spec_kernel)   one lane per POSITION p: the walk the reference would perform if one of its walks started at p (the state there by
//                              the generator's jump-ahead) -> val[p], used[p] = draws it consumes (1 .. 2000).  All positions are simulated, most
//                              for nothing: the price of the speculation is the mean number of draws per walk (tens).
//   B1 (sl_walk_chunk_kernel)  the chain p -> p + used[p] is what the reference follows.  Per chunk of 4096 positions, back to front in LDS:
//                              tab[p] = (walks started inside the chunk from p on) << 16 | (offset in the NEXT chunk where the chain leaves to);
//                              a jump is at most 2000 < 4096 positions, so a chain never skips a chunk.
//   B2 (sl_walk_chain_kernel)  one lane follows the chain chunk by chunk (M / 4096 steps): where it enters each chunk, how many walks came before.
//   B3 (sl_walk_emit_kernel)   one lane per chunk follows the chain inside it and writes the walks' values IN WALK ORDER; the lane that writes the
//                              last wanted walk also says where the stream stands after it.
// The host repeats windows until the wanted number of walks is there (a window always yields walks), then sl_walk_seq_stats_kernel adds mean
// and variance in walk order (64 values loaded at once, added one after the other): the same values, the same order, the same bits as the
// one-lane kernel above (tests compare the two, and both with the reference's own TypeScript: G10 / G11).
#define SL_WALK_SPEC_CHUNK 4096u
struct sl_walk_meta { unsigned long long done, next_rel, active_chunks, pad; };
__global__ __launch_bounds__(256) void sl_walk_spec_kernel(uint64_t M, uint32_t seed, uint64_t base, uint32_t row, const uint32_t *row_ptr, const uint32_t *col_idx,
                                                           const double *val, const double *b, double *out_val, uint32_t *out_used, const double *rd, const double *rs)
{
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    uint64_t state = walk_lcg_jump(seed, base + j);
    uint32_t draws = 0;
    out_val[j] = walk_one(row, state, row_ptr, col_idx, val, b, draws, rd, rs);
    out_used[j] = draws;
}
__global__ __launch_bounds__(256) void sl_walk_chunk_kernel(uint64_t M, const uint32_t *used, uint32_t *tab)
{
    __shared__ uint32_t u[SL_WALK_SPEC_CHUNK], t[SL_WALK_SPEC_CHUNK];
    const uint64_t c0 = (uint64_t)blockIdx.x * SL_WALK_SPEC_CHUNK;
    const uint32_t len = (uint32_t)(M - c0 < SL_WALK_SPEC_CHUNK ? M - c0 : SL_WALK_SPEC_CHUNK);
    for (uint32_t e = threadIdx.x; e < len; e += 256) u[e] = used[c0 + e];
    __syncthreads();
    if (threadIdx.x == 0)
        for (uint32_t e = len; e-- > 0;) {
            const uint32_t to = e + u[e];
            t[e] = to >= len ? (1u << 16) | (to - len) : ((t[to] >> 16) + 1u) << 16 | (t[to] & 0xffffu);
        }
    __syncthreads();
    for (uint32_t e = threadIdx.x; e < len; e += 256) tab[c0 + e] = t[e];
}
__global__ void sl_walk_chain_kernel(uint64_t M, uint64_t need, const uint32_t *tab, uint32_t *chunk_entry, unsigned long long *chunk_base, sl_walk_meta *meta)
{
    if (blockIdx.x || threadIdx.x) return;
    const uint64_t nchunks = (M + SL_WALK_SPEC_CHUNK - 1) / SL_WALK_SPEC_CHUNK;
    unsigned long long done = 0, next_rel = 0, active = 0;
    uint32_t e = 0;
    bool limited = false;
    for (uint64_t c = 0; c < nchunks; ++c) {
        const uint64_t c0 = c * SL_WALK_SPEC_CHUNK;
        const uint32_t len = (uint32_t)(M - c0 < SL_WALK_SPEC_CHUNK ? M - c0 : SL_WALK_SPEC_CHUNK);
        if (e >= len) { next_rel = c0 + e; active = c; goto out; }      // (a short last chunk the chain jumps over)
        const uint32_t tv = tab[c0 + e], cnt = tv >> 16;
        chunk_entry[c] = e; chunk_base[c] = done;
        active = c + 1;
        if (done + cnt >= need) { done = need; limited = true; break; }   // the last wanted walk starts in this chunk: the emitting lane says where the stream stands then
        done += cnt;
        e = tv & 0xffffu;
    }
    if (!limited) next_rel = M + e;
out:
    meta->done = done; meta->active_chunks = active;
    meta->next_rel = limited ? 0 : next_rel;          // (limited: the emitting lane of the last walk writes it)
}
__global__ __launch_bounds__(64) void sl_walk_emit_kernel(uint64_t M, uint64_t need, const sl_walk_meta *meta, const uint32_t *chunk_entry, const unsigned long long *chunk_base,
                                                          const uint32_t *used, const double *val, double *values, sl_walk_meta *meta_out)
{
    const uint64_t c = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (c >= meta->active_chunks) return;
    const uint64_t c0 = c * SL_WALK_SPEC_CHUNK, c1 = c0 + SL_WALK_SPEC_CHUNK < M ? c0 + SL_WALK_SPEC_CHUNK : M;
    uint64_t p = c0 + chunk_entry[c], k = chunk_base[c];
    while (p < c1 && k < need) {
        values[k] = val[p];
        p += used[p];
        if (++k == need) meta_out->next_rel = p;
    }
}
// mean and sample variance added in walk order, as Array.reduce adds them (solver.ts:630-633, :313-316): one wave, 64 values loaded at a time,
// every lane adding them one after the other (uniform: all lanes hold the same running sums)
__global__ __launch_bounds__(64) void sl_walk_seq_stats_kernel(uint64_t W, const double *v, double *mean_out, double *var_out)
{
    const uint32_t lane = threadIdx.x;
    double m = 0.0;
    for (uint64_t b0 = 0; b0 < W; b0 += 64) {
        const double x = b0 + lane < W ? v[b0 + lane] : 0.0;
        const int cnt = (int)(W - b0 < 64 ? W - b0 : 64);
        for (int j = 0; j < cnt; ++j) m = __dadd_rn(m, __shfl(x, j));
    }
    m = m / (double)W;
    double q2 = 0.0;
    for (uint64_t b0 = 0; b0 < W; b0 += 64) {
        const double x = b0 + lane < W ? v[b0 + lane] : 0.0;
        const int cnt = (int)(W - b0 < 64 ? W - b0 : 64);
        for (int j = 0; j < cnt; ++j) { const double q = __dadd_rn(__shfl(x, j), -m); q2 = __dadd_rn(q2, __dmul_rn(q, q)); }
    }
    if (lane == 0) { mean_out[0] = m; var_out[0] = W > 1 ? q2 / (double)(W - 1) : 0.0; }
}

// Host side of the pipeline: `need` walks from `row`, the stream standing at absolute draw `*pos` (advanced to where it stands afterwards);
// values[need] in walk order (device).  Buffers are the caller's (sized for max_m positions).
struct sl_walk_spec_bufs { double *val; uint32_t *used, *tab, *chunk_entry; unsigned long long *chunk_base; sl_walk_meta *meta; uint64_t max_m; const double *rd, *rs; };
static sl_status sl_walk_spec_run(const sl_matrix *m, const double *db, uint32_t seed, uint32_t row, uint64_t need, uint64_t *pos, double *d_values,
                                  const sl_walk_spec_bufs &w, hipStream_t s)
{
    uint64_t done = 0, used = 0;                        // walks kept so far, draws they used: the next window is sized from their ratio
    while (done < need) {
        // positions the missing walks are expected to span (+ 1/8 and a chunk); before anything is known: 16 per walk (a walk of the
        // S-DD recipe uses ~20 draws, one on a unit diagonal 1) — a window that falls short is followed by another, one that overshoots
        // simulated positions for nothing
        uint64_t M = done ? (uint64_t)((double)(need - done) * (double)used / (double)done) : (need - done) * 16;
        M += M / 8 + SL_WALK_SPEC_CHUNK;
        M = (M + SL_WALK_SPEC_CHUNK - 1) / SL_WALK_SPEC_CHUNK * SL_WALK_SPEC_CHUNK;
        M = std::min<uint64_t>(std::max<uint64_t>(M, SL_WALK_SPEC_CHUNK), w.max_m);
        const uint64_t nchunks = M / SL_WALK_SPEC_CHUNK;
        hipLaunchKernelGGL(sl_walk_spec_kernel, dim3((uint32_t)((M + 255) / 256)), dim3(256), 0, s, M, seed, *pos, row, m->d_row_ptr, m->d_col_idx, m->d_values, db, w.val, w.used, w.rd, w.rs);
        hipLaunchKernelGGL(sl_walk_chunk_kernel, dim3((uint32_t)nchunks), dim3(256), 0, s, M, w.used, w.tab);
        hipLaunchKernelGGL(sl_walk_chain_kernel, dim3(1), dim3(1), 0, s, M, need - done, w.tab, w.chunk_entry, w.chunk_base, w.meta);
        hipLaunchKernelGGL(sl_walk_emit_kernel, dim3((uint32_t)((nchunks + 63) / 64)), dim3(64), 0, s, M, need - done, w.meta, w.chunk_entry, w.chunk_base, w.used, w.val,
                           d_values + done, w.meta);
        SL_HIP(hipGetLastError());
        sl_walk_meta h;
        SL_TRY(sl_read_back(&h, w.meta, sizeof(h), s));
        if (!h.done || !h.next_rel) return sl_fail(SL_ALGORITHM_ERROR, "internal: a window of the serial-stream pipeline yielded no walk");
        done += h.done;
        used += h.next_rel;
        *pos += h.next_rel;
    }
    return SL_OK;
}
static sl_status sl_walk_spec_alloc(uint64_t need, DevBuf &val, DevBuf &used, DevBuf &tab, DevBuf &ce, DevBuf &cb, DevBuf &meta, sl_walk_spec_bufs *w)
{
    uint64_t cap = 1ull << 22;                       // positions per window: 4 Mi (80 MB of scratch); SL_WALK_SPEC_WINDOW = a smaller one (tests: many windows on few walks)
    if (const char *e = getenv("SL_WALK_SPEC_WINDOW")) { const uint64_t v = strtoull(e, nullptr, 10); if (v) cap = std::min<uint64_t>(cap, (v + SL_WALK_SPEC_CHUNK - 1) / SL_WALK_SPEC_CHUNK * SL_WALK_SPEC_CHUNK); }
    uint64_t max_m = std::min<uint64_t>(std::max<uint64_t>((need * 64 + SL_WALK_SPEC_CHUNK - 1) / SL_WALK_SPEC_CHUNK * SL_WALK_SPEC_CHUNK, SL_WALK_SPEC_CHUNK), cap);
    const uint64_t nchunks = max_m / SL_WALK_SPEC_CHUNK;
    SL_TRY(val.alloc(max_m * 8)); SL_TRY(used.alloc(max_m * 4)); SL_TRY(tab.alloc(max_m * 4)); SL_TRY(ce.alloc(nchunks * 4)); SL_TRY(cb.alloc(nchunks * 8)); SL_TRY(meta.alloc(sizeof(sl_walk_meta)));
    *w = sl_walk_spec_bufs{val.as<double>(), used.as<uint32_t>(), tab.as<uint32_t>(), ce.as<uint32_t>(), cb.as<unsigned long long>(), static_cast<sl_walk_meta *>(meta.p), max_m, nullptr, nullptr};
    return SL_OK;
}
// The per-row table of a call (rd, rs: n doubles each) when the walks are expected to visit at least as many rows as the matrix has —
// expected_visits = walks (or simulated stream positions) x a few steps; SL_WALK_TABLE=0 / 1 forces it off / on (tests: same bits).
static sl_status walk_table(const sl_matrix *m, uint64_t expected_visits, DevBuf &rd, DevBuf &rs, const double **prd, const double **prs, hipStream_t s)
{
    *prd = *prs = nullptr;
    const char *e = getenv("SL_WALK_TABLE");
    const bool on = e && (*e == '0' || *e == '1') ? *e == '1' : expected_visits >= m->n_rows;
    if (!on || !m->n_rows) return SL_OK;
    SL_TRY(rd.alloc(m->n_rows * 8)); SL_TRY(rs.alloc(m->n_rows * 8));
    hipLaunchKernelGGL(sl_walk_table_kernel, dim3((uint32_t)((m->n_rows + 255) / 256)), dim3(256), 0, s, m->n_rows, m->d_row_ptr, m->d_col_idx, m->d_values,
                       rd.as<double>(), rs.as<double>());
    *prd = rd.as<double>(); *prs = rs.as<double>();
    return SL_OK;
}
// SL_WALK_SERIAL_PLAIN=1: the one-lane kernel instead (the cross-check of the tests)
static bool walk_serial_plain() { const char *e = getenv("SL_WALK_SERIAL_PLAIN"); return e && *e == '1'; }

// sum of (v - mean)^2 partials
__global__ __launch_bounds__(256) void sl_walk_var_kernel(uint64_t n, const double *v, double mean, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        const double q = v[i] - mean;
        acc += q * q;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}
__global__ __launch_bounds__(256) void sl_walk_sum_kernel(uint64_t n, const double *v, double *partials)
{
    __shared__ double red[4];
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) acc += v[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = ((red[0] + red[1]) + red[2]) + red[3];
}

extern "C" sl_status sl_estimate_entry_random_walk(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double epsilon,
                                                   uint32_t seed, sl_walk_stream stream, uint64_t num_samples, double *walk_values,
                                                   sl_walk_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (row >= m->n_rows) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)row, (unsigned long long)m->n_rows);
    if (!(epsilon > 0.0) && num_samples == 0) return sl_fail(SL_INVALID_INPUT, "epsilon must be positive");
    if (!m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "random-walk estimation needs the raw CSR (create with SL_MATRIX_KEEP_CSR)");
    if (stream != SL_WALK_STREAM_BLOCKS && stream != SL_WALK_STREAM_SERIAL) return sl_fail(SL_INVALID_INPUT, "unknown sl_walk_stream %d", (int)stream);
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    unsigned long long hs[4];
    SL_HIP(hipGetLastError());
    {   // createTransitionMatrix rejects a zero diagonal anywhere (solver.ts:368-371)
        sl_status st = sl_matrix_diag_pass(m, nullptr, hs);
        if (st != SL_OK) return st;
        if (hs[0] & 6ull) return sl_fail(SL_NUMERICAL_INSTABILITY, "Zero diagonal at position %llu", (hs[0] & 2ull) ? hs[2] : hs[3]);
    }
    if (num_samples == 0) {
        const double ns = std::ceil(1.0 / (epsilon * epsilon));             // solver.ts:586
        if (!(ns < 9.0e15)) return sl_fail(SL_INVALID_INPUT, "epsilon %g asks for %g walks", epsilon, ns);      // (beyond 2^53 the count is no integer any more; NaN lands here too)
        num_samples = ns > 100.0 ? (uint64_t)ns : 100;
    }
    if (stream == SL_WALK_STREAM_BLOCKS && num_samples > SL_WALK_MAX_TOTAL)
        return sl_fail(SL_INVALID_INPUT, "%llu walks in one call: the generator's period (2^32 draws) leaves blocks of fewer than 16 draws beyond 2^28 walks",
                       (unsigned long long)num_samples);
    DevBuf bbuf, vbuf;
    const double *db = b;
    if (where == SL_MEM_HOST) {
        SL_TRY(bbuf.alloc(n * 8));
        SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, hipMemcpyHostToDevice, s));
        db = bbuf.as<double>();
    }
    SL_TRY(vbuf.alloc(num_samples * 8));
    double *d_vals = vbuf.as<double>();
    sl_timer timer;
    SL_TRY(timer.start(s));
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    DevBuf rdbuf, rsbuf;
    const double *rd = nullptr, *rs = nullptr;
    if (!(stream == SL_WALK_STREAM_SERIAL && walk_serial_plain()))       // (serial pipeline: 16+ simulated positions per walk kept)
        SL_TRY(walk_table(m, num_samples * (stream == SL_WALK_STREAM_SERIAL ? 64 : 4), rdbuf, rsbuf, &rd, &rs, s));
    sl_status st = SL_OK;
    double h_sum = 0.0, h_var = 0.0;
    if (!scr) st = sl_fail(SL_ALLOCATION, "scratch");
    if (st == SL_OK && stream == SL_WALK_STREAM_SERIAL) {
        if (walk_serial_plain())
            hipLaunchKernelGGL(sl_walk_serial_kernel, dim3(1), dim3(1), 0, s, (uint64_t)1, num_samples, seed, (uint32_t)row, 1, m->d_row_ptr, m->d_col_idx,
                               m->d_values, db, d_vals, scr, scr + 1);
        else {
            DevBuf wv, wu, wt, wce, wcb, wm;
            sl_walk_spec_bufs wb;
            SL_TRY(sl_walk_spec_alloc(num_samples, wv, wu, wt, wce, wcb, wm, &wb));
            wb.rd = rd; wb.rs = rs;
            uint64_t pos = 0;
            SL_TRY(sl_walk_spec_run(m, db, seed, (uint32_t)row, num_samples, &pos, d_vals, wb, s));
            hipLaunchKernelGGL(sl_walk_seq_stats_kernel, dim3(1), dim3(64), 0, s, num_samples, d_vals, scr, scr + 1);
        }
        double h[2] = {0.0, 0.0};
        hipMemcpyAsync(h, scr, sizeof(h), hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        res->estimate = h[0]; res->variance = h[1]; res->num_samples = num_samples;
    } else if (st == SL_OK) {
        hipLaunchKernelGGL(sl_walk_kernel, dim3((uint32_t)((num_samples + 255) / 256)), dim3(256), 0, s, num_samples, seed, (uint32_t)row, (uint64_t)0,
                           walk_stride(num_samples), m->d_row_ptr, m->d_col_idx, m->d_values, db, d_vals, rd, rs);
        const uint32_t g = (uint32_t)std::min<uint64_t>((num_samples + 255) / 256, 2048);
        hipLaunchKernelGGL(sl_walk_sum_kernel, dim3(g), dim3(256), 0, s, num_samples, d_vals, scr);
        std::vector<double> part(g);
        hipMemcpyAsync(part.data(), scr, g * 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        for (uint32_t k = 0; k < g; ++k) h_sum += part[k];
        const double mean = h_sum / (double)num_samples;
        hipLaunchKernelGGL(sl_walk_var_kernel, dim3(g), dim3(256), 0, s, num_samples, d_vals, mean, scr);
        hipMemcpyAsync(part.data(), scr, g * 8, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        for (uint32_t k = 0; k < g; ++k) h_var += part[k];
        res->estimate = mean;
        res->variance = num_samples > 1 ? h_var / (double)(num_samples - 1) : 0.0;
        res->num_samples = num_samples;
    }
    const float ms = timer.stop();
    res->device_time_ms = ms;
    if (st == SL_OK && walk_values) {
        hipMemcpyAsync(walk_values, d_vals, num_samples * 8, where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s);
        hipStreamSynchronize(s);
    }
    hipError_t le = hipGetLastError();
    if (st == SL_OK && le != hipSuccess) st = sl_fail(SL_DEVICE_ERROR, "random-walk kernels failed: %s", hipGetErrorString(le));
    return st;
    SL_ABI_END
}


// ---- solveRandomWalk: the `random-walk` method of SublinearSolver.solve (src/core/solver.ts:278-357) ----------------------------
// for every coordinate i: numWalks = max(100, ceil(1 / eps^2)) walks from i (the kernel above), solution[i] = their mean,
// totalVariance += their sample variance (N - 1 denominator); then residual = ||A solution - b||_2, converged = residual < eps, and
// the reference THROWS when it is not (CONVERGENCE_FAILED, :335-341) — here SL_CONVERGENCE_FAILURE with x and the result filled.
// Streams: the reference draws all walks of all coordinates from ONE LCG stream (serial by construction); as in estimateEntry above
// walk w of coordinate i is walk number i * numWalks + w of the solve and reads the stream from that number's block (the CPU checker
// of the tests restates both this form and the reference's serial walk of the stream).
// one block per coordinate: mean and sample variance of its W walk values, fixed strided order + fixed tree = deterministic
__global__ __launch_bounds__(256) void sl_walk_rows_kernel(uint64_t W, const double *values, double *x, double *var)
{
    __shared__ double red[256];
    const double *v = values + (uint64_t)blockIdx.x * W;
    double acc = 0.0;
    for (uint64_t k = threadIdx.x; k < W; k += 256) acc = __dadd_rn(acc, v[k]);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = __dadd_rn(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    const double mean = red[0] / (double)W;
    __syncthreads();
    acc = 0.0;
    for (uint64_t k = threadIdx.x; k < W; k += 256) { const double q = __dadd_rn(v[k], -mean); acc = __dadd_rn(acc, __dmul_rn(q, q)); }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = __dadd_rn(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) { x[blockIdx.x] = mean; var[blockIdx.x] = red[0] / (double)(W - 1); }
}

extern "C" sl_status sl_solve_random_walk(const sl_matrix *m, const double *b, sl_mem where, double epsilon, uint32_t seed, sl_walk_stream stream,
                                          uint64_t num_walks, double *x, double *variances, sl_random_walk_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !x || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (!(epsilon > 0.0)) return sl_fail(SL_INVALID_INPUT, "epsilon must be positive");
    if (!m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "random-walk solve needs the raw CSR (create with SL_MATRIX_KEEP_CSR)");
    if (stream != SL_WALK_STREAM_BLOCKS && stream != SL_WALK_STREAM_SERIAL) return sl_fail(SL_INVALID_INPUT, "unknown sl_walk_stream %d", (int)stream);
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    unsigned long long hs[4];
    SL_HIP(hipGetLastError());
    SL_TRY(sl_matrix_diag_pass(m, nullptr, hs));
    if (hs[0] & 6ull) return sl_fail(SL_NUMERICAL_INSTABILITY, "Zero diagonal at position %llu", (hs[0] & 2ull) ? hs[2] : hs[3]);     // solver.ts:368-371
    if (num_walks == 0) {
        const double ns = std::ceil(1.0 / (epsilon * epsilon));             // solver.ts:303
        if (!(ns < 9.0e15)) return sl_fail(SL_INVALID_INPUT, "epsilon %g asks for %g walks per coordinate", epsilon, ns);
        num_walks = ns > 100.0 ? (uint64_t)ns : 100;
    }
    if (num_walks < 2) return sl_fail(SL_INVALID_INPUT, "at least two walks per coordinate (the sample variance divides by N - 1)");
    if (num_walks > (1ull << 26)) return sl_fail(SL_INVALID_INPUT, "%llu walks per coordinate (epsilon %g): more than 2^26", (unsigned long long)num_walks, epsilon);
    const bool serial = stream == SL_WALK_STREAM_SERIAL;
    if (!serial && n * num_walks > SL_WALK_MAX_TOTAL)
        return sl_fail(SL_INVALID_INPUT, "%llu coordinates x %llu walks: the generator's period (2^32 draws) leaves blocks of fewer than 16 draws beyond 2^28 walks",
                       (unsigned long long)n, (unsigned long long)num_walks);
    const uint64_t stride = walk_stride(n * num_walks);
    const uint64_t batch = serial ? 1 : std::max<uint64_t>(1, std::min<uint64_t>(n, (1ull << 26) / num_walks));      // at most 512 MB of walk values at a time
    DevBuf bbuf, vbuf, xbuf, varbuf, ybuf;
    const double *db = b;
    if (where == SL_MEM_HOST) {
        SL_TRY(bbuf.alloc(n * 8));
        SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, hipMemcpyHostToDevice, s));
        db = bbuf.as<double>();
    }
    SL_TRY(vbuf.alloc(batch * num_walks * 8)); SL_TRY(xbuf.alloc(n * 8)); SL_TRY(varbuf.alloc(n * 8)); SL_TRY(ybuf.alloc(n * 8));
    sl_timer timer;
    SL_TRY(timer.start(s));
    DevBuf rdbuf, rsbuf;
    const double *rd = nullptr, *rs = nullptr;
    if (!(serial && walk_serial_plain())) SL_TRY(walk_table(m, n * num_walks, rdbuf, rsbuf, &rd, &rs, s));      // (every row starts >= 100 walks)
    if (serial && n && walk_serial_plain())       // the reference as written: one lane, one stream, coordinate after coordinate
        hipLaunchKernelGGL(sl_walk_serial_kernel, dim3(1), dim3(1), 0, s, n, num_walks, seed, 0u, 0, m->d_row_ptr, m->d_col_idx, m->d_values, db,
                           vbuf.as<double>(), xbuf.as<double>(), varbuf.as<double>());
    else if (serial && n) {                       // the same stream, every position of it simulated in parallel, coordinate after coordinate (pipeline above)
        DevBuf wv, wu, wt, wce, wcb, wm;
        sl_walk_spec_bufs wb;
        SL_TRY(sl_walk_spec_alloc(num_walks, wv, wu, wt, wce, wcb, wm, &wb));
        wb.rd = rd; wb.rs = rs;
        uint64_t pos = 0;
        for (uint64_t i = 0; i < n; ++i) {
            SL_TRY(sl_walk_spec_run(m, db, seed, (uint32_t)i, num_walks, &pos, vbuf.as<double>(), wb, s));
            hipLaunchKernelGGL(sl_walk_seq_stats_kernel, dim3(1), dim3(64), 0, s, num_walks, vbuf.as<double>(), xbuf.as<double>() + i, varbuf.as<double>() + i);
        }
    }
    for (uint64_t i0 = 0; i0 < n && !serial; i0 += batch) {
        const uint64_t rows = std::min(batch, n - i0), walks = rows * num_walks;
        hipLaunchKernelGGL(sl_walk_kernel, dim3((uint32_t)((walks + 255) / 256)), dim3(256), 0, s, walks, seed, (uint32_t)i0, num_walks, stride,
                           m->d_row_ptr, m->d_col_idx, m->d_values, db, vbuf.as<double>(), rd, rs);
        hipLaunchKernelGGL(sl_walk_rows_kernel, dim3((uint32_t)rows), dim3(256), 0, s, num_walks, vbuf.as<double>(), xbuf.as<double>() + i0, varbuf.as<double>() + i0);
    }
    SL_HIP(hipGetLastError());
    // residual = ||A x - b||_2 (solver.ts:328-333) through the library's own primitives, device-resident
    SL_TRY(sl_spmv(m, xbuf.as<double>(), ybuf.as<double>(), SL_ORDER_CSR_SEQUENTIAL, SL_MEM_DEVICE));
    SL_TRY(sl_axpy(n, -1.0, db, ybuf.as<double>(), SL_MEM_DEVICE));
    double resn = 0.0, tv = 0.0;
    SL_TRY(sl_l2_norm(n, ybuf.as<double>(), &resn, SL_MEM_DEVICE));
    {   // totalVariance: the coordinates' variances added in coordinate order, as the reference adds them (:319)
        std::vector<double> hv(n);
        if (n) SL_TRY(sl_read_back(hv.data(), varbuf.p, n * 8, s));
        for (uint64_t i = 0; i < n; ++i) tv += hv[i];
        if (variances && where == SL_MEM_HOST) memcpy(variances, hv.data(), n * 8);
    }
    res->device_time_ms = timer.stop();
    if (n) {
        SL_HIP(hipMemcpyAsync(x, xbuf.p, n * 8, where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
        if (variances && where != SL_MEM_HOST) SL_HIP(hipMemcpyAsync(variances, varbuf.p, n * 8, hipMemcpyDeviceToDevice, s));
        SL_HIP(hipStreamSynchronize(s));
    }
    res->iterations = n; res->num_walks = num_walks; res->residual = resn; res->total_variance = tv;
    res->converged = resn < epsilon ? 1 : 0;
    if (!res->converged)
        return sl_fail(SL_CONVERGENCE_FAILURE, "Random walk sampling failed to achieve desired accuracy (final residual %.6g, variance %.6g)", resn, std::sqrt(tv));
    return SL_OK;
    SL_ABI_END
}
