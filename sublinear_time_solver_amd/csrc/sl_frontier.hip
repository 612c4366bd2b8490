// sl_frontier.hip — frontier machinery of the synchronous thresholded push:
//   * ordered stream compaction by wavefront ballot + popcount + prefix scan (ascending frontier),
//   * sparse rounds: candidate-row expansion over columns, pull update of candidate rows only,
//   * the host round loop (sl_push_solve) switching between sparse rounds and the dense
//     row-slice kernel, and sl_estimate_entry (local push on A^T).
//
// Spec: ForwardPushSolver::push_node (solver/forward_push.rs:179-216) and TS solveForwardPush
// (src/core/solver.ts:437-522) — invariant r = b - A x; here every round pushes ALL rows above
// the threshold at once; each candidate row's update is a pull over its own CSR row in column
// order (product rounded, then added), so the result does not depend on scheduling: frontier
// sets and values are bit-identical to the sequential CPU restatement.
#include "sl_internal.hpp"
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <vector>

sl_status sl_sort_keys_u32(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n, hipStream_t s);

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DSUB(a, b) __dsub_rn((a), (b))

namespace {

// operator view: CSR rows of the operator B, and CSR rows of B^T (= columns of B)
struct op_view {
    const uint32_t *ptr, *idx; const double *val;   // rows of B
    const uint32_t *tptr, *tidx;                    // columns of B (pattern)
    const uint32_t *tk;                             // for each column entry: index of the same entry in the row arrays
};
} // namespace

// ---- ordered compaction ---------------------------------------------------------------------
// tile = 2048 consecutive indices per 256-thread block; wave w owns 512 of them, 8 passes of 64.
#define SL_CTILE 2048

__device__ __forceinline__ bool sl_pred(const double *delta, sl_theta theta, uint64_t i)
{
    return theta.everything() ? true : (delta[i] != 0.0);
}

__global__ __launch_bounds__(256) void sl_compact_count_kernel(uint64_t n, const double *delta, sl_theta theta, uint32_t *block_count)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SL_CTILE + (uint64_t)wave * 512;
    uint32_t cnt = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        const bool f = i < n && sl_pred(delta, theta, i);
        cnt += (uint32_t)__popcll(__ballot(f));
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single-block exclusive scan of block counts (nb <= a few million); total -> *total_out
__global__ __launch_bounds__(1024) void sl_scan_kernel(uint32_t nb, const uint32_t *in, uint32_t *out, uint32_t *total_out)
{
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (nb + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * chunk;
    const uint32_t hi = lo + chunk < nb ? lo + chunk : nb;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; ++k) s += in[k];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int t = 0; t < 1024; ++t) { const uint32_t v = part[t]; part[t] = run; run += v; }
        *total_out = run;
    }
    __syncthreads();
    uint32_t run = part[threadIdx.x];
    for (uint32_t k = lo; k < hi; ++k) { const uint32_t v = in[k]; out[k] = run; run += v; }
}

__global__ __launch_bounds__(256) void sl_compact_write_kernel(uint64_t n, const double *delta, sl_theta theta,
                                                               const uint32_t *block_off, uint32_t *list)
{
    __shared__ uint32_t wsum[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * SL_CTILE + (uint64_t)wave * 512;
    unsigned long long masks[8];
    uint32_t cnt = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        const bool f = i < n && sl_pred(delta, theta, i);
        masks[p] = __ballot(f);
        cnt += (uint32_t)__popcll(masks[p]);
    }
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    uint32_t off = block_off[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const uint64_t i = base + (uint64_t)p * 64 + lane;
        if ((masks[p] >> lane) & 1ull) list[off + (uint32_t)__popcll(masks[p] & lt)] = (uint32_t)i;
        off += (uint32_t)__popcll(masks[p]);
    }
}

// ---- dense select (round 0): delta_i = r_i*dinv_i if |.| >= theta else 0 --------------------
__global__ __launch_bounds__(256) void sl_select_kernel(uint64_t n, const double *r, const double *dinv, sl_theta theta,
                                                        double *delta)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double p = DMUL(r[i], dinv[i]);
        delta[i] = (fabs(p) >= theta.at(i)) ? p : 0.0;
    }
}

// ---- sparse round ------------------------------------------------------------------------------
// Sparse rounds are DEVICE-DRIVEN: the list sizes live in a control block in HBM, every kernel of a round reads
// them there and strides over its list with a fixed grid, and a one-thread kernel closes the round (statistics,
// list swap, stop rule).  The host enqueues several rounds back to back and reads the block once per batch; rounds
// enqueued past the stop are empty launches.  A local query of a dozen rounds costs two host round trips.
struct sl_push_ctl {
    // every wave of a round reserves list slots through these counters: each sits on its own 128-byte line, so the atomics on
    // different counters do not queue behind one another in one L2 channel
    alignas(128) uint32_t nrec;           // hit records of the running round
    alignas(128) uint32_t nc;             // candidate rows of the running round
    alignas(128) uint32_t n_touched;      // rows whose x / r may be non-zero (query sessions; not reset per batch)
    alignas(128) uint32_t nn;             // next frontier
    alignas(128) unsigned long long next_hits;   // column entries under the next frontier
    alignas(128) uint32_t n_long_cols;    // (column, piece) work items of the running round
    alignas(128) uint32_t n_heavy;        // candidate rows with more than SL_MAX_HITS hits
    alignas(128) uint32_t nf;             // |frontier|
    uint32_t stop;                        // 0 running, 1 frontier empty (converged), 2 frontier too large for a sparse round
    uint32_t rounds, done_blocks;         // rounds executed in this batch; block counter of the closing kernel
    uint32_t nf_prev;                     // |frontier| of the round before (cleared at the head of the next expansion)
    unsigned long long hits;              // column entries under the frontier (= records a round writes)
    unsigned long long pushes, rows_touched;
};
#define SL_FLAG_TOUCHED 2u            // row is in the session's touched list
#define SL_EMPTY 0xffffffffu          // head[] of a row without hits
#define SL_MAX_HITS 8                 // rows hit by more frontier columns than this are pulled over their whole length
#define SL_PIECE 256u                 // column entries one wave expands at a time; longer frontier columns are cut into pieces

// one frontier column entry (i, j): the product B_ij * delta_j, the position of the entry in B's row arrays (the order
// key of the row sum) and the previous hit of the same row (a per-row linked list, newest first)
struct sl_hit { double prod; uint32_t k, next; };

// everything a sparse round touches.  Round k of a batch reads frontier[k & 1] / delta[k & 1] and writes the other pair: the
// parity comes from the round counter in the control block, so the same argument block serves every round of the batch
struct sl_round_io {
    sl_push_ctl *c;
    uint32_t *frontier[2];
    double *delta[2];
    op_view op;
    double *x, *r;
    const double *dinv;
    sl_hit *recs;
    uint32_t *head, *cand, *flag, *touched, *heavy;
    uint2 *long_cols;                          // (column, piece) work items of the running round
    sl_theta theta;
    int order;
    uint32_t dense_threshold, round_limit;     // a batch runs rounds while c->rounds < round_limit
    unsigned long long hit_limit;
    uint32_t nf0;                              // |frontier| the batch starts from
    unsigned long long rec_cap;                // record buffer capacity (hard limit on the entries under a frontier)
};
// The argument block of a batch lives in device memory (written by one small kernel per batch): every kernel of the batch takes
// the same 8-byte pointer instead of ~200 bytes of arguments (launches got 15 % cheaper; a local query is a few dozen launches).
__global__ void sl_set_io_kernel(sl_round_io io, sl_round_io *dst) { *dst = io; }
#define SL_PICK(pair, which) ((which) ? (pair)[1] : (pair)[0])      // no dynamic indexing into kernel arguments
// where a thread sits in the machine (the round phases are written against this, not against blockIdx / gridDim directly)
struct sl_worker { uint32_t tid, nthreads, wave, nwaves, lane; };
__device__ __forceinline__ sl_worker sl_worker_here()
{
    sl_worker w;
    w.tid = blockIdx.x * blockDim.x + threadIdx.x; w.nthreads = gridDim.x * blockDim.x;
    w.wave = w.tid >> 6; w.nwaves = w.nthreads >> 6; w.lane = threadIdx.x & 63u;
    return w;
}
template <typename T> __device__ __forceinline__ T sl_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// The control block and the round limit are kernel ARGUMENTS of their own (round 4): the gate words and the argument block are then
// fetched side by side — one round trip to memory at the head of every launch of a train instead of two (block, then block->c->...).
__device__ __forceinline__ bool sl_round_open(const sl_push_ctl *c, uint32_t round_limit, uint32_t &par)
{
    const uint32_t stop = c->stop, rounds = c->rounds;
    if (stop || rounds >= round_limit) return false;
    par = rounds & 1u;
    return true;
}

// (every kernel of a batch takes its argument block at iop[blockIdx.y]: one launch serves one query — grid.y = 1 — or the W queries of a
// wide batch, each with a state of its own)
__global__ void sl_push_ctl_reset_kernel(const sl_round_io *iop)
{
    iop += blockIdx.y;
    sl_push_ctl *c = iop->c;
    // (an empty first frontier — the seed below its threshold — is a query that is over before its first round: only a wide batch asks)
    c->nf = iop->nf0; c->nn = 0; c->n_long_cols = 0; c->nrec = 0; c->nc = 0; c->n_heavy = 0; c->stop = iop->nf0 ? 0u : 1u; c->rounds = 0; c->done_blocks = 0; c->nf_prev = 0;
    c->hits = 0; c->next_hits = 0; c->pushes = 0; c->rows_touched = 0;
}
// column entries under the first frontier of a batch (later rounds accumulate the figure as their frontier forms)
__global__ __launch_bounds__(256) void sl_frontier_hits_kernel(const sl_round_io *iop)
{
    iop += blockIdx.y;
    sl_push_ctl *c = iop->c;
    const uint32_t *frontier = iop->frontier[0], *tptr = iop->op.tptr;
    const uint32_t nf = iop->nf0;
    unsigned long long acc = 0;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < nf; t += gridDim.x * 256) { const uint32_t j = frontier[t]; acc += tptr[j + 1] - tptr[j]; }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63u) == 0 && acc) atomicAdd(&c->hits, acc);
}
__global__ void sl_hits_gate_kernel(const sl_round_io *iop)
{
    iop += blockIdx.y;
    if (iop->c->hits > iop->rec_cap) iop->c->stop = 2u;
}

// (1) expansion.  Sparse rounds are HIT-DRIVEN: a frontier column j reaches row i through the entry B_ij, and only
//     those entries contribute to (B delta)_i.  A wave takes four short frontier columns per step (columns of more than 64
//     entries: 256-entry pieces, second kernel), emits a record per entry and links it into its row's list with one
//     atomicExch; a row whose list was empty becomes a candidate.  Slots in the record, candidate and touched lists are
//     reserved per wave (ballot + popcount, one atomic for up to 256 entries).  The work of a round follows the column
//     entries under the frontier, not the lengths of the rows they touch.
// Latency, not bandwidth, bounds a sparse round, so the piece is written as three round trips to memory: (1) the slot
// reservation (it needs only the entry count) next to the loads of row and position; (2) the list link (needs the row and the
// slot), the value and the row's session flag together; (3) the candidate / touched reservations, then plain stores.
// A piece = four segments of up to 64 column entries, each with its own delta: four short columns, or 256 consecutive entries
// of one long column.  pa[u] + lane < pe[u] selects the lanes of segment u.
__device__ __forceinline__ void sl_expand_piece(const uint32_t pa[4], const uint32_t pe[4], const double dj[4], uint32_t lane, const op_view &op,
                                                sl_hit *recs, uint32_t *head, uint32_t *cand, uint32_t *flag, uint32_t *touched, sl_push_ctl *c)
{
    const unsigned long long below = (1ull << lane) - 1ull;
    bool ok[4];
    unsigned long long m[4];
    uint32_t off[4], total = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) { ok[u] = pa[u] + lane < pe[u]; m[u] = __ballot(ok[u]); off[u] = total; total += (uint32_t)__popcll(m[u]); }
    if (total == 0) return;                                                    // wave-uniform
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&c->nrec, total);
    uint32_t row[4], kb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t p = pa[u] + lane;
        row[u] = ok[u] ? op.tidx[p] : 0u;
        kb[u] = ok[u] ? op.tk[p] : 0u;
    }
    base = __shfl(base, 0);
    uint32_t prev[4], rec[4], fl[4];
    double bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        rec[u] = base + off[u] + (uint32_t)__popcll(m[u] & below);
        prev[u] = ok[u] ? atomicExch(&head[row[u]], rec[u]) : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        bv[u] = ok[u] ? op.val[kb[u]] : 0.0;
        fl[u] = (touched && ok[u]) ? flag[row[u]] : 0u;      // only the lane that links a row first this round may change its flag
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) if (ok[u]) recs[rec[u]] = sl_hit{DMUL(bv[u], dj[u]), kb[u], prev[u]};
    // rows reached for the first time this round; among them, rows reached for the first time in this query
    bool fresh[4], first[4];
    unsigned long long mf[4], mt[4];
    uint32_t offf[4], offt[4], nfresh = 0, nfirst = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        fresh[u] = ok[u] && prev[u] == SL_EMPTY;
        first[u] = fresh[u] && touched && !(fl[u] & SL_FLAG_TOUCHED);
        mf[u] = __ballot(fresh[u]); offf[u] = nfresh; nfresh += (uint32_t)__popcll(mf[u]);
        mt[u] = __ballot(first[u]); offt[u] = nfirst; nfirst += (uint32_t)__popcll(mt[u]);
    }
    if (nfresh == 0) return;                                                   // wave-uniform
    uint32_t basec = 0, baset = 0;
    if (lane == 0) { basec = atomicAdd(&c->nc, nfresh); if (nfirst) baset = atomicAdd(&c->n_touched, nfirst); }
    basec = __shfl(basec, 0); baset = __shfl(baset, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (fresh[u]) cand[basec + offf[u] + (uint32_t)__popcll(mf[u] & below)] = row[u];
        if (first[u]) { flag[row[u]] = fl[u] | SL_FLAG_TOUCHED; touched[baset + offt[u] + (uint32_t)__popcll(mt[u] & below)] = row[u]; }
    }
}

__device__ __forceinline__ void sl_phase_expand(const sl_round_io &io, uint32_t par, uint32_t nf, const sl_worker &w)
{
    const uint32_t *frontier = SL_PICK(io.frontier, par);
    const double *delta = SL_PICK(io.delta, par);
    // four frontier columns per wave and step (lanes 0..3 fetch one each): most columns are short, and the slot reservations of a
    // piece — the only traffic all waves aim at the same few addresses — are shared by four columns
    for (uint32_t t0 = w.wave * 4u; t0 < nf; t0 += w.nwaves * 4u) {
        uint32_t j = 0, a = 0, e = 0;
        double d = 0.0;
        if (w.lane < 4u && t0 + w.lane < nf) {
            j = frontier[t0 + w.lane];
            d = delta[j];
            io.x[j] = DADD(io.x[j], d);
            a = io.op.tptr[j]; e = io.op.tptr[j + 1];
        }
        const bool is_long = e - a > 64u;                                       // more than one segment: (column, piece) items for the second phase
        const uint32_t pieces = is_long ? (e - a + SL_PIECE - 1u) / SL_PIECE : 0u;
        const unsigned long long ml = __ballot(is_long);
        if (ml) {
            uint32_t pre = 0, tot = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t pu = __shfl(pieces, u); if ((uint32_t)u < w.lane) pre += pu; tot += pu; }
            uint32_t base = 0;
            if (w.lane == 0) base = atomicAdd(&io.c->n_long_cols, tot);
            base = __shfl(base, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t pu = __shfl(pieces, u), ju = __shfl(j, u), bu = base + __shfl(pre, u);
                for (uint32_t q = w.lane; q < pu; q += 64u) io.long_cols[bu + q] = uint2{ju, q};
            }
        }
        uint32_t pa[4], pe[4];
        double dj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool lu = (ml >> u) & 1ull;
            pa[u] = lu ? 0u : __shfl(a, u); pe[u] = lu ? 0u : __shfl(e, u); dj[u] = __shfl(d, u);
        }
        sl_expand_piece(pa, pe, dj, w.lane, io.op, io.recs, io.head, io.cand, io.flag, io.touched, io.c);
    }
}

// the pieces of the longer columns, one wave each: a hub column is spread over the machine
__device__ __forceinline__ void sl_phase_expand_long(const sl_round_io &io, uint32_t par, uint32_t n_items, const sl_worker &w)
{
    const double *delta = SL_PICK(io.delta, par);
    for (uint32_t t = w.wave; t < n_items; t += w.nwaves) {
        const uint2 item = io.long_cols[t];
        const uint32_t j = item.x;
        const double d = delta[j];
        const uint32_t p0 = io.op.tptr[j] + item.y * SL_PIECE, p1 = io.op.tptr[j + 1];
        uint32_t pa[4], pe[4];
        double dj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { pa[u] = p0 + (uint32_t)u * 64u; pe[u] = p1; dj[u] = d; }
        sl_expand_piece(pa, pe, dj, w.lane, io.op, io.recs, io.head, io.cand, io.flag, io.touched, io.c);
    }
}

__device__ __forceinline__ double sl_csr_row_dot(const op_view &op, uint32_t i, const double *v, int order)
{
    const uint32_t s = op.ptr[i], e = op.ptr[i + 1];
    const uint32_t len = e - s;
    if (order == SL_ORDER_SIMD4 && len >= 8u) {       // simd_ops.rs:41-77
        const uint32_t chunks = len >> 2;
        double l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
        for (uint32_t q = 0; q < chunks; ++q) {
            const uint32_t k = s + 4 * q;
            l0 = DADD(l0, DMUL(op.val[k], v[op.idx[k]]));
            l1 = DADD(l1, DMUL(op.val[k + 1], v[op.idx[k + 1]]));
            l2 = DADD(l2, DMUL(op.val[k + 2], v[op.idx[k + 2]]));
            l3 = DADD(l3, DMUL(op.val[k + 3], v[op.idx[k + 3]]));
        }
        double y = DADD(DADD(DADD(l0, l1), l2), l3);
        for (uint32_t k = s + 4 * chunks; k < e; ++k) y = DADD(y, DMUL(op.val[k], v[op.idx[k]]));
        return y;
    }
    double acc = 0.0;                                   // sparse.rs:194-202
    for (uint32_t k = s; k < e; ++k) acc = DADD(acc, DMUL(op.val[k], v[op.idx[k]]));
    return acc;
}

// Ordered accumulation of the non-zero products of one row.  Only NON-ZERO products are added, in entry order —
// leaving out an exactly-zero product cannot change the running sum (s + (+-0) == s, and s is never -0), so the value
// equals the sequential reference sum over the whole row bit for bit.  `add` carries the reference's two summation
// orders (sparse.rs:194-202 and the 4-lane order of simd_ops.rs:41-77); positions must arrive ascending.
struct sl_row_acc {
    double sum = 0.0, l0 = 0.0, l1 = 0.0, l2 = 0.0, l3 = 0.0;
    uint32_t chunks4 = 0;
    bool lanes4 = false, merged = false;
    __device__ __forceinline__ void init(uint32_t len, int order)
    {
        lanes4 = (order == SL_ORDER_SIMD4 && len >= 8u);
        chunks4 = lanes4 ? ((len >> 2) << 2) : 0u;
    }
    __device__ __forceinline__ void add(uint32_t q, double pv)              // q = position of the entry in its row
    {
        if (q < chunks4) {
            const uint32_t ln = q & 3u;
            if (ln == 0) l0 = DADD(l0, pv); else if (ln == 1) l1 = DADD(l1, pv); else if (ln == 2) l2 = DADD(l2, pv); else l3 = DADD(l3, pv);
        } else {
            if (lanes4 && !merged) { sum = DADD(DADD(DADD(l0, l1), l2), l3); merged = true; }
            sum = DADD(sum, pv);
        }
    }
    __device__ __forceinline__ double finish()
    {
        if (lanes4 && !merged) sum = DADD(DADD(DADD(l0, l1), l2), l3);
        return sum;
    }
};

// (2) candidate rows: r_i -= (B delta_old)_i ; next frontier from |r_i dinv_i| >= theta.  Appends to the next frontier
//     (and the count of column entries under it) are reserved per wave.
__device__ __forceinline__ void sl_pull_finish_wave(bool live, uint32_t i, double acc, double r_old, double dinv_i, sl_theta theta, uint32_t lane,
                                                    const uint32_t *tptr, double *r, double *delta_new, uint32_t *next, sl_push_ctl *c)
{
    double p = 0.0;
    if (live) {
        const double rn = DSUB(r_old, acc);
        r[i] = rn;
        p = DMUL(rn, dinv_i);
    }
    const bool pass = live && fabs(p) >= theta.at(i);
    const unsigned long long m = __ballot(pass);
    if (!m) return;
    unsigned long long colw = pass ? (unsigned long long)(tptr[i + 1] - tptr[i]) : 0ull;
    for (int off = 32; off > 0; off >>= 1) colw += __shfl_xor(colw, off);
    uint32_t base = 0;
    const int leader = __builtin_ctzll(m);
    if ((int)lane == leader) { base = atomicAdd(&c->nn, (uint32_t)__popcll(m)); atomicAdd(&c->next_hits, colw); }
    base = __shfl(base, leader);
    if (pass) { delta_new[i] = p; next[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i; }
}

// whole-row pull of one row by a whole wave (rows hit by more frontier columns than the insertion network holds): 1024 entries
// (16 per lane) in flight per step, non-zero products added in entry order
__device__ __forceinline__ double sl_wave_row_pull(const op_view &op, uint32_t i, const double *delta_old, int order, uint32_t lane)
{
    const uint32_t s = op.ptr[i], len = op.ptr[i + 1] - s;
    sl_row_acc acc;
    acc.init(len, order);
    for (uint32_t base = 0; base < len; base += 1024u) {
        double pr[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t q = base + (uint32_t)u * 64u + lane;
            pr[u] = q < len ? DMUL(op.val[s + q], delta_old[op.idx[s + q]]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            unsigned long long seg = __ballot(pr[u] != 0.0);
            while (seg) {
                const int l = __builtin_ctzll(seg);
                seg &= seg - 1;
                acc.add(base + (uint32_t)u * 64u + (uint32_t)l, __shfl(pr[u], l));
            }
        }
    }
    return acc.finish();
}

// thread per candidate row: walk the row's hit list (newest first), order the hits by position with a static
// insertion network, add them up.  Rows hit more than SL_MAX_HITS times go to the heavy list (whole-row pull, next launch:
// pulling them in place was tried — the waves that find several of them hold the round up, 2x slower on large frontiers).
__device__ __forceinline__ void sl_phase_pull_hits(const sl_round_io &io, uint32_t par, uint32_t nc, const sl_worker &w)
{
    const op_view &op = io.op;
    double *delta_new = SL_PICK(io.delta, 1u - par);
    uint32_t *next = SL_PICK(io.frontier, 1u - par);
    const int order = io.order;
    const uint32_t trips = (nc + w.nthreads - 1) / w.nthreads;                 // wave-uniform (ballots in the finish)
    for (uint32_t it = 0; it < trips; ++it) {
        const uint32_t t = it * w.nthreads + w.tid;
        bool live = t < nc;
        const uint32_t i = live ? io.cand[t] : 0u;
        uint32_t h = SL_EMPTY;
        double r_old = 0.0, dv = 0.0;
        uint32_t s = 0, len = 0;
        if (live) {
            h = io.head[i]; io.head[i] = SL_EMPTY;
            r_old = io.r[i]; dv = io.dinv[i];
            if (order == SL_ORDER_SIMD4) { s = op.ptr[i]; len = op.ptr[i + 1] - s; }
        }
        uint32_t hk[SL_MAX_HITS]; double hp[SL_MAX_HITS];
#pragma unroll
        for (int u = 0; u < SL_MAX_HITS; ++u) { hk[u] = SL_EMPTY; hp[u] = 0.0; }
#pragma unroll
        for (int hop = 0; hop < SL_MAX_HITS; ++hop) {
            if (h != SL_EMPTY) {
                const sl_hit rec = io.recs[h];
                h = rec.next;
                uint32_t nk = rec.k; double np = rec.prod;
#pragma unroll
                for (int u = 0; u < SL_MAX_HITS; ++u) {                        // keep hk ascending: bubble the new hit into place
                    if (nk < hk[u]) { const uint32_t tk = hk[u]; const double tp = hp[u]; hk[u] = nk; hp[u] = np; nk = tk; np = tp; }
                }
            }
        }
        if (live && h != SL_EMPTY) { io.heavy[atomicAdd(&io.c->n_heavy, 1u)] = i; live = false; }   // more hits than the network holds
        sl_row_acc acc;
        acc.init(len, order);
#pragma unroll
        for (int u = 0; u < SL_MAX_HITS; ++u) if (hk[u] != SL_EMPTY) acc.add(hk[u] - s, hp[u]);
        sl_pull_finish_wave(live, i, acc.finish(), r_old, dv, io.theta, w.lane, op.tptr, io.r, delta_new, next, io.c);
    }
}

// heavy rows: one WAVE per row over the whole row
__device__ __forceinline__ void sl_phase_pull_heavy(const sl_round_io &io, uint32_t par, uint32_t nh, const sl_worker &w)
{
    const double *delta_old = SL_PICK(io.delta, par);
    for (uint32_t t = w.wave; t < nh; t += w.nwaves) {
        const uint32_t i = io.heavy[t];
        double r_old = 0.0, dv = 0.0;
        if (w.lane == 0) { r_old = io.r[i]; dv = io.dinv[i]; }
        const double sum = sl_wave_row_pull(io.op, i, delta_old, io.order, w.lane);
        sl_pull_finish_wave(w.lane == 0, i, sum, r_old, dv, io.theta, w.lane, io.op.tptr, io.r, SL_PICK(io.delta, 1u - par), SL_PICK(io.frontier, 1u - par), io.c);
    }
}

// delta[j] = 0 over a consumed frontier (keeps the buffer all-zero outside a frontier).  It runs at the head of the NEXT round's
// expansion — which reads the other buffer, while the list is still intact — and once after the last round of a batch.
__device__ __forceinline__ void sl_clear_frontier(const uint32_t *frontier, double *delta, uint32_t nf, const sl_worker &w)
{
    for (uint32_t t = w.tid; t < nf; t += w.nthreads) delta[frontier[t]] = 0.0;
}
// ONE thread closes the round once every pull has finished: statistics, the next frontier becomes the frontier, stop rule
__device__ __forceinline__ void sl_close_round(const sl_round_io &io, uint32_t nf)
{
    sl_push_ctl *c = io.c;
    const uint32_t nn = sl_ld(&c->nn), nc = sl_ld(&c->nc);
    const unsigned long long nh = sl_ld(&c->next_hits);
    c->rounds += 1; c->pushes += nf; c->rows_touched += nc;
    c->nf_prev = nf;
    c->nf = nn; c->nn = 0; c->nc = 0; c->n_heavy = 0; c->n_long_cols = 0; c->nrec = 0; c->hits = nh; c->next_hits = 0;
    if (nn == 0) c->stop = 1u;
    else if (nn > io.dense_threshold || nh > io.hit_limit) c->stop = 2u;
}

// A round is four launches: expansion of the short columns (+ clearing the previous frontier), the pieces of the long
// columns, the pull over hit lists, and the whole-row pull of heavy rows, whose last block closes the round.
__global__ __launch_bounds__(256) void sl_expand_kernel(const sl_round_io *iop, const sl_push_ctl *cg, uint32_t round_limit)
{
    const sl_round_io io = iop[blockIdx.y];
    uint32_t par;
    if (!sl_round_open(cg ? cg : io.c, round_limit, par)) return;      // (cg: the one query's control block as an argument of its own; a wide batch passes null)
    const sl_worker w = sl_worker_here();
    if (io.c->rounds) sl_clear_frontier(SL_PICK(io.frontier, 1u - par), SL_PICK(io.delta, 1u - par), io.c->nf_prev, w);
    sl_phase_expand(io, par, io.c->nf, w);
}
__global__ __launch_bounds__(256) void sl_expand_long_kernel(const sl_round_io *iop, const sl_push_ctl *cg, uint32_t round_limit)
{
    const sl_round_io io = iop[blockIdx.y];
    uint32_t par;
    if (!sl_round_open(cg ? cg : io.c, round_limit, par)) return;      // (cg: the one query's control block as an argument of its own; a wide batch passes null)
    sl_phase_expand_long(io, par, io.c->n_long_cols, sl_worker_here());
}
__global__ __launch_bounds__(256) void sl_pull_hits_kernel(const sl_round_io *iop, const sl_push_ctl *cg, uint32_t round_limit)
{
    const sl_round_io io = iop[blockIdx.y];
    uint32_t par;
    if (!sl_round_open(cg ? cg : io.c, round_limit, par)) return;      // (cg: the one query's control block as an argument of its own; a wide batch passes null)
    sl_phase_pull_hits(io, par, io.c->nc, sl_worker_here());
}
// the block that finishes last closes the round (a small grid: every block pays an agent-scope fence on the way out)
__global__ __launch_bounds__(256) void sl_pull_heavy_kernel(const sl_round_io *iop, const sl_push_ctl *cg, uint32_t round_limit)
{
    const sl_round_io io = iop[blockIdx.y];
    uint32_t par;
    if (!sl_round_open(cg ? cg : io.c, round_limit, par)) return;      // (cg: the one query's control block as an argument of its own; a wide batch passes null)
    sl_push_ctl *c = io.c;
    const uint32_t nf = c->nf;
    sl_phase_pull_heavy(io, par, c->n_heavy, sl_worker_here());
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence();
    if (atomicAdd(&c->done_blocks, 1u) != gridDim.x - 1) return;
    c->done_blocks = 0;
    sl_close_round(io, nf);
}
// SMALL rounds, several of them, in ONE launch of ONE workgroup (round 4, SL_PUSH_SMALL=1).  A local query starts — and ends — with
// frontiers of a few columns: each such round is microseconds of work behind four kernel boundaries of ~4.5 us.  Here one 8-wave
// workgroup runs the four phases of a round back to back behind block barriers, round after round, for as long as the round at hand is
// small (|F| <= max_nf and column entries under it <= max_hits) and the batch's round limit allows; a round that is too large — it
// wants the whole machine — is left untouched for the launch train behind this kernel, which is followed by this kernel again.
// The phases are the SAME device functions as the launch train's (written against sl_worker, not against the grid), so every value —
// records, hit order, sums, frontier SETS — is the same; the order of a frontier LIST differs as it does between two runs of the train
// (slots are reserved by atomics), which nothing downstream depends on.  Between phases: block barrier (workgroup release: the waves'
// stores and atomics have left) + an agent-scope ACQUIRE fence in every wave (its CU's L1 forgets what it held: lists and heads that
// another wave changed through an L2 atomic are re-read) — no release at agent scope, no other workgroup takes part.
#define SL_SMALL_THREADS 512      // 8 waves, up to 256 VGPRs each: the phases' register arrays do not spill; a small round is latency, not throughput
__global__ __launch_bounds__(SL_SMALL_THREADS) void sl_small_rounds_kernel(const sl_round_io *iop, uint32_t max_nf, unsigned long long max_hits)
{
    __shared__ uint32_t s_go, s_par, s_nf, s_prev, s_cnt;
    const sl_round_io io = iop[blockIdx.y];
    sl_push_ctl *c = io.c;
    sl_worker w;
    w.tid = threadIdx.x; w.nthreads = SL_SMALL_THREADS; w.wave = threadIdx.x >> 6; w.nwaves = SL_SMALL_THREADS / 64u; w.lane = threadIdx.x & 63u;
    auto phase_sync = [&]() { __syncthreads(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); };
    for (;;) {
        if (threadIdx.x == 0) {
            const uint32_t stop = sl_ld(&c->stop), rounds = sl_ld(&c->rounds), nf = sl_ld(&c->nf);
            const unsigned long long hits = sl_ld(&c->hits);
            s_go = (!stop && rounds < io.round_limit && nf <= max_nf && hits <= max_hits) ? 1u : 0u;
            s_par = rounds & 1u; s_nf = nf; s_prev = rounds ? sl_ld(&c->nf_prev) : 0u;
        }
        __syncthreads();
        if (!s_go) return;                                                      // block-uniform
        const uint32_t par = s_par, nf = s_nf;
        if (s_prev) sl_clear_frontier(SL_PICK(io.frontier, 1u - par), SL_PICK(io.delta, 1u - par), s_prev, w);
        sl_phase_expand(io, par, nf, w);
        phase_sync();
        if (threadIdx.x == 0) s_cnt = sl_ld(&c->n_long_cols);
        __syncthreads();
        if (s_cnt) sl_phase_expand_long(io, par, s_cnt, w);
        phase_sync();
        if (threadIdx.x == 0) s_cnt = sl_ld(&c->nc);
        __syncthreads();
        sl_phase_pull_hits(io, par, s_cnt, w);
        phase_sync();
        if (threadIdx.x == 0) s_cnt = sl_ld(&c->n_heavy);
        __syncthreads();
        if (s_cnt) sl_phase_pull_heavy(io, par, s_cnt, w);
        phase_sync();
        if (threadIdx.x == 0) sl_close_round(io, nf);
        phase_sync();
    }
}

// after the last round of a batch: the frontier that round consumed is still marked in its delta buffer
__global__ __launch_bounds__(256) void sl_batch_end_kernel(const sl_round_io *iop)
{
    const sl_round_io io = iop[blockIdx.y];
    const uint32_t rounds = io.c->rounds;
    if (rounds == 0) return;
    const uint32_t par = (rounds - 1u) & 1u;                                   // parity of the last round run
    sl_clear_frontier(SL_PICK(io.frontier, par), SL_PICK(io.delta, par), io.c->nf_prev, sl_worker_here());
}

// generic (any operator given as CSR) dense round, one thread per row — used by estimate_entry
// when the frontier of A^T grows past the dense switch (A^T has no row-slice layout).
__global__ __launch_bounds__(256) void sl_dense_csr_round_kernel(uint64_t n, op_view op, const double *delta_old, const double *dinv,
                                                                 sl_theta theta, int order, double *r, double *x, double *delta_new)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double dself = delta_old[i];
    if (dself != 0.0) x[i] = DADD(x[i], dself);
    const double acc = sl_csr_row_dot(op, (uint32_t)i, delta_old, order);
    const double rn = DSUB(r[i], acc);
    r[i] = rn;
    const double p = DMUL(rn, dinv[i]);
    delta_new[i] = (fabs(p) >= theta.at(i)) ? p : 0.0;
}

namespace {

struct push_state {
    uint64_t n = 0;
    op_view op{};
    double *x = nullptr, *r = nullptr, *dinv = nullptr;
    double *delta[2] = {nullptr, nullptr};
    uint32_t *frontier[2] = {nullptr, nullptr}; // cur / next
    uint32_t *cand = nullptr, *heavy = nullptr; // candidate rows of a sparse round; those pulled over their whole length
    uint32_t *head = nullptr;                   // per row: newest hit record of the running round (SL_EMPTY between rounds)
    uint32_t *cand_flag = nullptr;              // per row: SL_FLAG_TOUCHED (query sessions)
    sl_hit *recs = nullptr;                     // hit records of the running round
    uint64_t rec_cap = 0, op_nnz = 0;
    uint32_t *counters = nullptr;               // [2] compaction total
    sl_push_ctl *ctl = nullptr;                 // device-driven sparse rounds
    uint2 *long_list = nullptr;                 // (column, piece) work items of frontier columns longer than one piece
    uint32_t *touched = nullptr;                // query sessions: rows whose state must be cleaned up afterwards
    bool flooded = false;                       // a dense round ran: every row may be touched
    uint32_t *block_count = nullptr, *block_off = nullptr;
    uint32_t nblocks = 0;
    sl_round_io *io_dev = nullptr;              // argument block of the running batch
};

sl_status compact(push_state &ps, const double *delta, sl_theta theta, uint32_t *list, uint32_t *h_count, hipStream_t s)
{
    if (ps.n == 0) { *h_count = 0; return SL_OK; }
    hipLaunchKernelGGL(sl_compact_count_kernel, dim3(ps.nblocks), dim3(256), 0, s, ps.n, delta, theta, ps.block_count);
    hipLaunchKernelGGL(sl_scan_kernel, dim3(1), dim3(1024), 0, s, ps.nblocks, ps.block_count, ps.block_off, ps.counters + 2);
    hipLaunchKernelGGL(sl_compact_write_kernel, dim3(ps.nblocks), dim3(256), 0, s, ps.n, delta, theta, ps.block_off, list);
    SL_HIP(hipMemcpyAsync(h_count, ps.counters + 2, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
}

struct push_log {
    uint32_t *log = nullptr; uint64_t cap = 0, words = 0;
    std::vector<uint32_t> tmp;
    sl_status append(push_state &ps, uint32_t *d_list, uint32_t nf, bool sorted, hipStream_t s)
    {
        if (!log) return SL_OK;
        if (words < cap) log[words] = nf;
        ++words;
        if (nf == 0) return SL_OK;
        uint32_t *src = d_list;
        DevBuf sorted_buf;
        if (!sorted && nf > 1) {
            SL_TRY(sorted_buf.alloc((size_t)nf * 4));
            SL_TRY(sl_sort_keys_u32(d_list, sorted_buf.as<uint32_t>(), nf, s));
            src = sorted_buf.as<uint32_t>();
        }
        const uint64_t room = words < cap ? cap - words : 0;
        const uint64_t take = nf < room ? nf : room;
        if (take) {
            SL_HIP(hipMemcpyAsync(log + words, src, take * 4, hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
        }
        words += nf;
        (void)ps;
        return SL_OK;
    }
};

struct round_stats { uint64_t rounds = 0, pushes = 0, rows_touched = 0, dense_rounds = 0; uint32_t n_touched = 0; bool converged = false; };

// The round loop.  `m` != null enables the row-slice dense kernel (operator = A itself).
// preseeded: the caller has already written the round-0 frontier (list in frontier[0], values in delta[0], delta[1]
// all zero) and passes its size — a query session seeds one row without touching the other n - 1.
sl_status run_push(push_state &ps, const sl_matrix *m, sl_theta theta, uint64_t max_rounds, int order, double dense_switch,
                   push_log &plog, round_stats &rs, float *device_ms, bool preseeded = false, uint32_t nf0 = 0,
                   const std::function<void(hipStream_t, bool, uint32_t)> *tail = nullptr)
{
    // tail (query sessions): work to enqueue behind every sparse batch, BEFORE the host knows how the batch ended — its kernels
    // check the control block themselves and only act if the batch finished the query (frontier empty, or — second argument —
    // the round limit the caller set was reached).  A local query that converges inside its first batch is one host round trip.
    hipStream_t s = sl_context().stream;
    const uint64_t n = ps.n;
    sl_timer timer;
    SL_TRY(timer.start(s));

    int cur = 0;                 // delta[cur] holds the frontier values, delta[1-cur] is all zero
    // round 0 frontier from r (dense select), ascending list by compaction
    uint32_t nf = nf0;
    if (!preseeded) {
        hipLaunchKernelGGL(sl_select_kernel, dim3((uint32_t)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, s, n, ps.r, ps.dinv, theta, ps.delta[cur]);
        SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s));
        SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s));
    }
    bool list_valid = true, list_sorted = true;
    double *scr = nullptr;
    if (m) { scr = static_cast<double *>(sl_scratch((((size_t)sl_row_grid(m->n_slices) + m->n_long) * 2 + 4096) * sizeof(double))); if (!scr) return sl_fail(SL_ALLOCATION, "scratch"); }
    DevBuf resbuf;
    SL_TRY(resbuf.alloc(64));

    static const int cfg_batch = [] { const char *e = getenv("SL_PUSH_BATCH"); const int v = e ? atoi(e) : 12; return v < 1 ? 1 : v; }();
    const double dense_limit = dense_switch * (double)n;               // nf > dense_limit  <=>  nf > floor(dense_limit)
    const uint32_t dense_threshold = dense_limit >= 4294967295.0 ? 0xffffffffu : (dense_limit > 0.0 ? (uint32_t)dense_limit : 0u);

    // a sparse round costs 40-60x a dense round per matrix entry it touches (records, atomics, random sectors); 64 after the dense
    // rounds of graph-like matrices got the column-panel kernel (swept 16..96 on the 10^7-node PageRank queries)
    static const unsigned long long hit_div = [] { const char *e = getenv("SL_PUSH_HIT_DIV"); const unsigned long long v = e ? strtoull(e, nullptr, 10) : 64; return v ? v : 64ull; }();
    // (dense_switch >= 1: the caller asked for sparse rounds throughout; only the record buffer limits them)
    const unsigned long long hit_limit = dense_switch >= 1.0 ? ps.rec_cap
                                         : std::min<unsigned long long>(ps.rec_cap, std::max<unsigned long long>(ps.op_nnz / hit_div, 4096));
    bool force_dense = false, need_log = true;
    int sparse_batches_done = 0;
    static const bool idx_only = [] { const char *e = getenv("SL_PW_INDEX_ONLY"); return e && *e == '1'; }();
    DevBuf zbuf[2];
    bool z_valid = false;
    const double mean_col = n ? (double)ps.op_nnz / (double)n : 0.0;

    sl_status st = SL_OK;
    while (rs.rounds < max_rounds) {
        if (list_valid && need_log) { SL_TRY(plog.append(ps, ps.frontier[0], nf, list_sorted, s)); need_log = false; }
        if (nf == 0) { rs.converged = true; break; }
        // dense when the frontier is large — by rows, or by the column entries it is expected to cover (|F| x mean column length
        // against the limit the device-side gate of a sparse batch would trip on: no point enqueueing that batch)
        const bool dense = force_dense || (double)nf > dense_switch * (double)n || (dense_switch < 1.0 && (double)nf * mean_col > (double)hit_limit);
        force_dense = false;
        if (dense) {
            uint32_t nf_next = 0;
            if (m) {
                sl_row_args a = sl_matrix_row_args(m);
                a.gather = ps.delta[cur]; a.dinv = ps.dinv; a.out = ps.delta[1 - cur]; a.x = ps.x; a.r = ps.r; a.theta = theta.s; a.theta_rows = theta.rows;
                a.partials = scr; a.partials_slack = 4096; a.result = resbuf.as<double>();
                // column-constant operator (sl_matrix::d_colval; SL_PW_INDEX_ONLY=1): the paced kernel reads the index words of its stream alone
                // and gathers ready-made products from z = colval (.) delta; the epilogue leaves the next round's z beside the next delta.
                // z follows delta's parity; a sparse batch or a fresh frontier invalidates it (one elementwise pass brings it back).
                if (idx_only && m->d_colval && m->d_pw_idx && order == SL_ORDER_CSR_SEQUENTIAL) {
                    if (!zbuf[0].p) { st = zbuf[0].alloc(n * 8); if (st == SL_OK) st = zbuf[1].alloc(n * 8); if (st != SL_OK) break; }
                    if (!z_valid) { st = sl_launch_scale_rows(n, m->d_colval, ps.delta[cur], zbuf[cur].as<double>(), s); if (st != SL_OK) break; }
                    a.zgather = zbuf[cur].as<double>(); a.zcol = m->d_colval; a.zout = zbuf[1 - cur].as<double>();
                    z_valid = true;
                }
                st = sl_launch_rows(a, (sl_order)order, SL_EPI_PUSH, s);
                if (st != SL_OK) break;
                double h[2];
                SL_HIP(hipMemcpyAsync(h, resbuf.p, 16, hipMemcpyDeviceToHost, s));
                SL_HIP(hipStreamSynchronize(s));
                nf_next = (uint32_t)h[1];
            } else {
                hipLaunchKernelGGL(sl_dense_csr_round_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, ps.op, ps.delta[cur],
                                   ps.dinv, theta, order, ps.r, ps.x, ps.delta[1 - cur]);
            }
            rs.rounds += 1; rs.pushes += nf; rs.rows_touched += n; rs.dense_rounds += 1;
            ps.flooded = true;
            cur = 1 - cur;                       // delta[cur] now dense-valid; delta[1-cur] holds stale values
            const bool next_dense_likely = m && !plog.log && ((double)nf_next > dense_switch * (double)n || (dense_switch < 1.0 && (double)nf_next * mean_col > (double)hit_limit));
            if (next_dense_likely) { nf = nf_next; list_valid = false; continue; } // no list needed: the dense kernel overwrites everything
            SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s));
            SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s));
            list_valid = true; list_sorted = true; need_log = true;
        } else {
            z_valid = false;                                             // sparse rounds write delta without its pre-multiplied twin
            if (!list_valid) { SL_TRY(compact(ps, ps.delta[cur], theta, ps.frontier[0], &nf, s)); list_valid = true; list_sorted = true; need_log = true;
                               SL_HIP(hipMemsetAsync(ps.delta[1 - cur], 0, n * 8, s)); }
            // a batch of device-driven sparse rounds (one round when the frontier lists are being logged); the first batch is the
            // long one — a query that outlives it usually needs a few rounds more, and every round enqueued past the stop still
            // costs four empty launches
            uint64_t batch = plog.log ? 1 : (uint64_t)(sparse_batches_done == 0 ? cfg_batch : std::max(cfg_batch / 2, 1));
            ++sparse_batches_done;
            if (batch > max_rounds - rs.rounds) batch = max_rounds - rs.rounds;
            // SL_PUSH_SMALL=1 (round 4; not with a frontier log, which wants every round's list on the host): a one-workgroup kernel that runs
            // small rounds back to back stands before every launch train; a batch then enqueues fewer trains (SL_PUSH_SMALL_TRAINS, 6) and
            // allows more rounds (each small kernel may run many)
            static const int small_on = [] { const char *e = getenv("SL_PUSH_SMALL"); return e && *e == '1' ? 1 : 0; }();
            static const uint32_t small_nf = [] { const char *e = getenv("SL_PUSH_SMALL_NF"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 256u; }();
            static const unsigned long long small_hits = [] { const char *e = getenv("SL_PUSH_SMALL_HITS"); return e ? strtoull(e, nullptr, 10) : 8192ull; }();
            static const uint64_t small_trains = [] { const char *e = getenv("SL_PUSH_SMALL_TRAINS"); const long v = e ? atol(e) : 6; return (uint64_t)(v < 1 ? 1 : v); }();
            const bool small = small_on && !plog.log;
            uint64_t trains = batch;
            if (small) { trains = std::min<uint64_t>(batch, small_trains); batch = std::min<uint64_t>(max_rounds - rs.rounds, 64); }
            sl_round_io io{};
            io.c = ps.ctl;
            io.frontier[0] = ps.frontier[0]; io.frontier[1] = ps.frontier[1];
            io.delta[0] = ps.delta[cur]; io.delta[1] = ps.delta[1 - cur];
            io.op = ps.op; io.x = ps.x; io.r = ps.r; io.dinv = ps.dinv; io.recs = ps.recs; io.head = ps.head; io.cand = ps.cand;
            io.flag = ps.cand_flag; io.long_cols = ps.long_list; io.touched = ps.touched; io.heavy = ps.heavy;
            io.theta = theta; io.order = order; io.dense_threshold = dense_threshold; io.round_limit = (uint32_t)batch; io.hit_limit = hit_limit;
            io.nf0 = nf; io.rec_cap = ps.rec_cap;
            const sl_round_io *iop = ps.io_dev;
            hipLaunchKernelGGL(sl_set_io_kernel, dim3(1), dim3(1), 0, s, io, ps.io_dev);
            hipLaunchKernelGGL(sl_push_ctl_reset_kernel, dim3(1), dim3(1), 0, s, iop);
            hipLaunchKernelGGL(sl_frontier_hits_kernel, dim3((uint32_t)std::min<uint64_t>((nf + 255) / 256, 512)), dim3(256), 0, s, iop);
            hipLaunchKernelGGL(sl_hits_gate_kernel, dim3(1), dim3(1), 0, s, iop);              // hard limit: the record buffer
            if (small) hipLaunchKernelGGL(sl_small_rounds_kernel, dim3(1), dim3(SL_SMALL_THREADS), 0, s, iop, small_nf, small_hits);
            for (uint64_t b = 0; b < trains; ++b) {
                hipLaunchKernelGGL(sl_expand_kernel, dim3(1024), dim3(256), 0, s, iop, ps.ctl, io.round_limit);
                hipLaunchKernelGGL(sl_expand_long_kernel, dim3(512), dim3(256), 0, s, iop, ps.ctl, io.round_limit);
                hipLaunchKernelGGL(sl_pull_hits_kernel, dim3(512), dim3(256), 0, s, iop, ps.ctl, io.round_limit);
                hipLaunchKernelGGL(sl_pull_heavy_kernel, dim3(128), dim3(256), 0, s, iop, ps.ctl, io.round_limit);
                if (small) hipLaunchKernelGGL(sl_small_rounds_kernel, dim3(1), dim3(SL_SMALL_THREADS), 0, s, iop, small_nf, small_hits);
            }
            hipLaunchKernelGGL(sl_batch_end_kernel, dim3(128), dim3(256), 0, s, iop);
            if (tail && !ps.flooded) (*tail)(s, rs.rounds + batch >= max_rounds, (uint32_t)batch);
            sl_push_ctl h;
            SL_HIP(hipMemcpyAsync(&h, ps.ctl, sizeof(h), hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
            rs.rounds += h.rounds; rs.pushes += h.pushes; rs.rows_touched += h.rows_touched;
            if (h.rounds & 1u) { std::swap(ps.frontier[0], ps.frontier[1]); cur = 1 - cur; }
            nf = h.nf; list_valid = true; list_sorted = false;
            rs.n_touched = h.n_touched;
            force_dense = (h.stop == 2u);                              // too many column entries (or rows) for a sparse round
            need_log = h.rounds != 0;
            if (h.rounds == 0 && h.stop == 0) return sl_fail(SL_DEVICE_ERROR, "sparse push batch made no progress");
        }
    }
    const float ms = timer.stop();
    if (device_ms) *device_ms = ms;
    return st;
}

// owned: plain allocations that may be released by another thread (long-lived sessions); otherwise the calling thread's pool
sl_status alloc_state(push_state &ps, uint64_t n, uint64_t nnz, DevBuf bufs[20], bool owned = false)
{
    auto get = [owned](DevBuf &b, size_t bytes) { return owned ? b.alloc_owned(bytes) : b.alloc(bytes); };
    ps.n = n;
    ps.op_nnz = nnz;
    ps.rec_cap = std::min<uint64_t>(nnz, std::max<uint64_t>(2 * n + nnz / 16, 1u << 22));   // 16 B per record; small systems: every entry
    ps.nblocks = (uint32_t)((n + SL_CTILE - 1) / SL_CTILE);
    if (ps.nblocks == 0) ps.nblocks = 1;
    hipStream_t s = sl_context().stream;
    size_t k = 0;
    SL_TRY(get(bufs[k], n * 8)); ps.x = bufs[k++].as<double>();
    SL_TRY(get(bufs[k], n * 8)); ps.r = bufs[k++].as<double>();
    SL_TRY(get(bufs[k], n * 8)); ps.dinv = bufs[k++].as<double>();
    SL_TRY(get(bufs[k], n * 8)); ps.delta[0] = bufs[k++].as<double>();
    SL_TRY(get(bufs[k], n * 8)); ps.delta[1] = bufs[k++].as<double>();
    SL_TRY(get(bufs[k], n * 4)); ps.frontier[0] = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], n * 4)); ps.frontier[1] = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], n * 4)); ps.cand = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], n * 4)); ps.heavy = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], n * 4)); ps.head = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], n * 4)); ps.cand_flag = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], (ps.rec_cap / SL_PIECE + n + 64) * sizeof(uint2))); ps.long_list = bufs[k++].as<uint2>();   // pieces <= hits / 256 + |frontier|
    SL_TRY(get(bufs[k], (ps.rec_cap + 256) * sizeof(sl_hit))); ps.recs = bufs[k++].as<sl_hit>();
    SL_TRY(get(bufs[k], 64)); ps.counters = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], sizeof(sl_push_ctl))); ps.ctl = bufs[k++].as<sl_push_ctl>();
    SL_TRY(get(bufs[k], sizeof(sl_round_io))); ps.io_dev = bufs[k++].as<sl_round_io>();
    SL_TRY(get(bufs[k], (size_t)ps.nblocks * 4)); ps.block_count = bufs[k++].as<uint32_t>();
    SL_TRY(get(bufs[k], (size_t)ps.nblocks * 4)); ps.block_off = bufs[k++].as<uint32_t>();
    SL_HIP(hipMemsetAsync(ps.ctl, 0, sizeof(sl_push_ctl), s));
    SL_HIP(hipMemsetAsync(ps.cand_flag, 0, n * 4, s));
    SL_HIP(hipMemsetAsync(ps.head, 0xff, n * 4, s));                    // SL_EMPTY
    SL_HIP(hipMemsetAsync(ps.counters, 0, 64, s));
    return SL_OK;
}

} // namespace

extern "C" {

void sl_push_options_default(sl_push_options *o)
{
    memset(o, 0, sizeof(*o));
    o->theta = 1e-6;               // ForwardPushConfig.epsilon, forward_push.rs:40-49
    o->max_rounds = 1000000;       // ForwardPushConfig.max_pushes
    o->order = SL_ORDER_CSR_SEQUENTIAL;
    o->mem = SL_MEM_HOST;
    o->dense_switch = 1.0 / 16.0;
}

sl_status sl_push_solve(const sl_matrix *m, const double *b, const sl_push_options *o, double *x, double *r_out,
                        uint32_t *frontier_log, uint64_t frontier_cap, uint64_t *frontier_words, sl_push_result *res)
{
    SL_ABI_BEGIN
    if (!m || !b || !o || !x || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (frontier_words) *frontier_words = 0;
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (!m->d_tptr || !m->d_row_ptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "push needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    const uint64_t n = m->n_rows;
    if (n == 0) { res->converged = 1; return SL_OK; }             // nothing to push
    hipStream_t s = sl_context().stream;
    const hipMemcpyKind in_kind = o->mem == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    const hipMemcpyKind out_kind = o->mem == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;

    sl_range trace_range("push solve");
    push_state ps;
    DevBuf bufs[20], bbuf, ax;
    SL_TRY(alloc_state(ps, n, m->nnz, bufs));
    ps.op = op_view{m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_trow, m->d_tent};
    SL_TRY(bbuf.alloc(n * 8));
    SL_HIP(hipMemcpyAsync(bbuf.p, b, n * 8, in_kind, s));
    SL_HIP(hipMemcpyAsync(ps.x, x, n * 8, in_kind, s));
    unsigned long long hs[4];
    SL_TRY(sl_matrix_diag_pass(m, ps.dinv, hs));
    if (hs[0] & 2ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (hs[0] & 4ull) return sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    // r = b - A x0
    SL_TRY(ax.alloc(n * 8));
    {
        sl_row_args a = sl_matrix_row_args(m);
        a.gather = ps.x; a.out = ax.as<double>();
        SL_TRY(sl_launch_rows(a, (sl_order)o->order, SL_EPI_SPMV, s));
        SL_TRY(sl_launch_sub(n, bbuf.as<double>(), ax.as<double>(), ps.r, s));
    }
    push_log plog;
    plog.log = frontier_log; plog.cap = frontier_cap;
    round_stats rs;
    float ms = 0.f;
    sl_theta th{o->theta, nullptr};
    DevBuf thbuf;
    if (o->theta_rows) {
        if (o->mem == SL_MEM_HOST) {
            SL_TRY(thbuf.alloc(n * 8));
            SL_HIP(hipMemcpyAsync(thbuf.p, o->theta_rows, n * 8, hipMemcpyHostToDevice, s));
            th.rows = thbuf.as<double>();
        } else th.rows = o->theta_rows;
    }
    sl_status st = run_push(ps, m, th, o->max_rounds, o->order, o->dense_switch > 0 ? o->dense_switch : 1.0 / 16.0, plog, rs, &ms);
    if (st != SL_OK) return st;
    res->rounds = rs.rounds; res->pushes = rs.pushes; res->rows_touched = rs.rows_touched; res->dense_rounds = rs.dense_rounds;
    res->converged = rs.converged ? 1 : 0; res->device_time_ms = ms;
    double *scr = static_cast<double *>(sl_scratch(4096 * sizeof(double)));
    if (!scr) return sl_fail(SL_ALLOCATION, "scratch");
    SL_TRY(sl_launch_sumsq(n, ps.r, scr, scr + 4000, s));
    double h = 0.0;
    SL_HIP(hipMemcpyAsync(&h, scr + 4000, 8, hipMemcpyDeviceToHost, s));
    SL_HIP(hipMemcpyAsync(x, ps.x, n * 8, out_kind, s));
    if (r_out) SL_HIP(hipMemcpyAsync(r_out, ps.r, n * 8, out_kind, s));
    SL_HIP(hipStreamSynchronize(s));
    res->residual_norm = std::sqrt(h);
    if (frontier_words) *frontier_words = plog.words;
    return SL_OK;
    SL_ABI_END
}

// ---- query sessions: single-entry queries whose cost follows the rows the push touches ----------------------
// ForwardPushSolver holds its graph and answers query after query (forward_push.rs:52-66, 224-231).  The session
// is that object: state vectors, D^-1, the device copy of b and the transpose live across queries and stay all-zero
// between them; a query seeds one row, runs device-driven sparse rounds, sums over the rows it touched and zeroes
// exactly those again.  Nothing in a query is O(n) unless its frontier floods past the dense switch.
} // extern "C"

struct sl_query_pool;
struct sl_query_session {
    const sl_matrix *m = nullptr;
    bool given_is_transpose = false;
    uint64_t n = 0;
    int device = 0;
    push_state ps;
    DevBuf bufs[20], bbuf, touched, sums, tinv;
    const double *db = nullptr;
    std::vector<double> h_dinv;           // host copy: the seed's threshold test needs dinv[row] only
    double dense_switch = 0.25;
    sl_query_pool *pool = nullptr;        // lanes of sl_query_session_estimate_batch (created on first use)
    struct sl_query_wide *wide = nullptr; // slots of the wide batch (SL_QUERY_WIDE; created on first use)
    ~sl_query_session();
};

__global__ void sl_invert_perm_kernel(uint64_t nnz, const uint32_t *perm, uint32_t *inv)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (uint64_t p = (uint64_t)blockIdx.x * 256 + threadIdx.x; p < nnz; p += stride) inv[perm[p]] = (uint32_t)p;
}

// r[row] = 1 and, if it passes the threshold, the one-entry frontier {row}
__global__ void sl_seed_kernel(uint32_t row, double p, int in_frontier, double *r, double *delta, uint32_t *frontier, uint32_t *flag,
                               uint32_t *touched, sl_push_ctl *c)
{
    r[row] = 1.0;
    if (in_frontier) { delta[row] = p; frontier[0] = row; }
    flag[row] = SL_FLAG_TOUCHED;
    touched[0] = row;
    c->n_touched = 1;
}

// estimate = sum x_i b_i and ||r||_1 over the touched rows.  The list comes in whatever order the rounds appended it, and the
// answer must not depend on that: a binned (pre-rounded) sum.  Pass 1 takes the largest magnitude M of either set of terms; with
// 2^E > M every term is cut into SL_BINS slices, slice b a multiple of 2^(E - 20 (b + 1)) below 2^(E - 20 b) — sums of such
// multiples are EXACT in fp64 for up to 2^32 terms, so any reduction order (shuffles, atomics) gives the same bits — and the
// host adds the bins smallest first.  80 bits below the largest term; no sort, two launches.
#define SL_BINS 4
#define SL_BIN_BITS 20
#define SL_SUM_STRIDE 16               // accumulators 128 bytes apart: atomics on different values do not share an L2 line
__device__ __forceinline__ void sl_touched_terms(uint32_t i, const double *x, const double *b, const double *r, double &e, double &l)
{
    e = DMUL(x[i], b[i]);
    l = fabs(r[i]);
}
// The three kernels of a query's tail run speculatively behind a batch (run_push): they act only if the batch finished the query.
// gate: 0 = unconditional (the host already knows), 1 = frontier empty, 2 = frontier empty or round limit reached
__device__ __forceinline__ bool sl_tail_open(const sl_push_ctl *c, int gate, uint32_t round_limit, uint32_t &nt)
{
    nt = c->n_touched;
    if (gate == 0) return true;
    return c->stop == 1u || (gate == 2 && c->rounds >= round_limit);
}
__device__ __forceinline__ void sl_touched_max_body(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, const double *x,
                                                    const double *b, const double *r, unsigned long long *mx)
{
    uint32_t nt;
    if (!sl_tail_open(c, gate, round_limit, nt)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) mx[2 * SL_SUM_STRIDE - 1] = 1ull;   // marker: the sums below are those of the finished query
    unsigned long long me = 0, ml = 0;                     // bit patterns of non-negative doubles order like the doubles
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < nt; t += gridDim.x * 256) {
        double e, l;
        sl_touched_terms(touched[t], x, b, r, e, l);
        me = max(me, (unsigned long long)__double_as_longlong(fabs(e)));
        ml = max(ml, (unsigned long long)__double_as_longlong(l));
    }
    for (int off = 32; off > 0; off >>= 1) { me = max(me, __shfl_xor(me, off)); ml = max(ml, __shfl_xor(ml, off)); }
    __shared__ unsigned long long red[8];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = me; red[4 + (threadIdx.x >> 6)] = ml; }
    __syncthreads();
    if (threadIdx.x == 0) {                                // one atomic per block and value: they all aim at the same two lines
        me = max(max(red[0], red[1]), max(red[2], red[3])); ml = max(max(red[4], red[5]), max(red[6], red[7]));
        if (me) atomicMax(&mx[0], me);
        if (ml) atomicMax(&mx[SL_SUM_STRIDE], ml);
    }
}
// slices of v against the top exponent E (2^E > every |v|): slice k = v rounded to a multiple of 2^(E - 20 (k + 1)), then removed
// from v (scaling by a power of two, rounding to an integer and subtracting the slice are all exact)
__device__ __forceinline__ void sl_bin_slices(double v, int E, double out[SL_BINS])
{
#pragma unroll
    for (int k = 0; k < SL_BINS; ++k) {
        const int eq = E - SL_BIN_BITS * (k + 1);
        const double p = ldexp(rint(ldexp(v, -eq)), eq);
        out[k] = p;
        v = DSUB(v, p);
    }
}
__device__ __forceinline__ int sl_top_exponent(unsigned long long bits)
{
    if (bits == 0) return 0;
    int e = 0;
    (void)frexp(__longlong_as_double((long long)bits), &e);     // value = m 2^e, 0.5 <= m < 1
    return e < -900 ? -900 : e;                                   // keep every quantum a normal number
}
__global__ __launch_bounds__(256) void sl_touched_max_kernel(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, const double *x,
                                                             const double *b, const double *r, unsigned long long *mx)
{
    sl_touched_max_body(c, gate, round_limit, touched, x, b, r, mx);
}
__device__ __forceinline__ void sl_touched_bins_body(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, const double *x,
                                                     const double *b, const double *r, const unsigned long long *mx, double *bins)
{
    uint32_t nt;
    if (!sl_tail_open(c, gate, round_limit, nt)) return;
    const unsigned long long be = mx[0], bl = mx[SL_SUM_STRIDE];
    if ((be >> 52) == 0x7ffull || (bl >> 52) == 0x7ffull) return;           // a non-finite term: the host reports it from the maxima
    const int Ee = sl_top_exponent(be), El = sl_top_exponent(bl);
    double acc[2 * SL_BINS];
#pragma unroll
    for (int k = 0; k < 2 * SL_BINS; ++k) acc[k] = 0.0;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < nt; t += gridDim.x * 256) {
        double e, l, pe[SL_BINS], pl[SL_BINS];
        sl_touched_terms(touched[t], x, b, r, e, l);
        sl_bin_slices(e, Ee, pe);
        sl_bin_slices(l, El, pl);
#pragma unroll
        for (int k = 0; k < SL_BINS; ++k) { acc[k] = DADD(acc[k], pe[k]); acc[SL_BINS + k] = DADD(acc[SL_BINS + k], pl[k]); }
    }
    __shared__ double red[4][2 * SL_BINS];
#pragma unroll
    for (int k = 0; k < 2 * SL_BINS; ++k) {
        double v = acc[k];
        for (int off = 32; off > 0; off >>= 1) v = DADD(v, __shfl_xor(v, off));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 2 * SL_BINS) {                        // exact additions: neither tree nor arrival order matters
        const double v = DADD(DADD(red[0][threadIdx.x], red[1][threadIdx.x]), DADD(red[2][threadIdx.x], red[3][threadIdx.x]));
        if (v != 0.0) atomicAdd(&bins[threadIdx.x * SL_SUM_STRIDE], v);
    }
}
__global__ __launch_bounds__(256) void sl_touched_bins_kernel(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, const double *x,
                                                              const double *b, const double *r, const unsigned long long *mx, double *bins)
{
    sl_touched_bins_body(c, gate, round_limit, touched, x, b, r, mx, bins);
}
__device__ __forceinline__ void sl_touched_cleanup_body(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, double *x, double *r,
                                                        double *d0, double *d1, uint32_t *flag)
{
    uint32_t nt;
    if (!sl_tail_open(c, gate, round_limit, nt)) return;
    const uint32_t stride = gridDim.x * 256;
    for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < nt; t += stride) {
        const uint32_t i = touched[t];
        x[i] = 0.0; r[i] = 0.0; d0[i] = 0.0; d1[i] = 0.0; flag[i] = 0u;
    }
}
__global__ __launch_bounds__(256) void sl_touched_cleanup_kernel(const sl_push_ctl *c, int gate, uint32_t round_limit, const uint32_t *touched, double *x, double *r,
                                                                 double *d0, double *d1, uint32_t *flag)
{
    sl_touched_cleanup_body(c, gate, round_limit, touched, x, r, d0, d1, flag);
}

// ---- the same tail for the W queries of a WIDE batch (one launch each, slot = blockIdx.y) ---------------------------------------------
#define SL_WIDE_STATS 8                                     // u64 per slot: rounds, pushes, rows_touched, stop, nf, n_touched, marker, -
#define SL_WIDE_OUT ((size_t)SL_WIDE_STATS + (2 + 2 * SL_BINS) * SL_SUM_STRIDE)     // 8-byte words a slot leaves for the host: stats, then the sums block
struct sl_wide_aux { unsigned long long *out; uint32_t row; int32_t in_frontier; double p; };
__global__ void sl_seed_wide_kernel(const sl_round_io *iov, const sl_wide_aux *aux)
{
    const sl_round_io &io = iov[blockIdx.x];
    const sl_wide_aux a = aux[blockIdx.x];
    io.r[a.row] = 1.0;
    if (a.in_frontier) { io.delta[0][a.row] = a.p; io.frontier[0][0] = a.row; }
    io.flag[a.row] = SL_FLAG_TOUCHED;
    io.touched[0] = a.row;
    io.c->n_touched = 1;
}
__global__ __launch_bounds__(256) void sl_touched_max_wide_kernel(const sl_round_io *iov, const sl_wide_aux *aux, int gate, uint32_t round_limit, const double *b)
{
    const sl_round_io &io = iov[blockIdx.y];
    sl_touched_max_body(io.c, gate, round_limit, io.touched, io.x, b, io.r, aux[blockIdx.y].out + SL_WIDE_STATS);
}
__global__ __launch_bounds__(256) void sl_touched_bins_wide_kernel(const sl_round_io *iov, const sl_wide_aux *aux, int gate, uint32_t round_limit, const double *b)
{
    const sl_round_io &io = iov[blockIdx.y];
    unsigned long long *mx = aux[blockIdx.y].out + SL_WIDE_STATS;
    sl_touched_bins_body(io.c, gate, round_limit, io.touched, io.x, b, io.r, mx, reinterpret_cast<double *>(mx) + 2 * SL_SUM_STRIDE);
}
__global__ __launch_bounds__(256) void sl_touched_cleanup_wide_kernel(const sl_round_io *iov, int gate, uint32_t round_limit)
{
    const sl_round_io &io = iov[blockIdx.y];
    sl_touched_cleanup_body(io.c, gate, round_limit, io.touched, io.x, io.r, io.delta[0], io.delta[1], io.flag);
}
__global__ void sl_wide_stats_kernel(const sl_round_io *iov, const sl_wide_aux *aux)
{
    const sl_push_ctl *c = iov[blockIdx.x].c;
    unsigned long long *o = aux[blockIdx.x].out;
    o[0] = c->rounds; o[1] = c->pushes; o[2] = c->rows_touched; o[3] = c->stop; o[4] = c->nf; o[5] = c->n_touched;
}

extern "C" {

static sl_status session_create(const sl_matrix *m, int matrix_is_transpose, const double *b, sl_mem where, bool owned, sl_query_session **out)
{
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!m || !b) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (!m->d_tptr || !m->d_row_ptr || (m->nnz && !m->d_tent)) return sl_fail(SL_UNSUPPORTED_FORMAT, "estimate_entry needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    const uint64_t n = m->n_rows;
    hipStream_t s = sl_context().stream;
    sl_query_session *q = new (std::nothrow) sl_query_session();
    if (!q) return sl_fail(SL_ALLOCATION, "out of host memory");
    q->m = m; q->n = n; q->given_is_transpose = matrix_is_transpose != 0;
    (void)hipGetDevice(&q->device);
    q->dense_switch = q->given_is_transpose ? 1.0 / 16.0 : 0.25;
    auto get = [owned](DevBuf &buf, size_t bytes) { return owned ? buf.alloc_owned(bytes) : buf.alloc(bytes); };
    sl_status st = alloc_state(q->ps, n, m->nnz, q->bufs, owned);
    if (st == SL_OK) st = get(q->touched, (n ? n : 1) * 4);
    if (st == SL_OK) st = get(q->sums, (2 + 2 * SL_BINS) * SL_SUM_STRIDE * sizeof(double));
    unsigned long long hs[4] = {0, 0, 0, 0};
    if (st == SL_OK) {
        q->ps.touched = q->touched.as<uint32_t>();
        if (q->given_is_transpose) {
            // operator B = the matrix itself (= A^T); its dense rounds can use the row-slice kernels
            q->ps.op = op_view{m->d_row_ptr, m->d_col_idx, m->d_values, m->d_tptr, m->d_trow, m->d_tent};
            st = sl_matrix_diag_pass(m, q->ps.dinv, hs);
        } else {
            // operator B = A^T: rows of B = columns of A (sorted transpose), columns of B = rows of A
            // the column entries of B are A's CSR entries k; their place in B's row arrays is the inverse of the transpose permutation
            st = get(q->tinv, (m->nnz ? m->nnz : 1) * 4);
            if (st == SL_OK && m->nnz)
                hipLaunchKernelGGL(sl_invert_perm_kernel, dim3((uint32_t)std::min<uint64_t>((m->nnz + 255) / 256, 65535)), dim3(256), 0, s, m->nnz, m->d_tent,
                                   q->tinv.as<uint32_t>());
            q->ps.op = op_view{m->d_tptr, m->d_trow, m->d_tval, m->d_row_ptr, m->d_col_idx, q->tinv.as<uint32_t>()};
            if (st == SL_OK) st = sl_csr_diag_pass(n, m->d_tptr, m->d_trow, m->d_tval, q->ps.dinv, hs);
        }
    }
    if (st == SL_OK && (hs[0] & 2ull)) st = sl_fail(SL_INVALID_SPARSE_MATRIX, "Missing diagonal element at position %llu", hs[2]);
    if (st == SL_OK && (hs[0] & 4ull)) st = sl_fail(SL_INVALID_SPARSE_MATRIX, "Zero or near-zero diagonal element at position %llu", hs[3]);
    if (st == SL_OK) {
        q->h_dinv.resize(n);
        hipError_t e = hipMemsetAsync(q->ps.x, 0, n * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(q->ps.r, 0, n * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(q->ps.delta[0], 0, n * 8, s);
        if (e == hipSuccess) e = hipMemsetAsync(q->ps.delta[1], 0, n * 8, s);
        if (e == hipSuccess) e = hipMemcpyAsync(q->h_dinv.data(), q->ps.dinv, n * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && where == SL_MEM_HOST) {
            st = get(q->bbuf, n * 8);
            if (st == SL_OK) { e = hipMemcpyAsync(q->bbuf.p, b, n * 8, hipMemcpyHostToDevice, s); q->db = q->bbuf.as<double>(); }
        } else {
            q->db = b;                     // device vector owned by the caller; must outlive the session
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess && st == SL_OK) st = sl_fail(SL_DEVICE_ERROR, "session setup failed: %s", hipGetErrorString(e));
    }
    if (st != SL_OK) { delete q; return st; }
    *out = q;
    return SL_OK;
}

// a session may be handed to and released by another thread: its buffers are plain allocations, not pool blocks
sl_status sl_query_session_create(const sl_matrix *m, int matrix_is_transpose, const double *b, sl_mem where, sl_query_session **out)
{
    SL_ABI_BEGIN
    return session_create(m, matrix_is_transpose, b, where, true, out);
    SL_ABI_END
}

void sl_query_session_destroy(sl_query_session *q) { delete q; }

} // extern "C"

// ---- many independent queries at once: lanes ------------------------------------------------------------------------------------------
// A local query is a train of tiny launches (four per round, a dozen rounds) with nothing else on the GPU: 0.4 ms of which the device
// computes for a few microseconds.  Independent queries (ForwardPushSolver::query_single_entry called for many pairs,
// forward_push.rs:224-231; TS estimateEntry per request, core/solver.ts:550-659) do not depend on each other, so the batch entry runs
// them on LANES: each lane = a state of its own (the session's vectors cloned; D^-1, b and the matrix are shared read-only), a HIP
// stream of its own and a host thread that drives it — the launch trains of different lanes overlap on the device (up to the
// runtime's hardware queues: GPU_MAX_HW_QUEUES, 4 by default), and one lane's host round trip hides under the others' launches.
// Every query runs exactly the code of sl_query_session_estimate on a state that is all-zero between queries: its result does not
// depend on the lane or on what ran beside it (tests: bitwise equal to the one-at-a-time answers).
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
struct sl_query_pool {
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    std::vector<std::thread> threads;
    // the job (valid while active > 0)
    const uint64_t *rows = nullptr;
    uint64_t count = 0, max_rounds = 0, generation = 0;
    double theta = 0.0;
    sl_estimate_result *results = nullptr;
    std::atomic<uint64_t> next{0};
    int active = 0, ready = 0;
    bool quit = false;
    sl_status first_error = SL_OK;
    std::string first_msg;
};

static void query_lane_main(sl_query_session *q, sl_query_pool *p)
{
    sl_query_session *sub = nullptr;
    hipStream_t st = nullptr;
    sl_status init = SL_OK;
    if (hipSetDevice(q->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) init = sl_fail(SL_DEVICE_ERROR, "lane: no stream");
    if (init == SL_OK) {
        sl_context().stream = st;
        init = session_create(q->m, q->given_is_transpose ? 1 : 0, q->db, SL_MEM_DEVICE, true, &sub);     // shares b (device) and the matrix; own state vectors
    }
    uint64_t seen = 0;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (init != SL_OK && p->first_error == SL_OK) { p->first_error = init; p->first_msg = sl_context().last_error; }
        // a lane that joins a pool which has already run jobs starts at the pool's CURRENT generation: the jobs before it were never
        // published to it (with seen = 0 its first wait returned at once, for a job it had no part in, and it then took `active` down a
        // second time — the caller could return while another lane still wrote its results).  The creator publishes the next job only
        // after it has seen `ready` rise, so the generation read here is the last one this lane did NOT take part in.
        seen = p->generation;
        ++p->ready;
        p->cv_done.notify_all();
    }
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(p->mu);
            p->cv_job.wait(lk, [&] { return p->quit || p->generation != seen; });
            if (p->quit) break;
            seen = p->generation;
        }
        if (sub) {
            for (;;) {
                const uint64_t i = p->next.fetch_add(1);
                if (i >= p->count) break;
                const sl_status e = sl_query_session_estimate(sub, p->rows[i], p->theta, p->max_rounds, &p->results[i]);
                if (e != SL_OK) {
                    std::unique_lock<std::mutex> lk(p->mu);
                    if (p->first_error == SL_OK) { p->first_error = e; p->first_msg = sl_context().last_error; }
                }
            }
        }
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->active > 0 && --p->active == 0) p->cv_done.notify_all();      // one decrement per lane and generation (seen == generation here)
    }
    delete sub;
    sl_release_workspace();
    sl_ctx &c = sl_context();
    if (c.scratch) { (void)hipFree(c.scratch); c.scratch = nullptr; c.scratch_bytes = 0; }
    if (c.pinned) { (void)hipHostFree(c.pinned); c.pinned = nullptr; c.pinned_bytes = 0; }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
}

// ---- many independent queries at once: a WIDE batch (round 4, SL_QUERY_WIDE=W; opt-in until measured) --------------------------------------
// The lanes above overlap the launch trains of different queries; every query still pays its own ~56 launches.  A wide batch runs W
// queries through ONE launch train: W slots — a state of its own each, like a lane's — and every kernel of the batch launched once with
// grid.y = W, slot = blockIdx.y (argument blocks iop[blockIdx.y]); the rounds of the W queries proceed in lockstep, a query that is done
// gates its blocks off through its own control block.  Seeds, argument blocks and results travel as ONE upload and ONE read-back per
// group of W.  Every query runs the same phases on the same kind of state as sl_query_session_estimate: the same bits.  A query that the
// first batch does not finish (a frontier that wants dense rounds, more rounds than the batch holds) has its slot cleaned and is answered
// again by the ordinary path on that slot — a deterministic function of the query, so the answer is the one it would have had.
struct sl_query_wide {
    std::vector<sl_query_session *> slot;
    DevBuf io_dev, aux_dev, out_dev;
    std::vector<sl_round_io> h_io;
    std::vector<sl_wide_aux> h_aux;
    std::vector<unsigned long long> h_out;
    ~sl_query_wide() { for (sl_query_session *q : slot) delete q; }
};

sl_query_session::~sl_query_session()
{
    delete wide;
    if (!pool) return;
    { std::unique_lock<std::mutex> lk(pool->mu); pool->quit = true; pool->cv_job.notify_all(); }
    for (std::thread &t : pool->threads) if (t.joinable()) t.join();
    delete pool;
}

static sl_status wide_batch(sl_query_session *q, uint64_t count, const uint64_t *rows, double theta, uint64_t max_rounds, uint32_t W, sl_estimate_result *results)
{
    hipStream_t s = sl_context().stream;
    if (!q->wide) q->wide = new sl_query_wide();
    sl_query_wide *wd = q->wide;
    while (wd->slot.size() < W) {
        sl_query_session *sub = nullptr;
        SL_TRY(session_create(q->m, q->given_is_transpose ? 1 : 0, q->db, SL_MEM_DEVICE, true, &sub));     // shares b (device) and the matrix; own state vectors
        wd->slot.push_back(sub);
    }
    if (wd->h_io.size() < W) {
        // device buffers first: were a later allocation to fail, the next call must come back here, not find the host vectors already
        // sized and launch with a null io_dev / aux_dev / out_dev
        SL_TRY(wd->io_dev.alloc_owned(W * sizeof(sl_round_io))); SL_TRY(wd->aux_dev.alloc_owned(W * sizeof(sl_wide_aux)));
        SL_TRY(wd->out_dev.alloc_owned((size_t)W * SL_WIDE_OUT * 8));
        wd->h_aux.resize(W); wd->h_out.resize((size_t)W * SL_WIDE_OUT); wd->h_io.resize(W);      // h_io last: its size is the "all of it is there" mark
    }
    static const int cfg_batch = [] { const char *e = getenv("SL_PUSH_BATCH"); const int v = e ? atoi(e) : 12; return v < 1 ? 1 : v; }();
    static const unsigned long long hit_div = [] { const char *e = getenv("SL_PUSH_HIT_DIV"); const unsigned long long v = e ? strtoull(e, nullptr, 10) : 64; return v ? v : 64ull; }();
    static const int small_on = [] { const char *e = getenv("SL_PUSH_SMALL"); return e && *e == '1' ? 1 : 0; }();
    static const uint32_t small_nf = [] { const char *e = getenv("SL_PUSH_SMALL_NF"); return e ? (uint32_t)strtoul(e, nullptr, 10) : 256u; }();
    static const unsigned long long small_hits = [] { const char *e = getenv("SL_PUSH_SMALL_HITS"); return e ? strtoull(e, nullptr, 10) : 8192ull; }();
    const uint64_t n = q->n;
    const uint64_t batch = std::min<uint64_t>((uint64_t)cfg_batch, max_rounds);
    const int gate = batch >= max_rounds ? 2 : 1;                           // the round limit of the caller falls inside this batch
    const double dense_limit = q->dense_switch * (double)n;
    const uint32_t dense_threshold = dense_limit >= 4294967295.0 ? 0xffffffffu : (dense_limit > 0.0 ? (uint32_t)dense_limit : 0u);
    const sl_round_io *iov = wd->io_dev.as<sl_round_io>();
    const sl_wide_aux *auxv = wd->aux_dev.as<sl_wide_aux>();
    for (uint64_t g0 = 0; g0 < count; g0 += W) {
        const uint32_t G = (uint32_t)std::min<uint64_t>(W, count - g0);
        if (batch == 0) {                                                   // max_rounds = 0: nothing runs; the ordinary path says what that means
            for (uint32_t j = 0; j < G; ++j) SL_TRY(sl_query_session_estimate(wd->slot[j], rows[g0 + j], theta, max_rounds, &results[g0 + j]));
            continue;
        }
        for (uint32_t j = 0; j < G; ++j) {
            push_state &ps = wd->slot[j]->ps;
            ps.flooded = false;
            sl_round_io io{};
            io.c = ps.ctl;
            io.frontier[0] = ps.frontier[0]; io.frontier[1] = ps.frontier[1];
            io.delta[0] = ps.delta[0]; io.delta[1] = ps.delta[1];
            io.op = ps.op; io.x = ps.x; io.r = ps.r; io.dinv = ps.dinv; io.recs = ps.recs; io.head = ps.head; io.cand = ps.cand;
            io.flag = ps.cand_flag; io.long_cols = ps.long_list; io.touched = ps.touched; io.heavy = ps.heavy;
            io.theta = sl_theta{theta, nullptr}; io.order = SL_ORDER_CSR_SEQUENTIAL; io.dense_threshold = dense_threshold; io.round_limit = (uint32_t)batch;
            io.hit_limit = q->dense_switch >= 1.0 ? ps.rec_cap : std::min<unsigned long long>(ps.rec_cap, std::max<unsigned long long>(ps.op_nnz / hit_div, 4096));
            io.rec_cap = ps.rec_cap;
            const double p = q->h_dinv[rows[g0 + j]];
            const int in_frontier = std::fabs(p) >= theta ? 1 : 0;
            io.nf0 = (uint32_t)in_frontier;
            wd->h_io[j] = io;
            wd->h_aux[j] = sl_wide_aux{wd->out_dev.as<unsigned long long>() + (size_t)j * SL_WIDE_OUT, (uint32_t)rows[g0 + j], in_frontier, p};
        }
        SL_TRY(sl_upload(wd->io_dev.p, wd->h_io.data(), G * sizeof(sl_round_io), s));
        SL_TRY(sl_upload(wd->aux_dev.p, wd->h_aux.data(), G * sizeof(sl_wide_aux), s));
        SL_HIP(hipMemsetAsync(wd->out_dev.p, 0, (size_t)G * SL_WIDE_OUT * 8, s));
        sl_timer timer;
        SL_TRY(timer.start(s));
        hipLaunchKernelGGL(sl_seed_wide_kernel, dim3(G), dim3(1), 0, s, iov, auxv);
        hipLaunchKernelGGL(sl_push_ctl_reset_kernel, dim3(1, G), dim3(1), 0, s, iov);
        hipLaunchKernelGGL(sl_frontier_hits_kernel, dim3(1, G), dim3(256), 0, s, iov);
        hipLaunchKernelGGL(sl_hits_gate_kernel, dim3(1, G), dim3(1), 0, s, iov);
        if (small_on) hipLaunchKernelGGL(sl_small_rounds_kernel, dim3(1, G), dim3(SL_SMALL_THREADS), 0, s, iov, small_nf, small_hits);
        for (uint64_t b = 0; b < batch; ++b) {                              // a quarter of a single query's grids per slot: W of them share the machine
            hipLaunchKernelGGL(sl_expand_kernel, dim3(256, G), dim3(256), 0, s, iov, (const sl_push_ctl *)nullptr, (uint32_t)batch);
            hipLaunchKernelGGL(sl_expand_long_kernel, dim3(128, G), dim3(256), 0, s, iov, (const sl_push_ctl *)nullptr, (uint32_t)batch);
            hipLaunchKernelGGL(sl_pull_hits_kernel, dim3(128, G), dim3(256), 0, s, iov, (const sl_push_ctl *)nullptr, (uint32_t)batch);
            hipLaunchKernelGGL(sl_pull_heavy_kernel, dim3(32, G), dim3(256), 0, s, iov, (const sl_push_ctl *)nullptr, (uint32_t)batch);
            if (small_on) hipLaunchKernelGGL(sl_small_rounds_kernel, dim3(1, G), dim3(SL_SMALL_THREADS), 0, s, iov, small_nf, small_hits);
        }
        hipLaunchKernelGGL(sl_batch_end_kernel, dim3(32, G), dim3(256), 0, s, iov);
        hipLaunchKernelGGL(sl_touched_max_wide_kernel, dim3(32, G), dim3(256), 0, s, iov, auxv, gate, (uint32_t)batch, q->db);
        hipLaunchKernelGGL(sl_touched_bins_wide_kernel, dim3(32, G), dim3(256), 0, s, iov, auxv, gate, (uint32_t)batch, q->db);
        hipLaunchKernelGGL(sl_wide_stats_kernel, dim3(G), dim3(1), 0, s, iov, auxv);              // before the cleanup: n_touched and the counters as the query left them
        hipLaunchKernelGGL(sl_touched_cleanup_wide_kernel, dim3(64, G), dim3(256), 0, s, iov, gate, (uint32_t)batch);
        SL_HIP(hipGetLastError());
        SL_TRY(sl_read_back(wd->h_out.data(), wd->out_dev.p, (size_t)G * SL_WIDE_OUT * 8, s));
        const float ms = timer.stop();
        for (uint32_t j = 0; j < G; ++j) {
            const unsigned long long *o = wd->h_out.data() + (size_t)j * SL_WIDE_OUT;
            const double *hs = reinterpret_cast<const double *>(o + SL_WIDE_STATS);
            unsigned long long marker = 0;
            memcpy(&marker, &hs[2 * SL_SUM_STRIDE - 1], 8);
            sl_estimate_result *res = &results[g0 + j];
            memset(res, 0, sizeof(*res));
            if (marker == 1ull) {                                           // the batch finished the query on the device's terms: sums taken, slot clean
                double h[2] = {0.0, 0.0};
                for (int v = 0; v < 2; ++v) {
                    const double m = hs[v * SL_SUM_STRIDE];                  // bit pattern of a non-negative double
                    double sum = 0.0;
                    if (!std::isfinite(m)) sum = m;
                    else for (int k = SL_BINS - 1; k >= 0; --k) sum += hs[(2 + v * SL_BINS + k) * SL_SUM_STRIDE];
                    h[v] = sum;
                }
                res->estimate = h[0]; res->residual_l1 = h[1];
                res->rounds = o[0]; res->pushes = o[1]; res->rows_touched = o[2];
                res->device_time_ms = ms / (float)G;          // a group's launches serve its G queries at once: each carries its share (their sum = the device time spent)
                res->converged = o[4] == 0 ? 1 : 0;
            } else {
                // not finished inside the batch: clean the slot (everything the query touched is on its list), answer it the ordinary way
                push_state &ps = wd->slot[j]->ps;
                hipLaunchKernelGGL(sl_touched_cleanup_kernel, dim3(256), dim3(256), 0, s, ps.ctl, 0, 0u, ps.touched, ps.x, ps.r, ps.delta[0], ps.delta[1], ps.cand_flag);
                SL_HIP(hipGetLastError());
                SL_TRY(sl_query_session_estimate(wd->slot[j], rows[g0 + j], theta, max_rounds, res));
            }
        }
    }
    return SL_OK;
}

extern "C" {

sl_status sl_query_session_estimate_batch(sl_query_session *q, uint64_t count, const uint64_t *rows, double theta, uint64_t max_rounds, uint32_t lanes,
                                          sl_estimate_result *results)
{
    SL_ABI_BEGIN
    if (!q || (count && (!rows || !results))) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (count == 0) return SL_OK;
    for (uint64_t i = 0; i < count; ++i)
        if (rows[i] >= q->n) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)rows[i], (unsigned long long)q->n);
    static const uint32_t wide_env = [] { const char *e = getenv("SL_QUERY_WIDE"); const long v = e ? atol(e) : 0; return (uint32_t)(v < 0 ? 0 : (v > 256 ? 256 : v)); }();
    if (wide_env >= 2 && count >= 2) return wide_batch(q, count, rows, theta, max_rounds, (uint32_t)std::min<uint64_t>(wide_env, count), results);
    if (lanes == 0) lanes = 8;
    lanes = (uint32_t)std::min<uint64_t>(std::min<uint32_t>(lanes, 64u), count);
    if (!q->pool) q->pool = new sl_query_pool();
    sl_query_pool *p = q->pool;
    // the calling thread is lane 0 (the session's own state, the caller's stream); lanes 1.. are threads that stay with the session
    while (p->threads.size() + 1 < lanes) {
        const size_t want = p->threads.size() + 1;
        p->threads.emplace_back(query_lane_main, q, p);
        std::unique_lock<std::mutex> lk(p->mu);
        p->cv_done.wait(lk, [&] { return (size_t)p->ready >= want; });
    }
    {
        std::unique_lock<std::mutex> lk(p->mu);
        if (p->first_error != SL_OK) { const sl_status e = p->first_error; p->first_error = SL_OK; return sl_fail(e, "a lane of the batch could not be set up: %s", p->first_msg.c_str()); }
        p->rows = rows; p->count = count; p->theta = theta; p->max_rounds = max_rounds; p->results = results;
        p->next.store(0);
        p->active = (int)p->threads.size();
        ++p->generation;
        p->cv_job.notify_all();
    }
    sl_status mine = SL_OK;
    std::string mine_msg;
    for (;;) {
        const uint64_t i = p->next.fetch_add(1);
        if (i >= count) break;
        const sl_status e = sl_query_session_estimate(q, rows[i], theta, max_rounds, &results[i]);
        if (e != SL_OK && mine == SL_OK) { mine = e; mine_msg = sl_context().last_error; }
    }
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->active == 0; });
    if (mine == SL_OK && p->first_error != SL_OK) { mine = p->first_error; mine_msg = p->first_msg; }
    p->first_error = SL_OK;
    if (mine != SL_OK) return sl_fail(mine, "%s", mine_msg.c_str());
    return SL_OK;
    SL_ABI_END
}

sl_status sl_query_session_estimate(sl_query_session *q, uint64_t row, double theta, uint64_t max_rounds, sl_estimate_result *res)
{
    SL_ABI_BEGIN
    if (!q || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    const uint64_t n = q->n;
    if (row >= n) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)row, (unsigned long long)n);
    hipStream_t s = sl_context().stream;
    push_state &ps = q->ps;
    // y0 = 0, r = e_row; round-0 frontier = {row} if |1 * dinv_row| >= theta
    const double p = q->h_dinv[row];
    const int in_frontier = std::fabs(p) >= theta ? 1 : 0;
    ps.flooded = false;
    hipLaunchKernelGGL(sl_seed_kernel, dim3(1), dim3(1), 0, s, (uint32_t)row, p, in_frontier, ps.r, ps.delta[0], ps.frontier[0], ps.cand_flag,
                       ps.touched, ps.ctl);
    push_log plog;
    round_stats rs;
    rs.n_touched = 1;
    float ms = 0.f;
    // estimate = y . b ; residual_l1 = ||r_y||_1 over the touched rows (binned sums: independent of the list order), then exactly
    // those rows are zeroed again.  Enqueued behind every sparse batch, acting only if that batch finished the query.
    unsigned long long *mx = q->sums.as<unsigned long long>();
    double *bins = q->sums.as<double>() + 2 * SL_SUM_STRIDE;
    double hs[(2 + 2 * SL_BINS) * SL_SUM_STRIDE] = {0.0};
    hipError_t tail_err = hipSuccess;
    const std::function<void(hipStream_t, bool, uint32_t)> tail = [&](hipStream_t ts, bool limit_is_final, uint32_t round_limit) {
        const int gate = round_limit == 0 ? 0 : (limit_is_final ? 2 : 1);
        hipError_t e = hipMemsetAsync(q->sums.p, 0, sizeof(hs), ts);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(sl_touched_max_kernel, dim3(128), dim3(256), 0, ts, ps.ctl, gate, round_limit, ps.touched, ps.x, q->db, ps.r, mx);
            hipLaunchKernelGGL(sl_touched_bins_kernel, dim3(128), dim3(256), 0, ts, ps.ctl, gate, round_limit, ps.touched, ps.x, q->db, ps.r, mx, bins);
            hipLaunchKernelGGL(sl_touched_cleanup_kernel, dim3(256), dim3(256), 0, ts, ps.ctl, gate, round_limit, ps.touched, ps.x, ps.r, ps.delta[0],
                               ps.delta[1], ps.cand_flag);
            e = hipMemcpyAsync(hs, q->sums.p, sizeof(hs), hipMemcpyDeviceToHost, ts);
        }
        if (e != hipSuccess) tail_err = e;
    };
    sl_status st = run_push(ps, q->given_is_transpose ? q->m : nullptr, sl_theta{theta, nullptr}, max_rounds, SL_ORDER_CSR_SEQUENTIAL, q->dense_switch, plog, rs, &ms,
                            true, (uint32_t)in_frontier, &tail);
    double h[2] = {0.0, 0.0};
    if (st == SL_OK && !ps.flooded) {
        unsigned long long marker = 0;
        memcpy(&marker, &hs[2 * SL_SUM_STRIDE - 1], 8);
        if (marker != 1ull) {                              // no batch ran, or the last one did not end the query on the device's terms
            tail(s, false, 0);
            if (tail_err == hipSuccess) tail_err = hipStreamSynchronize(s);
        }
        if (tail_err != hipSuccess) st = sl_fail(SL_DEVICE_ERROR, "query readback failed: %s", hipGetErrorString(tail_err));
        if (st == SL_OK) {
            for (int v = 0; v < 2; ++v) {
                const double m = hs[v * SL_SUM_STRIDE];                    // bit pattern of a non-negative double
                double sum = 0.0;
                if (!std::isfinite(m)) sum = m;                            // inf / nan term: report it
                else for (int k = SL_BINS - 1; k >= 0; --k) sum += hs[(2 + v * SL_BINS + k) * SL_SUM_STRIDE];   // smallest quantum first
                h[v] = sum;
            }
        }
    } else if (st == SL_OK) {
        // the frontier flooded the graph: whole-vector sums and a whole-vector reset
        double *scr = static_cast<double *>(sl_scratch(8192 * sizeof(double)));
        if (!scr) st = sl_fail(SL_ALLOCATION, "scratch");
        if (st == SL_OK) st = sl_launch_dot(n, ps.x, q->db, scr, scr + 4000, s);
        if (st == SL_OK) st = sl_launch_abs_sum(n, ps.r, scr + 4100, scr + 4001, s);
        if (st == SL_OK) {
            hipError_t e = hipMemcpyAsync(h, scr + 4000, 16, hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) st = sl_fail(SL_DEVICE_ERROR, "query readback failed: %s", hipGetErrorString(e));
        }
    }
    if (st != SL_OK || ps.flooded) {                       // leave the session all-zero whatever happened
        hipMemsetAsync(ps.x, 0, n * 8, s); hipMemsetAsync(ps.r, 0, n * 8, s);
        hipMemsetAsync(ps.delta[0], 0, n * 8, s); hipMemsetAsync(ps.delta[1], 0, n * 8, s);
        hipMemsetAsync(ps.cand_flag, 0, n * 4, s);
    }
    if (st != SL_OK) return st;
    res->estimate = h[0]; res->residual_l1 = h[1];
    res->rounds = rs.rounds; res->pushes = rs.pushes; res->rows_touched = rs.rows_touched;
    res->device_time_ms = ms; res->converged = rs.converged ? 1 : 0;
    return SL_OK;
    SL_ABI_END
}

// one-shot queries: a session for the duration of the call
static sl_status estimate_entry_impl(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double theta,
                                     uint64_t max_rounds, bool given_is_transpose, sl_estimate_result *res)
{
    if (!m || !b || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(res, 0, sizeof(*res));
    if (m->n_rows != m->n_cols || m->row_offset != 0) return sl_fail(SL_INVALID_INPUT, "Matrix must be square");
    if (row >= m->n_rows) return sl_fail(SL_INVALID_INPUT, "Row index %llu out of bounds. Matrix has %llu rows", (unsigned long long)row, (unsigned long long)m->n_rows);
    sl_query_session *q = nullptr;
    SL_TRY(session_create(m, given_is_transpose ? 1 : 0, b, where, false, &q));   // lives for this call: pool blocks
    const sl_status st = sl_query_session_estimate(q, row, theta, max_rounds, res);
    sl_query_session_destroy(q);
    return st;
}

sl_status sl_estimate_entry(const sl_matrix *m, const double *b, sl_mem where, uint64_t row, double theta,
                            uint64_t max_rounds, sl_estimate_result *res)
{
    SL_ABI_BEGIN
    return estimate_entry_impl(m, b, where, row, theta, max_rounds, false, res);
    SL_ABI_END
}

sl_status sl_estimate_entry_transposed(const sl_matrix *mt, const double *b, sl_mem where, uint64_t row, double theta,
                                       uint64_t max_rounds, sl_estimate_result *res)
{
    SL_ABI_BEGIN
    return estimate_entry_impl(mt, b, where, row, theta, max_rounds, true, res);
    SL_ABI_END
}

sl_status sl_matrix_transpose(const sl_matrix *m, uint32_t flags, sl_matrix **out)
{
    SL_ABI_BEGIN
    if (!m || !out) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (!m->d_tptr) return sl_fail(SL_UNSUPPORTED_FORMAT, "sl_matrix_transpose needs a matrix created with SL_MATRIX_WITH_TRANSPOSE");
    if (m->row_offset != 0) return sl_fail(SL_UNSUPPORTED_FORMAT, "cannot transpose a row slice");
    return sl_matrix_create_csr(m->n_cols, m->n_rows, m->nnz, m->d_tptr, m->d_trow, m->d_tval, SL_MEM_DEVICE, 0, flags, out);
    SL_ABI_END
}

} // extern "C"
