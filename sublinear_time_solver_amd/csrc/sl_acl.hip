// sl_acl.hip — the graph side of the orphan Rust push spec behind the C ABI:
//   PushGraph {adjacency, reverse_adjacency, degrees, reverse_degrees}    src/graph/adjacency.rs:199-277  -> sl_push_graph_*
//   the PageRank / PPR systems over it, I - (1 - alpha) P^T and I - (1 - alpha) P (P_uv = w_uv / deg_u, a dangling node keeps its
//   mass: forward_push.rs:210-215; TS computePageRank builds the same matrix densely, src/core/solver.ts:664-722)  -> sl_push_graph_system
//   ForwardPushSolver::{solve_single_source, solve_multi_source, solve_with_target} (src/solver/forward_push.rs:67-290) and
//   BackwardPushSolver::solve_single_target (src/solver/backward_push.rs:67-220) in THEIR visiting order: WorkQueue pops the largest
//   (priority, node id) (src/graph/mod.rs:132-213; the reference leaves the order of equal priorities open: the larger node id
//   first, the rule the tests' CPU restatement fixes too)                                                                                          -> sl_*_push_acl*
//
// The ACL push is sequential across pushes by definition (every pop depends on the residuals the push before it left), so the
// device parallelises INSIDE a push only: the maximum over the queue (one block, fixed-tie reduction — any priority queue over a
// strict total order pops the same sequence as the reference's binary heap) and the edge loop of push_node.  It exists for
// order-exact parity — push_count, nodes_visited, every estimate / residual bit equal to the sequential CPU restatement of the spec; the
// throughput path is the synchronous thresholded push on the system matrix (sl_push_solve with theta_rows, sl_frontier.hip).
#include "sl_internal.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

#define DMUL(a, b) __dmul_rn((a), (b))
#define DADD(a, b) __dadd_rn((a), (b))
#define DDIV(a, b) __ddiv_rn((a), (b))

sl_status sl_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, int end_bit, hipStream_t s);
sl_status sl_sort_pairs_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, int end_bit, hipStream_t s);

struct sl_push_graph {
    uint64_t n = 0, nnz = 0;
    int device = 0;
    uint32_t *d_rp = nullptr, *d_ci = nullptr;          // adjacency, entries in the caller's order (forward_neighbors walks it as stored)
    double *d_w = nullptr;
    uint32_t *d_trp = nullptr, *d_tci = nullptr;        // reverse adjacency = CompressedSparseRow::transpose (graph/mod.rs:92-130): sources of a node ascending
    double *d_tw = nullptr;
    double *d_deg = nullptr, *d_rdeg = nullptr;         // row sums of the two, added left to right (graph/mod.rs:81-89)
    uint8_t *d_dup = nullptr, *d_tdup = nullptr;        // a row names the same neighbour twice: its edge loop runs on one thread
    ~sl_push_graph()
    {
        hipFree(d_rp); hipFree(d_ci); hipFree(d_w); hipFree(d_trp); hipFree(d_tci); hipFree(d_tw); hipFree(d_deg); hipFree(d_rdeg); hipFree(d_dup); hipFree(d_tdup);
    }
};

namespace {

__global__ void pg_validate_kernel(uint64_t n, uint64_t nnz, const uint32_t *rp, const uint32_t *ci, const double *w, uint32_t *err)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    if (t == 0 && (rp[0] != 0 || rp[n] != nnz)) atomicOr(err, 1u);
    for (uint64_t i = t; i < n; i += stride) if (rp[i] > rp[i + 1]) atomicOr(err, 1u);
    for (uint64_t k = t; k < nnz; k += stride) { if (ci[k] >= n) atomicOr(err, 2u); if (!isfinite(w[k])) atomicOr(err, 4u); }
}

__global__ void pg_row_sums_kernel(uint64_t n, const uint32_t *rp, const double *w, double *out)     // graph/mod.rs:81-89: s = s + w, left to right
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (uint32_t k = rp[i]; k < rp[i + 1]; ++k) s = DADD(s, w[k]);
    out[i] = s;
}

__global__ void pg_rows_of_entries_kernel(uint64_t n, const uint32_t *rp, uint32_t *row_of)
{
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i = wave; i < n; i += nw)
        for (uint32_t k = rp[i] + lane; k < rp[i + 1]; k += 64u) row_of[k] = (uint32_t)i;
}

__global__ void pg_count_kernel(uint64_t nnz, const uint32_t *ci, uint32_t *count)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x) atomicAdd(&count[ci[k] + 1], 1u);
}

__global__ void pg_iota_kernel(uint64_t n, uint32_t *out)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) out[k] = (uint32_t)k;
}

// transposed entry e = entry perm[e] of the adjacency: source row and weight
__global__ void pg_transpose_fill_kernel(uint64_t nnz, const uint32_t *perm, const uint32_t *row_of, const double *w, uint32_t *tci, double *tw)
{
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t k = perm[e];
        tci[e] = row_of[k];
        tw[e] = w[k];
    }
}

__global__ void pg_keys64_kernel(uint64_t nnz, const uint32_t *row_of, const uint32_t *ci, unsigned long long *key)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (uint64_t)gridDim.x * blockDim.x)
        key[k] = ((unsigned long long)row_of[k] << 32) | ci[k];
}

__global__ void pg_dup_kernel(uint64_t nnz, const unsigned long long *sorted, uint8_t *dup)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; k < nnz; k += (uint64_t)gridDim.x * blockDim.x)
        if (sorted[k] == sorted[k - 1]) dup[sorted[k] >> 32] = 1;
}

// rows of the reverse adjacency ascend in their sources, so duplicates are neighbours there
__global__ void pg_tdup_kernel(uint64_t n, const uint32_t *trp, const uint32_t *tci, uint8_t *tdup)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t d = 0;
    for (uint32_t k = trp[i] + 1; k < trp[i + 1]; ++k) d |= (tci[k] == tci[k - 1]) ? 1 : 0;
    tdup[i] = d;
}

uint32_t grid_of(uint64_t n, uint32_t block = 256, uint32_t cap = 8192) { const uint64_t g = (n + block - 1) / block; return (uint32_t)std::min<uint64_t>(g ? g : 1, cap); }

// ---- system assembly -------------------------------------------------------------------------------------------------------------
// Row v of the system over the SORTED source rows `srp / sci / sw` (forward: the reverse adjacency, whose row v lists the u -> v edges
// by ascending u; backward: the adjacency sorted by column): one entry -(1 - alpha) * (w / deg_u) per edge with the other end != v,
// and ONE diagonal entry 1 - (1 - alpha) * (sum over the self loops of w / deg + [v dangling]) — the diagonal is what D^-1 is read
// from, it must not be split.  deg_of = the degrees P divides by: of the SOURCE node u (forward: the column index of the entry,
// backward: the row itself).
template <bool COUNT>
__global__ __launch_bounds__(256) void pg_system_kernel(uint64_t n, int backward, int dangling_identity, double alpha, const uint32_t *srp, const uint32_t *sci, const double *sw, const double *deg,
                                                        const uint32_t *out_rp, uint32_t *counts, uint32_t *oci, double *ova)
{
    const uint64_t v = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t s = srp[v], e = srp[v + 1];
    const double oma = 1.0 - alpha;
    if (COUNT) {
        uint32_t c = 1;
        for (uint32_t k = s; k < e; ++k) c += (sci[k] != (uint32_t)v) ? 1u : 0u;
        counts[v + 1] = c;
        return;
    }
    uint32_t o = out_rp[v];
    double self = (deg[v] > 0.0 || dangling_identity) ? 0.0 : 1.0;   // dangling: P_vv = 1 (forward_push.rs:210-215), unless the caller wants TS's rule
    bool diag_done = false;
    auto emit_diag = [&]() { oci[o] = (uint32_t)v; ova[o] = 0.0; ++o; diag_done = true; };     // value filled in below
    uint32_t diag_at = 0;
    for (uint32_t k = s; k < e; ++k) {
        const uint32_t u = sci[k];
        const double du = backward ? deg[v] : deg[u];
        const double p = du > 0.0 ? DDIV(sw[k], du) : 0.0;      // a dangling source spreads nothing along its (zero-sum) edges
        if (u == (uint32_t)v) { self = DADD(self, p); continue; }
        if (!diag_done && u > (uint32_t)v) { diag_at = o; emit_diag(); }
        oci[o] = u;
        ova[o] = -DMUL(oma, p);
        ++o;
    }
    if (!diag_done) { diag_at = o; emit_diag(); }
    ova[diag_at] = 1.0 - DMUL(oma, self);
}

// ---- the ACL push in the spec's order ---------------------------------------------------------------------------------------------
struct acl_ctl {
    unsigned long long pushes, nvis;
    uint32_t qlen;
    uint32_t state;              // 0 running, 1 queue empty, 2 max_pushes reached, 3 target precision reached
    double threshold;            // WorkQueue::threshold (adaptive)
};

#define SL_ACL_THREADS 1024
__global__ void acl_init_kernel(uint64_t n, uint64_t nsrc, const uint64_t *src, const double *qdeg, double queue_threshold, double *res, uint8_t *inq,
                                uint32_t *qnode, double *qprio, acl_ctl *ctl)
{
    if (blockIdx.x || threadIdx.x) return;
    ctl->pushes = 0; ctl->nvis = 0; ctl->qlen = 0; ctl->state = 0; ctl->threshold = queue_threshold;
    // forward_push.rs:72-83 (one source: unit mass) / :131-137 (several: 1 / len each, added up where a source repeats)
    if (nsrc == 1) { if (src[0] < n) res[src[0]] = 1.0; }
    else { const double mass = 1.0 / (double)nsrc; for (uint64_t k = 0; k < nsrc; ++k) if (src[k] < n) res[src[k]] = DADD(res[src[k]], mass); }
    for (uint64_t k = 0; k < nsrc; ++k) {
        const uint64_t u = src[k];
        if (u >= n) continue;
        const double d = fmax(qdeg[u], 1.0);
        const double pr = d > 0.0 ? DDIV(res[u], d) : res[u];                      // graph/mod.rs:172
        if (pr >= ctl->threshold && !inq[u]) { const uint32_t q = ctl->qlen++; qnode[q] = (uint32_t)u; qprio[q] = pr; inq[u] = 1; }
    }
}

// backward = 0: rows of the adjacency, transfer (rem * w) / deg_u, queue degrees = out-degrees (forward_push.rs:179-216)
// backward = 1: rows of the reverse adjacency, transfer rem * (w / max(outdeg[pred], 1)), queue degrees = in-degrees (backward_push.rs:179-220)
__global__ __launch_bounds__(SL_ACL_THREADS) void acl_kernel(uint32_t n, int backward, const uint32_t *__restrict__ rp, const uint32_t *__restrict__ ci,
                                                             const double *__restrict__ w, const double *__restrict__ qdeg, const double *__restrict__ outdeg,
                                                             const uint8_t *__restrict__ dup, double alpha, double epsilon, int adaptive,
                                                             unsigned long long max_pushes, unsigned long long budget, int with_target, uint32_t target,
                                                             double target_precision, double *est, double *res, uint8_t *inq, uint8_t *visited, uint32_t *qnode,
                                                             double *qprio, uint32_t *log, unsigned long long log_cap, acl_ctl *ctl)
{
    __shared__ double sp[SL_ACL_THREADS / 64];
    __shared__ uint32_t sn[SL_ACL_THREADS / 64], spos[SL_ACL_THREADS / 64];
    __shared__ uint32_t s_qlen, s_node, s_mode;      // mode 0: stop, 1: popped but skipped, 2: push with edges, 3: push of a dangling node
    __shared__ double s_rem, s_deg, s_thr;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (ctl->state != 0) return;
    if (tid == 0) { s_qlen = ctl->qlen; s_thr = ctl->threshold; }
    __syncthreads();
    unsigned long long pushes = ctl->pushes;
    for (unsigned long long step = 0; step < budget; ++step) {
        const uint32_t qlen = s_qlen;
        // ---- WorkQueue::pop: the largest (priority, node id) of the queue ----
        double bp = -INFINITY;
        uint32_t bn = 0, bpos = 0xffffffffu;
        for (uint32_t q = tid; q < qlen; q += SL_ACL_THREADS) {
            const double p = qprio[q];
            const uint32_t nd = qnode[q];
            if (bpos == 0xffffffffu || p > bp || (p == bp && nd > bn)) { bp = p; bn = nd; bpos = q; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double op = __shfl_xor(bp, o);
            const uint32_t on = __shfl_xor(bn, o), opos = __shfl_xor(bpos, o);
            if (opos != 0xffffffffu && (bpos == 0xffffffffu || op > bp || (op == bp && on > bn))) { bp = op; bn = on; bpos = opos; }
        }
        if (lane == 0) { sp[wave] = bp; sn[wave] = bn; spos[wave] = bpos; }
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < SL_ACL_THREADS / 64; ++wv)
                if (spos[wv] != 0xffffffffu && (bpos == 0xffffffffu || sp[wv] > bp || (sp[wv] == bp && sn[wv] > bn))) { bp = sp[wv]; bn = sn[wv]; bpos = spos[wv]; }
            uint32_t mode = 0;
            if (qlen == 0) ctl->state = 1;                                                   // while !work_queue.is_empty()
            else if (pushes >= max_pushes) ctl->state = 2;                                   // && push_count < max_pushes
            else if (with_target && est[target] > target_precision && res[target] < target_precision * 0.1) ctl->state = 3;   // forward_push.rs:262-265
            else {
                const uint32_t u = bn;
                qnode[bpos] = qnode[qlen - 1]; qprio[bpos] = qprio[qlen - 1];                // out of the queue
                inq[u] = 0;
                s_qlen = qlen - 1;
                mode = 1;
                const double du = fmax(qdeg[u], 1.0);
                if (!(res[u] < epsilon * du)) {                                              // forward_push.rs:96-99
                    mode = 2;
                    const double ru = res[u];
                    double rem = 0.0, deg = qdeg[u];
                    if (!(ru <= 0.0)) {                                                      // push_node, :186-188
                        est[u] = DADD(est[u], DMUL(alpha, ru));
                        rem = DMUL(1.0 - alpha, ru);
                        res[u] = 0.0;
                        if (!(deg > 0.0)) {                                                  // :210-215: the mass stays on the node
                            res[u] = DADD(res[u], rem);
                            const double pr = res[u];                                        // degree 1
                            if (pr >= s_thr && !inq[u]) { const uint32_t q = s_qlen++; qnode[q] = u; qprio[q] = pr; inq[u] = 1; }
                            mode = 3;
                        }
                    } else mode = 3;                                                         // returned early: counted, nothing moves
                    s_rem = rem; s_deg = deg;
                    if (!visited[u]) { visited[u] = 1; ++ctl->nvis; }
                    if (log && pushes < log_cap) log[pushes] = u;
                }
                s_node = u;
            }
            s_mode = mode;
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t mode = s_mode;
        if (mode == 0) break;
        if (mode == 2) {
            const uint32_t u = s_node, e0 = rp[u], e1 = rp[u + 1];
            const double rem = s_rem, deg = s_deg, thr = s_thr;
            auto edge = [&](uint32_t k) {
                const uint32_t v = ci[k];
                const double m = backward ? DMUL(rem, DDIV(w[k], fmax(outdeg[v], 1.0))) : DDIV(DMUL(rem, w[k]), deg);
                const double rv = DADD(res[v], m);
                res[v] = rv;
                const double dv = fmax(qdeg[v], 1.0);
                const double pr = DDIV(rv, dv);
                if (pr >= thr && !inq[v]) { const uint32_t q = atomicAdd(&s_qlen, 1u); qnode[q] = v; qprio[q] = pr; inq[v] = 1; }
            };
            if (!dup[u]) { for (uint32_t k = e0 + tid; k < e1; k += SL_ACL_THREADS) edge(k); }
            else if (tid == 0) { for (uint32_t k = e0; k < e1; ++k) edge(k); }               // the same neighbour twice: one after the other
        }
        __threadfence_block();
        __syncthreads();
        if (mode >= 2) {
            ++pushes;
            if (tid == 0 && adaptive && pushes % 1000ull == 0ull) {                          // WorkQueue::adaptive_threshold(10000, 100), graph/mod.rs:204-212
                if (s_qlen > 10000u) s_thr = DMUL(s_thr, 1.1);
                else if (s_qlen < 100u && s_thr > 1e-12) s_thr = DMUL(s_thr, 0.9);
            }
            __syncthreads();
        }
    }
    if (tid == 0) { ctl->pushes = pushes; ctl->qlen = s_qlen; ctl->threshold = s_thr; }
}

sl_status upload(const void *src, size_t bytes, sl_mem where, void **out, hipStream_t s)
{
    SL_HIP(sl_malloc(out, bytes ? bytes : 8));
    if (bytes) SL_HIP(hipMemcpyAsync(*out, src, bytes, where == SL_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s));
    return SL_OK;
}

sl_status acl_run(const sl_push_graph *g, int backward, uint64_t nsrc, const uint64_t *sources, const sl_acl_options *o, int with_target, uint64_t target,
                  double target_precision, double *estimate, double *residual, uint32_t *push_log, uint64_t log_cap, sl_acl_result *out)
{
    if (!g || !o || !estimate || !residual || !out || (nsrc && !sources)) return sl_fail(SL_INVALID_INPUT, "null argument");
    memset(out, 0, sizeof(*out));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sl_fail(SL_DEVICE_ERROR, "no HIP device available; libsublinear_hip has no CPU fallback");
    const uint64_t n = g->n;
    hipStream_t s = sl_context().stream;
    const sl_mem where = (sl_mem)o->mem;
    const hipMemcpyKind out_kind = where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    sl_range trace_range(backward ? "acl backward push" : "acl forward push");
    DevBuf est, res, inq, vis, qn, qp, ctlb, srcb, logb;
    SL_TRY(est.alloc((n ? n : 1) * 8)); SL_TRY(res.alloc((n ? n : 1) * 8)); SL_TRY(inq.alloc(n ? n : 1)); SL_TRY(vis.alloc(n ? n : 1));
    SL_TRY(qn.alloc((n ? n : 1) * 4)); SL_TRY(qp.alloc((n ? n : 1) * 8)); SL_TRY(ctlb.alloc(sizeof(acl_ctl))); SL_TRY(srcb.alloc((nsrc ? nsrc : 1) * 8));
    const uint64_t log_dev = push_log ? std::min<uint64_t>(log_cap, o->max_pushes) : 0;
    if (log_dev) SL_TRY(logb.alloc(log_dev * 4));
    SL_HIP(hipMemsetAsync(est.p, 0, (n ? n : 1) * 8, s)); SL_HIP(hipMemsetAsync(res.p, 0, (n ? n : 1) * 8, s));
    SL_HIP(hipMemsetAsync(inq.p, 0, n ? n : 1, s)); SL_HIP(hipMemsetAsync(vis.p, 0, n ? n : 1, s));
    acl_ctl h;
    memset(&h, 0, sizeof(h));
    // forward_push.rs:75-83 / :238-246: a single source (or the target) out of range is an empty result, not an error
    const bool empty = n == 0 || nsrc == 0 || (nsrc == 1 && sources[0] >= n) || (with_target && target >= n);
    sl_timer timer;
    SL_TRY(timer.start(s));
    if (!empty) {
        // sources live on the host in both memory modes (a handful of indices)
        SL_HIP(hipMemcpyAsync(srcb.p, sources, nsrc * 8, hipMemcpyHostToDevice, s));
        const double *qdeg = backward ? g->d_rdeg : g->d_deg;
        hipLaunchKernelGGL(acl_init_kernel, dim3(1), dim3(1), 0, s, n, nsrc, srcb.as<uint64_t>(), qdeg, o->queue_threshold, res.as<double>(), inq.as<uint8_t>(),
                           qn.as<uint32_t>(), qp.as<double>(), ctlb.as<acl_ctl>());
        SL_HIP(hipGetLastError());
        const unsigned long long budget = 4096;          // pops per launch: bounded kernel time, the host relaunches until the loop ends
        do {
            hipLaunchKernelGGL(acl_kernel, dim3(1), dim3(SL_ACL_THREADS), 0, s, (uint32_t)n, backward, backward ? g->d_trp : g->d_rp, backward ? g->d_tci : g->d_ci,
                               backward ? g->d_tw : g->d_w, qdeg, g->d_deg, backward ? g->d_tdup : g->d_dup, o->alpha, o->epsilon, o->adaptive_threshold,
                               (unsigned long long)o->max_pushes, budget, with_target, (uint32_t)target, target_precision, est.as<double>(), res.as<double>(),
                               inq.as<uint8_t>(), vis.as<uint8_t>(), qn.as<uint32_t>(), qp.as<double>(), logb.as<uint32_t>(), (unsigned long long)log_dev,
                               ctlb.as<acl_ctl>());
            SL_HIP(hipGetLastError());
            SL_HIP(hipMemcpyAsync(&h, ctlb.p, sizeof(h), hipMemcpyDeviceToHost, s));
            SL_HIP(hipStreamSynchronize(s));
        } while (h.state == 0);
    }
    out->device_time_ms = timer.stop();
    out->push_count = h.pushes;
    out->nodes_visited = h.nvis;
    out->stopped_by = empty ? 0 : (int32_t)h.state;
    double hsum = 0.0;
    if (n) {
        double *scr = static_cast<double *>(sl_scratch(4200 * sizeof(double)));
        if (!scr) return sl_fail(SL_ALLOCATION, "scratch");
        if (!empty) {
            SL_TRY(sl_launch_sumsq(n, res.as<double>(), scr, scr + 4100, s));      // compute_residual_norm, forward_push.rs:219-221 (tree sum: equal to ~1e-16)
            SL_HIP(hipMemcpyAsync(&hsum, scr + 4100, 8, hipMemcpyDeviceToHost, s));
        }
        SL_HIP(hipMemcpyAsync(estimate, est.p, n * 8, out_kind, s));
        SL_HIP(hipMemcpyAsync(residual, res.p, n * 8, out_kind, s));
    }
    if (log_dev && h.pushes) SL_HIP(hipMemcpyAsync(push_log, logb.p, std::min<uint64_t>(log_dev, h.pushes) * 4, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    out->residual_norm = std::sqrt(hsum);
    return SL_OK;
}

} // namespace

extern "C" {

// PushGraph::from_matrix, src/graph/adjacency.rs:212-224 (from_edges, :226-238, is an edge list sorted into this CSR by the host)
sl_status sl_push_graph_create(uint64_t n, const uint32_t *row_ptr, const uint32_t *col_idx, const double *weights, sl_mem where, sl_push_graph **out)
{
    SL_ABI_BEGIN
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!row_ptr || (n > 0xfffffffeull)) return sl_fail(SL_INVALID_INPUT, "null row_ptr or more than 2^32 - 2 nodes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sl_fail(SL_DEVICE_ERROR, "no HIP device available; libsublinear_hip has no CPU fallback");
    hipStream_t s = sl_context().stream;
    sl_range trace_range("push graph build");
    uint32_t last = 0;
    if (where == SL_MEM_HOST) last = row_ptr[n];
    else { SL_HIP(hipMemcpyAsync(&last, row_ptr + n, 4, hipMemcpyDeviceToHost, s)); SL_HIP(hipStreamSynchronize(s)); }
    const uint64_t nnz = last;
    if (nnz && (!col_idx || !weights)) return sl_fail(SL_INVALID_INPUT, "null col_idx / weights");
    std::unique_ptr<sl_push_graph> g(new sl_push_graph());
    g->n = n; g->nnz = nnz;
    (void)hipGetDevice(&g->device);
    SL_TRY(upload(row_ptr, (n + 1) * 4, where, (void **)&g->d_rp, s));
    SL_TRY(upload(col_idx, nnz * 4, where, (void **)&g->d_ci, s));
    SL_TRY(upload(weights, nnz * 8, where, (void **)&g->d_w, s));
    DevBuf err, rowof, perm, iota, keys, keys_out, k64, k64o;
    SL_TRY(err.alloc(16));
    SL_HIP(hipMemsetAsync(err.p, 0, 16, s));
    hipLaunchKernelGGL(pg_validate_kernel, dim3(grid_of(std::max(n, nnz))), dim3(256), 0, s, n, nnz, g->d_rp, g->d_ci, g->d_w, err.as<uint32_t>());
    uint32_t herr = 0;
    SL_HIP(hipMemcpyAsync(&herr, err.p, 4, hipMemcpyDeviceToHost, s));
    SL_HIP(hipStreamSynchronize(s));
    if (herr & 1u) return sl_fail(SL_INVALID_SPARSE_MATRIX, "row_ptr is not a monotone 0..nnz prefix array");
    if (herr & 2u) return sl_fail(SL_INDEX_OUT_OF_BOUNDS, "edge endpoint >= num_nodes (%llu)", (unsigned long long)n);
    if (herr & 4u) return sl_fail(SL_INVALID_INPUT, "non-finite edge weight");
    SL_HIP(sl_malloc(&g->d_deg, (n ? n : 1) * 8)); SL_HIP(sl_malloc(&g->d_rdeg, (n ? n : 1) * 8));
    SL_HIP(sl_malloc(&g->d_dup, n ? n : 1)); SL_HIP(sl_malloc(&g->d_tdup, n ? n : 1));
    SL_HIP(sl_malloc(&g->d_trp, (n + 1) * 4)); SL_HIP(sl_malloc(&g->d_tci, (nnz ? nnz : 1) * 4)); SL_HIP(sl_malloc(&g->d_tw, (nnz ? nnz : 1) * 8));
    SL_HIP(hipMemsetAsync(g->d_dup, 0, n ? n : 1, s));
    SL_HIP(hipMemsetAsync(g->d_trp, 0, (n + 1) * 4, s));
    if (n) hipLaunchKernelGGL(pg_row_sums_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, g->d_rp, g->d_w, g->d_deg);
    if (nnz) {
        SL_TRY(rowof.alloc(nnz * 4)); SL_TRY(perm.alloc(nnz * 4)); SL_TRY(iota.alloc(nnz * 4)); SL_TRY(keys_out.alloc(nnz * 4));
        hipLaunchKernelGGL(pg_rows_of_entries_kernel, dim3(grid_of(n * 64)), dim3(256), 0, s, n, g->d_rp, rowof.as<uint32_t>());
        hipLaunchKernelGGL(pg_count_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, g->d_ci, g->d_trp);
        // the column histogram -> row pointers of the reverse adjacency (host prefix sum: one-off, n + 1 words)
        std::vector<uint32_t> tp(n + 1);
        SL_TRY(sl_read_back(tp.data(), g->d_trp, (n + 1) * 4, s));                                   // (control-plane transfers: the library's pinned staging)
        uint64_t run = 0;
        for (uint64_t j = 0; j <= n; ++j) { run += tp[j]; tp[j] = (uint32_t)run; }
        if (run != nnz) return sl_fail(SL_DEVICE_ERROR, "push graph: the in-degree histogram as read back sums to %llu, the graph holds %llu edges", (unsigned long long)run, (unsigned long long)nnz);
        SL_TRY(sl_upload(g->d_trp, tp.data(), (n + 1) * 4, s));
        hipLaunchKernelGGL(pg_iota_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, iota.as<uint32_t>());
        int bits = 1;
        while (bits < 32 && (1ull << bits) < n) ++bits;
        // stable sort by target: entries of one target keep their storage order = ascending source (graph/mod.rs:92-130)
        SL_TRY(sl_sort_pairs_u32(g->d_ci, keys_out.as<uint32_t>(), iota.as<uint32_t>(), perm.as<uint32_t>(), nnz, bits, s));
        hipLaunchKernelGGL(pg_transpose_fill_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, perm.as<uint32_t>(), rowof.as<uint32_t>(), g->d_w, g->d_tci, g->d_tw);
        // duplicates (u, v): adjacent after a sort by (row, column)
        SL_TRY(k64.alloc(nnz * 8)); SL_TRY(k64o.alloc(nnz * 8));
        hipLaunchKernelGGL(pg_keys64_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, rowof.as<uint32_t>(), g->d_ci, k64.as<unsigned long long>());
        SL_TRY(sl_sort_pairs_u64(k64.as<uint64_t>(), k64o.as<uint64_t>(), iota.as<uint32_t>(), keys_out.as<uint32_t>(), nnz, 64, s));
        hipLaunchKernelGGL(pg_dup_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, k64o.as<unsigned long long>(), g->d_dup);
    }
    if (n) {
        hipLaunchKernelGGL(pg_row_sums_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, g->d_trp, g->d_tw, g->d_rdeg);
        hipLaunchKernelGGL(pg_tdup_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, n, g->d_trp, g->d_tci, g->d_tdup);
    }
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(s));
    sl_log(1, "push graph: %llu nodes, %llu edges, reverse adjacency and degrees on the device", (unsigned long long)n, (unsigned long long)nnz);
    *out = g.release();
    return SL_OK;
    SL_ABI_END
}

void sl_push_graph_destroy(sl_push_graph *g) { delete g; }

sl_status sl_push_graph_size(const sl_push_graph *g, uint64_t *num_nodes, uint64_t *num_edges)
{
    if (!g) return sl_fail(SL_INVALID_INPUT, "null graph");
    if (num_nodes) *num_nodes = g->n;
    if (num_edges) *num_edges = g->nnz;
    return SL_OK;
}

// PushGraph::degrees / reverse_degrees (row sums / column sums, adjacency.rs:214-217); either pointer may be null
sl_status sl_push_graph_degrees(const sl_push_graph *g, double *out_degrees, double *in_degrees, sl_mem where)
{
    SL_ABI_BEGIN
    if (!g) return sl_fail(SL_INVALID_INPUT, "null graph");
    hipStream_t s = sl_context().stream;
    const hipMemcpyKind k = where == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    if (out_degrees && g->n) SL_HIP(hipMemcpyAsync(out_degrees, g->d_deg, g->n * 8, k, s));
    if (in_degrees && g->n) SL_HIP(hipMemcpyAsync(in_degrees, g->d_rdeg, g->n * 8, k, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
    SL_ABI_END
}

// SL_SYSTEM_FORWARD: A = I - (1 - alpha) P^T, the system whose solution for b = alpha e_s is the personalised PageRank vector of s and for
// b = alpha / n the PageRank vector (core/solver.ts:664-722 with d = 1 - alpha); backward = 1: A = I - (1 - alpha) P.  Assembled on the
// device in CSR (sorted columns, one diagonal entry per row), handed to sl_matrix_create_csr with `matrix_flags`.
sl_status sl_push_graph_system(const sl_push_graph *g, double alpha, uint32_t system_flags, uint32_t matrix_flags, sl_matrix **out)
{
    const int backward = (system_flags & SL_SYSTEM_BACKWARD) ? 1 : 0, dangling_identity = (system_flags & SL_SYSTEM_DANGLING_IDENTITY) ? 1 : 0;
    SL_ABI_BEGIN
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (!g) return sl_fail(SL_INVALID_INPUT, "null graph");
    if (!(alpha > 0.0 && alpha <= 1.0)) return sl_fail(SL_INVALID_INPUT, "alpha must lie in (0, 1]");
    const uint64_t n = g->n, nnz = g->nnz;
    hipStream_t s = sl_context().stream;
    sl_range trace_range("push graph system assembly");
    // sorted source rows: forward = the reverse adjacency (sources ascending by construction); backward = the adjacency sorted by column
    const uint32_t *srp = g->d_trp, *sci = g->d_tci;
    const double *sw = g->d_tw;
    DevBuf rowof, k64, k64o, iota, perm, sci_b, sw_b;
    if (backward) {
        srp = g->d_rp;
        if (nnz) {
            SL_TRY(rowof.alloc(nnz * 4)); SL_TRY(k64.alloc(nnz * 8)); SL_TRY(k64o.alloc(nnz * 8)); SL_TRY(iota.alloc(nnz * 4)); SL_TRY(perm.alloc(nnz * 4));
            SL_TRY(sci_b.alloc(nnz * 4)); SL_TRY(sw_b.alloc(nnz * 8));
            hipLaunchKernelGGL(pg_rows_of_entries_kernel, dim3(grid_of(n * 64)), dim3(256), 0, s, n, g->d_rp, rowof.as<uint32_t>());
            hipLaunchKernelGGL(pg_keys64_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, rowof.as<uint32_t>(), g->d_ci, k64.as<unsigned long long>());
            hipLaunchKernelGGL(pg_iota_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, iota.as<uint32_t>());
            SL_TRY(sl_sort_pairs_u64(k64.as<uint64_t>(), k64o.as<uint64_t>(), iota.as<uint32_t>(), perm.as<uint32_t>(), nnz, 64, s));
            // entry e of the sorted rows = entry perm[e]: its column and weight (pg_transpose_fill_kernel with `ci` in the place of the row ids)
            hipLaunchKernelGGL(pg_transpose_fill_kernel, dim3(grid_of(nnz)), dim3(256), 0, s, nnz, perm.as<uint32_t>(), g->d_ci, g->d_w, sci_b.as<uint32_t>(), sw_b.as<double>());
            sci = sci_b.as<uint32_t>(); sw = sw_b.as<double>();
        }
    }
    DevBuf orp, oci, ova;
    SL_TRY(orp.alloc((n + 1) * 4));
    SL_HIP(hipMemsetAsync(orp.p, 0, (n + 1) * 4, s));
    const uint32_t grid = (uint32_t)((n + 255) / 256);
    if (n) hipLaunchKernelGGL((pg_system_kernel<true>), dim3(grid), dim3(256), 0, s, n, backward, dangling_identity, alpha, srp, sci, sw, g->d_deg, (const uint32_t *)nullptr, orp.as<uint32_t>(),
                              (uint32_t *)nullptr, (double *)nullptr);
    std::vector<uint32_t> hp(n + 1);
    SL_TRY(sl_read_back(hp.data(), orp.p, (n + 1) * 4, s));
    uint64_t run = 0;
    for (uint64_t i = 0; i <= n; ++i) { run += hp[i]; if (run > 0xfffffff0ull) return sl_fail(SL_ALLOCATION, "system too large for 32-bit entry counts"); hp[i] = (uint32_t)run; }
    const uint64_t onnz = run;
    SL_TRY(sl_upload(orp.p, hp.data(), (n + 1) * 4, s));
    SL_TRY(oci.alloc((onnz ? onnz : 1) * 4)); SL_TRY(ova.alloc((onnz ? onnz : 1) * 8));
    if (n) hipLaunchKernelGGL((pg_system_kernel<false>), dim3(grid), dim3(256), 0, s, n, backward, dangling_identity, alpha, srp, sci, sw, g->d_deg, orp.as<uint32_t>(), (uint32_t *)nullptr,
                              oci.as<uint32_t>(), ova.as<double>());
    SL_HIP(hipGetLastError());
    SL_HIP(hipStreamSynchronize(s));
    return sl_matrix_create_csr(n, n, onnz, orp.as<uint32_t>(), oci.as<uint32_t>(), ova.as<double>(), SL_MEM_DEVICE, 0, matrix_flags, out);
    SL_ABI_END
}

void sl_acl_options_default(sl_acl_options *o)      // ForwardPushConfig::default, forward_push.rs:38-47
{
    memset(o, 0, sizeof(*o));
    o->alpha = 0.15; o->epsilon = 1e-6; o->max_pushes = 1000000; o->queue_threshold = 1e-8; o->adaptive_threshold = 1; o->mem = SL_MEM_HOST;
}

sl_status sl_forward_push_acl(const sl_push_graph *g, uint64_t n_sources, const uint64_t *sources, const sl_acl_options *o, double *estimate, double *residual,
                              uint32_t *push_log, uint64_t log_cap, sl_acl_result *res)
{
    SL_ABI_BEGIN
    return acl_run(g, 0, n_sources, sources, o, 0, 0, 0.0, estimate, residual, push_log, log_cap, res);
    SL_ABI_END
}

sl_status sl_backward_push_acl(const sl_push_graph *g, uint64_t n_targets, const uint64_t *targets, const sl_acl_options *o, double *estimate, double *residual,
                               uint32_t *push_log, uint64_t log_cap, sl_acl_result *res)
{
    SL_ABI_BEGIN
    return acl_run(g, 1, n_targets, targets, o, 0, 0, 0.0, estimate, residual, push_log, log_cap, res);
    SL_ABI_END
}

sl_status sl_forward_push_acl_with_target(const sl_push_graph *g, uint64_t source, uint64_t target, double target_precision, const sl_acl_options *o,
                                          double *estimate, double *residual, uint32_t *push_log, uint64_t log_cap, sl_acl_result *res)
{
    SL_ABI_BEGIN
    return acl_run(g, 0, 1, &source, o, 1, target, target_precision, estimate, residual, push_log, log_cap, res);
    SL_ABI_END
}

// BackwardPushSolver::solve_with_source, backward_push.rs:238-293: the same queue kernel over the reverse adjacency; the mass starts at
// `target`, the node whose precision ends the loop is `source` (:262-264); either out of range: the empty result (:243-251)
sl_status sl_backward_push_acl_with_source(const sl_push_graph *g, uint64_t source, uint64_t target, double source_precision, const sl_acl_options *o,
                                           double *estimate, double *residual, uint32_t *push_log, uint64_t log_cap, sl_acl_result *res)
{
    SL_ABI_BEGIN
    return acl_run(g, 1, 1, &target, o, 1, source, source_precision, estimate, residual, push_log, log_cap, res);
    SL_ABI_END
}

// {Forward,Backward}PushSolver::extrapolated_solution, forward_push.rs:292-301 / backward_push.rs:302-311
sl_status sl_acl_extrapolated_solution(uint64_t count, double alpha, const double *estimate, const double *residual, double *solution, sl_mem where)
{
    SL_ABI_BEGIN
    if (count && (!estimate || !residual || !solution)) return sl_fail(SL_INVALID_INPUT, "null argument");
    if (!count) return SL_OK;
    if (where == SL_MEM_HOST) memcpy(solution, estimate, count * 8);                       // solution = estimate.clone()
    else {
        hipStream_t s = sl_context().stream;
        SL_HIP(hipMemcpyAsync(solution, estimate, count * 8, hipMemcpyDeviceToDevice, s));
    }
    return sl_axpy(count, alpha, residual, solution, where);                               // solution[i] += alpha * res   (device; mul, then add)
    SL_ABI_END
}

// BackwardPushSolver::reachability_probabilities, backward_push.rs:296-299
sl_status sl_backward_push_acl_reachability(const sl_push_graph *g, uint64_t target, const sl_acl_options *o, double *solution, sl_acl_result *res)
{
    SL_ABI_BEGIN
    if (!g || !o || !solution || !res) return sl_fail(SL_INVALID_INPUT, "null argument");
    const uint64_t n = g->n;
    DevBuf est, rsd;
    SL_TRY(est.alloc((n ? n : 1) * 8)); SL_TRY(rsd.alloc((n ? n : 1) * 8));
    sl_acl_options od = *o;
    od.mem = SL_MEM_DEVICE;
    SL_TRY(acl_run(g, 1, 1, &target, &od, 0, 0, 0.0, est.as<double>(), rsd.as<double>(), nullptr, 0, res));
    if (!n) return SL_OK;
    hipStream_t s = sl_context().stream;
    SL_TRY(sl_launch_axpy(n, o->alpha, rsd.as<double>(), est.as<double>(), s));
    SL_HIP(hipMemcpyAsync(solution, est.p, n * 8, o->mem == SL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, s));
    SL_HIP(hipStreamSynchronize(s));
    return SL_OK;
    SL_ABI_END
}

} // extern "C"
