// sl_comm.hip — the multi-GPU side of the C ABI: one process per GPU of ONE node, row-range partition.
//
// Precedent in the reference: simd_ops::parallel_matrix_vector_multiply (src/simd_ops.rs:201-239) hides row chunks behind one
// call; here a rank = a row chunk on its own GPU, and the pieces below hide the exchange behind sl_neumann_state_* so that a host
// in any language reaches N GPUs through include/sublinear_hip.h alone.
//
// Transport (what moves, and how it is kept coherent BY CONSTRUCTION — nothing here has run on two devices in the build image):
//   * rendezvous + control words: ONE POSIX shared-memory file per communicator (/dev/shm/slcomm_<name>), mapped by every rank and
//     registered with the HIP runtime (hipHostRegister, mapped): host memory is never cached on a device, so a system-scope store by
//     a kernel on one GPU is what a system-scope load on another GPU (or a host) sees.  It carries the host barrier, the exchange
//     of IPC handles, and per rank one 128-byte line {ticket} plus a ring of published doubles.
//   * vectors: the full-length gathered vectors (two term buffers + the solution, n_global doubles each) are ordinary device
//     allocations exported with hipIpcGetMemHandle; a rank PULLS the pieces it needs out of its peers' buffers with
//     hipMemcpyAsync (device to device over xGMI between GPUs; a plain copy when two ranks share a GPU — the test setup): the
//     runtime's copy path reads what the producer's finished kernels wrote back (kernel boundaries write dirty L2 lines back) and
//     the destination is written locally, so no rank ever reads lines a peer wrote behind its own caches.
//   * order: every collective point is a TICKET (a counter that all ranks advance in the same order).  A rank publishes
//     {value, ticket} after the work the ticket stands for (stream order), then waits until every rank's ticket has arrived, sums
//     the values in rank order (the same bits on every rank) and applies the stop rule to the sum — the all-reduce of ||t||^2 and
//     the "data ready" handshake of the halo in one kernel of 64 lanes.  Because every ticket waits for ALL ranks, a buffer is
//     never overwritten while a peer still pulls from it (its next writer has passed a later ticket than the puller's copy).
//     A wait is bounded (SL_COMM_TIMEOUT_MS, default 20 s): a dead peer turns into SL_DEVICE_ERROR, not into a hung GPU.
//   RCCL: the exchange through torch.distributed / RCCL (all-gather, grouped send/recv, all-reduce form) stays available above the
//   ABI in sublinear_time_solver_amd/distributed.py; this file needs neither RCCL nor torch.
#include "sl_internal.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <time.h>
#include <unistd.h>
#define SL_COMM_FRESH_S 120u

static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the blob slots assume");

// ---- host side of the shared block --------------------------------------------------------------------------------------------
namespace {
inline uint64_t ld_acq(const volatile uint64_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void st_rel(volatile uint64_t *p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

long comm_timeout_ms()
{
    static const long v = [] { const char *e = getenv("SL_COMM_TIMEOUT_MS"); const long t = e && *e ? atol(e) : 20000; return t > 0 ? t : 20000; }();
    return v;
}

template <class F> bool wait_until(F cond)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0; !cond(); ++spin) {
        if (spin > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > (double)comm_timeout_ms()) return false;
    }
    return true;
}
} // namespace

sl_status sl_comm_host_barrier(sl_comm *c)
{
    sl_comm_shm *h = c->h_shm;
    const uint64_t k = ++c->barrier_count;
    st_rel(&h->arrive[c->rank], k);
    const bool ok = wait_until([&] { for (int p = 0; p < c->world; ++p) if (ld_acq(&h->arrive[p]) < k) return false; return true; });
    if (!ok) return sl_fail(SL_DEVICE_ERROR, "communicator barrier timed out after %ld ms (rank %d of %d)", comm_timeout_ms(), c->rank, c->world);
    return SL_OK;
}

// every rank contributes `bytes` (<= SL_COMM_BLOB) and receives all contributions, rank order
sl_status sl_comm_allgather_blob(sl_comm *c, const void *mine, size_t bytes, void *all)
{
    if (bytes > SL_COMM_BLOB) return sl_fail(SL_INVALID_INPUT, "blob of %zu bytes exceeds %d", bytes, SL_COMM_BLOB);
    sl_comm_shm *h = c->h_shm;
    const uint64_t k = ++c->blob_count;
    memcpy(const_cast<unsigned char *>(h->blob[c->rank]), mine, bytes);
    st_rel(&h->blob_seq[c->rank], k);
    const bool ok = wait_until([&] { for (int p = 0; p < c->world; ++p) if (ld_acq(&h->blob_seq[p]) < k) return false; return true; });
    if (!ok) return sl_fail(SL_DEVICE_ERROR, "communicator exchange timed out after %ld ms (rank %d of %d)", comm_timeout_ms(), c->rank, c->world);
    for (int p = 0; p < c->world; ++p) memcpy(static_cast<unsigned char *>(all) + (size_t)p * bytes, const_cast<unsigned char *>(h->blob[p]), bytes);
    return sl_comm_host_barrier(c);                 // nobody overwrites its slot before everybody has read it
}

// ---- device side: publish / wait for all / sum in rank order / stop rule --------------------------------------------------------
__device__ __forceinline__ uint64_t sys_load(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// One wave.  `local` (device) holds this rank's contribution (may be null: a pure "data ready" ticket, value 0).  With ctl: gated
// like every launch of a speculative batch; the sum is logged and judged exactly as sl_judge_reduce_kernel does for one GPU.
// Channel 1 ("halo ready", the boundary-first step) carries no value and judges nothing: it has its own ticket counter and its own
// word of the rank's line, so that it may run on a second stream beside the sums of channel 0.
__global__ __launch_bounds__(64) void sl_comm_ticket_kernel(sl_comm_shm *shm, int rank, int world, uint64_t ticket, const double *local,
                                                             double *result, sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot, int mode, double thr,
                                                             unsigned long long timeout_ticks, int chan)
{
    if (ctl && gate_it > ctl->stop_after) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t ring = (uint32_t)(ticket % SL_COMM_RING);
    if (__hip_atomic_load(&shm->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) return;      // a rank gave up: drain the queue without waiting
    if (lane == 0) {
        if (chan == 0) {
            const double v = local ? *local : 0.0;
            __hip_atomic_store(&shm->value[ring][rank], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __hip_atomic_store(&shm->ready[rank][chan], ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // everything before this launch is done (stream order)
    }
    bool ok = true;
    if (lane < (uint32_t)world) {
        const unsigned long long t0 = wall_clock64();
        while (sys_load(&shm->ready[lane][chan]) < ticket) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > timeout_ticks) { ok = false; break; }
        }
    }
    const bool all_ok = __ballot(!ok) == 0ull;
    if (lane != 0) return;
    if (!all_ok) {
        __hip_atomic_store(&shm->error, (uint64_t)(rank + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (ctl) ctl->stop_after = 0;                                    // nothing enqueued behind this runs
        return;
    }
    if (chan != 0) return;
    double t = 0.0;
    for (int p = 0; p < world; ++p) t += __hip_atomic_load(&shm->value[ring][p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // rank order: same bits everywhere
    if (result) result[0] = t;
    if (ctl && mode != SL_JUDGE_LOCAL) {
        ctl->log[slot] = t;
        ctl->n_done = slot + 1;
        bool stop = false;
        if (mode == SL_JUDGE_LT) stop = t < thr;
        else if (mode == SL_JUDGE_LE_OR_NONFINITE) stop = (t <= thr) || (t != t) || (fabs(t) == INFINITY);
        if (stop && gate_it < ctl->stop_after) ctl->stop_after = gate_it;
    }
}

sl_status sl_comm_launch_ticket(sl_comm *c, const double *local, double *result, sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot, int mode,
                                double thr, hipStream_t s, int channel)
{
    const uint64_t ticket = ++c->ticket[channel ? 1 : 0];
    const unsigned long long ticks = (unsigned long long)comm_timeout_ms() * 100000ull;            // wall_clock64: 100 MHz
    hipLaunchKernelGGL(sl_comm_ticket_kernel, dim3(1), dim3(64), 0, s, c->d_shm, c->rank, c->world, ticket, local, result, ctl, gate_it, slot, mode,
                       thr, ticks, channel ? 1 : 0);
    SL_HIP(hipGetLastError());
    return SL_OK;
}

bool sl_comm_failed(const sl_comm *c) { return __atomic_load_n(&c->h_shm->error, __ATOMIC_ACQUIRE) != 0; }

// ---- partitioned vectors --------------------------------------------------------------------------------------------------------
// A full-length vector (n_global doubles) per rank, exported to the peers; pull() copies the pieces this rank needs from their
// owners' copies into its own.
sl_status sl_dist_vector_create(sl_comm *c, uint64_t n_global, sl_dist_vector *v)
{
    v->n = n_global;
    v->peer.assign((size_t)c->world, nullptr);
    SL_HIP(hipMalloc(&v->mine, (n_global ? n_global : 1) * sizeof(double)));
    SL_HIP(hipMemset(v->mine, 0, (n_global ? n_global : 1) * sizeof(double)));
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof(h));
    if (c->world > 1) SL_HIP(hipIpcGetMemHandle(&h, v->mine));
    std::vector<hipIpcMemHandle_t> all((size_t)c->world);
    SL_TRY(sl_comm_allgather_blob(c, &h, sizeof(h), all.data()));
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) { v->peer[p] = v->mine; continue; }
        void *q = nullptr;
        SL_HIP(hipIpcOpenMemHandle(&q, all[p], hipIpcMemLazyEnablePeerAccess));
        v->peer[p] = static_cast<double *>(q);
    }
    return SL_OK;
}

void sl_dist_vector_destroy(sl_comm *c, sl_dist_vector *v)
{
    for (int p = 0; p < (int)v->peer.size(); ++p)
        if (p != c->rank && v->peer[p]) (void)hipIpcCloseMemHandle(v->peer[p]);
    v->peer.clear();
    if (v->mine) (void)hipFree(v->mine);
    v->mine = nullptr;
}

#define SL_PULL_PARALLEL_BYTES (4u << 20)      // below this an exchange is two halo strips: one stream, no forks
sl_status sl_dist_pull(sl_dist *d, sl_dist_vector *v, hipStream_t s)
{
    static const bool parallel_ok = [] { const char *e = getenv("SL_PULL_STREAMS"); return !(e && *e && atoi(e) == 0); }();
    auto copy = [&](const sl_dist::piece &pc, hipStream_t q) {
        return hipMemcpyAsync(v->mine + pc.lo, v->peer[pc.rank] + pc.lo, (pc.hi - pc.lo) * sizeof(double), hipMemcpyDeviceToDevice, q);
    };
    if (!parallel_ok || d->need.size() < 2 || d->pull_bytes < SL_PULL_PARALLEL_BYTES) {
        for (const sl_dist::piece &pc : d->need)
            if (pc.hi > pc.lo) SL_HIP(copy(pc, s));
        return SL_OK;
    }
    const size_t ns = std::min<size_t>(d->need.size(), 8);
    if (d->pull_streams.size() < ns) {
        if (!d->pull_fork) SL_HIP(hipEventCreateWithFlags(&d->pull_fork, hipEventDisableTiming));
        while (d->pull_streams.size() < ns) {
            hipStream_t q = nullptr;
            hipEvent_t e = nullptr;
            SL_HIP(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
            if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipStreamDestroy(q); return sl_fail(SL_DEVICE_ERROR, "event creation failed"); }
            d->pull_streams.push_back(q); d->pull_done.push_back(e);
        }
    }
    SL_HIP(hipEventRecord(d->pull_fork, s));
    for (size_t q = 0; q < ns; ++q) SL_HIP(hipStreamWaitEvent(d->pull_streams[q], d->pull_fork, 0));
    for (size_t i = 0; i < d->need.size(); ++i)
        if (d->need[i].hi > d->need[i].lo) SL_HIP(copy(d->need[i], d->pull_streams[i % ns]));
    for (size_t q = 0; q < ns; ++q) {
        SL_HIP(hipEventRecord(d->pull_done[q], d->pull_streams[q]));
        SL_HIP(hipStreamWaitEvent(s, d->pull_done[q], 0));
    }
    return SL_OK;
}

// ---- the communicator ----------------------------------------------------------------------------------------------------------
extern "C" {

sl_status sl_comm_create(int rank, int world, const char *rendezvous, sl_comm **out)
{
    SL_ABI_BEGIN
    if (!out) return sl_fail(SL_INVALID_INPUT, "out is null");
    *out = nullptr;
    if (world < 1 || world > SL_COMM_MAX_RANKS || rank < 0 || rank >= world)
        return sl_fail(SL_INVALID_INPUT, "rank %d of %d: a communicator spans 1..%d ranks of one node", rank, world, SL_COMM_MAX_RANKS);
    if (!rendezvous || !*rendezvous || strchr(rendezvous, '/')) return sl_fail(SL_INVALID_INPUT, "rendezvous must be a name without '/'");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return sl_fail(SL_DEVICE_ERROR, "no HIP device available; libsublinear_hip has no CPU fallback");
    sl_range trace_range("communicator create");
    sl_comm *c = new sl_comm();
    c->rank = rank; c->world = world;
    c->path = std::string("/dev/shm/slcomm_") + rendezvous;
    (void)hipGetDevice(&c->device);
    const size_t bytes = (sizeof(sl_comm_shm) + 4095) & ~(size_t)4095;
    auto fail = [&](sl_status st) { sl_comm_destroy(c); return st; };
    // rank 0 creates the block (zero-filled by ftruncate) and stamps it last; the others wait for a FRESH stamped block — a block
    // left behind under the same name by a job that died before all its ranks had joined (the name is unlinked at that point) is
    // older than SL_COMM_FRESH_S and is not joined
    const uint64_t now = (uint64_t)time(nullptr);
    if (rank == 0) {
        (void)unlink(c->path.c_str());
        c->fd = open(c->path.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (c->fd < 0 || ftruncate(c->fd, (off_t)bytes) != 0) return fail(sl_fail(SL_DEVICE_ERROR, "cannot create %s", c->path.c_str()));
        void *map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);
        if (map == MAP_FAILED) return fail(sl_fail(SL_DEVICE_ERROR, "mmap of %s failed", c->path.c_str()));
        c->h_shm = static_cast<sl_comm_shm *>(map);
        c->shm_bytes = bytes;
        c->h_shm->world = (uint64_t)world; c->h_shm->created_unix = now;
        st_rel(&c->h_shm->magic, SL_COMM_MAGIC);
    } else {
        const bool ok = wait_until([&] {
            const int fd = open(c->path.c_str(), O_RDWR);
            if (fd < 0) return false;
            struct stat sb;
            if (fstat(fd, &sb) != 0 || (size_t)sb.st_size < bytes) { close(fd); return false; }
            void *map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (map == MAP_FAILED) { close(fd); return false; }
            sl_comm_shm *h = static_cast<sl_comm_shm *>(map);
            if (ld_acq(&h->magic) == SL_COMM_MAGIC && h->created_unix + SL_COMM_FRESH_S >= now) { c->fd = fd; c->h_shm = h; c->shm_bytes = bytes; return true; }
            munmap(map, bytes); close(fd);
            return false;
        });
        if (!ok) return fail(sl_fail(SL_DEVICE_ERROR, "rank %d: no fresh rendezvous block %s within %ld ms", rank, c->path.c_str(), comm_timeout_ms()));
    }
    void *map = c->h_shm;
    if (c->h_shm->world != (uint64_t)world) return fail(sl_fail(SL_INVALID_INPUT, "rank %d joins a communicator of %llu ranks, asked for %d", rank,
                                                                  (unsigned long long)c->h_shm->world, world));
    if (hipHostRegister(map, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess)
        return fail(sl_fail(SL_DEVICE_ERROR, "hipHostRegister of the rendezvous block failed"));
    c->registered = true;
    void *dptr = nullptr;
    if (hipHostGetDevicePointer(&dptr, map, 0) != hipSuccess) return fail(sl_fail(SL_DEVICE_ERROR, "no device pointer for the rendezvous block"));
    c->d_shm = static_cast<sl_comm_shm *>(dptr);
    const sl_status bs = sl_comm_host_barrier(c);
    if (bs != SL_OK) return fail(bs);
    if (rank == 0) (void)unlink(c->path.c_str());      // everybody has it mapped: the name can go (the memory lives while mapped)
    sl_log(1, "communicator '%s': rank %d of %d on device %d", rendezvous, rank, world, c->device);
    *out = c;
    return SL_OK;
    SL_ABI_END
}

void sl_comm_destroy(sl_comm *c)
{
    if (!c) return;
    if (c->registered) (void)hipHostUnregister(c->h_shm);
    if (c->h_shm) munmap(c->h_shm, c->shm_bytes);
    if (c->fd >= 0) close(c->fd);
    delete c;
}

sl_status sl_comm_barrier(sl_comm *c)
{
    SL_ABI_BEGIN
    if (!c) return sl_fail(SL_INVALID_INPUT, "null communicator");
    SL_HIP(hipStreamSynchronize(sl_context().stream));
    return sl_comm_host_barrier(c);
    SL_ABI_END
}

sl_status sl_comm_rank(const sl_comm *c, int *rank, int *world)
{
    if (!c) return sl_fail(SL_INVALID_INPUT, "null communicator");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return SL_OK;
}

// contiguous row ranges balanced by stored entries — rank r starts at the first row whose prefix sum of row_ptr
// reaches r * nnz / world.  Host arithmetic on a host row_ptr; every rank derives the same bounds from the same row_ptr.
sl_status sl_balanced_row_bounds(uint64_t n_rows, const uint32_t *row_ptr, int world, uint64_t *bounds)
{
    if (!row_ptr || !bounds || world < 1) return sl_fail(SL_INVALID_INPUT, "null argument or world < 1");
    const uint64_t nnz = row_ptr[n_rows];
    bounds[0] = 0;
    for (int r = 1; r < world; ++r) {
        const uint64_t target = (uint64_t)r * nnz / (uint64_t)world;
        uint64_t lo = 0, hi = n_rows + 1;                         // first index i in [0, n_rows] with row_ptr[i] >= target
        while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (row_ptr[mid] < target) lo = mid + 1; else hi = mid; }
        bounds[r] = std::max<uint64_t>(std::min<uint64_t>(lo, n_rows), bounds[r - 1]);
    }
    bounds[world] = n_rows;
    return SL_OK;
}

sl_status sl_comm_allgather_u64(sl_comm *c, uint64_t mine, uint64_t *all)
{
    SL_ABI_BEGIN
    if (!c || !all) return sl_fail(SL_INVALID_INPUT, "null argument");
    return sl_comm_allgather_blob(c, &mine, sizeof(mine), all);
    SL_ABI_END
}

} // extern "C"

// ---- the partition of one solve -------------------------------------------------------------------------------------------------
// bounds from the ranks' row counts (consecutive ranges in rank order); need = for every peer the part of ITS rows this rank's
// columns can reach: [lo - w, hi + w) with w = the measured bandwidth of the local rows (sl_matrix::bandwidth), everything for
// matrices without a usable bandwidth.
sl_status sl_dist_create(sl_comm *c, const sl_matrix *local, sl_dist **out)
{
    *out = nullptr;
    sl_dist *d = new sl_dist();
    d->c = c;
    std::vector<uint64_t> rows((size_t)c->world), offs((size_t)c->world);
    sl_status st = sl_comm_allgather_blob(c, &local->n_rows, sizeof(uint64_t), rows.data());
    if (st == SL_OK) st = sl_comm_allgather_blob(c, &local->row_offset, sizeof(uint64_t), offs.data());
    if (st != SL_OK) { delete d; return st; }
    d->bounds.assign((size_t)c->world + 1, 0);
    for (int p = 0; p < c->world; ++p) {
        if (offs[p] != d->bounds[p]) { delete d; return sl_fail(SL_DIMENSION_MISMATCH, "rank %d's rows start at %llu, the ranks before it hold %llu rows", p,
                                                                 (unsigned long long)offs[p], (unsigned long long)d->bounds[p]); }
        d->bounds[p + 1] = d->bounds[p] + rows[p];
    }
    d->n_global = d->bounds[c->world];
    if (d->n_global != local->n_cols) { const uint64_t ng = d->n_global; delete d; return sl_fail(SL_DIMENSION_MISMATCH, "the ranks hold %llu rows in total, the matrix has %llu columns",
                                                                       (unsigned long long)ng, (unsigned long long)local->n_cols); }
    d->lo = d->bounds[c->rank]; d->hi = d->bounds[c->rank + 1];
    const uint64_t w = (local->bandwidth == ~0ull || local->nnz == 0) ? (local->nnz ? d->n_global : 0) : local->bandwidth;
    d->reach = w;
    {
        std::vector<uint64_t> reaches((size_t)c->world);
        st = sl_comm_allgather_blob(c, &w, sizeof(uint64_t), reaches.data());
        if (st != SL_OK) { delete d; return st; }
        d->max_reach = *std::max_element(reaches.begin(), reaches.end());
    }
    const uint64_t a = d->lo > w ? d->lo - w : 0, b = std::min<uint64_t>(d->n_global, d->hi + w);
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        const uint64_t plo = std::max(a, d->bounds[p]), phi = std::min(b, d->bounds[p + 1]);
        if (phi > plo) { d->need.push_back({p, plo, phi}); d->pull_bytes += (phi - plo) * 8; }
    }
    sl_log(1, "partition: rank %d holds rows [%llu, %llu) of %llu, reach %llu columns, pulls %.1f KB from %zu peers per exchange", c->rank,
           (unsigned long long)d->lo, (unsigned long long)d->hi, (unsigned long long)d->n_global, (unsigned long long)w, (double)d->pull_bytes / 1e3, d->need.size());
    *out = d;
    return SL_OK;
}

void sl_dist_destroy(sl_dist *d)
{
    if (!d) return;
    for (hipStream_t q : d->pull_streams) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    for (hipEvent_t e : d->pull_done) (void)hipEventDestroy(e);
    if (d->pull_fork) (void)hipEventDestroy(d->pull_fork);
    for (sl_dist_vector *v : {&d->t[0], &d->t[1], &d->x}) sl_dist_vector_destroy(d->c, v);
    delete d;
}
