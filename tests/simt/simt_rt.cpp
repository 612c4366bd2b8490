// tests/simt/simt_rt.cpp — the scheduler and the host API of the SIMT emulator (TEST INFRASTRUCTURE; see hip/hip_runtime.h).
//
// Execution model.  A launch runs its blocks one after another (SIMT_THREADS > 1: on that many host threads, a block per thread at a
// time).  A block's work-items are fibers with stacks of their own; the scheduler runs the lanes of a wavefront in lane order until
// each blocks — at a cross-lane collective, at the block barrier, in s_sleep — or returns.  When no lane of a wavefront can run, the
// lanes waiting at collectives are grouped by CALL SITE; a group is the set of lanes the hardware would have active at that
// instruction (lanes in another branch wait at another site or have left), and completes with exactly those lanes as its EXEC mask:
//   shfl / shfl_xor / up / down   value of the source lane if it is in the group, else the lane's own value
//   ballot                        mask of the group's lanes whose predicate is set
//   readfirstlane                 value of the group's lowest lane
//   readlane                      value of the named lane (0 if it is not in the group)
//   update_dpp                    quad_perm, row_shl/shr, wave_shl/shr, row_mirror, row_half_mirror; source invalid or inactive:
//                                 bound_ctrl ? 0 : old
//   wave_barrier                  nothing: the lanes continue together
// The block barrier opens when every lane that has not returned waits at it.  A block in which nothing can run and nothing resolves is a
// deadlock: the process aborts with the state of every wave.
#include "hip/hip_runtime.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <map>
#include <mutex>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <thread>
#include <vector>

namespace simt {
namespace {

enum { ST_RUN = 0, ST_COLL = 1, ST_BARRIER = 2, ST_DONE = 3, ST_YIELD = 4 };
constexpr size_t STACK_BYTES = 192 * 1024;
constexpr uint32_t MAX_THREADS = 1024;
constexpr size_t DYN_LDS_BYTES = 192 * 1024;

struct fiber {
    void *sp = nullptr;
    lane_ctx ctx{};
    int state = ST_DONE;
    int op = 0;
    const void *site = nullptr;
    uint64_t value = 0, aux = 0, aux2 = 0, result = 0;
};

struct worker {
    char *stacks = nullptr;            // MAX_THREADS stacks, mapped once
    std::vector<fiber> f;
    void *sched_sp = nullptr;
    fiber *cur = nullptr;
    block_ctx blk{};
    char *dyn = nullptr;
    void (*body)(void *) = nullptr;
    void *arg = nullptr;
    uint32_t n = 0, n_waves = 0;
};
thread_local worker *tl_worker = nullptr;
thread_local lane_ctx tl_host_lane{};          // host code that touches threadIdx by mistake reads zeros, not garbage
thread_local block_ctx tl_host_block{};

extern "C" void simt_switch(void **from_sp, void *to_sp);
asm(R"(
    .text
    .globl simt_switch
    .type simt_switch,@function
simt_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size simt_switch,.-simt_switch
)");

void fiber_entry()
{
    worker *w = tl_worker;
    w->body(w->arg);
    fiber *me = w->cur;
    me->state = ST_DONE;
    simt_switch(&me->sp, w->sched_sp);
    __builtin_trap();
}

void to_scheduler(worker *w)
{
    fiber *me = w->cur;
    simt_switch(&me->sp, w->sched_sp);
}

void prepare_fiber(worker *w, uint32_t t)
{
    char *top = w->stacks + (size_t)(t + 1) * STACK_BYTES;
    void **sp = reinterpret_cast<void **>(top);
    *--sp = nullptr;                                   // where fiber_entry's caller's return address would be
    *--sp = reinterpret_cast<void *>(&fiber_entry);    // simt_switch's `ret` lands here: rsp % 16 == 8 at entry, as after a call
    for (int i = 0; i < 6; ++i) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
    w->f[t].sp = sp;
}

const char *op_name(int op)
{
    switch (op) { case OP_SHFL: return "shfl"; case OP_BALLOT: return "ballot"; case OP_READFIRST: return "readfirstlane"; case OP_READLANE: return "readlane";
                  case OP_DPP: return "dpp"; case OP_WAVE_BARRIER: return "wave_barrier"; }
    return "?";
}

// source lane of a DPP control word for lane l (gfx9 encodings); -1: no valid source
int dpp_source(uint32_t ctrl, uint32_t l)
{
    const uint32_t row = l & ~15u, pos = l & 15u;
    if (ctrl <= 0xFF) return (int)((l & ~3u) | ((ctrl >> (2 * (l & 3u))) & 3u));                     // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const uint32_t k = ctrl & 15u; return pos + k <= 15u ? (int)(row | (pos + k)) : -1; }      // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const uint32_t k = ctrl & 15u; return pos >= k ? (int)(row | (pos - k)) : -1; }            // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) { const uint32_t k = ctrl & 15u; return (int)(row | ((pos + 16u - k) & 15u)); }             // row_ror
    if (ctrl == 0x130) return l + 1u < 64u ? (int)(l + 1u) : -1;                                      // wave_shl:1  (lane i reads lane i + 1)
    if (ctrl == 0x134) return (int)((l + 1u) & 63u);                                                  // wave_rol:1
    if (ctrl == 0x138) return l >= 1u ? (int)(l - 1u) : -1;                                           // wave_shr:1  (lane i reads lane i - 1)
    if (ctrl == 0x13C) return (int)((l + 63u) & 63u);                                                 // wave_ror:1
    if (ctrl == 0x140) return (int)(row | (15u - pos));                                               // row_mirror
    if (ctrl == 0x141) return (int)((l & ~7u) | (7u - (l & 7u)));                                     // row_half_mirror
    fprintf(stderr, "simt: DPP control 0x%x is not modelled\n", ctrl);
    abort();
}

void resolve_wave(worker *w, uint32_t wv)
{
    const uint32_t lo = wv * 64, hi = std::min(w->n, lo + 64);
    bool taken[64] = {false};
    for (uint32_t a = lo; a < hi; ++a) {
        fiber &fa = w->f[a];
        if (fa.state != ST_COLL || taken[a - lo]) continue;
        // the group: every waiting lane at the same call site
        uint32_t members[64], nm = 0;
        uint64_t exec = 0;
        for (uint32_t b = a; b < hi; ++b) {
            fiber &fb = w->f[b];
            if (fb.state == ST_COLL && !taken[b - lo] && fb.site == fa.site) {
                if (fb.op != fa.op) { fprintf(stderr, "simt: lanes at one call site with different operations (%s / %s)\n", op_name(fa.op), op_name(fb.op)); abort(); }
                members[nm++] = b; exec |= 1ull << (b - lo); taken[b - lo] = true;
            }
        }
        auto active = [&](int l) { return l >= 0 && l < 64 && ((exec >> l) & 1ull); };
        switch (fa.op) {
        case OP_SHFL:
            for (uint32_t i = 0; i < nm; ++i) { fiber &f = w->f[members[i]]; const int j = (int)f.aux; f.result = active(j) ? w->f[lo + j].value : f.value; }
            break;
        case OP_BALLOT: {
            uint64_t m = 0;
            for (uint32_t i = 0; i < nm; ++i) if (w->f[members[i]].value) m |= 1ull << (members[i] - lo);
            for (uint32_t i = 0; i < nm; ++i) w->f[members[i]].result = m;
            break; }
        case OP_READFIRST:
            for (uint32_t i = 0; i < nm; ++i) w->f[members[i]].result = w->f[members[0]].value;
            break;
        case OP_READLANE:
            for (uint32_t i = 0; i < nm; ++i) { fiber &f = w->f[members[i]]; const int j = (int)(f.aux & 63u); f.result = active(j) ? w->f[lo + j].value : 0; }
            break;
        case OP_DPP:
            for (uint32_t i = 0; i < nm; ++i) {
                fiber &f = w->f[members[i]];
                const bool bound = (f.aux >> 16) & 1u;
                const int j = dpp_source((uint32_t)(f.aux & 0xffffu), members[i] - lo);
                f.result = active(j) ? w->f[lo + j].value : (bound ? 0 : f.aux2);
            }
            break;
        case OP_WAVE_BARRIER:
            break;
        default:
            fprintf(stderr, "simt: unknown collective %d\n", fa.op); abort();
        }
        for (uint32_t i = 0; i < nm; ++i) w->f[members[i]].state = ST_RUN;
    }
}

[[noreturn]] void deadlock(worker *w)
{
    fprintf(stderr, "simt: DEADLOCK in block (%u,%u,%u) of grid (%u,%u,%u), %u threads\n", w->blk.block_idx.x, w->blk.block_idx.y, w->blk.block_idx.z,
            w->blk.grid_dim.x, w->blk.grid_dim.y, w->blk.grid_dim.z, w->n);
    for (uint32_t wv = 0; wv < w->n_waves; ++wv) {
        int cnt[5] = {0};
        for (uint32_t t = wv * 64; t < std::min(w->n, wv * 64 + 64); ++t) cnt[w->f[t].state]++;
        fprintf(stderr, "  wave %u: run %d, collective %d, barrier %d, done %d, yielded %d\n", wv, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4]);
    }
    abort();
}

void run_block(worker *w)
{
    for (uint32_t t = 0; t < w->n; ++t) {
        fiber &f = w->f[t];
        f.state = ST_RUN;
        const uint32_t bx = w->blk.block_dim.x, by = w->blk.block_dim.y;
        f.ctx.tid = uint3{t % bx, (t / bx) % by, t / (bx * by)};
        f.ctx.lane = t & 63u;
        f.ctx.wave = t >> 6;
        prepare_fiber(w, t);
    }
    uint32_t live = w->n;
    while (live) {
        bool progressed = false;
        for (uint32_t wv = 0; wv < w->n_waves; ++wv) {
            const uint32_t lo = wv * 64, hi = std::min(w->n, lo + 64);
            for (;;) {
                bool ran = false;
                for (uint32_t t = lo; t < hi; ++t) {
                    fiber &f = w->f[t];
                    if (f.state != ST_RUN && f.state != ST_YIELD) continue;
                    f.state = ST_RUN;
                    w->cur = &f;
                    simt_switch(&w->sched_sp, f.sp);
                    w->cur = nullptr;
                    ran = true;
                    if (f.state == ST_DONE) --live;
                }
                if (ran) progressed = true;
                bool any_coll = false, any_runnable = false;
                for (uint32_t t = lo; t < hi; ++t) {
                    any_coll |= w->f[t].state == ST_COLL;
                    any_runnable |= w->f[t].state == ST_RUN || w->f[t].state == ST_YIELD;
                }
                if (any_coll && !any_runnable) { resolve_wave(w, wv); progressed = true; continue; }     // the group(s) continue at once
                break;                                                                                    // yielded / at the barrier / done: next wave
            }
        }
        if (!live) break;
        bool all_at_barrier = true, any_barrier = false;
        for (uint32_t t = 0; t < w->n; ++t) {
            if (w->f[t].state == ST_DONE) continue;
            if (w->f[t].state == ST_BARRIER) any_barrier = true; else all_at_barrier = false;
        }
        if (any_barrier && all_at_barrier) {
            for (uint32_t t = 0; t < w->n; ++t) if (w->f[t].state == ST_BARRIER) w->f[t].state = ST_RUN;
            progressed = true;
        }
        if (!progressed) deadlock(w);
    }
}

worker *make_worker()
{
    worker *w = new worker();
    w->stacks = static_cast<char *>(mmap(nullptr, (size_t)MAX_THREADS * STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
    if (w->stacks == MAP_FAILED) { perror("simt: mmap of the fiber stacks"); abort(); }
    w->f.resize(MAX_THREADS);
    w->dyn = static_cast<char *>(aligned_alloc(256, DYN_LDS_BYTES));
    return w;
}

int host_threads()
{
    static const int n = [] { const char *e = getenv("SIMT_THREADS"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 64 ? 64 : v); }();
    return n;
}

std::atomic<unsigned long long> g_launches{0}, g_blocks{0};

}   // namespace

lane_ctx &self() { worker *w = tl_worker; return (w && w->cur) ? w->cur->ctx : tl_host_lane; }
block_ctx &block() { worker *w = tl_worker; return (w && w->cur) ? w->blk : tl_host_block; }

uint64_t collective(int op, uint64_t value, uint64_t aux, uint64_t aux2, const void *site)
{
    worker *w = tl_worker;
    if (!w || !w->cur) { fprintf(stderr, "simt: cross-lane operation outside a kernel\n"); abort(); }
    fiber *me = w->cur;
    me->op = op; me->value = value; me->aux = aux; me->aux2 = aux2; me->site = site; me->state = ST_COLL;
    to_scheduler(w);
    return me->result;
}
void barrier()
{
    worker *w = tl_worker;
    if (!w || !w->cur) { fprintf(stderr, "simt: __syncthreads outside a kernel\n"); abort(); }
    w->cur->state = ST_BARRIER;
    to_scheduler(w);
}

void yield()
{
    worker *w = tl_worker;
    if (!w || !w->cur) return;
    w->cur->state = ST_YIELD;
    to_scheduler(w);
}

void launch(dim3 grid, dim3 blockdim, size_t dyn_lds, void (*body)(void *), void *arg)
{
    const uint64_t nthreads = (uint64_t)blockdim.x * blockdim.y * blockdim.z;
    if (nthreads == 0 || nthreads > MAX_THREADS) { fprintf(stderr, "simt: block of %llu threads\n", (unsigned long long)nthreads); abort(); }
    if (dyn_lds > 160u * 1024u) { fprintf(stderr, "simt: %zu bytes of dynamic LDS requested, a CU has 160 KiB\n", dyn_lds); abort(); }
    const uint64_t nblocks = (uint64_t)grid.x * grid.y * grid.z;
    g_launches++; g_blocks += nblocks;
    if (nblocks == 0) return;
    struct job_t { dim3 grid, blockdim; size_t dyn_lds; void (*body)(void *); void *arg; uint64_t nblocks, nthreads; std::atomic<uint64_t> next{0}; };
    job_t job;
    job.grid = grid; job.blockdim = blockdim; job.dyn_lds = dyn_lds; job.body = body; job.arg = arg; job.nblocks = nblocks; job.nthreads = nthreads;
    auto work = [](job_t *j) {
        static thread_local worker *mine = nullptr;
        if (!mine) mine = make_worker();
        worker *w = mine;
        worker *outer = tl_worker;
        tl_worker = w;
        w->body = j->body; w->arg = j->arg;
        w->n = (uint32_t)j->nthreads; w->n_waves = (uint32_t)((j->nthreads + 63) / 64);
        for (;;) {
            const uint64_t b = j->next.fetch_add(1);
            if (b >= j->nblocks) break;
            w->blk.block_idx = dim3((uint32_t)(b % j->grid.x), (uint32_t)((b / j->grid.x) % j->grid.y), (uint32_t)(b / ((uint64_t)j->grid.x * j->grid.y)));
            w->blk.block_dim = j->blockdim; w->blk.grid_dim = j->grid; w->blk.dyn_lds = w->dyn;
            // LDS is not zero on entry: fill the dynamic window with a pattern, so that a kernel that relies on zeros shows it
            if (j->dyn_lds) memset(w->dyn, 0xA5, j->dyn_lds);
            run_block(w);
        }
        tl_worker = outer;
    };
    const int nt = (int)std::min<uint64_t>((uint64_t)host_threads(), nblocks);
    if (nt <= 1 || tl_worker) { work(&job); return; }
    // a pool of host threads that lives as long as the process (each keeps its fiber stacks): one launch at a time
    struct pool_t {
        std::mutex mu, launch_mu;
        std::condition_variable cv_go, cv_done;
        std::vector<std::thread> threads;
        job_t *job = nullptr;
        void (*fn)(job_t *) = nullptr;
        uint64_t epoch = 0;
        int wanted = 0, running = 0;
    };
    static pool_t *pool = new pool_t();
    std::lock_guard<std::mutex> one_launch(pool->launch_mu);
    {
        std::unique_lock<std::mutex> lk(pool->mu);
        while ((int)pool->threads.size() < nt - 1) {
            pool->threads.emplace_back([] {
                uint64_t seen = 0;
                for (;;) {
                    job_t *j; void (*fn)(job_t *);
                    {
                        std::unique_lock<std::mutex> lk(pool->mu);
                        pool->cv_go.wait(lk, [&] { return pool->epoch != seen && pool->wanted > 0; });
                        seen = pool->epoch; --pool->wanted; j = pool->job; fn = pool->fn;
                    }
                    fn(j);
                    { std::unique_lock<std::mutex> lk(pool->mu); if (--pool->running == 0) pool->cv_done.notify_all(); }
                }
            });
            pool->threads.back().detach();
        }
        pool->job = &job; pool->fn = work; pool->wanted = nt - 1; pool->running = nt - 1; ++pool->epoch;
    }
    pool->cv_go.notify_all();
    work(&job);
    { std::unique_lock<std::mutex> lk(pool->mu); pool->cv_done.wait(lk, [&] { return pool->running == 0; }); }
}

}   // namespace simt

void *simt_dyn_lds() { return simt::block().dyn_lds; }

int __syncthreads_or(int pred)
{
    // two barriers around a flag in "LDS": all lanes contribute, all read
    static thread_local int flag;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) flag = 0;
    __syncthreads();
    if (pred) flag = 1;
    __syncthreads();
    return flag;
}
int __syncthreads_count(int pred)
{
    static thread_local int count;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) count = 0;
    __syncthreads();
    if (pred) ++count;
    __syncthreads();
    return count;
}

unsigned long long wall_clock64()
{
    const auto t = std::chrono::steady_clock::now().time_since_epoch();
    return (unsigned long long)(std::chrono::duration_cast<std::chrono::nanoseconds>(t).count() / 10);      // 100 MHz
}

// ---- host API: device memory is host memory, streams are synchronous ------------------------------------------------------------------
struct simt_stream { int dummy; };
struct simt_event { std::chrono::steady_clock::time_point t; };
namespace {
std::mutex g_mu;
struct alloc_info { size_t bytes, rounded; int fd; };
std::map<void *, alloc_info> g_alloc;
std::map<void *, size_t> g_ipc_open;
size_t g_bytes = 0;
}

extern "C" {
// SIMT_DEVICES=N: pretend N devices (all of them this host) — what lets a rehearsal take the "one GPU per rank" branches of bench.py
static int simt_device_count() { static const int n = [] { const char *e = getenv("SIMT_DEVICES"); const int v = e && *e ? atoi(e) : 1; return v > 0 ? v : 1; }(); return n; }
static thread_local int simt_current_device = 0;
hipError_t hipGetDeviceCount(int *n) { *n = simt_device_count(); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = simt_current_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= simt_device_count()) return hipErrorInvalidValue; simt_current_device = d; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int)
{
    memset(p, 0, sizeof(*p));
    snprintf(p->name, sizeof(p->name), "simt-emulator (host fibers; tests only)");
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950-simt");
    p->totalGlobalMem = (size_t)16 << 30;
    p->sharedMemPerBlock = 64 * 1024; p->sharedMemPerBlockOptin = 160 * 1024; p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
    const char *e = getenv("SIMT_CUS");
    p->multiProcessorCount = e ? atoi(e) : 256;        // as an MI355X: the layout decisions (which kernel a matrix gets) follow the CU count
    p->warpSize = 64; p->maxThreadsPerBlock = 1024; p->clockRate = 2400000; p->l2CacheSize = 4 << 20;
    return hipSuccess;
}
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "error (simt emulator)"; }
// SIMT_IPC=1: "device" allocations are memfd-backed shared mappings, so that hipIpcGetMemHandle / hipIpcOpenMemHandle work between the
// processes of a multi-rank test exactly as the ipc transport uses them (one process per rank, peers' vectors mapped, pulled by copies).
static bool ipc_on() { static const bool on = [] { const char *e = getenv("SIMT_IPC"); return e && *e == '1'; }(); return on; }
hipError_t hipMalloc(void **p, size_t bytes)
{
    const size_t rounded = (std::max<size_t>(bytes, 1) + 4095) & ~(size_t)4095;
    void *q = nullptr;
    int fd = -1;
    if (ipc_on()) {
        fd = memfd_create("simt_device_memory", 0);
        if (fd < 0 || ftruncate(fd, (off_t)rounded) != 0) { if (fd >= 0) close(fd); *p = nullptr; return hipErrorOutOfMemory; }
        q = mmap(nullptr, rounded, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (q == MAP_FAILED) { close(fd); *p = nullptr; return hipErrorOutOfMemory; }
    } else {
#ifdef SIMT_EXACT_ALLOC      // the AddressSanitizer build: the allocation ends where the request ends, so the first byte past a device buffer is poisoned
        if (posix_memalign(&q, 256, std::max<size_t>(bytes, 1)) != 0) q = nullptr;
#else
        q = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
#endif
        if (!q) { *p = nullptr; return hipErrorOutOfMemory; }
    }
    memset(q, 0xCD, bytes);                               // fresh device memory is not zero either
    { std::lock_guard<std::mutex> lk(g_mu); g_alloc[q] = alloc_info{bytes, rounded, fd}; g_bytes += bytes; }
    *p = q;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    alloc_info ai;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_alloc.find(p); if (it == g_alloc.end()) return hipErrorInvalidValue; ai = it->second; g_bytes -= ai.bytes; g_alloc.erase(it); }
    if (ai.fd >= 0) { munmap(p, ai.rounded); close(ai.fd); } else free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { *dev = host; return hipSuccess; }
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind, hipStream_t) { if (bytes) memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemset(void *dst, int value, size_t bytes) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t) { if (bytes) memset(dst, value, bytes); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = new simt_stream(); return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = new simt_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = new simt_event(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new simt_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
// inter-process device memory (SIMT_IPC=1): the handle names the exporter's memfd through /proc/<pid>/fd/<fd>; the opener maps the same pages
struct simt_ipc_handle { char magic[8]; int pid, fd; unsigned long long bytes; };
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_alloc.find(p);                            // the base address of an allocation, as the real runtime demands
    if (it == g_alloc.end() || it->second.fd < 0) return it == g_alloc.end() ? hipErrorInvalidValue : hipErrorNotSupported;
    simt_ipc_handle sh;
    memset(&sh, 0, sizeof(sh));
    memcpy(sh.magic, "SIMTIPC1", 8);
    sh.pid = (int)getpid(); sh.fd = it->second.fd; sh.bytes = it->second.rounded;
    memset(h, 0, sizeof(*h));
    memcpy(h->reserved, &sh, sizeof(sh));
    return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned)
{
    simt_ipc_handle sh;
    memcpy(&sh, h.reserved, sizeof(sh));
    if (memcmp(sh.magic, "SIMTIPC1", 8) != 0) return hipErrorInvalidValue;
    char path[64];
    snprintf(path, sizeof(path), "/proc/%d/fd/%d", sh.pid, sh.fd);
    const int fd = open(path, O_RDWR);
    if (fd < 0) return hipErrorInvalidValue;
    // SIMT_IPC_FAULT=zeros (tests of the library's own checks): the importer gets a private page of zeros instead of the exporter's memory —
    // what a transport that does not deliver looks like; the ipc self-test must notice it on every rank and name it
    static const bool fault_zeros = [] { const char *e = getenv("SIMT_IPC_FAULT"); return e && !strcmp(e, "zeros"); }();
    void *q = fault_zeros ? mmap(nullptr, sh.bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0)
                          : mmap(nullptr, sh.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (q == MAP_FAILED) return hipErrorOutOfMemory;
    { std::lock_guard<std::mutex> lk(g_mu); g_ipc_open[q] = sh.bytes; }
    *p = q;
    return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void *p)
{
    size_t bytes = 0;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_ipc_open.find(p); if (it == g_ipc_open.end()) return hipErrorInvalidValue; bytes = it->second; g_ipc_open.erase(it); }
    munmap(p, bytes);
    return hipSuccess;
}

// SIMT_REPORT=1: one line at exit — the tests assert on it that kernels really ran under the emulator (and not nothing at all)
namespace { struct simt_report { ~simt_report() { if (getenv("SIMT_REPORT")) fprintf(stderr, "simt: %llu launches, %llu blocks executed as host fibers\n", simt::g_launches.load(), simt::g_blocks.load()); } } g_report; }
// what the tests read to see that kernels really ran under the emulator
void simt_counters(unsigned long long *launches, unsigned long long *blocks) { *launches = simt::g_launches.load(); *blocks = simt::g_blocks.load(); }
}
