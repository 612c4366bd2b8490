// split_bench.hip — would the uniform-column step get faster if the CUs that GATHER never issued a load at HBM latency?   (round 5)
//
// Where sl_pw_kernel's 0.945 ms go (profiles/r03_order_any.txt): 1.6e8 divergent 8-byte gathers (L2 hits) 0.57 ms, the
// 1.92 GB matrix stream 0.25-0.30 ms, and the two ADD — every request of a CU, hit or miss, waits in the same queue of its L1, and a
// stream line at HBM latency holds its place several times longer than a gather that hits the L2 (0.25 / 0.57 of the time for 1.5e7 /
// 1.6e8 of the lines: ~4.7 x per line).  MI355X_MICROARCH.md's "gather-pass" row says the same from the other side: a gather pass
// queued behind its own CU's refill burst takes 2-3 x as long as with that CU's loader quiet.
// Hypothesis: give the HBM latency to CUs that do nothing else.  A few CUs of every XCD ("warmers") read the stream of their XCD's
// gathering CUs a step or two AHEAD of them — plain loads, results discarded — so that the lines are in the XCD's L2 when the
// consumers ask for them; a consumer's stream load then costs what a gather costs (1.5e7 more L2 hits on top of 1.6e8).
// What has to be true for that to pay, each measured here on synthetic traffic of exactly the kernel's shape (256-entry steps of
// 3 KB stream + 4 divergent 8-byte gathers per lane from a 512 KB panel of an 80 MB vector; 1.6e8 entries in all):
//   A  gathers only, all of them carried by 256 / 224 / 192 / 160 / 128 blocks: is the gather rate a property of the CHIP (the L2s'
//      2.7e11 requests/s) or of the CU count?  (chip: the time stays; CU: it grows as 256 / blocks)
//   B  stream only (plain loads, discarded) from 32 / 48 / 64 / 96 blocks: how few CUs carry 1.92 GB in ~0.6 ms (3.2 TB/s)?
//   C  both on every CU as today, but the stream FOLDED into 1 MB per XCD (every stream load an L2 hit) or into 16 MB per XCD
//      (128 MB in all: hits in the memory-side cache, misses in the L2): what a perfectly warmed stream would cost the consumers
//   D  the split itself: per XCD `warm` blocks warm, 32 - warm blocks gather + read the stream; consumers publish the step they are
//      at (one word per block, agent-scope store), a warmer waits while it is more than `lead` steps ahead of the slowest consumer
//      of its XCD (consumers never wait: a late warmer only means the old behaviour).  Block b runs on XCD b % 8.
// Decision rule: D at its best (warm, lead) <= 0.70 ms  =>  build the split into sl_pw_kernel (layout for 32 - warm
// consumer blocks per XCD, a warmer role in the same launch); otherwise record A-D in profiles/ and close the item.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/split_bench tools/split_bench.hip ;  run: tools/split_bench [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define WAVES 16
#define XCDS 8u
#define PANEL_WORDS 65536u          // 512 KB of doubles
#define SLOT 3072u                  // bytes of one wave's step: 256 entries x 12 B
#define MAXB 32u                    // blocks per XCD

struct args_t {
    const double *vec; uint32_t n_panels;
    const char *stream;
    uint32_t steps;                 // steps per gathering wave
    uint32_t fold_bytes;            // C: the stream of an XCD folded into this many bytes (power of two); 0 = not folded
    uint32_t warm, lead;            // D: warmer blocks per XCD, steps a warmer may run ahead of the consumers' own loads
    uint32_t *progress;             // D: [XCDS][MAXB] the step each consumer block is at (+ 1)
    uint64_t total_bytes;           // B: the whole stream
    double *out;
};

__device__ __forceinline__ void gather4(const args_t &a, uint32_t step, uint32_t steps, uint32_t &h, double (&g)[4])
{
    const uint32_t pan = (uint32_t)(((uint64_t)step * a.n_panels) / steps);      // all waves walk the panels together
    const double *base = a.vec + (uint64_t)pan * PANEL_WORDS;
#pragma unroll
    for (int u = 0; u < 4; ++u) { h = h * 1664525u + 1013904223u; g[u] = base[(h >> 8) & (PANEL_WORDS - 1u)]; }
}

// MODE 0: gathers only   1: stream only, discarded (plain loads)   2: both on every CU (stream by non-temporal loads; fold_bytes folds it)
// MODE 3: the split (consumers: gathers + non-temporal stream loads; warmers: plain loads of the same bytes ahead of them)
template <int MODE, bool NT = true>      // NT: the consumers' stream loads carry the non-temporal hint (as sl_pw_kernel's do)
__global__ __launch_bounds__(WAVES * 64) void k(args_t a)
{
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t xcd = blockIdx.x % XCDS, local = blockIdx.x / XCDS, per_xcd = gridDim.x / XCDS;
    const uint32_t ncons = MODE == 3 ? per_xcd - a.warm : per_xcd;             // gathering blocks of this XCD
    const uint32_t cwaves = ncons * WAVES;                                      // gathering waves of this XCD
    // the stream of an XCD: [step][gathering wave][3 KB] — one step of the whole XCD is contiguous (what its warmers read in one go)
    const uint64_t xcd_bytes = (uint64_t)a.steps * cwaves * SLOT;
    const char *xs = a.stream + (uint64_t)xcd * (a.fold_bytes ? a.fold_bytes : xcd_bytes);
    uint32_t h = (uint32_t)((blockIdx.x * WAVES + wave) * 64 + lane) * 2654435761u + 12345u;
    double acc = 0.0;
    if constexpr (MODE == 0) {
        double g0[4], g1[4];
        gather4(a, 0, a.steps, h, g0);
        for (uint32_t s = 0; s < a.steps; ++s) {
            gather4(a, s + 1 < a.steps ? s + 1 : s, a.steps, h, g1);
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc += g0[u]; g0[u] = g1[u]; }
        }
    } else if constexpr (MODE == 1) {
        // every wave of the launch takes 1 KB pieces of the whole stream in turn, 8 in flight, results folded into one word
        const uint64_t pieces = a.total_bytes / 1024u, nw = (uint64_t)gridDim.x * WAVES, me = (uint64_t)blockIdx.x * WAVES + wave;
        uint32_t x = 0;
        for (uint64_t p = me; p < pieces; p += 8 * nw) {
            u32x4 q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint64_t pp = p + (uint64_t)u * nw; q[u] = *(reinterpret_cast<const u32x4 *>(a.stream + (pp < pieces ? pp : p) * 1024u) + lane); }
#pragma unroll
            for (int u = 0; u < 8; ++u) x ^= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
        }
        acc = (double)x;
    } else {
        const bool warmer = MODE == 3 && local >= ncons;
        if (!warmer) {
            const uint32_t cw = local * WAVES + wave;
            const uint32_t fslots = a.fold_bytes ? a.fold_bytes / SLOT : 1u;
            auto ld = [&](uint32_t s, u32x4 &q, f64x2 &va, f64x2 &vb) {
                uint64_t off = ((uint64_t)s * cwaves + cw) * SLOT;
                if (a.fold_bytes) off = ((off / SLOT) % fslots) * SLOT;                  // a whole slot inside the folded region
                const char *p = xs + off;
                if constexpr (NT) {
                    q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p) + lane);
                    va = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + 1024) + lane);
                    vb = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + 2048) + lane);
                } else {
                    q = *(reinterpret_cast<const u32x4 *>(p) + lane);
                    va = *(reinterpret_cast<const f64x2 *>(p + 1024) + lane);
                    vb = *(reinterpret_cast<const f64x2 *>(p + 2048) + lane);
                }
            };
            u32x4 q0, q1, q2; f64x2 a0, a1, a2, b0, b1, b2;
            double g0[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0};
            const uint32_t last = a.steps - 1;
            ld(0, q0, a0, b0); ld(last < 1 ? last : 1, q1, a1, b1);
            gather4(a, 0, a.steps, h, g0);
            for (uint32_t s = 0; s < a.steps; ++s) {
                if (MODE == 3 && wave == 0 && lane == 0)            // "this block is at step s": its loads reach s + 2
                    __hip_atomic_store(&a.progress[xcd * MAXB + local], s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ld(s + 2 < a.steps ? s + 2 : last, q2, a2, b2);
                gather4(a, s + 1 < a.steps ? s + 1 : s, a.steps, h, g1);
                acc += (double)(q0.x ^ q0.y ^ q0.z ^ q0.w) + a0.x * g0[0] + a0.y * g0[1] + b0.x * g0[2] + b0.y * g0[3];
                q0 = q1; a0 = a1; b0 = b1; q1 = q2; a1 = a2; b1 = b2;
#pragma unroll
                for (int u = 0; u < 4; ++u) g0[u] = g1[u];
            }
            if (MODE == 3 && wave == 0 && lane == 0)
                __hip_atomic_store(&a.progress[xcd * MAXB + local], 0xffffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            // warmer wave ww of the XCD's nww: in step s the 1 KB pieces ww, ww + nww, ... of the XCD's step-s block (cwaves x 3 KB, contiguous)
            const uint32_t nww = a.warm * WAVES, ww = (local - ncons) * WAVES + wave;
            const uint32_t pieces = cwaves * 3u;
            uint32_t x = 0, seen = 0, spins = 0;
            bool alive = true;                                                   // the pace is a hint with a bounded wait that switches itself off
            for (uint32_t s = 0; s < a.steps; ++s) {
                // the consumers' own loads are at their step + 2; stay at most `lead` steps beyond that
                while (alive && s > seen + 2u + a.lead) {
                    uint32_t m = lane < ncons ? __hip_atomic_load(&a.progress[xcd * MAXB + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xffffffffu;
#pragma unroll
                    for (int o = 32; o; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)m, o); m = t < m ? t : m; }
                    seen = m ? (m - 1u < 0x3fffffffu ? m - 1u : 0x3fffffffu) : 0u;
                    if (s > seen + 2u + a.lead) { __builtin_amdgcn_s_sleep(8); if (++spins > (1u << 15)) alive = false; }
                }
                const char *blk = xs + (uint64_t)s * cwaves * SLOT;
                for (uint32_t p = ww; p < pieces; p += 4 * nww) {
                    u32x4 q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t pp = p + (uint32_t)u * nww; q[u] = *(reinterpret_cast<const u32x4 *>(blk + (uint64_t)(pp < pieces ? pp : p) * 1024u) + lane); }
#pragma unroll
                    for (int u = 0; u < 4; ++u) x ^= q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
                }
            }
            acc = (double)x;
        }
    }
    if (acc == 1.2345e-300) a.out[0] = acc;
}

template <int MODE, bool NT = true>
static int run(const char *name, args_t a, int blocks, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0;
    for (int i = 0; i < reps + 3; ++i) {
        if (a.progress) CK(hipMemsetAsync(a.progress, 0, XCDS * MAXB * sizeof(uint32_t)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(WAVES * 64), 0, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-112s mean %.3f  best %.3f ms\n", name, sum / reps, best);
    return 0;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int cus = pr.multiProcessorCount;
    if (cus != 256) { printf("written for 256 CUs in 8 XCDs; this device has %d\n", cus); return 1; }
    const uint64_t entries = 160000000ull, vec_words = 10000000ull;
    const uint64_t stream_bytes = entries * 12ull + (64ull << 20);               // slack: the step counts below round up
    double *vec, *out; char *stream; uint32_t *progress;
    CK(hipMalloc(&vec, vec_words * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&stream, stream_bytes)); CK(hipMalloc(&progress, XCDS * MAXB * sizeof(uint32_t)));
    CK(hipMemset(vec, 0, vec_words * 8)); CK(hipMemset(stream, 1, stream_bytes));
    args_t a{};
    a.total_bytes = entries * 12ull;
    a.vec = vec; a.n_panels = (uint32_t)(vec_words / PANEL_WORDS); a.stream = stream; a.out = out;
    auto steps_for = [&](int gathering_blocks) { return (uint32_t)((entries / 256 + (uint64_t)gathering_blocks * WAVES - 1) / ((uint64_t)gathering_blocks * WAVES)); };
    char name[256];
    printf("%d CUs; 1.6e8 entries = 1.92 GB of stream + 1.6e8 gathers from 512 KB panels of an 80 MB vector; ms per launch\n", cus);
    printf("A  gathers only, the same 1.6e8 gathers carried by fewer blocks (one 16-wave block per CU)\n");
    for (int b : {256, 224, 192, 160, 128}) {
        a.steps = steps_for(b);
        snprintf(name, sizeof name, "   %3d blocks x %u steps", b, a.steps);
        if (run<0>(name, a, b, reps)) return 1;
    }
    printf("B  stream only (1.92 GB, plain 16-byte loads, discarded) from few blocks\n");
    a.steps = steps_for(256);
    for (int b : {32, 48, 64, 96, 128, 256}) {
        snprintf(name, sizeof name, "   %3d blocks", b);
        if (run<1>(name, a, b, reps)) return 1;
    }
    printf("C  both on every CU (today's kernel); the stream as it is / folded into the L2 / folded into the memory-side cache\n");
    a.steps = steps_for(256);
    a.fold_bytes = 0;        if (run<2>("   stream from HBM (mode 2 of ldsdma_bench)", a, 256, reps)) return 1;
                             if (run<2, false>("   stream from HBM, plain loads instead of non-temporal ones", a, 256, reps)) return 1;
    a.fold_bytes = 1u << 20; if (run<2>("   stream folded into 1 MB per XCD: every stream load an L2 hit (non-temporal loads)", a, 256, reps)) return 1;
                             if (run<2, false>("   stream folded into 1 MB per XCD, plain loads", a, 256, reps)) return 1;
    a.fold_bytes = 16u << 20; if (run<2>("   stream folded into 16 MB per XCD (128 MB): memory-side cache hits, L2 misses (non-temporal loads)", a, 256, reps)) return 1;
                              if (run<2, false>("   stream folded into 16 MB per XCD, plain loads", a, 256, reps)) return 1;
    a.fold_bytes = 0;
    printf("D  the split: per XCD `warm` blocks read the stream ahead (plain loads), 32 - warm blocks gather and read it again (non-temporal)\n");
    a.progress = progress;
    for (uint32_t warm : {4u, 6u, 8u, 10u, 12u})
        for (uint32_t lead : {1u, 2u, 4u}) {
            a.warm = warm; a.lead = lead; a.steps = steps_for((int)(256 - 8 * warm));
            snprintf(name, sizeof name, "   warm %2u of 32 blocks per XCD, lead %u steps (%u steps per gathering wave, %.2f MB of stream per XCD and step)", warm, lead, a.steps,
                     (32 - warm) * WAVES * SLOT / 1e6);
            if (run<3>(name, a, 256, reps)) return 1;
            if (lead == 2u) {
                snprintf(name, sizeof name, "   warm %2u, lead %u, the consumers' stream loads plain instead of non-temporal", warm, lead);
                if (run<3, false>(name, a, 256, reps)) return 1;
            }
        }
    // control: the split's consumers with NO warmers at work (warm blocks idle): what the smaller number of gathering CUs costs by itself
    a.progress = nullptr;
    printf("E  control: 256 - 8 warm gathering blocks do everything themselves, the other blocks absent\n");
    for (uint32_t warm : {4u, 8u, 12u}) {
        const int b = (int)(256 - 8 * warm);
        a.warm = 0; a.steps = steps_for(b);
        snprintf(name, sizeof name, "   %3d blocks, gathers + stream", b);
        if (run<2>(name, a, b, reps)) return 1;
    }
    return 0;
}
