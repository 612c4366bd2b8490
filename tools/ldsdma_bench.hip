// ldsdma_bench.hip — does the matrix STREAM of the paced column-panel kernel stop queueing behind the gathers when it travels by LDS-DMA
// (global_load_lds: global -> LDS, no VGPRs held) instead of vector loads into registers?   
//
// The uniform-column step (sl_pw_kernel, n = 10^7 x 16) spends 0.59 ms on 1.6e8 divergent 8-byte gathers (L2 hits, one request each) and
// 0.25 ms on its 1.92 GB stream, and the two ADD (0.77 ms together; round 3) because both wait in the same ~106-request window of the
// CU's L1.  This emulates exactly those two classes of traffic, one persistent 16-wave block per CU, per wave `chunks` steps of 256 entries:
//   gathers : 4 divergent 8-byte loads per lane and step from a 512 KB panel of an 80 MB vector (all waves walk the panels together: L2 hits)
//   stream  : 3 KB per wave and step (the 12-byte entries), read once, far larger than every cache
// modes:  0 gathers only            1 stream only, vector loads (16 B, non-temporal)     2 both, vector loads (what the kernel does today)
//         3 both; every wave brings ITS stream in by global_load_lds into a ring of its own in LDS (2 steps ahead) and reads it back (ds_read)
//         4 both; ONE loader wave per block streams the whole block's bytes by global_load_lds into a 32 KB ring, 15 waves gather
//         5 stream only, one loader wave per block (what that wave sustains alone)         6 stream only, every wave by global_load_lds
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ldsdma_bench tools/ldsdma_bench.hip ;  run: tools/ldsdma_bench [reps]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define WAVES 16
#define PANEL_WORDS 65536u          // 512 KB of doubles
#define RING_SLOT 3072u             // bytes of one wave's step: 256 entries x 12 B

__device__ __forceinline__ void dma16(const void *g, void *lds_wave_base)
{   // 64 lanes x 16 B = 1 KiB from g (per-lane address) to LDS at lds_wave_base + lane * 16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g, (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int MODE>
__global__ __launch_bounds__(WAVES * 64) void k(const double *__restrict__ vec, uint32_t n_panels, const char *__restrict__ stream, uint32_t chunks, double *out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_id = (uint64_t)blockIdx.x * WAVES + wave;
    const char *my = stream + wave_id * (uint64_t)chunks * RING_SLOT;           // this wave's stream, contiguous
    uint32_t h = (uint32_t)(wave_id * 64 + lane) * 2654435761u + 12345u;
    double acc = 0.0;
    constexpr bool GATHER = MODE == 0 || MODE == 2 || MODE == 3 || MODE == 4;
    auto gather4 = [&](uint32_t ch, double (&g)[4]) {
        const uint32_t pan = (uint32_t)(((uint64_t)ch * n_panels) / chunks);
        const double *base = vec + (uint64_t)pan * PANEL_WORDS;
#pragma unroll
        for (int u = 0; u < 4; ++u) { h = h * 1664525u + 1013904223u; g[u] = base[(h >> 8) & (PANEL_WORDS - 1u)]; }
    };
    if constexpr (MODE == 0) {
        double g0[4], g1[4];
        gather4(0, g0);
        for (uint32_t ch = 0; ch < chunks; ++ch) {
            gather4(ch + 1 < chunks ? ch + 1 : ch, g1);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += g0[u];
#pragma unroll
            for (int u = 0; u < 4; ++u) g0[u] = g1[u];
        }
    } else if constexpr (MODE == 1 || MODE == 2) {
        // stream two steps ahead in registers, gathers one step ahead (the kernel's pipeline)
        auto ld = [&](uint32_t ch, u32x4 &q, f64x2 &a, f64x2 &b) {
            const char *p = my + (uint64_t)ch * RING_SLOT;
            q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p) + lane);
            a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + 1024) + lane);
            b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p + 2048) + lane);
        };
        u32x4 q0, q1, q2; f64x2 a0, a1, a2, b0, b1, b2;
        double g0[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0};
        const uint32_t last = chunks - 1;
        ld(0, q0, a0, b0); ld(last < 1 ? last : 1, q1, a1, b1);
        if (GATHER) gather4(0, g0);
        for (uint32_t ch = 0; ch < chunks; ++ch) {
            ld(ch + 2 < chunks ? ch + 2 : last, q2, a2, b2);
            if (GATHER) gather4(ch + 1 < chunks ? ch + 1 : ch, g1);
            acc += (double)(q0.x ^ q0.y ^ q0.z ^ q0.w) + a0.x * g0[0] + a0.y * g0[1] + b0.x * g0[2] + b0.y * g0[3];
            q0 = q1; a0 = a1; b0 = b1; q1 = q2; a1 = a2; b1 = b2;
#pragma unroll
            for (int u = 0; u < 4; ++u) g0[u] = g1[u];
        }
    } else if constexpr (MODE == 3 || MODE == 6) {
        // every wave: its own ring of 3 slots x 3 KB in LDS, filled by LDS-DMA two steps ahead, read back by ds_read
        char *ring = lds + (size_t)wave * 3 * RING_SLOT;
        auto fill = [&](uint32_t ch, uint32_t slot) {
            const char *p = my + (uint64_t)ch * RING_SLOT + lane * 16;
            char *d = ring + slot * RING_SLOT;
            dma16(p, d); dma16(p + 1024, d + 1024); dma16(p + 2048, d + 2048);
        };
        const uint32_t last = chunks - 1;
        fill(0, 0); fill(last < 1 ? last : 1, 1);
        double g0[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0};
        if (GATHER) gather4(0, g0);
        uint32_t slot = 0;
        for (uint32_t ch = 0; ch < chunks; ++ch) {
            const uint32_t s2 = slot + 2 >= 3 ? slot + 2 - 3 : slot + 2;
            fill(ch + 2 < chunks ? ch + 2 : last, s2);
            if (GATHER) gather4(ch + 1 < chunks ? ch + 1 : ch, g1);
            // the oldest fill (3 DMA instructions) must have landed: everything issued after it may still fly: 3 (newest fill) + 4 (gathers) + 3 (middle fill)
            if (GATHER) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            const char *d = ring + slot * RING_SLOT;
            const u32x4 q = *(reinterpret_cast<const u32x4 *>(d) + lane);
            const f64x2 a = *(reinterpret_cast<const f64x2 *>(d + 1024) + lane), b = *(reinterpret_cast<const f64x2 *>(d + 2048) + lane);
            acc += (double)(q.x ^ q.y ^ q.z ^ q.w) + a.x * g0[0] + a.y * g0[1] + b.x * g0[2] + b.y * g0[3];
#pragma unroll
            for (int u = 0; u < 4; ++u) g0[u] = g1[u];
            slot = slot + 1 >= 3 ? 0 : slot + 1;
        }
    } else {        // MODE 4 / 5: wave 0 is the loader of the whole block's stream, the others gather (4) or idle (5)
        if (wave == 0) {
            const char *blk = stream + (uint64_t)blockIdx.x * WAVES * (uint64_t)chunks * RING_SLOT;
            const uint64_t kib = (uint64_t)WAVES * chunks * 3;                   // 1 KiB pieces of the block's stream
            for (uint64_t p = 0; p < kib; ++p) {
                dma16(blk + p * 1024 + lane * 16, lds + (p & 31u) * 1024);       // a 32 KB ring, nobody waits for it: the loader's rate alone
                if ((p & 7u) == 7u) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");   // at most 32 KB in flight
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (MODE == 4) {
            // 15 waves carry the block's 16 x chunks gather steps
            const uint32_t steps = (chunks * WAVES + 14) / 15;
            double g0[4], g1[4];
            gather4(0, g0);
            for (uint32_t s = 0; s < steps; ++s) {
                const uint32_t ch = (uint32_t)(((uint64_t)s * chunks) / steps);
                gather4(ch, g1);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc += g0[u];
#pragma unroll
                for (int u = 0; u < 4; ++u) g0[u] = g1[u];
            }
        }
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

template <int MODE>
static int run(const char *name, const double *vec, uint32_t n_panels, const char *stream, uint32_t chunks, double *out, int blocks, int reps, size_t lds)
{
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(WAVES * 64), lds, 0, vec, n_panels, stream, chunks, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(WAVES * 64), lds, 0, vec, n_panels, stream, chunks, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mode %d  %-78s %.3f ms\n", MODE, name, ms / reps);
    return 0;
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    const int blocks = pr.multiProcessorCount;
    const uint64_t entries = 160000000ull;
    const uint32_t chunks = (uint32_t)(entries / 256 / ((uint64_t)blocks * WAVES));          // steps per wave
    const uint64_t stream_bytes = (uint64_t)blocks * WAVES * chunks * RING_SLOT;
    const uint64_t vec_words = 10000000ull;
    const uint32_t n_panels = (uint32_t)(vec_words / PANEL_WORDS);
    double *vec, *out; char *stream;
    CK(hipMalloc(&vec, vec_words * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&stream, stream_bytes));
    CK(hipMemset(vec, 0, vec_words * 8)); CK(hipMemset(stream, 1, stream_bytes));
    printf("%d CUs, %u steps of 256 entries per wave, stream %.2f GB, vector %.0f MB in %u panels, gathers %.3g\n", blocks, chunks, stream_bytes / 1e9, vec_words * 8 / 1e6,
           n_panels, (double)blocks * WAVES * chunks * 256.0);
    if (run<0>("gathers only (L2 hits, 4 per lane and step)", vec, n_panels, stream, chunks, out, blocks, reps, 0)) return 1;
    if (run<1>("stream only, vector loads into registers", vec, n_panels, stream, chunks, out, blocks, reps, 0)) return 1;
    if (run<2>("both, vector loads (today's kernel)", vec, n_panels, stream, chunks, out, blocks, reps, 0)) return 1;
    if (run<6>("stream only, every wave by global_load_lds into its own ring, read back", vec, n_panels, stream, chunks, out, blocks, reps, WAVES * 3 * RING_SLOT)) return 1;
    if (run<3>("both; every wave's stream by global_load_lds into its own ring, read back", vec, n_panels, stream, chunks, out, blocks, reps, WAVES * 3 * RING_SLOT)) return 1;
    if (run<5>("stream only, ONE loader wave per block (32 KB ring)", vec, n_panels, stream, chunks, out, blocks, reps, 32768)) return 1;
    if (run<4>("both; ONE loader wave per block streams, 15 waves gather", vec, n_panels, stream, chunks, out, blocks, reps, 32768)) return 1;
    return 0;
}
