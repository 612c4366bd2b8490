// sgather_bench.hip — can the SCALAR memory path gather beside the vector one?  The uniform-column step is bound by the CUs'
// texture-address path: ~2.3 cycles per lane of a divergent 64-lane load (tools/gather_bench.hip: 265-280 G gathers/s from L2-resident
// tables).  Every wave also owns a scalar unit whose loads (s_load_dwordx2: one 8-byte element per instruction, through the scalar data
// cache to the L2) do not pass that path.  This measures what it sustains: each wave issues batches of independent s_load_dwordx2 at
// hashed (wave-uniform) addresses; elements per second for a table that fits the L2s and for the 80 MB vector of the headline run.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/sgather_bench tools/sgather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int BATCH>
__global__ __launch_bounds__(256) void sgather_kernel(const double *__restrict__ t, uint32_t mask, uint32_t iters, unsigned long long *out)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t state = wave * 2654435761u + 12345u;
    unsigned long long acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        unsigned long long v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) {
            state = state * 1664525u + 1013904223u;
            const uint32_t off = ((state >> 4) & mask) * 8u;           // byte offset of a random element
            asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v[b]) : "s"(t), "s"(off));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < BATCH; ++b) { asm volatile("" : "+s"(v[b])); acc ^= v[b]; }
    }
    if ((threadIdx.x & 63) == 0) out[wave] = acc;
}

// the vector path on the same footing: 64 lanes, BATCH divergent loads in flight per lane
template <int BATCH>
__global__ __launch_bounds__(256) void vgather_kernel(const double *__restrict__ t, uint32_t mask, uint32_t iters, unsigned long long *out)
{
    uint32_t state = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    double acc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        double v[BATCH];
#pragma unroll
        for (int b = 0; b < BATCH; ++b) { state = state * 1664525u + 1013904223u; v[b] = t[(state >> 4) & mask]; }
#pragma unroll
        for (int b = 0; b < BATCH; ++b) acc += v[b];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned long long)acc;
}

// both at once: every wave gathers 64 x VB elements by vector loads and SB elements by scalar loads per iteration
// vector loads of an iteration in flight, then NB batches of 14 scalar loads each (the scalar counter holds 15), then the vector results
template <int VB, int NB>
__global__ __launch_bounds__(256) void both_batched_kernel(const double *__restrict__ t, uint32_t mask, uint32_t iters, unsigned long long *out)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t vstate = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, sstate = wave * 40503u + 7u;
    double acc = 0;
    unsigned long long sacc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        double v[VB];
#pragma unroll
        for (int b = 0; b < VB; ++b) { vstate = vstate * 1664525u + 1013904223u; v[b] = t[(vstate >> 4) & mask]; }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            unsigned long long sv[14];
#pragma unroll
            for (int b = 0; b < 14; ++b) {
                sstate = sstate * 1664525u + 1013904223u;
                const uint32_t off = ((sstate >> 4) & mask) * 8u;
                asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(sv[b]) : "s"(t), "s"(off));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < 14; ++b) { asm volatile("" : "+s"(sv[b])); sacc ^= sv[b]; }
        }
#pragma unroll
        for (int b = 0; b < VB; ++b) acc += v[b];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned long long)acc ^ sacc;
}

template <int VB, int SB>
__global__ __launch_bounds__(256) void both_kernel(const double *__restrict__ t, uint32_t mask, uint32_t iters, unsigned long long *out)
{
    const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    uint32_t vstate = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, sstate = wave * 40503u + 7u;
    double acc = 0;
    unsigned long long sacc = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        double v[VB];
        unsigned long long sv[SB];
#pragma unroll
        for (int b = 0; b < VB; ++b) { vstate = vstate * 1664525u + 1013904223u; v[b] = t[(vstate >> 4) & mask]; }
#pragma unroll
        for (int b = 0; b < SB; ++b) {
            sstate = sstate * 1664525u + 1013904223u;
            const uint32_t off = ((sstate >> 4) & mask) * 8u;
            asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(sv[b]) : "s"(t), "s"(off));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int b = 0; b < SB; ++b) { asm volatile("" : "+s"(sv[b])); sacc ^= sv[b]; }
#pragma unroll
        for (int b = 0; b < VB; ++b) acc += v[b];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned long long)acc ^ sacc;
}

template <class K> static int run(const char *name, K kernel, const double *t, uint32_t mask, double elems_per_wave_iter, unsigned long long *out, uint32_t blocks)
{
    const uint32_t iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, t, mask, 200u, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, t, mask, iters, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double waves = (double)blocks * 4.0;
    printf("%-44s table %6.1f MB: %8.3f ms  %7.1f G elements/s\n", name, (mask + 1.0) * 8 / 1e6, ms, waves * iters * elems_per_wave_iter / (ms * 1e-3) / 1e9);
    fflush(stdout);
    return 0;
}

int main()
{
    const uint32_t nmax = 1u << 24;                      // 128 MB
    double *t; unsigned long long *out;
    CK(hipMalloc(&t, (size_t)nmax * 8)); CK(hipMemset(t, 0, (size_t)nmax * 8));
    CK(hipMalloc(&out, (size_t)8192 * 256 * 8));
    for (uint32_t mask : {(1u << 18) - 1u, (1u << 23) - 1u + (1u << 23)}) {       // 2 MB (L2-resident), 128 MB
        for (uint32_t blocks : {256u * 4u, 256u * 8u}) {
            printf("-- %u blocks of 4 waves\n", blocks);
            run("scalar, 4 loads in flight per wave", sgather_kernel<4>, t, mask, 4, out, blocks);
            run("scalar, 8 loads in flight per wave", sgather_kernel<8>, t, mask, 8, out, blocks);
            run("scalar, 14 loads in flight per wave", sgather_kernel<14>, t, mask, 14, out, blocks);
            run("vector, 4 loads in flight per lane", vgather_kernel<4>, t, mask, 4 * 64, out, blocks);
            run("vector 4 per lane + scalar 8 per wave", both_kernel<4, 8>, t, mask, 4 * 64 + 8, out, blocks);
            run("vector 4 per lane + scalar 14 per wave", both_kernel<4, 14>, t, mask, 4 * 64 + 14, out, blocks);
            run("vector 4 per lane + scalar 2 x 14 per wave", both_batched_kernel<4, 2>, t, mask, 4 * 64 + 28, out, blocks);
            run("vector 4 per lane + scalar 4 x 14 per wave", both_batched_kernel<4, 4>, t, mask, 4 * 64 + 56, out, blocks);
            run("vector 8 per lane + scalar 6 x 14 per wave", both_batched_kernel<8, 6>, t, mask, 8 * 64 + 84, out, blocks);
        }
    }
    return 0;
}
