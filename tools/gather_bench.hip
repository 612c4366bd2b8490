// gather_bench.hip — how fast can gfx950 serve random 8-byte gathers from an 80 MB vector (uniform columns, n = 1e7)?
// Variants: plain loads vs non-temporal loads, gathers in flight per lane, block size / occupancy, table size.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/gather_bench tools/gather_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

template <int G, int MODE, int BLOCK>   // MODE 0 plain, 1 nontemporal, 2 volatile-ish (glc via atomic load relaxed agent)
__global__ __launch_bounds__(BLOCK) void gather_kernel(const double *__restrict__ table, uint64_t table_n, uint64_t rows, double *sink)
{
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * BLOCK) {
        double v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const double *p = table + mix64(i * 64 + g) % table_n;
            if (MODE == 0) v[g] = *p;
            else if (MODE == 1) v[g] = __builtin_nontemporal_load(p);
            else v[g] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) acc += v[g];
    }
    if (acc == 12345.678) *sink = acc;
}

template <int G, int MODE, int BLOCK>
static int run(const char *name, const double *table, uint64_t table_n, double *sink, int grid)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint64_t rows = 160000000ull / G;
    gather_kernel<G, MODE, BLOCK><<<grid, BLOCK>>>(table, table_n, rows, sink); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) gather_kernel<G, MODE, BLOCK><<<grid, BLOCK>>>(table, table_n, rows, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("%-28s table %5.0f MB G=%2d block %4d grid %6d : %7.3f ms per 1.6e8 gathers = %7.2f G/s\n", name, table_n * 8 / 1e6, G, BLOCK, grid, ms,
           1.6e8 / (ms * 1e-3) / 1e9);
    return 0;
}

int main(int argc, char **argv)
{
    double *table, *sink;
    const uint64_t big = 1ull << 27;                 // 1 GiB
    CK(hipMalloc(&table, big * 8)); CK(hipMalloc(&sink, 64)); CK(hipMemset(table, 0, big * 8));
    if (argc > 1 && argv[1][0] == 's') {             // small tables: how fast are gathers that stay inside an XCD's L2 (4 MB)?
        for (uint64_t kb : {64ull, 256ull, 512ull, 1024ull, 2048ull, 4096ull, 16384ull, 32768ull}) {
            run<16, 0, 256>("plain", table, kb * 128, sink, 16384);
            run<32, 0, 256>("plain", table, kb * 128, sink, 16384);
        }
        return 0;
    }
    for (uint64_t tn : {10000000ull, 1000000ull, 100000000ull}) {
        for (int grid : {4096, 16384, 65536}) {
            run<16, 0, 256>("plain", table, tn, sink, grid);
            run<16, 1, 256>("nontemporal", table, tn, sink, grid);
        }
        run<16, 2, 256>("atomic-load(agent)", table, tn, sink, 16384);
        run<4, 0, 256>("plain", table, tn, sink, 65536);
        run<32, 0, 256>("plain", table, tn, sink, 16384);
        run<32, 1, 256>("nontemporal", table, tn, sink, 16384);
        run<16, 0, 1024>("plain", table, tn, sink, 4096);
        run<16, 0, 64>("plain", table, tn, sink, 65536);
    }
    return 0;
}
