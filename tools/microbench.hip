// tools/microbench.hip — measured ceilings that bound the push-SpMV kernel on MI355X:
//   copy    : 16 B/lane streaming copy                       -> achievable HBM rate
//   stream  : nt streaming read of a matrix-sized buffer     -> read-only rate
//   gather  : 8-byte random gathers from a table of T bytes, window W (T = whole table = uniform)
//             -> gathers/s: the L2-request / Infinity-Cache bound of the irregular part.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *__restrict__ in, f32x4 *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ __launch_bounds__(256) void stream_kernel(const f32x4 *__restrict__ in, float *sink, size_t n)
{
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc += __builtin_nontemporal_load(&in[i]);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}
__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}
// each lane = one "row" at position i in [0, rows); G gathers per row from window [c-w, c+w] around
// c = i * table / rows (w = 0: whole table); indices computed in registers (no index stream).
template <int G>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ table, uint64_t table_n, uint64_t rows, uint64_t w,
                                                     double *sink)
{
    double acc = 0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < rows; i += (uint64_t)gridDim.x * 256) {
        uint64_t lo = 0, span = table_n;
        if (w) {
            const uint64_t c = (uint64_t)((__uint128_t)i * table_n / rows);
            span = 2 * w + 1; lo = c > w ? c - w : 0; if (lo + span > table_n) lo = table_n - span;
        }
        double v[G];
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = table[lo + mix64(i * 64 + g) % span];
#pragma unroll
        for (int g = 0; g < G; ++g) acc += v[g];
    }
    if (acc == 12345.678) *sink = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main()
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t bytes = 2ull << 30;                       // 2 GiB buffers (past the 256 MiB Infinity Cache)
    f32x4 *a, *b; float *sink; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    const size_t n16 = bytes / 16;
    for (int grid : {2048, 8192, 65536}) {
        copy_kernel<<<grid, 256>>>(a, b, n16); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) copy_kernel<<<grid, 256>>>(a, b, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("copy   grid %6d : %8.1f GB/s (read+write)\n", grid, 2.0 * bytes * 10 / (time_ms(e0, e1) * 1e-3) / 1e9);
        stream_kernel<<<grid, 256>>>(a, sink, n16); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 10; ++r) stream_kernel<<<grid, 256>>>(a, sink, n16); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        printf("stream grid %6d : %8.1f GB/s (nt read)\n", grid, 1.0 * bytes * 10 / (time_ms(e0, e1) * 1e-3) / 1e9);
    }
    // gathers: rows = 10M "rows" x 16 gathers
    const uint64_t rows = 10'000'000;
    double *table = reinterpret_cast<double *>(a), *dsink = reinterpret_cast<double *>(sink);
    struct { uint64_t table_n; uint64_t w; const char *name; } cfg[] = {
        {1'000'000, 0, "table 8 MB uniform"}, {10'000'000, 0, "table 80 MB uniform (C3)"},
        {80'000'000, 0, "table 640 MB uniform (C5 per GPU)"}, {10'000'000, 312'500, "80 MB, window +-312500 (n/32)"},
        {10'000'000, 32'768, "80 MB, window +-32768"}, {10'000'000, 4'096, "80 MB, window +-4096"},
        {10'000'000, 512, "80 MB, window +-512"}, {10'000'000, 64, "80 MB, window +-64"}};
    for (auto &c : cfg) {
        for (int grid : {8192, 39064}) {
            gather_kernel<16><<<grid, 256>>>(table, c.table_n, rows, c.w, dsink); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) gather_kernel<16><<<grid, 256>>>(table, c.table_n, rows, c.w, dsink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            const double ms = time_ms(e0, e1) / 5;
            printf("gather %-36s grid %6d : %7.3f ms per 1.6e8 gathers = %7.2f Ggather/s\n", c.name, grid, ms, 1.6e8 / (ms * 1e-3) / 1e9);
        }
    }
    return 0;
}
