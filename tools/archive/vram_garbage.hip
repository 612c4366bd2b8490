// Fill (nearly) all free device memory with pseudo-random bits and exit: the next processes' fresh allocations then hold garbage instead of
// whatever the last tenant left (often zeros or well-formed data).  A debugging aid for "relies on fresh memory" faults across processes.
// usage: vram_garbage [fraction of free memory, default 0.9] [seed]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void junk(unsigned long long *p, size_t n, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + i * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}
int main(int argc, char **argv)
{
    const double frac = argc > 1 ? atof(argv[1]) : 0.9;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) return 1;
    size_t left = (size_t)(fr * frac), chunk = 8ull << 30, filled = 0;
    while (left >= (64u << 20)) {
        const size_t b = left < chunk ? left : chunk;
        unsigned long long *p = nullptr;
        if (hipMalloc(&p, b) != hipSuccess) { chunk /= 2; if (chunk < (64u << 20)) break; continue; }
        hipLaunchKernelGGL(junk, dim3(4096), dim3(256), 0, 0, p, b / 8, seed + filled);
        filled += b; left -= b;                                  // (never freed: everything goes back at exit)
    }
    hipDeviceSynchronize();
    printf("filled %.1f GB of %.1f GB free with junk\n", filled / 1e9, fr / 1e9);
    return 0;
}
