// Does an asynchronous device-to-host copy into PAGEABLE memory, followed by hipStreamSynchronize, ever hand the host stale data when
// several processes share the GPU?  (The layout build's one unexplained failure looked like it: tools/../profiles/r03_fuzz_campaign.json.)
// Each process: kernel writes a fresh pattern (iteration number mixed in) into a device array of a few KB -> hipMemcpyAsync into a
// std::vector -> hipStreamSynchronize -> host checks every word; also the reverse direction (pageable H2D, then a kernel checks).
// usage: d2h_stress <processes> <iterations> <words>        build: hipcc --offload-arch=gfx950 -O2 tools/d2h_stress.hip -o tools/d2h_stress
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>

__global__ void fill(unsigned *p, unsigned n, unsigned it) { for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = it * 2654435761u + i; }
__global__ void check(const unsigned *p, unsigned n, unsigned it, unsigned *bad) { for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) if (p[i] != it * 40503u + i) atomicAdd(bad, 1u); }

static int worker(int id, int iters, unsigned words)
{
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned *d = nullptr, *d2 = nullptr, *dbad = nullptr;
    hipMalloc(&d, words * 4); hipMalloc(&d2, words * 4); hipMalloc(&dbad, 4);
    hipMemset(dbad, 0, 4);
    long stale_d2h = 0;
    for (int it = 1; it <= iters; ++it) {
        std::vector<unsigned> h(words), h2(words);                       // fresh pageable buffers every time, like the build's temporaries
        hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, s, d, words, (unsigned)it);
        hipMemcpyAsync(h.data(), d, words * 4, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
        for (unsigned i = 0; i < words; ++i) if (h[i] != (unsigned)it * 2654435761u + i) { ++stale_d2h; break; }
        for (unsigned i = 0; i < words; ++i) h2[i] = (unsigned)it * 40503u + i;
        hipMemcpyAsync(d2, h2.data(), words * 4, hipMemcpyHostToDevice, s);
        hipLaunchKernelGGL(check, dim3(4), dim3(256), 0, s, d2, words, (unsigned)it, dbad);
        if ((it & 255) == 0) hipStreamSynchronize(s);
        else hipStreamSynchronize(s);                                      // (h2 must outlive the copy)
    }
    unsigned bad = 0;
    hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
    printf("process %d: %d iterations of %u words: stale D2H read-backs %ld, words wrong after pageable H2D %u\n", id, iters, words, stale_d2h, bad);
    return (stale_d2h || bad) ? 1 : 0;
}

int main(int argc, char **argv)
{
    const int procs = argc > 1 ? atoi(argv[1]) : 8, iters = argc > 2 ? atoi(argv[2]) : 20000;
    const unsigned words = argc > 3 ? (unsigned)atoi(argv[3]) : 900;
    std::vector<pid_t> pid(procs);
    for (int p = 0; p < procs; ++p) { pid[p] = fork(); if (pid[p] == 0) return worker(p, iters, words); }
    int rc = 0;
    for (int p = 0; p < procs; ++p) { int st = 0; waitpid(pid[p], &st, 0); rc |= st; }
    return rc ? 1 : 0;
}
