// launch_floor.hip — what a gated-off launch of a push round's train costs: kernel boundary alone, + one dependent load (the gate word as
// a kernel argument's target), + two (argument block -> control block pointer -> gate word: the round-3 shape), at the grids the train uses.
// A local query is ~56 launches of which the device works for microseconds: this is its floor.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/launch_floor tools/launch_floor.hip ;  run: tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct ctl { uint32_t stop, rounds; };
struct io { ctl *c; uint32_t pad[48]; };
__global__ void k_empty() {}
__global__ void k_one(const ctl *c, uint32_t *sink) { if (c->stop) return; if (threadIdx.x == 9999) *sink = 1; }
__global__ void k_two(const io *p, uint32_t *sink) { const io v = *p; if (v.c->stop) return; if (threadIdx.x == 9999) *sink = v.pad[3]; }
__global__ void k_both(const io *p, const ctl *c, uint32_t *sink) { const io v = *p; if (c->stop) return; if (threadIdx.x == 9999) *sink = v.pad[3]; }
template <class F> static float run(F launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}
int main()
{
    ctl *c; io *p; uint32_t *sink;
    CK(hipMalloc(&c, sizeof(ctl))); CK(hipMalloc(&p, sizeof(io))); CK(hipMalloc(&sink, 4));
    ctl hc = {1u, 0u};                                       // stop set: every launch is gated off
    io hp = {};
    hp.c = c;
    CK(hipMemcpy(c, &hc, sizeof(hc), hipMemcpyHostToDevice)); CK(hipMemcpy(p, &hp, sizeof(hp), hipMemcpyHostToDevice));
    const int reps = 2000;
    for (int grid : {1, 128, 512, 1024}) {
        const float a = run([&] { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, 0); }, reps);
        const float b = run([&] { hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, 0, c, sink); }, reps);
        const float d = run([&] { hipLaunchKernelGGL(k_two, dim3(grid), dim3(256), 0, 0, p, sink); }, reps);
        const float e = run([&] { hipLaunchKernelGGL(k_both, dim3(grid), dim3(256), 0, 0, p, c, sink); }, reps);
        printf("grid %4d x 256: empty %.2f us   gate word by argument %.2f us   block -> pointer -> gate word %.2f us   block and gate word side by side %.2f us  (per launch, back to back on one stream)\n",
               grid, a, b, d, e);
    }
    return 0;
}
