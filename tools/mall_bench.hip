// mall_bench.hip — does the Infinity Cache (256 MB, memory side) keep what a kernel WROTE for the next kernel to read?
// Decides whether products routed through a 128 MB buffer (write in one launch, read in the next) cost HBM bandwidth or not.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mall_bench tools/mall_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double f64x2 __attribute__((ext_vector_type(2)));

__global__ void write_k(f64x2 *p, size_t n2, double v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { f64x2 w; w.x = v; w.y = v + 1; p[i] = w; } }
__global__ void read_k(const f64x2 *p, size_t n2, double *out) { double acc = 0; for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { f64x2 w = __builtin_nontemporal_load(&p[i]); acc += w.x + w.y; } if (acc == 1.2345) out[0] = acc; }
__global__ void copy_k(const f64x2 *a, f64x2 *b, size_t n2) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) b[i] = __builtin_nontemporal_load(&a[i]); }

int main()
{
    const size_t big = (size_t)2 << 30;
    f64x2 *thrash, *buf; double *out;
    CK(hipMalloc(&thrash, big)); CK(hipMalloc(&buf, (size_t)1 << 30)); CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mb : {32, 64, 128, 192, 256, 512}) {
        const size_t bytes = mb << 20, n2 = bytes / 16;
        float t_cold = 0, t_warm = 0, t_afterw = 0, t_w = 0, t_wrep = 0;
        for (int rep = 0; rep < 3; ++rep) {
            // cold read: thrash the caches with a 2 GB read first
            read_k<<<4096, 256>>>(thrash, big / 16, out);
            CK(hipEventRecord(e0)); read_k<<<4096, 256>>>(buf, n2, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_cold, e0, e1));
            // warm read: the same buffer again right away (read-allocated in the Infinity Cache?)
            CK(hipEventRecord(e0)); read_k<<<4096, 256>>>(buf, n2, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_warm, e0, e1));
            // write, then read in the next launch
            read_k<<<4096, 256>>>(thrash, big / 16, out);
            CK(hipEventRecord(e0)); write_k<<<4096, 256>>>(buf, n2, 1.0); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_w, e0, e1));
            CK(hipEventRecord(e0)); read_k<<<4096, 256>>>(buf, n2, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_afterw, e0, e1));
            // the same buffer written again and again (do writes die in the cache?)
            CK(hipEventRecord(e0)); for (int k = 0; k < 8; ++k) write_k<<<4096, 256>>>(buf, n2, 2.0 + k); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&t_wrep, e0, e1));
        }
        auto gbs = [&](float ms, double b) { return b / (ms * 1e-3) / 1e9; };
        printf("%4zu MB: cold read %6.0f GB/s | re-read %6.0f GB/s | write %6.0f GB/s | read after write %6.0f GB/s | 8 rewrites %6.0f GB/s\n", mb,
               gbs(t_cold, bytes), gbs(t_warm, bytes), gbs(t_w, bytes), gbs(t_afterw, bytes), gbs(t_wrep, 8.0 * bytes));
    }
    return 0;
}
