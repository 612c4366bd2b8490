// Which XCD does workgroup b of a launch run on?  The paced layout's XCD-local spans (csrc/sl_kernels.hip, sl_pw_kernel: lblock) assume the
// hardware deals workgroups to the XCDs round robin — block b on XCD b mod 8.  This reads the XCC_ID hardware register in every block of
// launches shaped like the paced kernel's (one 1024-thread block per CU, the CU's whole LDS) and counts the blocks for which that holds.
// build: hipcc --offload-arch=gfx950 -O2 tools/xcd_map_check.hip -o tools/xcd_map_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) who(unsigned *xcc, unsigned *cu)
{
    extern __shared__ double lds[];
    if (threadIdx.x == 0) {
        // s_getreg_b32: id | offset << 6 | (size - 1) << 11;  HW_REG_XCC_ID = 20 (bits 3:0 = XCC), HW_REG_HW_ID1 = 23 (bits 11:8 = CU, 15:13 = SE)
        const unsigned x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        const unsigned h = __builtin_amdgcn_s_getreg(23 | (0 << 6) | (31 << 11));
        xcc[blockIdx.x] = x;
        cu[blockIdx.x] = h;
        lds[0] = (double)x;
    }
    // stay resident for a moment so that all blocks of the launch coexist (one per CU)
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned *dx, *dc;
    hipMalloc(&dx, 4096 * 4); hipMalloc(&dc, 4096 * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(who), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    for (int grid : {cus, cus / 2, 2 * cus}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(who, dim3(grid), dim3(1024), 160 * 1024 - 512, 0, dx, dc);
            std::vector<unsigned> x(grid), c(grid);
            hipMemcpy(x.data(), dx, grid * 4, hipMemcpyDeviceToHost);
            hipMemcpy(c.data(), dc, grid * 4, hipMemcpyDeviceToHost);
            int ok = 0, per[16] = {0};
            for (int b = 0; b < grid; ++b) { ok += (int)(x[b] == (unsigned)(b % 8)); if (x[b] < 16) ++per[x[b]]; }
            printf("grid %4d rep %d: blocks with XCC_ID == b mod 8: %d of %d;  blocks per XCC:", grid, rep, ok, grid);
            for (int k = 0; k < 8; ++k) printf(" %d", per[k]);
            printf("   first 16 XCC ids:");
            for (int b = 0; b < 16 && b < grid; ++b) printf(" %u", x[b]);
            printf("\n");
        }
    }
    return 0;
}
