#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t *out)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t v = 100 + lane;
    out[lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x130, 0xf, 0xf, false);        // wave_shl:1
    out[64 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x138, 0xf, 0xf, false);   // wave_shr:1
    out[128 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x140, 0xf, 0xf, false);  // row_mirror
    out[192 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x141, 0xf, 0xf, false);  // row_half_mirror
    out[256 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    out[320 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    const unsigned long long m = 0xF0F0F0F00F0F0F0Full;
    out[384 + lane] = __builtin_amdgcn_inverse_ballot_w64(m) ? 1u : 0u;
}
int main()
{
    uint32_t *d, h[448];
    hipMalloc(&d, sizeof(h));
    k<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"wave_shl1", "wave_shr1", "row_mirror", "row_half_mirror", "quad 1032", "quad 2301", "inv_ballot"};
    for (int t = 0; t < 7; ++t) { printf("%s:", names[t]); for (int l = 0; l < 64; ++l) printf(" %u", h[t * 64 + l]); printf("\n"); }
    return 0;
}
