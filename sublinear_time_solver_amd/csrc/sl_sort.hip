// sl_sort.hip — the only place that uses a library primitive: rocPRIM's radix sort (ROCm's own device primitives; no CUB-compat
// layer in between), for one-off layout construction (transpose, column panels) and — only when a frontier log is requested —
// ordering of the sparse frontier list.  LSD radix sort is stable, so equal keys keep their input order.
#include "sl_internal.hpp"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

sl_status sl_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                            uint64_t n, int end_bit, hipStream_t s)
{
    if (n == 0) return SL_OK;
    if (n > 0x7fffffffull) return sl_fail(SL_ALLOCATION, "sort too large");
    size_t tb = 0;
    SL_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned int)end_bit, s));
    DevBuf tmpbuf;
    SL_TRY(tmpbuf.alloc(tb));
    void *tmp = tmpbuf.p;
    hipError_t e = rocprim::radix_sort_pairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned int)end_bit, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return SL_OK;
}

sl_status sl_sort_keys_u32(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n, hipStream_t s)
{
    if (n == 0) return SL_OK;
    if (n > 0x7fffffffull) return sl_fail(SL_ALLOCATION, "sort too large");
    size_t tb = 0;
    SL_HIP(rocprim::radix_sort_keys(nullptr, tb, keys_in, keys_out, (size_t)n, 0u, 32u, s));
    DevBuf tmpbuf;
    SL_TRY(tmpbuf.alloc(tb));
    void *tmp = tmpbuf.p;
    hipError_t e = rocprim::radix_sort_keys(tmp, tb, keys_in, keys_out, (size_t)n, 0u, 32u, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return SL_OK;
}

// 64-bit keys (block tile << 32 | column: the order-free column stream), 32-bit payload
sl_status sl_sort_pairs_u64(const uint64_t *keys_in, uint64_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, uint64_t n, int end_bit, hipStream_t s)
{
    if (n == 0) return SL_OK;
    if (n > 0x7fffffffull) return sl_fail(SL_ALLOCATION, "sort too large");
    size_t tb = 0;
    SL_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned int)end_bit, s));
    DevBuf tmpbuf;
    SL_TRY(tmpbuf.alloc(tb));
    hipError_t e = rocprim::radix_sort_pairs(tmpbuf.p, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, (unsigned int)end_bit, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return SL_OK;
}
