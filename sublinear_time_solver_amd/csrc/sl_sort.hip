// sl_sort.hip — the only place that uses a library primitive (rocPRIM radix sort via hipCUB):
// one-off transpose construction and (only when a frontier log is requested) ordering of the
// sparse frontier list.  LSD radix sort is stable, so equal keys keep their input order.
#include "sl_internal.hpp"
#include <hipcub/hipcub.hpp>

sl_status sl_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out,
                            uint64_t n, int end_bit, hipStream_t s)
{
    if (n == 0) return SL_OK;
    if (n > 0x7fffffffull) return sl_fail(SL_ALLOCATION, "sort too large");
    size_t tb = 0;
    SL_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, s));
    DevBuf tmpbuf;
    SL_TRY(tmpbuf.alloc(tb));
    void *tmp = tmpbuf.p;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tb, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return SL_OK;
}

sl_status sl_sort_keys_u32(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n, hipStream_t s)
{
    if (n == 0) return SL_OK;
    if (n > 0x7fffffffull) return sl_fail(SL_ALLOCATION, "sort too large");
    size_t tb = 0;
    SL_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, keys_in, keys_out, (int)n, 0, 32, s));
    DevBuf tmpbuf;
    SL_TRY(tmpbuf.alloc(tb));
    void *tmp = tmpbuf.p;
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(tmp, tb, keys_in, keys_out, (int)n, 0, 32, s);
    hipError_t e2 = hipStreamSynchronize(s);
    if (e != hipSuccess || e2 != hipSuccess) return sl_fail(SL_DEVICE_ERROR, "radix sort failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return SL_OK;
}
