// tests/simt/rocprim/device/device_radix_sort.hpp — TEST INFRASTRUCTURE: the two rocPRIM entry points sl_sort.hip uses, on the host, for the
// SIMT-emulated build (tests/simt/hip/hip_runtime.h).  Same contract: first call with a null workspace returns its size; LSD radix sort
// is STABLE on the bits [begin_bit, end_bit) of the key — equal keys keep their input order, which the layout builds rely on.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>
#include <hip/hip_runtime.h>

namespace rocprim {
namespace simt_detail {
template <class K> inline K key_bits(K k, unsigned begin_bit, unsigned end_bit)
{
    const unsigned w = end_bit - begin_bit;
    const K m = w >= sizeof(K) * 8 ? ~(K)0 : (((K)1 << w) - (K)1);
    return (K)(k >> begin_bit) & m;
}
}
template <class K, class V>
inline hipError_t radix_sort_pairs(void *tmp, size_t &tmp_bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n, unsigned begin_bit,
                                   unsigned end_bit, hipStream_t = nullptr)
{
    if (!tmp) { tmp_bytes = 256; return hipSuccess; }
    std::vector<size_t> perm(n);
    std::iota(perm.begin(), perm.end(), (size_t)0);
    std::stable_sort(perm.begin(), perm.end(), [&](size_t a, size_t b) { return simt_detail::key_bits(keys_in[a], begin_bit, end_bit) < simt_detail::key_bits(keys_in[b], begin_bit, end_bit); });
    std::vector<K> ko(n);
    std::vector<V> vo(n);
    for (size_t i = 0; i < n; ++i) { ko[i] = keys_in[perm[i]]; vo[i] = vals_in[perm[i]]; }
    std::copy(ko.begin(), ko.end(), keys_out);
    std::copy(vo.begin(), vo.end(), vals_out);
    return hipSuccess;
}
template <class K>
inline hipError_t radix_sort_keys(void *tmp, size_t &tmp_bytes, const K *keys_in, K *keys_out, size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t = nullptr)
{
    if (!tmp) { tmp_bytes = 256; return hipSuccess; }
    std::vector<K> ko(keys_in, keys_in + n);
    std::stable_sort(ko.begin(), ko.end(), [&](K a, K b) { return simt_detail::key_bits(a, begin_bit, end_bit) < simt_detail::key_bits(b, begin_bit, end_bit); });
    std::copy(ko.begin(), ko.end(), keys_out);
    return hipSuccess;
}
}   // namespace rocprim
