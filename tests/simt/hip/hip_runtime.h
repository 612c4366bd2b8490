// tests/simt/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product.
//
// A SIMT emulator for the build container, which has no GPU: this header stands where <hip/hip_runtime.h> stands when tests/simt/build.py
// compiles the UNCHANGED kernel sources of sublinear_time_solver_amd/csrc/*.hip as host C++ into tests/simt/_build/libsublinear_hip_simt.so.
// Every work-item of a launch then runs as a fiber on the host: 64 consecutive work-items form a wavefront whose cross-lane operations
// (__shfl*, __ballot, readfirstlane, readlane, DPP, wave barrier) are collectives resolved by the scheduler in simt_rt.cpp, a block's
// fibers share its __shared__ arrays, __syncthreads is a block barrier, s_sleep yields to the other waves of the block.  "Device memory"
// is host memory; streams are in-order and synchronous.
//
// What it is for: executing the device code paths that were written or edited while the GPU pool was closed to this repository
// (rounds 4 and 5) against the CPU oracle, bit for bit — index arithmetic, LDS geometry, pacing, launch trains, epilogues.  What it is
// NOT: a fallback (the product library has none and fails with SL_DEVICE_ERROR without a HIP device), a performance model, or evidence
// about hardware behaviour (memory ordering, occupancy, ISA).  bench.py refuses to run on it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <tuple>
#include <utility>

#define SL_SIMT_EMULATOR 1

// ---- kernel language ------------------------------------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local      // one block at a time per host thread: its fibers share the array, as a workgroup shares LDS
#define __constant__ static

struct dim3 {
    uint32_t x, y, z;
    constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3 { uint32_t x, y, z; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
struct double2 { double x, y; };

namespace simt {
enum coll_op { OP_SHFL = 1, OP_BALLOT, OP_READFIRST, OP_READLANE, OP_DPP, OP_WAVE_BARRIER };
struct lane_ctx {            // what a work-item sees of itself
    uint3 tid;
    uint32_t lane;           // position in the wavefront
    uint32_t wave;           // wavefront in the block
};
struct block_ctx { dim3 block_idx, block_dim, grid_dim; void *dyn_lds; };
lane_ctx &self();
block_ctx &block();
// cross-lane collective: every active lane of the wavefront arrives with (op, value, aux); returns the lane's result
uint64_t collective(int op, uint64_t value, uint64_t aux, uint64_t aux2, const void *site);
void barrier();              // __syncthreads
void yield();                // s_sleep / spinning: let the other waves of the block run
void launch(dim3 grid, dim3 block, size_t dyn_lds, void (*body)(void *), void *arg);
}   // namespace simt

#define threadIdx (::simt::self().tid)
#define blockIdx (::simt::block().block_idx)
#define blockDim (::simt::block().block_dim)
#define gridDim (::simt::block().grid_dim)
static const int warpSize = 64;

// ---- runtime API (host) -----------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorNotSupported = 801, hipErrorUnknown = 999 };
typedef struct simt_stream *hipStream_t;
typedef struct simt_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostRegisterPortable = 1, hipHostRegisterMapped = 2,
       hipIpcMemLazyEnablePeerAccess = 1 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipIpcMemHandle_t { char reserved[64]; };
struct hipDeviceProp_t {
    char name[256];
    size_t totalGlobalMem, sharedMemPerBlock, sharedMemPerBlockOptin, maxSharedMemoryPerMultiProcessor;
    int multiProcessorCount, warpSize, maxThreadsPerBlock, clockRate, l2CacheSize;
    char gcnArchName[256];
};

extern "C" {
hipError_t hipGetDeviceCount(int *n);
hipError_t hipGetDevice(int *d);
hipError_t hipSetDevice(int d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int d);
hipError_t hipDeviceSynchronize(void);
hipError_t hipGetLastError(void);
const char *hipGetErrorString(hipError_t e);
hipError_t hipMalloc(void **p, size_t bytes);
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags);
hipError_t hipHostFree(void *p);
hipError_t hipHostRegister(void *p, size_t bytes, unsigned flags);
hipError_t hipHostUnregister(void *p);
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned flags);
hipError_t hipMemcpy(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipMemset(void *dst, int value, size_t bytes);
hipError_t hipMemsetAsync(void *dst, int value, size_t bytes, hipStream_t s);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned flags, int priority);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p);
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned flags);
hipError_t hipIpcCloseMemHandle(void *p);
}
// C++ conveniences of the real header
template <class T> inline hipError_t hipMalloc(T **p, size_t bytes) { return hipMalloc(reinterpret_cast<void **>(p), bytes); }
template <class T> inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned flags = 0) { return hipHostMalloc(reinterpret_cast<void **>(p), bytes, flags); }
inline hipError_t hipHostMalloc(void **p, size_t bytes) { return hipHostMalloc(p, bytes, 0u); }
inline hipError_t hipEventRecord(hipEvent_t e) { return hipEventRecord(e, nullptr); }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
template <class K> inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, K, int block, size_t lds)
{
    const size_t by_lds = lds ? (160u * 1024u) / lds : 8u;
    const size_t by_threads = block > 0 ? 2048u / (size_t)block : 8u;
    *n = (int)std::max<size_t>(1, std::min<size_t>({by_lds, by_threads, 8u}));
    return hipSuccess;
}

namespace simt {
template <class K, class Tup> struct launch_pack { K k; Tup args; };
template <class K, class Tup> void launch_thunk(void *p)
{
    auto *lp = static_cast<launch_pack<K, Tup> *>(p);
    std::apply(lp->k, lp->args);
}
template <class K, class... A> void launch_kernel(K k, dim3 grid, dim3 block, size_t lds, hipStream_t, A &&...a)
{
    using Tup = std::tuple<std::decay_t<A>...>;
    launch_pack<K, Tup> lp{k, Tup(std::forward<A>(a)...)};
    launch(grid, block, lds, &launch_thunk<K, Tup>, &lp);
}
}   // namespace simt
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) ::simt::launch_kernel(kernel, dim3(grid), dim3(block), (size_t)(lds), (stream), ##__VA_ARGS__)

using std::isfinite; using std::isnan; using std::isinf;

// ---- device intrinsics -------------------------------------------------------------------------------------------------------------
// A collective's identity is its SOURCE position (file, line, column of the call in the kernel), taken by defaulted arguments.  Not the return
// address: the optimiser duplicates call sites (jump threading turns `p = c ? f(x) : 0; ballot(p != 0)` into two calls, one per arm of a
// per-lane condition), and the lanes of one hardware instruction would then wait at two different addresses.
#define SIMT_AT int _sl = __builtin_LINE(), int _sc = __builtin_COLUMN(), const char *_sf = __builtin_FILE()
#define SIMT_SITE ((const void *)((uintptr_t)_sf * 1000003u + (uintptr_t)_sl * 4099u + (uintptr_t)_sc))
namespace simt {
template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "cross-lane values are at most 64 bits"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
inline uint64_t coll_site(int op, uint64_t value, uint64_t aux, uint64_t aux2, const void *site) { return collective(op, value, aux, aux2, site); }
}
template <class T> __forceinline__ T __shfl(T v, int src, int width = 64, SIMT_AT)
{
    const uint32_t l = ::simt::self().lane;
    const uint32_t j = (l & ~(uint32_t)(width - 1)) | ((uint32_t)src & (uint32_t)(width - 1));
    return ::simt::from_bits<T>(::simt::coll_site(::simt::OP_SHFL, ::simt::to_bits(v), j, 0, SIMT_SITE));
}
template <class T> __forceinline__ T __shfl_xor(T v, int mask, int width = 64, SIMT_AT)
{
    const uint32_t l = ::simt::self().lane;
    uint32_t j = l ^ (uint32_t)mask;
    if ((j & ~(uint32_t)(width - 1)) != (l & ~(uint32_t)(width - 1))) j = l;
    return ::simt::from_bits<T>(::simt::coll_site(::simt::OP_SHFL, ::simt::to_bits(v), j, 0, SIMT_SITE));
}
template <class T> __forceinline__ T __shfl_up(T v, unsigned delta, int width = 64, SIMT_AT)
{
    const uint32_t l = ::simt::self().lane;
    const uint32_t base = l & ~(uint32_t)(width - 1);
    const uint32_t j = (l - base) >= delta ? l - delta : l;
    return ::simt::from_bits<T>(::simt::coll_site(::simt::OP_SHFL, ::simt::to_bits(v), j, 0, SIMT_SITE));
}
template <class T> __forceinline__ T __shfl_down(T v, unsigned delta, int width = 64, SIMT_AT)
{
    const uint32_t l = ::simt::self().lane;
    const uint32_t base = l & ~(uint32_t)(width - 1);
    const uint32_t j = (l - base) + delta < (uint32_t)width ? l + delta : l;
    return ::simt::from_bits<T>(::simt::coll_site(::simt::OP_SHFL, ::simt::to_bits(v), j, 0, SIMT_SITE));
}
__forceinline__ unsigned long long __ballot(int pred, SIMT_AT) { return ::simt::coll_site(::simt::OP_BALLOT, pred ? 1u : 0u, 0, 0, SIMT_SITE); }
__forceinline__ void __syncthreads() { ::simt::barrier(); }
int __syncthreads_or(int pred);
int __syncthreads_count(int pred);
__forceinline__ int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
__forceinline__ int __popc(unsigned v) { return __builtin_popcount(v); }
__forceinline__ int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
__forceinline__ int __ffs(unsigned v) { return __builtin_ffs((int)v); }
__forceinline__ int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
__forceinline__ int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
__forceinline__ void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
__forceinline__ void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
__forceinline__ void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
// IEEE operations with round-to-nearest, never contracted (the emulated build is compiled with -ffp-contract=off as the device build is)
__forceinline__ double __dmul_rn(double a, double b) { return a * b; }
__forceinline__ double __dadd_rn(double a, double b) { return a + b; }
__forceinline__ double __dsub_rn(double a, double b) { return a - b; }
__forceinline__ double __ddiv_rn(double a, double b) { return a / b; }
__forceinline__ long long __double_as_longlong(double v) { return ::simt::from_bits<long long>(::simt::to_bits(v)); }
__forceinline__ double __longlong_as_double(long long v) { return ::simt::from_bits<double>(::simt::to_bits(v)); }
unsigned long long wall_clock64();          // 100 MHz, as on the device
__forceinline__ unsigned long long clock64() { return wall_clock64(); }

// amdgcn builtins the kernels call directly (clang only knows them for the amdgcn target): build.py rewrites `simt_amdgcn_` to `simt_amdgcn_`
__forceinline__ int simt_amdgcn_readfirstlane(int v, SIMT_AT) { return (int)::simt::coll_site(::simt::OP_READFIRST, (uint32_t)v, 0, 0, SIMT_SITE); }
__forceinline__ int simt_amdgcn_readlane(int v, int l, SIMT_AT) { return (int)::simt::coll_site(::simt::OP_READLANE, (uint32_t)v, (uint32_t)l, 0, SIMT_SITE); }
__forceinline__ int simt_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, SIMT_AT)
{
    (void)row_mask; (void)bank_mask;      // every site passes 0xf, 0xf
    return (int)::simt::coll_site(::simt::OP_DPP, (uint32_t)src, (uint32_t)ctrl | (bound_ctrl ? 0x10000u : 0u), (uint32_t)old, SIMT_SITE);
}
__forceinline__ bool simt_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> ::simt::self().lane) & 1ull; }
__forceinline__ uint32_t simt_amdgcn_mbcnt_lo(uint32_t m, uint32_t add)
{
    const uint32_t l = ::simt::self().lane;
    return add + (uint32_t)__builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u)));
}
__forceinline__ uint32_t simt_amdgcn_mbcnt_hi(uint32_t m, uint32_t add)
{
    const uint32_t l = ::simt::self().lane;
    return add + (l <= 32 ? 0u : (uint32_t)__builtin_popcount(m & ((1u << (l - 32)) - 1u)));
}
__forceinline__ void simt_amdgcn_wave_barrier(SIMT_AT) { (void)::simt::coll_site(::simt::OP_WAVE_BARRIER, 0, 0, 0, SIMT_SITE); }
__forceinline__ void simt_amdgcn_s_sleep(int) { ::simt::yield(); }
#define simt_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
// buffer_load ... lds: every lane moves `size` bytes from ITS global address to LDS base + lane * size + offset (M0 = the wave-uniform base)
__forceinline__ void simt_amdgcn_global_load_lds(const void *src, void *lds_base, unsigned size, unsigned offset, unsigned)
{
    memcpy(static_cast<char *>(lds_base) + (size_t)::simt::self().lane * size + offset, src, size);
}

// atomics: blocks may run on several host threads, so these are real atomics
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
// (__hip_atomic_load / _store / _fetch_add / _compare_exchange_strong are clang builtins on every target: used as they are)
template <class T> __forceinline__ T simt_atomic_rmw_add(T *p, T v)
{
    if constexpr (std::is_floating_point<T>::value) {
        T old, des;
        __atomic_load(p, &old, __ATOMIC_RELAXED);
        do { des = old + v; } while (!__atomic_compare_exchange(p, &old, &des, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
        return old;
    } else return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST);
}
template <class T, class V> __forceinline__ T atomicAdd(T *p, V v) { return simt_atomic_rmw_add(p, (T)v); }
template <class T, class V> __forceinline__ T atomicOr(T *p, V v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> __forceinline__ T atomicAnd(T *p, V v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class V> __forceinline__ T atomicExch(T *p, V v) { T val = (T)v, old; __atomic_exchange(p, &val, &old, __ATOMIC_SEQ_CST); return old; }
template <class T, class V> __forceinline__ T atomicMin(T *p, V v)
{
    T old, des;
    __atomic_load(p, &old, __ATOMIC_RELAXED);
    do { des = std::min<T>(old, (T)v); } while (!__atomic_compare_exchange(p, &old, &des, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
    return old;
}
template <class T, class V> __forceinline__ T atomicMax(T *p, V v)
{
    T old, des;
    __atomic_load(p, &old, __ATOMIC_RELAXED);
    do { des = std::max<T>(old, (T)v); } while (!__atomic_compare_exchange(p, &old, &des, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED));
    return old;
}
template <class T, class V, class W> __forceinline__ T atomicCAS(T *p, V cmp, W val)
{
    T expected = (T)cmp, desired = (T)val;
    __atomic_compare_exchange(p, &expected, &desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return expected;
}

// min / max of the device library, in the global namespace as HIP has them
#define SIMT_MINMAX(T) __forceinline__ T min(T a, T b) { return a < b ? a : b; } __forceinline__ T max(T a, T b) { return a > b ? a : b; }
SIMT_MINMAX(int) SIMT_MINMAX(unsigned) SIMT_MINMAX(long) SIMT_MINMAX(unsigned long) SIMT_MINMAX(long long) SIMT_MINMAX(unsigned long long)
SIMT_MINMAX(float) SIMT_MINMAX(double)
#undef SIMT_MINMAX
__forceinline__ unsigned long min(unsigned long a, unsigned b) { return a < b ? a : b; }
__forceinline__ unsigned long min(unsigned a, unsigned long b) { return a < b ? a : b; }
__forceinline__ unsigned long max(unsigned long a, unsigned b) { return a > b ? a : b; }
__forceinline__ unsigned long max(unsigned a, unsigned long b) { return a > b ? a : b; }
