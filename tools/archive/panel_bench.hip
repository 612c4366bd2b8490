// panel_bench.hip — prototype of the column-panel pass for matrices with uniformly random columns (the reference generators'
// recipe): does gathering from an L2-resident panel of the vector, panel after panel, beat 1.6e8 gathers from all over an 80 MB
// vector (58 G/s plateau = 2.74 ms)?  Entries are grouped by (tile of WT rows, panel of PC columns); one wave owns a tile, keeps
// its WT running sums in LDS and walks the panels in order.  Synthetic structure (hashed rows / columns), timing only.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/panel_bench tools/panel_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

// v2: the entries of a tile are ONE stream sorted by (panel, row, column); an entry carries its row inside the tile (u16) and its
// global column (u32) — 14 bytes per entry — so chunks of the stream are always full and no per-panel pointers exist
__global__ void fill_kernel(uint64_t total, uint32_t per_tile, uint32_t wt, uint32_t pc, uint32_t P, uint64_t ncols, uint16_t *rowl, uint32_t *col, double *val, bool lane_private)
{
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (uint64_t)gridDim.x * 256) {
        const uint32_t within = (uint32_t)(k % per_tile);
        const uint32_t panel = (uint32_t)(((uint64_t)within * P) / per_tile);
        const uint32_t seg0 = (uint32_t)(((uint64_t)panel * per_tile + P - 1) / P), seg1 = (uint32_t)(((uint64_t)(panel + 1) * per_tile + P - 1) / P);
        rowl[k] = lane_private ? (uint16_t)((k % 64) + 64 * (mix64(k + 5) % (wt / 64)))                 // every lane only ever meets its own rows
                               : (uint16_t)(((uint64_t)(within - seg0) * wt) / (seg1 - seg0));      // ascending rows inside a panel segment
        uint64_t c = (uint64_t)panel * pc + mix64(k) % pc;
        col[k] = (uint32_t)(c < ncols ? c : ncols - 1);
        val[k] = 1e-3 * (double)(mix64(k + 77) % 1000);
    }
}

template <int NW, int U, int VAR>
__global__ __launch_bounds__(NW * 64) void panel_kernel(uint32_t ntiles, uint32_t per_tile, uint32_t wt, const uint16_t *__restrict__ rowl_a,
                                                        const uint32_t *__restrict__ col_a, const double *__restrict__ val,
                                                        const double *__restrict__ t, const double *__restrict__ dinv, double *__restrict__ out,
                                                        double *__restrict__ x, double *partials, uint32_t *done, uint32_t P, uint32_t target)
{
    extern __shared__ double acc_all[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * NW + wave;
    double *acc = acc_all + (size_t)wave * wt;
    if (tile >= ntiles) return;
    for (uint32_t r = lane; r < wt; r += 64) acc[r] = 0.0;
    const uint64_t s = (uint64_t)tile * per_tile;
    uint32_t cur_panel = 0;
    bool tickets_on = true;
    {   // this wave's round: blocks are dispatched in index order, `target` arrives as the resident tile count
        const uint32_t round = tile / target, r0 = round * target, r1 = min(ntiles, r0 + target);
        target = r0 + (uint32_t)(0.85f * (float)(r1 - r0));
    }
    for (uint32_t c0 = 0; c0 < per_tile; c0 += 64 * U) {
        if (VAR & 4) {
            // tickets (a locality hint, never a correctness condition): a wave enters panel q only when most waves of its round have
            // left panel q - 2; a wait that times out switches the hint off for the rest of the tile
            const uint32_t pnl = (uint32_t)(((uint64_t)c0 * P) / per_tile);
            if (pnl != cur_panel) {
                if (lane == 0) {
                    for (uint32_t q = cur_panel; q < pnl; ++q) atomicAdd(&done[q * 32u], 1u);            // one counter per 128-byte line: same-line atomics queue up
                    if (pnl >= 2 && tickets_on) {
                        uint32_t spins = 0;
                        while (__hip_atomic_load(&done[(pnl - 2) * 32u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < 64) __builtin_amdgcn_s_sleep(8);
                        if (spins >= 64) tickets_on = false;
                    }
                }
                cur_panel = pnl;
            }
        }
        uint32_t rl[U], cl[U]; double v[U], tv[U]; bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t k = c0 + u * 64 + lane;
            ok[u] = true;                                      // per_tile is a multiple of 64 * U here (a real layout pads the stream)
            rl[u] = __builtin_nontemporal_load(&rowl_a[s + k]);
            cl[u] = __builtin_nontemporal_load(&col_a[s + k]);
            v[u] = __builtin_nontemporal_load(&val[s + k]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) tv[u] = t[(VAR & 1) ? (cl[u] & 0x1ffffu) : cl[u]];   // VAR&1: every gather from the first 1 MB
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t rowl = rl[u];
            const double prod = __dmul_rn(v[u], tv[u]);
            const uint32_t prev = __shfl_up(rowl, 1);
            const bool same = ok[u] && lane > 0 && prev == rowl;
            unsigned long long m = __ballot(same);
            if (VAR & 2) { if (ok[u]) acc[lane] = __dadd_rn(acc[lane], prod); }
            else if (!m) { if (ok[u]) acc[rowl] = __dadd_rn(acc[rowl], prod); }
            else {
                const unsigned long long starts = ~m & ((2ull << lane) - 1ull);
                const uint32_t pos = lane - (63u - (uint32_t)__builtin_clzll(starts));
                uint32_t maxpos = pos;
                for (int off = 32; off > 0; off >>= 1) maxpos = max(maxpos, (uint32_t)__shfl_xor(maxpos, off));
                for (uint32_t st = 0; st <= maxpos; ++st) if (ok[u] && pos == st) acc[rowl] = __dadd_rn(acc[rowl], prod);
            }
        }
    }
    double part = 0.0;
    for (uint32_t r = lane; r < wt; r += 64) {
        const uint64_t i = (uint64_t)tile * wt + r;
        const double tn = __dsub_rn(t[i], __dmul_rn(acc[r], dinv[i]));
        __builtin_nontemporal_store(tn, &out[i]);
        __builtin_nontemporal_store(__dadd_rn(x[i], tn), &x[i]);
        part = __dadd_rn(part, __dmul_rn(tn, tn));
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) partials[tile] = part;
}

template <int NW, int U, int VAR = 0>
static int run(uint32_t n, uint32_t k, uint32_t wt, int pcb)
{
    const uint32_t pc = 1u << pcb, P = (n + pc - 1) / pc, ntiles = n / wt;
    const uint32_t per_tile = wt * k;
    uint32_t *done; CK(hipMalloc(&done, (P + 8) * 128));
    const uint32_t target = 256u * (160u * 1024u / (uint32_t)(NW * wt * 8)) * NW;   // resident tiles: CUs x blocks per CU (LDS) x waves
    const uint64_t total = (uint64_t)ntiles * per_tile;
    uint16_t *rowl; uint32_t *col; double *val, *t, *dinv, *out, *x, *partials;
    CK(hipMalloc(&rowl, total * 2)); CK(hipMalloc(&col, total * 4)); CK(hipMalloc(&val, total * 8));
    CK(hipMalloc(&t, ((uint64_t)n + 64) * 8)); CK(hipMalloc(&dinv, (uint64_t)n * 8)); CK(hipMalloc(&out, (uint64_t)n * 8)); CK(hipMalloc(&x, (uint64_t)n * 8));
    CK(hipMalloc(&partials, (uint64_t)ntiles * 8));
    CK(hipMemset(t, 0, ((uint64_t)n + 64) * 8)); CK(hipMemset(dinv, 0, (uint64_t)n * 8)); CK(hipMemset(x, 0, (uint64_t)n * 8));
    fill_kernel<<<4096, 256>>>(total, per_tile, wt, pc, P, n, rowl, col, val, (VAR & 8) != 0);
    CK(hipDeviceSynchronize());
    CK(hipMemset(done, 0, (P + 8) * 128));
    const size_t lds = (size_t)NW * wt * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(panel_kernel<NW, U, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = (ntiles + NW - 1) / NW;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    panel_kernel<NW, U, VAR><<<grid, NW * 64, lds>>>(ntiles, per_tile, wt, rowl, col, val, t, dinv, out, x, partials, done, P, target); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) { CK(hipMemsetAsync(done, 0, (P + 8) * 128)); panel_kernel<NW, U, VAR><<<grid, NW * 64, lds>>>(ntiles, per_tile, wt, rowl, col, val, t, dinv, out, x, partials, done, P, target); }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    const double bytes = 12.0 * n * k + 44.0 * n;
    printf("n=%u k=%u tile=%u rows, panels of %u cols (P=%u) NW=%d U=%d var=%d: %.3f ms -> %.1f %% of 8 TB/s on %.2f GB (%.0f G gathers/s)\n",
           n, k, wt, pc, P, NW, U, VAR, ms, bytes / (ms * 1e-3) / 8e12 * 100, bytes / 1e9, (double)n * k / (ms * 1e-3) / 1e9);
    hipFree(rowl); hipFree(col); hipFree(val); hipFree(t); hipFree(dinv); hipFree(out); hipFree(x); hipFree(partials);
    return 0;
}

int main()
{
    const uint32_t n = 10000000 / 4096 * 4096;
    // var 0 = the real sweep; var 1 = every gather from the first panel (perfect locality: the ceiling of the design);
    // var 4 = tickets per panel (round-aware, bounded wait, self-disabling): 3.5 ms — waiting costs more than drifting;
    // var 8 = rows dealt to lanes (row mod 64 = lane): no two lanes ever meet the same row, LDS updates free of bank conflicts
    run<4, 4, 0>(n, 16, 2048, 16); run<4, 4, 8>(n, 16, 2048, 16); run<4, 4, 1>(n, 16, 2048, 16); run<4, 4, 9>(n, 16, 2048, 16);
    return 0;
}
