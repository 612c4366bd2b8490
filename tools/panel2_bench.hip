// panel2_bench.hip — prototype of the PACED column-panel pass (round 2) for uniformly random columns (SURVEY §8(d)'s S-DD).
//
// Round 1's panel kernel (sl_panel_kernel): a wave owns a tile of 2048 rows (running sums in LDS) and walks its stream sorted by
// (panel of 2^16 columns, row, column); gathers hit the L2 only while the waves of an XCD stay on the same few panels — nothing
// kept them there (L2 hit rate 46 %, 1.39 ms against 0.95 ms with perfect locality).  This prototype measures, on a synthetic
// stream of the same shape, what keeps them together:
//   * ONE persistent block per CU (16 waves, the whole LDS as running sums), rounds of resident tiles;
//   * pacing inside the block through LDS progress words; pacing across the CUs of an XCD through one 128-byte line of
//     progress words that lives in that XCD's L2 (plain store + L1-bypassing load: no fabric round trip, no atomics);
//     both are hints with a bounded wait — never a correctness condition;
//   * 12 bytes per entry: u32 {row in tile : 11 | super-panel step : 1 | column low 20 bits} + f64 value;
//   * software pipelining: stream loads two chunks ahead, gathers one chunk ahead of the LDS updates.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/panel2_bench tools/panel2_bench.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __host__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define SP_BITS 20            // super-panel = 2^20 columns: what the 20 column bits of an entry are relative to
struct ctl_block {
    uint32_t slots[8 * 32];   // [xcd] registration counters, one per 128-byte line
    uint32_t prog[8 * 64];    // [xcd][up to 64 blocks]: progress words of the blocks of an XCD, two 128-byte lines per XCD
};

// synthetic stream of one tile: per_tile entries, panel = position-proportional, rows ascending inside a panel segment
// Memory layout of a chunk of 256 entries (pre-transposed so that 16-byte loads hand lane l the entries l, l + 64, l + 128, l + 192):
//   idx [chunk][lane][4] u32          val [chunk][half][lane][2] f64   (entry u * 64 + lane sits at idx[lane][u], val[u / 2][lane][u % 2])
__global__ void fill_kernel(uint64_t total, uint32_t per_tile, uint32_t wt, uint32_t pcb, uint32_t P, uint64_t ncols, uint32_t *idx, double *val, int dup)
{
    for (uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x; k < total; k += (uint64_t)gridDim.x * 256) {
        const uint32_t within = (uint32_t)(k % per_tile);
        const uint32_t panel = (uint32_t)(((uint64_t)within * P) / per_tile);
        const uint32_t seg0 = (uint32_t)(((uint64_t)panel * per_tile + P - 1) / P), seg1 = (uint32_t)(((uint64_t)(panel + 1) * per_tile + P - 1) / P);
        uint32_t row = (uint32_t)(((uint64_t)(within - seg0) * wt) / (seg1 - seg0));
        if (dup && (within - seg0) % 3 == 1) row = (uint32_t)(((uint64_t)(within - 1 - seg0) * wt) / (seg1 - seg0));   // every third entry repeats its left neighbour's row
        uint64_t c = ((uint64_t)panel << pcb) + mix64(k) % (1ull << pcb);
        if (c >= ncols) c = ncols - 1;
        // super-panel step flag: set on the first entry of the tile's stream that lies in a new super-panel
        uint32_t flag = 0;
        if (within > 0) {
            const uint32_t pprev = (uint32_t)(((uint64_t)(within - 1) * P) / per_tile);
            uint64_t cprev_sp = ((uint64_t)pprev << pcb) >> SP_BITS;
            flag = ((c >> SP_BITS) != cprev_sp) ? 1u : 0u;
        }
        const uint64_t chunk0 = k & ~255ull;
        const uint32_t e = (uint32_t)(k & 255u), u = e >> 6, l = e & 63u;
        idx[chunk0 + l * 4 + u] = (row << 21) | (flag << 20) | (uint32_t)(c & ((1u << SP_BITS) - 1u));
        val[chunk0 + (u >> 1) * 128 + l * 2 + (u & 1u)] = 1e-3 * (double)(mix64(k + 77) % 1000);
    }
}

// VAR bits: 1 = every gather from the first panel (perfect locality ceiling); 2 = no LDS update (gather + stream only);
//           4 = pace inside the block; 8 = pace across the XCD; 16 = no gathers at all (stream + LDS only); 32 = gathers from 8 KB (L1 hits)
//           64 = no stream loads (indices synthesised in registers)
template <int NW, int U, int VAR>
__global__ __launch_bounds__(NW * 64) void panel2_kernel(uint32_t ntiles, uint32_t per_tile, uint32_t wt, uint32_t pcb, uint32_t P,
                                                         const uint32_t *__restrict__ idx_a, const double *__restrict__ val_a,
                                                         const double *__restrict__ t, const double *__restrict__ dinv, double *__restrict__ out,
                                                         double *__restrict__ x, double *partials, ctl_block *ctl, uint32_t slack_b, uint32_t slack_x,
                                                         uint32_t gen)
{
    extern __shared__ double acc_all[];
    __shared__ uint32_t prog[NW];
    __shared__ uint32_t xinfo[4];     // [0] xcd, [1] slot, [2] cached XCD minimum, [3] pacing alive
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    double *acc = acc_all + (size_t)wave * wt;
    if (threadIdx.x == 0) {
        uint32_t xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        xinfo[0] = xcc;
        xinfo[1] = (VAR & 8) ? ((atomicAdd(&ctl->slots[xcc * 32], 1u) - gen * 32u) & 63u) : 0u;   // 32 x blocks-per-CU blocks per XCD per launch (checked by the host)
        xinfo[2] = 0;
        xinfo[3] = 1;
    }
    if (threadIdx.x < NW) prog[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t xcd = xinfo[0], slot = xinfo[1];
    volatile uint32_t *vprog = prog;
    volatile uint32_t *vx = xinfo;
    uint32_t *xline = &ctl->prog[xcd * 64];
    const uint32_t nblocks = gridDim.x;
    const uint32_t chunks = per_tile / (64 * U);          // per_tile is a multiple of 64 * U here (a real layout pads the stream)
    const uint32_t rounds = (ntiles + nblocks * NW - 1) / (nblocks * NW);
    double part = 0.0;
    for (uint32_t round = 0; round < rounds; ++round) {
        const uint32_t tile = (round * nblocks + blockIdx.x) * NW + wave;
        const bool active = tile < ntiles;
        for (uint32_t r = lane; r < wt; r += 64) acc[r] = 0.0;
        const uint64_t s = (uint64_t)((VAR & 128) ? (tile & 63u) : (active ? tile : 0)) * per_tile;   // 128: every wave re-reads one of 64 streams (stream from L2)
        uint32_t sp_g = 0;                                  // super-panel at the gather stage (wave-uniform)
        uint32_t SI[3][U], GC[3][U]; double SV[3][U], GG[3][U];
        auto load_stream = [&](uint32_t ch, uint32_t (&ii)[U], double (&vv)[U]) {
            static_assert(U == 4, "chunk layout is 4 entries per lane");
            const uint64_t k0 = s + (uint64_t)ch * 256;
            if (VAR & 64) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { ii[u] = (uint32_t)((ch * 1315423911u + lane * 2654435761u + u * 97u) & 0xffefffffu) % ((wt << 21) | 0xfffff); vv[u] = 1.5; }
                return;
            }
            const u32x4 q = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(idx_a + k0) + lane);
            const f64x2 a = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val_a + k0) + lane);
            const f64x2 b = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(val_a + k0 + 128) + lane);
            ii[0] = q.x; ii[1] = q.y; ii[2] = q.z; ii[3] = q.w;
            vv[0] = a.x; vv[1] = a.y; vv[2] = b.x; vv[3] = b.y;
        };
        auto pace = [&](uint32_t pan) {
            if (!(VAR & 12)) return;
            const uint32_t me = round * P + pan + 1;        // monotone over the launch
            if (lane == 0) vprog[wave] = me;
            if (!vx[3]) return;
            // block minimum: lanes 0..NW-1 read the progress words
            uint32_t spins = 0;
            for (;;) {
                uint32_t m = lane < NW ? vprog[lane] : 0xffffffffu;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = min(m, (uint32_t)__shfl_xor(m, o));
                bool ok = me <= m + slack_b;
                if (VAR & 8) {
                    if (wave == 0) {                                               // the block's scout: publishes the block minimum, refreshes the XCD minimum
                        if (lane == 0) __hip_atomic_store(&xline[slot], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        uint32_t q = __hip_atomic_load(&xline[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q = q == 0 ? 0xffffffffu : q;                               // 0 = block not registered yet: no constraint
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) q = min(q, (uint32_t)__shfl_xor(q, o));
                        if (lane == 0) vx[2] = q;
                        ok = ok && me <= q + slack_x;
                    } else {
                        ok = ok && me <= vx[2] + slack_x;
                    }
                }
                if (ok) break;
                if (++spins > 4096) { if (lane == 0) vx[3] = 0; break; }           // give up: the hint switches itself off
                __builtin_amdgcn_s_sleep(4);
            }
        };
        auto gather = [&](const uint32_t (&ii)[U], double (&gg)[U], uint32_t (&cc)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long fl = __ballot((ii[u] >> 20) & 1u);
                const uint32_t sp = sp_g + (uint32_t)__popcll(fl & ((2ull << lane) - 1ull));
                sp_g += (uint32_t)__popcll(fl);
                const uint32_t col = (sp << SP_BITS) | (ii[u] & ((1u << SP_BITS) - 1u));
                cc[u] = col;
                if (VAR & 16) gg[u] = 1.0;
                else {
                    // 1024: ascending columns inside the wave-instruction, about two 128-byte lines apart (density 0.5 entries per line, what a
                    // CU-wide column-sorted panel segment would give): do lanes that fall into one line share an L2 request?
                    const uint32_t sorted = (((uint32_t)u * 64u + lane) * 32u + (col & 31u)) & 0xffffu;
                    const double *ga = &t[(VAR & 1024) ? sorted : (VAR & 32) ? (col & 0x3ffu) : (VAR & 1) ? (col & 0xffffu) : col];
                    if (VAR & 256) gg[u] = __hip_atomic_load(ga, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: served by the L2, no L1 line fill
                    else if (VAR & 512) gg[u] = __builtin_nontemporal_load(ga);
                    else gg[u] = *ga;
                }
            }
        };
        auto accumulate = [&](const uint32_t (&ii)[U], const double (&vv)[U], const double (&gg)[U], const uint32_t (&cc)[U]) {
            if (VAR & 2) {
#pragma unroll
                for (int u = 0; u < U; ++u) part += vv[u] * gg[u];
                return;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t row = ii[u] >> 21;
                const double prod = __dmul_rn(vv[u], gg[u]);
                const uint32_t pan = cc[u] >> pcb;
                const uint32_t prow = __shfl_up(row, 1), ppan = __shfl_up(pan, 1);
                const unsigned long long same = __ballot(lane > 0 && prow == row && ppan == pan);   // continues its left neighbour's run
                const unsigned long long cut = __ballot(lane > 0 && ppan != pan);
                if (!(same | cut)) {
                    acc[row] = __dadd_rn(acc[row], prod);
                } else {
                    // runs of one row inside one panel: the leader adds its followers' products in lane order (explicit shuffles),
                    // one LDS update per run; parts (panels) one after the other, because a row may come back in the next panel
                    const unsigned long long lead = ~same;
                    const unsigned long long above = lead & ~((2ull << lane) - 1ull);             // leaders to my right
                    const uint32_t next_lead = above ? (uint32_t)__builtin_ctzll(above) : 64u;
                    const uint32_t runlen = next_lead - lane;                                    // meaningful on leaders
                    uint32_t maxrun = ((lead >> lane) & 1ull) ? runlen : 1u;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) maxrun = max(maxrun, (uint32_t)__shfl_xor(maxrun, o));
                    const uint32_t partno = (uint32_t)__popcll(cut & ((2ull << lane) - 1ull)), nparts = (uint32_t)__popcll(cut) + 1u;
                    const bool leader = (lead >> lane) & 1ull;
                    double q[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) q[j] = __shfl_down(prod, j + 1);
                    for (uint32_t f = 0; f < nparts; ++f) {
                        if (leader && partno == f) {
                            double sacc = __dadd_rn(acc[row], prod);
                            if (runlen > 1) sacc = __dadd_rn(sacc, q[0]);
                            if (runlen > 2) sacc = __dadd_rn(sacc, q[1]);
                            if (runlen > 3) sacc = __dadd_rn(sacc, q[2]);
                            acc[row] = sacc;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    if (maxrun > 4) {                                                             // long runs: the rest, step by step
                        for (uint32_t st = 4; st < maxrun; ++st) {
                            const double qq = __shfl_down(prod, st);
                            for (uint32_t f = 0; f < nparts; ++f) {
                                if (leader && partno == f && runlen > st) acc[row] = __dadd_rn(acc[row], qq);
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
                        }
                    }
                }
            }
        };
        if (active) {
            // three stream buffers, (at most) two of the gather buffers live: the loop is unrolled by three so that the buffers change
            // ROLES instead of being copied (a copy at the end of an iteration waits for everything that is in flight)
            load_stream(0, SI[0], SV[0]);
            if (chunks > 1) load_stream(1, SI[1], SV[1]);
            pace(0);
            gather(SI[0], GG[0], GC[0]);
#define STEP(r, r1, r2)                                                                                              \
            if (ch + 2 < chunks) load_stream(ch + 2, SI[r2], SV[r2]);                                                \
            if (ch + 1 < chunks) {                                                                                   \
                const uint32_t firstcol = (sp_g << SP_BITS) | (__builtin_amdgcn_readfirstlane(SI[r1][0]) & ((1u << SP_BITS) - 1u)); \
                pace(firstcol >> pcb);                                                                               \
                gather(SI[r1], GG[r1], GC[r1]);                                                                      \
            }                                                                                                        \
            accumulate(SI[r], SV[r], GG[r], GC[r]);                                                                  \
            if (++ch >= chunks) break;
            for (uint32_t ch = 0;;) {
                STEP(0, 1, 2)
                STEP(1, 2, 0)
                STEP(2, 0, 1)
            }
#undef STEP
            for (uint32_t r = lane; r < wt; r += 64) {
                const uint64_t i = (uint64_t)tile * wt + r;
                const double tn = __dsub_rn(t[i], __dmul_rn(acc[r], dinv[i]));
                __builtin_nontemporal_store(tn, &out[i]);
                __builtin_nontemporal_store(__dadd_rn(x[i], tn), &x[i]);
                part = __dadd_rn(part, __dmul_rn(tn, tn));
            }
        }
        if (VAR & 12) { if (lane == 0) vprog[wave] = (round + 1) * P + 1; }      // done with this round: as far along as its end
    }
    if (VAR & 12) {
        if (lane == 0) vprog[wave] = 0xfffffff0u;
        __syncthreads();
        if ((VAR & 8) && threadIdx.x == 0) __hip_atomic_store(&xline[slot], 0xfffffff0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) partials[blockIdx.x * NW + wave] = part;
}

__global__ void ctl_reset_prog(ctl_block *c) { c->prog[threadIdx.x] = 0; }

template <int NW, int U, int VAR>
static int run(uint32_t n, uint32_t k, uint32_t wt, int pcb, uint32_t slack_b = 1, uint32_t slack_x = 2, int dup = 0, int blocks_per_cu = 1)
{
    const uint32_t P = (uint32_t)(((uint64_t)n + (1ull << pcb) - 1) >> pcb), ntiles = (n + wt - 1) / wt;
    uint32_t per_tile = wt * k;
    per_tile = (per_tile + 64 * U - 1) / (64 * U) * (64 * U);
    const uint64_t total = (uint64_t)ntiles * per_tile;
    uint32_t *idx; double *val, *t, *dinv, *out, *x, *partials; ctl_block *ctl;
    CK(hipMalloc(&idx, total * 4)); CK(hipMalloc(&val, total * 8));
    const uint64_t nr = (uint64_t)ntiles * wt;
    CK(hipMalloc(&t, (nr + 64 + (1u << 21)) * 8)); CK(hipMalloc(&dinv, nr * 8)); CK(hipMalloc(&out, nr * 8)); CK(hipMalloc(&x, nr * 8));
    CK(hipMalloc(&partials, 65536 * 8)); CK(hipMalloc(&ctl, sizeof(ctl_block)));
    CK(hipMemset(t, 0, (nr + 64 + (1u << 21)) * 8)); CK(hipMemset(dinv, 0, nr * 8)); CK(hipMemset(x, 0, nr * 8)); CK(hipMemset(ctl, 0, sizeof(ctl_block)));
    fill_kernel<<<8192, 256>>>(total, per_tile, wt, pcb, P, n, idx, val, dup);
    CK(hipDeviceSynchronize());
    const size_t lds = (size_t)NW * wt * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(panel2_kernel<NW, U, VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t gen = 0;
    auto launch = [&]() {
        ctl_reset_prog<<<1, 512>>>(ctl);
        panel2_kernel<NW, U, VAR><<<grid, NW * 64, lds>>>(ntiles, per_tile, wt, pcb, P, idx, val, t, dinv, out, x, partials, ctl, slack_b, slack_x, gen * (uint32_t)blocks_per_cu);
        ++gen;
    };
    launch(); CK(hipDeviceSynchronize());
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 6; ++r) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 6;
    uint32_t hs[256]; CK(hipMemcpy(hs, ctl->slots, sizeof(hs), hipMemcpyDeviceToHost));
    const double bytes = 12.0 * n * k + 44.0 * n;
    printf("NW=%d U=%d var=%2d tile=%u rows/wave panel=2^%d (P=%u) slack=%u/%u dup=%d grid=%u: %.3f ms -> %.1f %% of 8 TB/s (%.0f G gathers/s)  [xcd0 regs %u of %u]\n",
           NW, U, VAR, wt, pcb, P, slack_b, slack_x, dup, grid, ms, bytes / (ms * 1e-3) / 8e12 * 100, (double)n * k / (ms * 1e-3) / 1e9, hs[0], gen * 32 * blocks_per_cu);
    fflush(stdout);
    hipFree(idx); hipFree(val); hipFree(t); hipFree(dinv); hipFree(out); hipFree(x); hipFree(partials); hipFree(ctl);
    return 0;
}

int main(int argc, char **argv)
{
    const uint32_t n = 10000000;
    const int set = argc > 1 ? atoi(argv[1]) : 0;
    if (set == 0) {
        // 16 waves per CU, 1220 rows per wave (2 rounds cover 10^7 rows): ceilings, then pacing variants
        run<16, 4, 16>(n, 16, 1221, 16);            // stream + LDS only
        run<16, 4, 1>(n, 16, 1221, 16);             // perfect locality
        run<16, 4, 0>(n, 16, 1221, 16);             // free-running
        run<16, 4, 4>(n, 16, 1221, 16, 1, 0);       // block pacing
        run<16, 4, 12>(n, 16, 1221, 16, 1, 1);      // block + XCD pacing
        run<16, 4, 12>(n, 16, 1221, 16, 1, 2);
        run<16, 4, 12>(n, 16, 1221, 16, 2, 3);
        run<16, 4, 12>(n, 16, 1221, 15, 2, 4);      // smaller panels
        run<16, 4, 12>(n, 16, 1221, 17, 1, 1);      // larger panels
        run<16, 4, 12>(n, 16, 1221, 16, 1, 2, 1);   // with runs of equal rows
    } else if (set == 5) {
        run<16, 4, 67>(n, 16, 1221, 16);            // gathers only, random inside the panel
        run<16, 4, 1024 + 67>(n, 16, 1221, 16);     // gathers only, ascending ~2 lines apart
        run<16, 4, 1>(n, 16, 1221, 16);             // whole kernel, perfect locality, random
        run<16, 4, 1024 + 1>(n, 16, 1221, 16);      // whole kernel, ascending ~2 lines apart
    } else if (set == 4) {
        run<16, 4, 256 + 67>(n, 16, 1221, 16);      // gathers only, sc1
        run<16, 4, 512 + 67>(n, 16, 1221, 16);      // gathers only, nt
        run<16, 4, 256 + 1>(n, 16, 1221, 16);       // perfect locality, sc1 gathers
        run<16, 4, 512 + 1>(n, 16, 1221, 16);       // perfect locality, nt gathers
        run<16, 4, 256 + 4>(n, 16, 1221, 16, 1, 0); // block pacing, sc1 gathers
        run<16, 4, 512 + 4>(n, 16, 1221, 16, 1, 0); // block pacing, nt gathers
    } else if (set == 3) {
        run<16, 4, 129>(n, 16, 1221, 16);           // perfect locality + stream from L2 (64 tiles re-read)
        run<16, 4, 144>(n, 16, 1221, 16);           // stream from L2, no gathers
        run<16, 4, 128>(n, 16, 1221, 16);           // real gathers, stream from L2
    } else if (set == 2) {
        run<16, 4, 18>(n, 16, 1221, 16);            // stream only (no gathers, no LDS)
        run<16, 4, 3>(n, 16, 1221, 16);             // perfect locality, no LDS update
        run<16, 4, 33>(n, 16, 1221, 16);            // gathers from 8 KB (L1 hits) + LDS
        run<16, 4, 65>(n, 16, 1221, 16);            // no stream loads: gathers (first panel) + LDS
        run<16, 4, 67>(n, 16, 1221, 16);            // gathers only
        run<8, 4, 1>(n, 16, 1221, 16, 1, 2, 0, 2);  // perfect locality, 2 x 8 waves
        run<4, 4, 1>(n, 16, 1221, 16, 1, 2, 0, 4);
        run<4, 4, 1>(n, 16, 1221, 16, 1, 2, 0, 2);  // half the waves per CU
    } else if (set == 1) {
        run<8, 4, 0>(n, 16, 1221, 16, 1, 2, 0, 2);
        run<8, 4, 12>(n, 16, 1221, 16, 1, 2, 0, 2);     // two 8-wave blocks per CU
        run<4, 4, 12>(n, 16, 1221, 16, 1, 2, 0, 4);
    }
    return 0;
}
