// sl_internal.hpp — internal structures of libsublinear_hip (not part of the ABI).
//
// HBM layout of a matrix ("row-slice" layout): rows are grouped in
// slices of 64 consecutive rows = one wavefront, lane l owns row 64*s + l.  A slice whose longest row has
// L entries is stored in ceil(L / 2) "pair blocks" of 128 entries; two consecutive pair blocks form a quad
// (entries 4q..4q+3 of all 64 rows), an odd last pair block stands alone (rows of 5 take 6 slots, not 8):
//     vals   : [pair block][lane 0..63][2]  f64   -> 16-B loads, 1 KiB per wave (both halves of a quad alike)
//     cols   : [quad][lane][4] u32 (16-B loads) over two pair blocks; odd last block [lane][2] (8-B loads)
//     cols16 : the same slots as int16 offsets col - row (8-B / 4-B loads)
// slice_ptr[] counts pair blocks.  Every matrix byte is fetched by fully coalesced loads while each lane still
// walks ITS row left to right — the reference's summation order (sparse.rs:187-203) is kept bit for bit with
// no cross-lane reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#include "../../include/sublinear_hip.h"

// Every device allocation of the library goes through here.  SL_POISON_ALLOC=1 (debug): the new block — and every block the workspace
// cache hands out again — is filled with 0xA5 bytes first: nothing may rely on fresh device memory being zero (it is, on an idle box, and
// is not once the box has run other work: the kind of fault that shows once in a thousand runs).  The test suite is run under it.
hipError_t sl_malloc_checked(void **p, size_t bytes);
void sl_poison(void *p, size_t bytes);                      // no-op unless SL_POISON_ALLOC=1
template <class T> inline hipError_t sl_malloc(T **p, size_t bytes) { return sl_malloc_checked(reinterpret_cast<void **>(p), bytes); }   // (an explicit wrapper: no runtime name is redefined)

#define SL_SLICE 64
#ifndef SL_BLOCK
#define SL_BLOCK 256
#endif
#define SL_WAVES_PER_BLOCK (SL_BLOCK / SL_SLICE)
// upper limit of the per-matrix long-row threshold (sl_matrix::long_row): longer rows leave the slice layout (one hub row would
// stretch its whole 64-row slice) and are reduced by the long-row kernel, one block per row, products in parallel, additions in
// the reference order
#ifndef SL_LONG_ROW
#define SL_LONG_ROW 256u
#endif
#define SL_LONG_SENTINEL 0xffffffffu   // row_len[] value of such a row

struct sl_matrix {
    uint64_t n_rows = 0, n_cols = 0, nnz = 0, row_offset = 0;
    uint32_t flags = 0;
    int device = 0;
    // row-slice layout
    uint64_t n_slices = 0, padded_nnz = 0;
    uint32_t *d_slice_ptr = nullptr; // [n_slices+1] in pair blocks (128 entries)
    uint32_t *d_row_len = nullptr;   // [n_slices*64]
    uint32_t *d_cols = nullptr;      // [padded_nnz]
    uint16_t *d_cols16 = nullptr;    // [padded_nnz] col - row as int16 (uniform-width band matrices only)
    double *d_vals = nullptr;        // [padded_nnz]
    uint32_t max_row_nnz = 0, min_row_nnz = 0, uniform_width = 0;
    uint32_t long_row = SL_LONG_ROW;   // rows with more entries than this are served by the long-row kernel (per matrix: 2.5 x mean, in [24, 256])
    uint64_t bandwidth = 0;          // max |col - (row_offset + row)| over stored entries
    // raw CSR (SL_MATRIX_KEEP_CSR or needed by the sparse-frontier kernels)
    uint32_t *d_row_ptr = nullptr, *d_col_idx = nullptr;
    double *d_values = nullptr;
    // transpose as CSR of A^T (SL_MATRIX_WITH_TRANSPOSE): rows of each column ascending
    uint32_t *d_tptr = nullptr; // [n_cols+1]
    uint32_t *d_trow = nullptr; // [nnz]
    uint32_t *d_long_rows = nullptr; // ascending local ids of the rows with more than SL_LONG_ROW entries
    uint64_t n_long = 0;
    double *d_tval = nullptr;   // [nnz]
    uint32_t *d_tent = nullptr; // [nnz] CSR entry index of each transposed entry (the permutation of the transpose)
    // column-panel layout: for matrices whose columns are spread far beyond what the LDS window can hold. The entries
    // of each tile of SL_PANEL_TILE rows, regrouped by panel of 2^SL_PANEL_COL_BITS columns, then row, then column: one stream per tile.
    uint32_t *d_pan_tile_ptr = nullptr; // [n_pan_tiles + 1], in entries (every tile padded to a multiple of SL_PANEL_CHUNK)
    uint16_t *d_pan_row = nullptr;      // row inside the tile (SL_PANEL_TILE = padding)
    uint32_t *d_pan_col = nullptr;      // global column
    double *d_pan_val = nullptr;
    uint64_t n_pan_tiles = 0, pan_entries = 0;
    bool pan_balanced = false;          // every tile's stream within 10 % of the mean length: launched one resident round at a time
    // paced column-panel layout (round 2; balanced matrices — uniformly random columns): ONE persistent 16-wave block per CU, a wave
    // owns a tile of pw_rpw rows (running sums in LDS; the CU's whole LDS holds its 16 tiles), tiles are dealt in rounds.  Rows go to
    // tiles in groups of 16, round robin (group q -> tile q mod n_pw_tiles, slot (q / n_pw_tiles) * 16 + row mod 16): a tile's rows
    // are spread over the whole matrix, so whatever depends on the row index — the diagonal first of all — lands in all panels alike
    // and every tile carries the same work in every panel (with consecutive rows a tile's 1200 diagonal entries sit in ONE panel:
    // ten panel-times of extra work at a different place for every block, which tears the blocks of an XCD apart).  A tile's
    // entries are one stream sorted by (panel of 2^16 columns, row, column), 12 bytes per entry:
    //     pw_idx u32 = row in tile << 21 | super-panel step << 20 | column & 0xfffff      pw_val f64
    // stored in chunks of 256 entries, pre-transposed so that three 16-byte loads hand lane l the entries l, l+64, l+128, l+192:
    //     pw_idx [chunk][lane][4]      pw_val [chunk][half][lane][2]
    // (super-panel = 2^20 columns; the step bit is set on the first entry of a tile's stream that lies in the next super-panel,
    // an empty super-panel is bridged by a padding entry: value 0, row = the tile's spare slot.)
    uint32_t *d_pw_idx = nullptr;
    double *d_pw_val = nullptr;
    uint32_t *d_pw_tile_ptr = nullptr;  // [n_pw_tiles + 1] in chunks
    uint64_t n_pw_tiles = 0, pw_chunks = 0;
    uint32_t pw_rpw = 0, pw_blocks = 0;
    // the same layout for WIDE BANDS (windows beyond the LDS): groups of rows are dealt among the 16 tiles of ONE block (pw_deal = 16
    // instead of all tiles), so that a block owns a contiguous range of rows and its waves walk the same narrow panels
    // (2^pw_pbits columns, a few KB of the vector) at the same time: the gathers become hits in the CU's L1
    uint32_t pw_deal = 0, pw_pbits = 16;
    // XCD-local spans (locality-bounded columns far beyond the L2: |i - j| <= w with w in the 10^5..10^6s): pw_xcd = G > 0 deals a run of
    // row groups among the tiles of the cus / G blocks that share an L2 (block b runs on XCD b % G), so that an XCD's waves gather from
    // the columns of ITS rows +- w only — fewer first touches per L2 and more entries per panel; the kernel maps blockIdx.x to the
    // logical block (b % G) * (blocks / G) + b / G
    uint32_t pw_xcd = 0;
    // With pw_xcd the spans are explicit: d_pw_span_tab[2 p], [2 p + 1] = first row group and number of row groups of PHYSICAL span p
    // (the span that tiles [p * deal, (p + 1) * deal) carry; physical span p runs in round p / G on XCD p % G).  For a rank's rows of
    // a partition the order is EDGE FIRST: the spans holding the rows within the bandwidth of either end of the range (what the
    // neighbours pull) — sized to cover exactly that — take the first pw_edge_rounds rounds, the interior spans the rest.  The partitioned
    // step launches the edge rounds first and exchanges beside the interior rounds (sl_api.hip, dist_step).  pw_edge_rows = rows from
    // either end that the edge rounds are sure to cover; pw_edge_rounds = 0: row order, no such split.
    uint32_t *d_pw_span_tab = nullptr;
    uint32_t pw_edge_rounds = 0;
    uint64_t pw_edge_rows = 0;
    bool pw_band = false;               // the wide-band form (block-local rows, narrow panels)
    uint32_t pw_slack = 4;              // panels a wave may gather ahead of the slowest wave of its block: about two chunks of its stream
    // ORDER-FREE column stream (SL_MATRIX_ORDER_ANY / SL_ORDER_ANY, round 3): ONE persistent 16-wave block per CU owns a BLOCK tile of
    // pwr_rpb rows (running sums in LDS, the CU's whole LDS; groups of 16 rows dealt round robin among the block tiles), and the
    // off-diagonal entries of those rows form ONE stream sorted by COLUMN, cut into chunks of 64 entries that the block's 16 waves
    // take in turn: the 64 gathers of a wave-instruction walk ascending lines a few hundred bytes apart, all CUs sweep the vector at
    // the same pace (equal work), and the sums are added by LDS atomics (ds_add_f64) in whatever order they arrive.  12 bytes per entry:
    //     pwr_idx u32 = slot in the block tile << 17 | column - pwr_base[chunk]  (chunks are cut early where the offset would not fit)
    //     pwr_val f64              pwr_base[chunk] u32 = the chunk's first column
    // chunk layout [chunk][lane] for both arrays.  The diagonal is not in the stream: pwr_diag[row] (the sum of
    // the row's entries at its own column) enters in the epilogue, where the row's own vector entry is loaded anyway.
    uint32_t *d_pwr_idx = nullptr, *d_pwr_base = nullptr, *d_pwr_tile_ptr = nullptr;
    double *d_pwr_val = nullptr, *d_pwr_diag = nullptr;
    uint64_t n_pwr_tiles = 0, pwr_chunks = 0;
    uint32_t pwr_rpb = 0, pwr_blocks = 0;

    // COLUMN-CONSTANT operators (round 4): every off-diagonal entry of a column holds the same value and every diagonal entry is exactly
    // 1 — the PageRank / PPR systems I - (1 - alpha) P^T of an unweighted graph, a_iu = -(1 - alpha) / deg_u.  d_colval[u] = that value
    // (0 for a column nobody references).  A dense push round may then run the paced kernel on the INDEX words of its stream alone
    // (4 instead of 12 bytes per entry), gathering from the pre-multiplied vector z_u = colval_u * delta_u: the product of every entry of
    // column u is that one rounded product, so the row sums keep their bits (sl_pw_kernel<.., IDX>).
    double *d_colval = nullptr;
    uint64_t device_bytes = 0;
    // what the slice fill counted (selects the column streams; kept so that the in-place mutators can rebuild them): entries far from
    // their row, entries in the slice layout, entries continuing a diagonal; stream_bytes = the share of device_bytes the streams hold
    uint64_t far_entries = 0, slice_entries = 0, diagonal_entries = 0, stream_bytes = 0;
    bool caller_device_arrays = false;  // sl_matrix_create_csr was handed device pointers (diagnostics of the layout build)
};
#ifndef SL_PANEL_TILE
#define SL_PANEL_TILE 2048u          // rows per tile = per wave: their running sums live in LDS (16 KiB)
#endif
#ifndef SL_PANEL_COL_BITS
#define SL_PANEL_COL_BITS 16         // a panel = 2^16 columns = 512 KiB of the gathered vector: stays in an XCD's L2 while the tiles pass it
                                     // (n = 10^7 x 16, ms per step by panel size: 2^18 1.57, 2^17 1.49, 2^16 1.39, 2^15 1.36, 2^14 1.43)
#endif
#define SL_PANEL_CHUNK 256u          // entries a wave has in flight (4 x 64)
#ifndef SL_PW_WAVES
#define SL_PW_WAVES 16               // paced layout: waves per block = tiles per CU
#endif
#ifndef SL_PW_GROUP
#define SL_PW_GROUP 16u              // rows are dealt to tiles in groups of 16 consecutive rows (one 128-byte line of every vector)
#endif
#ifndef SL_PW_MAX_ROWS
#define SL_PW_MAX_ROWS 1264u         // rows per wave tile (79 groups): 16 x (1264 + 1 spare slot) x 8 B = 161 920 B of the 160 KiB LDS
#endif
#ifndef SL_PW_ROW_BITS
#define SL_PW_ROW_BITS 11            // bits of the row slot in an entry's index word (slots <= SL_PW_MAX_ROWS); the rest: step bit + column bits
#endif
#ifndef SL_PWR_MAX_ROWS
#define SL_PWR_MAX_ROWS 19968u       // rows per BLOCK tile of the order-free stream (1248 groups): 159 744 B of the 160 KiB LDS
#endif
#define SL_PWR_CHUNK 64u             // entries per chunk of the order-free stream = one wave-instruction (small chunks keep the part of
                                     // the vector a CU's 16 waves gather from at any time to a few hundred KB: 256-entry chunks spanned
                                     // 2.2 MB per CU at n = 10^7 and the L2 hit rate fell to 72 %)
#define SL_PWR_OFF_BITS 17           // column offset bits of an entry (relative to its chunk's base); slot bits = 15 (slots < 2^15 > SL_PWR_MAX_ROWS)
#define SL_PW_ROW_SHIFT (32 - SL_PW_ROW_BITS)
#define SL_PW_SP_BITS (SL_PW_ROW_SHIFT - 1)   // super-panel: the column bits an entry carries (20)
#ifndef SL_PANEL_WAVES
#define SL_PANEL_WAVES 4
#endif

// thread-local launch context
struct sl_ctx {
    hipStream_t stream = nullptr;
    std::string last_error;
    // grow-only device scratch for reductions (per thread, per device)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    int scratch_device = -1;
    // workspace pool (sl_ws_alloc / sl_ws_free)
    struct ws_block { void *p; size_t bytes; int device; bool in_use; };
    std::vector<ws_block> ws;
    // pinned staging of the control-plane transfers (sl_read_back / sl_upload): page-locked host memory of the library's own
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    bool pinned_failed = false;   // hipHostMalloc was refused once on this thread: pageable transfers, not a retry per call
    // side stream + fork / join events: the long-row kernel of a launch runs beside the slice kernel (sl_kernels.hip)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int side_device = -1;
};
// ---- tracing and logging (hook points of the reference: optimized_solver.rs:25-28,73-75, lib.rs:137-141) ----------
// roctx ranges around layout build / solve loops / exchange steps: resolved at run time from libroctx64.so (rocprofv3 --marker-trace
// shows them); without the library they cost one branch.  SL_LOG=1 (phases) / 2 (launch decisions) writes lines to stderr.
void sl_range_push(const char *name);
void sl_range_pop();
int sl_log_level();
void sl_log(int level, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
struct sl_range { explicit sl_range(const char *n) { sl_range_push(n); } ~sl_range() { sl_range_pop(); } sl_range(const sl_range &) = delete; sl_range &operator=(const sl_range &) = delete; };

// Control-plane transfers — every word the HOST computes a layout from (slice widths, per-tile counts, column histograms) and every
// table it sends back (slice pointers, tile pointers): through the library's own page-locked staging buffer, chunk by chunk, each chunk
// complete (stream synchronised) before the host touches it.  No asynchronous copy ever targets pageable memory of a std::vector: in the
// many-process campaigns of round 3 a row twice lost its diagonal because the slice pointers on the device did not match the row lengths
// there, and the pageable paths of the runtime (staging rings shared by a process's copies, on-the-fly pinning) were the one part of that
// chain the library did not control.  SL_STAGING=pageable restores the old calls (A/B in tests/fuzz_dist.py).
// sl_read_back returns with the data in host_dst; sl_upload returns when host_src may be reused AND the data is on the device.
sl_status sl_read_back(void *host_dst, const void *dev_src, size_t bytes, hipStream_t st);
sl_status sl_upload(void *dev_dst, const void *host_src, size_t bytes, hipStream_t st);
bool sl_side_stream(sl_ctx &c);      // lazily creates the side stream / events for the current device; false if that failed
sl_ctx &sl_context();
sl_status sl_fail(sl_status s, const char *fmt, ...);
void *sl_scratch(size_t bytes); // nullptr on failure

// Workspace pool: device buffers of the solve calls (term vectors, frontier lists, logs) are taken from a per-thread
// cache and returned to it instead of hipMalloc / hipFree on every call — hipFree synchronises the device and both
// cost 10^2 us, which is what a local push query or a small solve takes in total.  All work of a thread runs on its
// one in-order stream, so a buffer handed out again is only touched after the previous user's launches.
// sl_release_workspace() (ABI) returns the cached buffers to the driver.
void *sl_ws_alloc(size_t bytes);   // nullptr on failure
void sl_ws_free(void *p);

#define SL_HIP(call)                                                                            \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return sl_fail(SL_DEVICE_ERROR, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                                 \
    } while (0)

// HIP-event stopwatch on a stream; the events are released on every exit path
struct sl_timer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream = nullptr;
    sl_timer() = default;
    sl_timer(const sl_timer &) = delete;
    sl_timer &operator=(const sl_timer &) = delete;
    ~sl_timer() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    sl_status start(hipStream_t s)
    {
        stream = s;
        SL_HIP(hipEventCreate(&e0));
        SL_HIP(hipEventCreate(&e1));
        SL_HIP(hipEventRecord(e0, s));
        return SL_OK;
    }
    float stop()                       // milliseconds since start(); waits for the stream to reach this point
    {
        float ms = 0.f;
        if (!e0 || !e1) return ms;
        if (hipEventRecord(e1, stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    }
};

struct DevBuf {
    void *p = nullptr;
    bool pooled = true;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { reset(); }
    void reset() { if (p) { if (pooled) sl_ws_free(p); else (void)hipFree(p); p = nullptr; } }
    sl_status alloc(size_t bytes)
    {
        reset();
        pooled = true;
        p = sl_ws_alloc(bytes ? bytes : 8);
        if (!p) return sl_fail(SL_ALLOCATION, "device allocation of %zu bytes failed", bytes);
        return SL_OK;
    }
    // exact-size allocation outside the pool, for buffers whose ownership may move to a longer-lived object
    sl_status alloc_owned(size_t bytes)
    {
        reset();
        pooled = false;
        if (sl_malloc(&p, bytes ? bytes : 8) != hipSuccess) { p = nullptr; return sl_fail(SL_ALLOCATION, "sl_malloc(%zu) failed", bytes); }
        return SL_OK;
    }
    void *release() { void *q = p; p = nullptr; return q; }   // alloc_owned buffers only
    template <class T> T *as() const { return static_cast<T *>(p); }
};
// Exception fence of the extern "C" boundary: host-side containers (std::vector, std::function, new) may throw; nothing may
// unwind through a C caller (Rust: panic = abort; JS / ctypes: process abort).  Usage: the whole body of an entry point sits
// between SL_ABI_BEGIN and SL_ABI_END.
#include <new>
#include <exception>
#define SL_ABI_BEGIN try {
#define SL_ABI_END                                                                                                   \
    } catch (const std::bad_alloc &) { return sl_fail(SL_ALLOCATION, "host allocation failed (%s)", __func__); }     \
    catch (const std::exception &e_) { return sl_fail(SL_ALGORITHM_ERROR, "internal error in %s: %s", __func__, e_.what()); }
#define SL_TRY(expr) do { sl_status s_ = (expr); if (s_ != SL_OK) return s_; } while (0)

// entry slot of position k of `lane`'s row inside a slice occupying pair blocks [h0, h1)   (fill / diagnostic kernels)
__host__ __device__ inline uint64_t sl_col_slot(uint32_t h0, uint32_t h1, uint32_t k, uint32_t lane)
{
    const uint32_t pk = k >> 1, w2 = h1 - h0;
    if ((w2 & 1u) && pk == w2 - 1) return (uint64_t)(h0 + pk) * 128 + lane * 2 + (k & 1u);      // the odd last pair block
    return (uint64_t)(h0 + (pk & ~1u)) * 128 + lane * 4 + (k & 3u);                              // inside a quad
}
__host__ __device__ inline uint64_t sl_val_slot(uint32_t h0, uint32_t k, uint32_t lane)
{
    return (uint64_t)(h0 + (k >> 1)) * 128 + lane * 2 + (k & 1u);
}

// ---- kernel launchers (sl_kernels.hip) -------------------------------------------------
enum sl_epilogue { SL_EPI_SPMV = 0, SL_EPI_NEUMANN = 1, SL_EPI_RESIDUAL = 2, SL_EPI_PUSH = 3 };

// Control block of a speculatively enqueued solve loop (sl_neumann_solve): the host enqueues several iterations
// without reading anything back; every reducing launch logs its sum here and closes the gate (stop_after) when the
// reference's stop rule fires, and every launch of a later iteration then returns at once.  The host replays the
// reference's control flow over the log afterwards — same decisions, same results, one readback per batch.
#define SL_CTL_LOG 62
struct sl_solve_ctl {
    uint32_t stop_after;   // launches with gate_it > stop_after do nothing
    uint32_t n_done;       // judged reductions executed so far (a prefix of the enqueued ones)
    uint32_t pad[2];
    double log[SL_CTL_LOG];
};
enum { SL_JUDGE_NONE = 0, SL_JUDGE_LT = 1, SL_JUDGE_LE_OR_NONFINITE = 2,
       SL_JUDGE_LOCAL = 3 };   // gated like the others, but the sum only goes to `result`: a partitioned solve judges the sum over ALL ranks (sl_comm.hip)

// Frontier threshold of the push: one value for all rows, or (rows != null) one per row — the degree-scaled admission rule of the
// ACL push, r[u] >= epsilon * max(deg_u, 1) (forward_push.rs:93-99, graph/mod.rs:171-212), as a vector theta_u
struct sl_theta {
    double s;
    const double *rows;
    __host__ __device__ double at(uint64_t i) const { return rows ? rows[i] : s; }
    __host__ __device__ bool everything() const { return !rows && s <= 0.0; }      // theta <= 0: every row is in every frontier
};

struct sl_row_args {
    // matrix
    const uint32_t *slice_ptr, *row_len, *cols;
    const uint16_t *cols16;   // null unless the matrix carries 16-bit column offsets
    const uint32_t *csr_ptr, *csr_idx;   // raw CSR (long rows only)
    const double *csr_val;
    const uint32_t *long_rows;
    uint32_t n_long;
    uint32_t part_stride;     // partial slots per set (set by the launcher): main grid + n_long
    const double *vals;
    uint64_t n_rows, n_cols, n_slices, row_offset;
    uint64_t bandwidth;   // ~0 = unknown / do not use the LDS band kernel
    uint32_t uniform_width;
    uint32_t max_row_nnz;     // longest row kept in the slice layout (selects the multi-pass window kernel: rows of at most 16 entries)
    // column-panel layout (null unless the matrix carries one)
    const uint32_t *pan_tile_ptr; const uint16_t *pan_row; const uint32_t *pan_col; const double *pan_val;
    uint32_t n_pan_tiles;
    uint32_t pan_balanced;
    // paced column-panel layout (null unless the matrix carries one)
    const uint32_t *pw_idx; const double *pw_val; const uint32_t *pw_tile_ptr;
    uint32_t pw_tiles, pw_rpw, pw_blocks;
    uint32_t pw_deal, pw_pbits; // tiles a run of row groups is dealt among (all tiles, the tiles of one XCD's blocks, or the 16 of a block); log2 of the panel width
    uint32_t pw_xcd;          // > 0: blocks b, b + G, b + 2 G, ... (one XCD) are logical neighbours (sl_matrix::pw_xcd)
    const uint32_t *pw_span_tab;   // explicit spans (sl_matrix::d_pw_span_tab; null: spans of deal * rpw rows in row order); with it blk_lo / blk_cnt select ROUNDS of the paced kernel
    uint32_t pw_slack;        // panels a wave may run ahead of the slowest wave of its block (set by the launcher; >= 2^20: no pacing)
    // order-free column stream (null unless the matrix carries one)
    const uint32_t *pwr_idx, *pwr_base, *pwr_tile_ptr; const double *pwr_val, *pwr_diag;
    uint32_t pwr_tiles, pwr_rpb, pwr_blocks;
    // vectors
    const double *gather; // gathered vector (n_cols)
    const double *dinv;   // n_rows
    const double *aux;    // RESIDUAL: rhs (n_rows); PUSH: unused
    // index-only stream of a column-constant operator (PUSH epilogue on the paced layout; all three null otherwise): zgather = the
    // pre-multiplied gathered vector colval (.) gather, zcol = colval, zout = where the epilogue leaves colval_i * delta'_i for the next round
    const double *zgather, *zcol;
    double *zout;
    uint32_t aux_dot;     // RESIDUAL epilogue as "product + dot": out = A g, sum of aux_i * (A g)_i instead of the residual's sum of squares (CG: p . Ap in the SpMV's launch)
    double *out;          // SPMV: y; NEUMANN: t_out; RESIDUAL: r (may be null); PUSH: delta_out
    double *x;            // NEUMANN / PUSH: x in/out
    double *r;            // PUSH: r in/out
    double theta;         // PUSH
    const double *theta_rows; // PUSH: per-row thresholds (null: `theta` for every row)
    double *partials;     // per-block partial sums (norm^2); PUSH: also counts at partials + nblocks (as u64)
    uint32_t partials_slack; // doubles available behind the two partial sets (>= 512 enables the two-stage reduction of very many partials)
    double *result;       // device scalar(s): [0] = sum of squares, PUSH: [1] = frontier count (as double bits u64)
    // speculative solve loop (null ctl = plain launch)
    sl_solve_ctl *ctl;
    uint32_t gate_it, ctl_slot;
    int ctl_mode;         // SL_JUDGE_*: which comparison of the reduced sum against ctl_threshold closes the gate
    double ctl_threshold;
    // a RANGE of the launch's row blocks (blk_cnt != 0): logical blocks [blk_lo, blk_lo + blk_cnt) only, no closing reduction — the
    // partitioned step runs the blocks at the edges of a rank's row range apart from the interior (sl_rows_geometry, sl_launch_rows_reduce)
    uint32_t blk_lo, blk_cnt;
};
sl_status sl_launch_rows(const sl_row_args &a, sl_order order, sl_epilogue epi, hipStream_t s, uint32_t *n_partials = nullptr);
// rows per block and blocks (padding included) of the launch sl_launch_rows would make; rows_per_block = 0: this matrix / kernel has no range launches
sl_status sl_rows_geometry(const sl_row_args &a, sl_order order, sl_epilogue epi, uint32_t *rows_per_block, uint32_t *n_blocks);
// the closing reduction of a launch made in ranges (n_partials as sl_launch_rows reports it for the whole launch)
sl_status sl_launch_rows_reduce(const sl_row_args &a, sl_epilogue epi, uint32_t n_partials, hipStream_t s);
sl_status sl_launch_final_reduce(const double *partials, uint32_t n, double *result, hipStream_t s);
sl_status sl_launch_rows_add(const sl_row_args &a, hipStream_t s);   // a.out += A a.gather, running sums seeded with a.out (sparse.rs:192-203)
sl_row_args sl_matrix_row_args(const sl_matrix *m);   // matrix part filled, vectors null
uint32_t sl_row_grid(uint64_t n_slices);

sl_status sl_launch_sumsq(uint64_t n, const double *x, double *partials, double *result, hipStream_t s);
sl_status sl_launch_sumsq_judged(uint64_t n, const double *x, double *partials, sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot,
                                 int mode, double threshold, hipStream_t s, double *result = nullptr);
sl_status sl_launch_ctl_reset(sl_solve_ctl *ctl, hipStream_t s);
sl_status sl_launch_dot(uint64_t n, const double *x, const double *y, double *partials, double *result, hipStream_t s);
sl_status sl_launch_axpy(uint64_t n, double alpha, const double *x, double *y, hipStream_t s);
sl_status sl_launch_scale_rows(uint64_t n, const double *a, const double *b, double *out, hipStream_t s); // out = a*b
sl_status sl_launch_sub(uint64_t n, const double *a, const double *b, double *out, hipStream_t s);        // out = a-b
sl_status sl_launch_abs_sum(uint64_t n, const double *x, double *partials, double *result, hipStream_t s);
sl_status sl_launch_abs_max(uint64_t n, const double *x, double *partials, double *result, hipStream_t s);                    // max |x_i|, NaN entries skipped
sl_status sl_launch_diff_sumsq(uint64_t n, const double *x, const double *y, double *partials, double *result, hipStream_t s); // sum (x_i - y_i)^2

// matrix build (sl_matrix.hip)
sl_status sl_build_from_device_csr(sl_matrix *m, const uint32_t *d_row_ptr, const uint32_t *d_col_idx,
                                   const double *d_values, bool keep_csr_copy);
// a6 + a7 pass: h_status[0] bits 1 = not dominant, 2 = missing diagonal, 4 = near-zero diagonal;
// h_status[1..3] = first offending row of each class.  d_dinv may be null.
void sl_matrix_row_dominance(const sl_matrix *m, uint64_t row, double out[2]);     // |a_ii|, sum |a_ij| of one row (error messages)
sl_status sl_matrix_diag_pass(const sl_matrix *m, double *d_dinv, unsigned long long h_status[4]);
// ConditioningInfo's row statistics (matrix/mod.rs:487-514, 83-100): h_out[0] = min |a_ii| / sum |a_ij| over rows with off-diagonal
// weight (+inf when there is none), h_out[1] = max |a_ii| + sum |a_ij|
sl_status sl_matrix_cond_pass(const sl_matrix *m, double h_out[2]);
// element / iterator / norm side of trait Matrix (sl_matrix.hip): get, row_iter, col_iter, the sum under frobenius_norm
sl_status sl_matrix_get_entry(const sl_matrix *m, uint64_t row, uint64_t col, int *found, double *value);
sl_status sl_matrix_fetch_row(const sl_matrix *m, uint64_t row, uint64_t capacity, uint32_t *cols, double *values, uint64_t *count);
sl_status sl_matrix_fetch_col(const sl_matrix *m, uint64_t col, uint64_t capacity, uint32_t *rows, double *values, uint64_t *count);
sl_status sl_matrix_frobenius_sq(const sl_matrix *m, double *sum_sq);
sl_status sl_matrix_entry_bandwidth(const sl_matrix *m, uint64_t *bandwidth);   // max |row - col| over ALL stored entries (hub rows included)
// the `&mut self` methods (matrix/mod.rs:346-372 over sparse.rs:229-248): every layout copy updated (sl_matrix.hip)
sl_status sl_matrix_slices_to_csr(const sl_matrix *m, uint32_t *d_rp, uint32_t *d_ci, double *d_va);   // a matrix without raw CSR: its rows from the slice layout
sl_status sl_matrix_scale_values(sl_matrix *m, double factor);
sl_status sl_matrix_shift_diagonal(sl_matrix *m, double alpha);
// same rules over a plain CSR operator (used for A^T, which has no row-slice layout)
sl_status sl_csr_diag_pass(uint64_t n, const uint32_t *ptr, const uint32_t *idx, const double *val, double *d_dinv,
                           unsigned long long h_status[4]);

// ---- multi-GPU: communicator, partition, partitioned vectors (sl_comm.hip) -----------------------------------------------------
#define SL_COMM_MAX_RANKS 16
#define SL_COMM_RING 32
#define SL_COMM_BLOB 512
#define SL_COMM_MAGIC 0x534c434f4d4d3033ull      /* "SLCOMM03" */
// the shared block of a communicator (POSIX shared memory, mapped by every rank, registered with the HIP runtime)
struct sl_comm_shm {
    volatile uint64_t magic, world, created_unix;                // created_unix: when rank 0 made the block (stale blocks of crashed jobs are not joined)
    volatile uint64_t generation, confirmed;                     // rank 0's nonce of this job; confirmed = generation once every rank of THIS job has joined
    volatile uint64_t arrive[SL_COMM_MAX_RANKS];                 // host barrier
    volatile uint64_t blob_seq[SL_COMM_MAX_RANKS];               // host exchange of small blobs (IPC handles, row ranges)
    volatile unsigned char blob[SL_COMM_MAX_RANKS][SL_COMM_BLOB];
    uint64_t ready[SL_COMM_MAX_RANKS][16];                       // device side: one 128-byte line per rank, [c] = last published ticket of channel c (0: sums, 1: halo ready)
    double value[SL_COMM_RING][SL_COMM_MAX_RANKS];               // device side: the values published with the tickets (ring)
    uint64_t error;                                              // a wait timed out / a rank failed locally (rank + 1): no further collective on this communicator
};
// Transports of the vectors and sums (SL_COMM_TRANSPORT): IPC = peers' buffers mapped with hipIpc*, pulled with hipMemcpyAsync, sums
// and handshakes through tickets in the shared block (ranks may share a GPU); RCCL = the form BASELINE's north_star words:
// ncclAllGather / grouped ncclSend + ncclRecv (or ONE ncclAllReduce over a compact halo buffer, SL_COMM_HALO=allreduce) for the
// vectors, ncclAllGather of the ranks' partial sums (added in rank order by a one-wave kernel) for the norms.  The shared block
// stays the rendezvous (it carries the ncclUniqueId) and the host barrier in both.
enum { SL_TRANSPORT_IPC = 0, SL_TRANSPORT_RCCL = 1 };
struct sl_comm {
    int rank = 0, world = 1, device = 0, fd = -1;
    std::string path;
    sl_comm_shm *h_shm = nullptr, *d_shm = nullptr;
    size_t shm_bytes = 0;
    bool registered = false;
    uint64_t barrier_count = 0, blob_count = 0, ticket[2] = {0, 0};   // advanced in the same order on every rank; ticket[channel]
    int transport = SL_TRANSPORT_IPC;
    bool halo_allreduce = false;                                  // RCCL: halo strips as one all-reduce over a compact buffer instead of sends / receives
    void *nccl = nullptr;                                         // ncclComm_t
    double *d_sums = nullptr;                                     // RCCL: [SL_COMM_RING][world] all-gathered partial sums
    uint64_t sums_seq = 0;
    int refs = 0;                                                 // partitioned states alive on this communicator
    bool closed = false;                                          // sl_comm_destroy was called while states were alive: the last state frees it
};
struct sl_dist_vector { uint64_t n = 0; double *mine = nullptr; std::vector<double *> peer; };
struct sl_dist {
    sl_comm *c = nullptr;
    uint64_t n_global = 0, lo = 0, hi = 0, pull_bytes = 0;
    std::vector<uint64_t> bounds;                                // world + 1
    struct piece { int rank; uint64_t lo, hi; };
    std::vector<piece> need;                                     // what this rank pulls per exchange: global index ranges of its peers' rows
    std::vector<piece> give;                                     // what its peers pull from it: ranges of ITS rows (rank = the peer) — the send list of the RCCL transport
    std::vector<uint64_t> reaches;                               // every rank's reach
    uint64_t reach = 0, max_reach = 0;                           // columns this rank's rows reach beyond its range; the largest over all ranks
    sl_dist_vector t[2], x;                                      // gathered term vectors (ping-pong) and gathered solution
    // large exchanges (uniform columns: every peer's whole range): one copy stream per peer, so that the pulls travel over their
    // xGMI links side by side instead of one after the other (created on first use)
    std::vector<hipStream_t> pull_streams;
    std::vector<hipEvent_t> pull_done;
    hipEvent_t pull_fork = nullptr;
    // RCCL, all-reduce form of the halo: the strips every rank exports (its rows within max_reach of either end of its range), all
    // ranks' strips one after the other in one compact buffer; exp_off[p] = where rank p's strips start, exp_lo/exp_hi the two strips
    double *d_halo = nullptr;
    uint64_t halo_len = 0;
    struct strips { uint64_t off, lo0, hi0, lo1, hi1; };         // rows [lo0, hi0) at off, rows [lo1, hi1) at off + (hi0 - lo0)
    std::vector<strips> exports;
    bool equal_ranges = false, all_to_all = false;               // every rank needs every peer's whole range (uniform columns): ncclAllGather when the ranges are equal
    uint64_t *d_check = nullptr;                                 // sl_dist_verify: checksums of pieces
};
sl_status sl_comm_host_barrier(sl_comm *c);
sl_status sl_comm_allgather_blob(sl_comm *c, const void *mine, size_t bytes, void *all);
sl_status sl_comm_launch_ticket(sl_comm *c, const double *local, double *result, sl_solve_ctl *ctl, uint32_t gate_it, uint32_t slot, int mode,
                                double thr, hipStream_t s, int channel = 0);
bool sl_comm_failed(const sl_comm *c);
sl_status sl_dist_create(sl_comm *c, const sl_matrix *local, sl_dist **out);
void sl_dist_destroy(sl_dist *d);
sl_status sl_dist_vector_create(sl_comm *c, uint64_t n_global, sl_dist_vector *v);
void sl_dist_vector_destroy(sl_comm *c, sl_dist_vector *v);
sl_status sl_dist_pull(sl_dist *d, sl_dist_vector *v, hipStream_t s);
// collective: every piece this rank holds of its peers' rows of `v` against the owner's own copy (wrapping sums of the bit patterns);
// *n_bad = pieces that differ on THIS rank.  The stream must be idle on every rank (the caller synchronises).
sl_status sl_dist_verify(sl_dist *d, sl_dist_vector *v, uint64_t *n_bad);
sl_status sl_comm_agree(sl_comm *c, sl_status mine);              // collective: the first failing rank's status on every rank
void sl_comm_poison(sl_comm *c);                                 // a local failure between collectives: the peers stop waiting for this rank
void sl_comm_release(sl_comm *c);                                // a state lets go of its communicator
const char *sl_comm_transport_name(const sl_comm *c);
