// tools/l2_model.cpp — trace-driven L2 model of the headline kernel (sl_pw_kernel, paced column panels).  CPU only, no device.
//
// What it is for: the GPU pool has been closed to this repository since round 4, and the one question left on the
// headline — how many of the step's L2 requests miss, and what panel width / pacing window / dealing of rows would change that — does not
// need a device to answer: the kernel's request stream is a pure function of the matrix and the layout.  This program
//   1. regenerates S-DD(n, k, seed, w) row by row (the counter-based rule of generators.py / sl_synth.hip),
//   2. restates the paced layout's dealing of rows to tiles and a tile's stream order (sl_matrix.hip: sl_pw_keys_kernel + the stable sort
//      by (tile, panel); tools/l2_model_check.py compares this restatement with the layout the LIBRARY builds, entry for entry),
//   3. replays the launch: 256 persistent blocks of 16 waves, block b on XCD b % 8, every wave walking its tile's stream 256 entries at a
//      time — stream lines two chunks ahead, 4 x 64 gathers per chunk, the pacing rule of the kernel (a wave at most `slack` panels ahead of
//      the slowest wave of its block), the epilogue's vector lines at the end of a tile, round after round without a barrier —
//      through 8 L2s of 4 MiB, 16 ways, 128-byte lines, LRU,
// and counts requests, hits and line fills per launch by class (gathers / stream / epilogue), the way rocprofv3's TCC_HIT / TCC_MISS /
// FETCH_SIZE saw them on the MI355X (profiles/r03_uniform_pmc.txt: hit rate 0.832, 2.79e7 misses, 3.49 GB fetched at n = 1e7 x 16).
// GATE: it must reproduce those two figures before any of its rankings is believed (profiles/r06_l2_model.txt holds the run).
//
// What it is NOT: a timing model.  All waves advance in lockstep ticks (one chunk per tick unless the pacing rule holds a wave back);
// `--jitter p` lets every block sit out a tick with probability p, which is the only drift between blocks the model knows.  It says how many
// lines are filled, not how long the fills take.
//
//   g++ -O2 -std=c++17 -pthread tools/l2_model.cpp -o tools/l2_model
//   tools/l2_model --n 10000000 --k 16 [--w 0] [--pbits 16] [--slack 0=the layout's own] [--jitter 0.0] [--cus 256] [--xcds 8]
//                  [--xcd-spans 0|1] [--l2-mib 4] [--ways 16] [--nt 0|1] [--warm 0] [--lead 2] [--dump-tile t file]
//   --nt 1: the stream's loads are non-temporal in the kernel (__builtin_nontemporal_load); how the L2 treats them is not documented for
//   gfx950 — 0 = like any load (installed most recently used), 1 = installed as the set's next victim.  The gate holds under both.
#include <algorithm>
#include <atomic>
#include <cinttypes>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {
constexpr uint32_t WAVES = 16, GROUP = 16, MAX_ROWS = 1264, CHUNK = 256, LINE = 128;

struct Cfg {
    uint64_t n = 10000000, seed = 1, w = 0;
    uint32_t k = 16, cus = 256, xcds = 8, pbits = 16, slack = 0, ways = 16, warm = 0, lead = 2;
    double jitter = 0.0, l2_mib = 4.0;
    int xcd_spans = 0, nt = 0;
    long dump_tile = -1;
    std::string dump_file;
};

inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
// the k columns of row i of S-DD(n, k, seed, w), ascending (generators.py sdd_rows_at: m = k - 1 stratified off-diagonals + the diagonal)
void row_columns(const Cfg &c, uint64_t i, uint32_t *out)
{
    const uint64_t G = 0x9E3779B97F4A7C15ull, K = 0xD1B54A32D192ED03ull;
    const uint32_t m = c.k - 1;
    const bool banded = c.w && 2 * c.w + 1 < c.n;
    uint64_t lo = 0, sw = c.n / m;
    if (banded) { lo = i > c.w ? i - c.w : 0; const uint64_t hi = std::min<uint64_t>(i + c.w + 1, c.n); sw = (hi - lo) / m; }
    uint32_t cols[64];
    for (uint32_t j = 0; j < m; ++j) {
        const uint64_t key = c.seed * G + (i * 64 + j + 1) * K, h1 = mix64(key), off = h1 % sw;
        uint64_t col = lo + j * sw + off;
        if (col == i) col = off + 1 < sw ? col + 1 : col - 1;
        cols[j] = (uint32_t)col;
    }
    uint32_t p = 0;
    while (p < m && cols[p] < i) ++p;
    for (uint32_t j = 0; j < p; ++j) out[j] = cols[j];
    out[p] = (uint32_t)i;
    for (uint32_t j = p; j < m; ++j) out[j + 1] = cols[j];
}

// the paced layout's geometry (sl_matrix.hip, sl_build_paced_panels; uniform columns and XCD-local spans in row order, no edge-first rounds)
struct Layout {
    uint64_t n_groups, n_tiles, rounds, waves;
    uint32_t deal, gpt, rpw, n_panels, slack, xcd;
    std::vector<uint64_t> span_g0;     // xcd spans: first group of span s
};
Layout make_layout(const Cfg &c)
{
    Layout L{};
    L.waves = (uint64_t)c.cus * WAVES;
    L.n_groups = (c.n + GROUP - 1) / GROUP;
    const uint64_t max_groups = MAX_ROWS / GROUP;
    L.rounds = (L.n_groups + L.waves * max_groups - 1) / (L.waves * max_groups);
    L.n_tiles = std::min<uint64_t>(L.rounds * L.waves, L.n_groups);
    L.xcd = c.xcd_spans ? c.xcds : 0;
    if (L.xcd) L.n_tiles = (L.n_tiles + L.waves / L.xcd - 1) / (L.waves / L.xcd) * (L.waves / L.xcd);
    L.deal = L.xcd ? (uint32_t)(L.waves / L.xcd) : (uint32_t)L.n_tiles;
    L.gpt = (uint32_t)((L.n_groups + L.n_tiles - 1) / L.n_tiles);
    if (L.xcd) {
        const uint64_t S = L.n_tiles / L.deal, nominal = (L.n_groups + S - 1) / S;
        L.span_g0.resize(S + 1);
        for (uint64_t q = 0; q <= S; ++q) L.span_g0[q] = std::min<uint64_t>(q * nominal, L.n_groups);
        L.gpt = (uint32_t)((nominal + L.deal - 1) / L.deal);
    }
    L.rpw = L.gpt * GROUP;
    L.n_panels = (uint32_t)((c.n + (1ull << c.pbits) - 1) >> c.pbits);
    // sl_matrix.hip: the lead a wave is allowed = two thirds of the panels it crosses per chunk, at least one
    const uint64_t total = c.n * c.k;
    const uint64_t tile_cols = c.w ? std::min<uint64_t>(c.n, (L.xcd ? (uint64_t)L.deal * L.rpw : c.n) + 2 * c.w) : c.n;
    const uint64_t tile_panels = (tile_cols >> c.pbits) + 1;
    L.slack = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(1, (2 * 256 * tile_panels * L.n_tiles + total * 3 / 2) / (total * 3)));
    if (c.slack) L.slack = c.slack;
    return L;
}
// the row groups of tile t, ascending
void tile_groups(const Layout &L, uint64_t t, std::vector<uint64_t> &groups)
{
    groups.clear();
    if (!L.xcd) { for (uint64_t q = t; q < L.n_groups; q += L.n_tiles) groups.push_back(q); return; }
    const uint64_t sp = t / L.deal, td = t % L.deal;
    for (uint64_t q = L.span_g0[sp] + td; q < L.span_g0[sp + 1]; q += L.deal) groups.push_back(q);
}
// a tile's stream: its entries in (panel, row, column) order = a stable sort of the CSR order by panel; cols only (what the gathers see)
void tile_stream(const Cfg &c, const Layout &L, uint64_t t, std::vector<uint32_t> &stream, std::vector<uint64_t> &rows)
{
    std::vector<uint64_t> groups;
    tile_groups(L, t, groups);
    rows.clear();
    for (uint64_t q : groups) for (uint32_t r = 0; r < GROUP; ++r) { const uint64_t i = q * GROUP + r; if (i < c.n) rows.push_back(i); }
    std::vector<uint32_t> cols(rows.size() * c.k);
    for (size_t r = 0; r < rows.size(); ++r) row_columns(c, rows[r], cols.data() + r * c.k);
    std::vector<uint32_t> count(L.n_panels + 1, 0);
    for (uint32_t col : cols) ++count[(col >> c.pbits) + 1];
    for (uint32_t p = 0; p < L.n_panels; ++p) count[p + 1] += count[p];
    stream.resize(cols.size());
    for (uint32_t col : cols) stream[count[col >> c.pbits]++] = col;
    // bridging entries (an empty super-panel of 2^20 columns between two entries) and the padding of the last chunk gather too — from column
    // 0 of the current super-panel; they are a handful per tile and are left out of the model
}

struct L2 {
    uint32_t sets, ways;
    std::vector<uint64_t> tag;      // sets * ways, MRU first; ~0 = empty
    L2(double mib, uint32_t w) : sets((uint32_t)(mib * 1048576.0 / LINE / w)), ways(w), tag((size_t)sets * w, ~0ull) {}
    // returns true on hit.  streaming = true (a non-temporal load, --nt 1): a miss installs the line at the LRU end of its set, a hit does not
    // refresh it — the line is the set's next victim, so a stream never pushes the gathered vector's lines out
    bool access(uint64_t line, bool streaming = false)
    {
        uint64_t *s = tag.data() + (size_t)(line % sets) * ways;      // (sets is not a power of two for odd sizes: plain modulo)
        for (uint32_t i = 0; i < ways; ++i)
            if (s[i] == line) { if (!streaming) { for (uint32_t j = i; j > 0; --j) s[j] = s[j - 1]; s[0] = line; } return true; }
        if (streaming) s[ways - 1] = line;
        else { for (uint32_t j = ways - 1; j > 0; --j) s[j] = s[j - 1]; s[0] = line; }
        return false;
    }
};

struct Counts {
    uint64_t gather_req = 0, gather_hit = 0, stream_req = 0, stream_hit = 0, epi_rd_req = 0, epi_rd_hit = 0, epi_wr_req = 0, epi_wr_hit = 0, stalls = 0, ticks = 0;
    void add(const Counts &o)
    {
        gather_req += o.gather_req; gather_hit += o.gather_hit; stream_req += o.stream_req; stream_hit += o.stream_hit; epi_rd_req += o.epi_rd_req;
        epi_rd_hit += o.epi_rd_hit; epi_wr_req += o.epi_wr_req; epi_wr_hit += o.epi_wr_hit; stalls += o.stalls; ticks = std::max(ticks, o.ticks);
    }
};

// address map (byte addresses, far apart): the gathered vector, the three epilogue vectors, the two stream arrays
constexpr uint64_t VEC = 0, DINV = 1ull << 40, XV = 2ull << 40, TOUT = 3ull << 40, SIDX = 4ull << 40, SVAL = 5ull << 40;

struct Wave {
    uint32_t round = 0, chunk = 0, chunks = 0, loaded = 0, warmed = 0;    // loaded / warmed = stream chunks requested so far by the wave / by a warmer block
    uint64_t tile = 0, ch0 = 0;                               // ch0 = first chunk of the tile's stream in the stream arrays
    std::vector<uint32_t> stream;
    std::vector<uint64_t> rows;
    bool done = false;
};

void simulate_xcd(const Cfg &c, const Layout &L, uint32_t xcd, const std::vector<uint64_t> &tile_ch0, Counts &out, std::vector<uint64_t> *touched)
{
    const uint32_t blocks_here = c.cus / c.xcds;
    L2 l2(c.l2_mib, c.ways);
    std::mt19937_64 rng(12345 + xcd);
    std::uniform_real_distribution<double> uni(0.0, 1.0);
    Counts ct;
    // warmers: the first `warm` blocks of the XCD take no tiles; they read the consumers' stream `lead` chunks ahead (the split of DESIGN §10)
    const uint32_t consumers = blocks_here - std::min(c.warm, blocks_here - 1);
    const uint32_t nblocks_total = consumers * c.xcds;           // consumer blocks of the launch: they share the tiles
    const uint64_t waves_c = (uint64_t)nblocks_total * WAVES;
    const uint32_t rounds = (uint32_t)((L.n_tiles + waves_c - 1) / waves_c);
    std::vector<std::vector<Wave>> blk(consumers, std::vector<Wave>(WAVES));
    std::vector<std::vector<uint32_t>> prog(consumers, std::vector<uint32_t>(WAVES, 0));
    auto lblock_of = [&](uint32_t local) {                        // global consumer block id -> logical block (sl_pw_kernel: XCD-local spans)
        const uint32_t bidx = local * c.xcds + xcd;
        return L.xcd ? (bidx % L.xcd) * (nblocks_total / L.xcd) + bidx / L.xcd : bidx;
    };
    auto start_tile = [&](Wave &wv, uint32_t local, uint32_t wave) {
        for (;; ++wv.round) {
            if (wv.round >= rounds) { wv.done = true; return; }
            wv.tile = ((uint64_t)wv.round * nblocks_total + lblock_of(local)) * WAVES + wave;
            if (wv.tile < L.n_tiles) break;
        }
        tile_stream(c, L, wv.tile, wv.stream, wv.rows);
        wv.chunks = (uint32_t)((wv.stream.size() + CHUNK - 1) / CHUNK);
        wv.chunk = 0; wv.loaded = 0; wv.warmed = 0; wv.ch0 = tile_ch0[wv.tile];
    };
    auto touch = [&](uint64_t line) { if (touched) touched->push_back(line); };
    auto stream_load = [&](Wave &wv, uint32_t ch, bool by_warmer) {
        // one chunk of the stream: 1 KiB of index words + 2 KiB of values = 8 + 16 lines
        const uint64_t ci = wv.ch0 + ch;
        for (uint32_t l = 0; l < 8; ++l) { const uint64_t line = (SIDX + ci * 1024) / LINE + l; ++ct.stream_req; if (l2.access(line, c.nt != 0)) ++ct.stream_hit; else touch(line); }
        for (uint32_t l = 0; l < 16; ++l) { const uint64_t line = (SVAL + ci * 2048) / LINE + l; ++ct.stream_req; if (l2.access(line, c.nt != 0)) ++ct.stream_hit; else touch(line); }
        (void)by_warmer;
    };
    for (uint32_t b = 0; b < consumers; ++b) for (uint32_t w = 0; w < WAVES; ++w) start_tile(blk[b][w], b, w);
    uint64_t live = 0;
    for (auto &bw : blk) for (auto &wv : bw) live += !wv.done;
    uint32_t line_buf[64];
    while (live) {
        ++ct.ticks;
        for (uint32_t b = 0; b < consumers; ++b) {
            if (c.jitter > 0.0 && uni(rng) < c.jitter) continue;
            for (uint32_t w = 0; w < WAVES; ++w) {
                Wave &wv = blk[b][w];
                if (wv.done) continue;
                // pace(): publish where this wave is about to gather, then compare with the slowest wave of the block
                const uint32_t pan = wv.chunk < wv.chunks ? wv.stream[(size_t)wv.chunk * CHUNK] >> c.pbits : 0;
                const uint32_t me = (wv.round << 20) + pan + 1u;
                prog[b][w] = me;
                uint32_t mn = 0xffffffffu;
                for (uint32_t v = 0; v < WAVES; ++v) mn = std::min(mn, prog[b][v]);
                if (L.slack < (1u << 20) && me > mn + L.slack) { ++ct.stalls; continue; }
                // stream: two chunks ahead of the gathers (three buffers); a warmer block, if any, has read it `lead` chunks before that
                // (its requests are requests too: with warmers every stream line is asked for twice, once by each role)
                if (c.warm) { const uint32_t ahead = std::min(wv.chunks, wv.chunk + 3u + c.lead); while (wv.warmed < ahead) stream_load(wv, wv.warmed++, true); }
                const uint32_t want = std::min(wv.chunks, wv.chunk + 3u);
                while (wv.loaded < want) stream_load(wv, wv.loaded++, false);
                // gathers of this chunk: 4 instructions of 64 lanes; lanes of one instruction that fall into the same line are one request
                const size_t e0 = (size_t)wv.chunk * CHUNK, e1 = std::min(wv.stream.size(), e0 + CHUNK);
                for (size_t u0 = e0; u0 < e1; u0 += 64) {
                    uint32_t nl = 0;
                    for (size_t e = u0; e < std::min(e1, u0 + 64); ++e) line_buf[nl++] = (uint32_t)(((uint64_t)wv.stream[e] * 8) / LINE);
                    std::sort(line_buf, line_buf + nl);
                    nl = (uint32_t)(std::unique(line_buf, line_buf + nl) - line_buf);
                    for (uint32_t q = 0; q < nl; ++q) { ++ct.gather_req; if (l2.access(VEC / LINE + line_buf[q])) ++ct.gather_hit; else touch(VEC / LINE + line_buf[q]); }
                }
                if (++wv.chunk < wv.chunks) continue;
                // epilogue: per group of 16 rows one line each of t_in (the gathered vector), dinv, x read; t_out, x written
                prog[b][w] = (wv.round + 1u) << 20;
                for (size_t r = 0; r < wv.rows.size(); r += GROUP) {
                    const uint64_t li = wv.rows[r] * 8 / LINE;
                    for (uint64_t base : {VEC, DINV, XV}) { ++ct.epi_rd_req; if (l2.access(base / LINE + li)) ++ct.epi_rd_hit; else touch(base / LINE + li); }
                    for (uint64_t base : {TOUT, XV}) { ++ct.epi_wr_req; if (l2.access(base / LINE + li)) ++ct.epi_wr_hit; }
                }
                ++wv.round;
                start_tile(wv, b, w);
                if (wv.done) { prog[b][w] = 0xffffffffu; --live; }
            }
        }
    }
    out = ct;
}
} // namespace

int main(int argc, char **argv)
{
    Cfg c;
    for (int i = 1; i < argc; ++i) {
        auto is = [&](const char *s) { return !strcmp(argv[i], s) && i + 1 < argc; };
        if (is("--n")) c.n = strtoull(argv[++i], nullptr, 10);
        else if (is("--k")) c.k = (uint32_t)atoi(argv[++i]);
        else if (is("--seed")) c.seed = strtoull(argv[++i], nullptr, 10);
        else if (is("--w")) c.w = strtoull(argv[++i], nullptr, 10);
        else if (is("--cus")) c.cus = (uint32_t)atoi(argv[++i]);
        else if (is("--xcds")) c.xcds = (uint32_t)atoi(argv[++i]);
        else if (is("--pbits")) c.pbits = (uint32_t)atoi(argv[++i]);
        else if (is("--slack")) c.slack = (uint32_t)atoi(argv[++i]);
        else if (is("--jitter")) c.jitter = atof(argv[++i]);
        else if (is("--l2-mib")) c.l2_mib = atof(argv[++i]);
        else if (is("--ways")) c.ways = (uint32_t)atoi(argv[++i]);
        else if (is("--xcd-spans")) c.xcd_spans = atoi(argv[++i]);
        else if (is("--nt")) c.nt = atoi(argv[++i]);
        else if (is("--warm")) c.warm = (uint32_t)atoi(argv[++i]);
        else if (is("--lead")) c.lead = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--dump-tile") && i + 2 < argc) { c.dump_tile = atol(argv[++i]); c.dump_file = argv[++i]; }
        else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
    }
    if (c.k < 2 || c.k > 64 || c.cus % c.xcds) { fprintf(stderr, "k in [2, 64], cus a multiple of xcds\n"); return 2; }
    const Layout L = make_layout(c);
    if (c.dump_tile >= 0) {      // the restated layout of one tile, for tools/l2_model_check.py: rows (u64) then the stream's columns (u32)
        std::vector<uint32_t> s; std::vector<uint64_t> rows;
        tile_stream(c, L, (uint64_t)c.dump_tile, s, rows);
        FILE *f = fopen(c.dump_file.c_str(), "wb");
        if (!f) return 1;
        const uint64_t hdr[6] = {L.n_tiles, L.rpw, L.deal, L.slack, rows.size(), s.size()};
        fwrite(hdr, 8, 6, f); fwrite(rows.data(), 8, rows.size(), f); fwrite(s.data(), 4, s.size(), f);
        fclose(f);
        return 0;
    }
    // where every tile's stream starts (in chunks), as dst[] of the build: tiles in order, each padded to whole chunks
    std::vector<uint64_t> tile_ch0(L.n_tiles + 1, 0);
    {
        std::vector<uint64_t> groups;
        for (uint64_t t = 0; t < L.n_tiles; ++t) {
            tile_groups(L, t, groups);
            uint64_t rows = 0;
            for (uint64_t q : groups) rows += std::min<uint64_t>(GROUP, c.n - q * GROUP);
            tile_ch0[t + 1] = tile_ch0[t] + (rows * c.k + CHUNK - 1) / CHUNK;
        }
    }
    std::vector<Counts> per(c.xcds);
    std::vector<std::vector<uint64_t>> touched(c.xcds);
    std::vector<std::thread> th;
    for (uint32_t x = 0; x < c.xcds; ++x) th.emplace_back([&, x] { simulate_xcd(c, L, x, tile_ch0, per[x], &touched[x]); });
    for (auto &t : th) t.join();
    Counts s;
    for (auto &p : per) s.add(p);
    // lines filled by more than one L2 (the other XCDs' fills of a line may be served by the memory-side cache, not by HBM)
    std::vector<uint64_t> all;
    for (auto &v : touched) { all.insert(all.end(), v.begin(), v.end()); std::vector<uint64_t>().swap(v); }
    const uint64_t fills = all.size();
    std::sort(all.begin(), all.end());
    const uint64_t distinct = (uint64_t)(std::unique(all.begin(), all.end()) - all.begin());
    const uint64_t req = s.gather_req + s.stream_req + s.epi_rd_req + s.epi_wr_req, hit = s.gather_hit + s.stream_hit + s.epi_rd_hit + s.epi_wr_hit;
    const uint64_t rd_miss = (s.gather_req - s.gather_hit) + (s.stream_req - s.stream_hit) + (s.epi_rd_req - s.epi_rd_hit);
    printf("{\"n\": %" PRIu64 ", \"k\": %u, \"w\": %" PRIu64 ", \"cus\": %u, \"xcds\": %u, \"pbits\": %u, \"slack\": %u, \"jitter\": %.3f, \"xcd_spans\": %d, \"nt\": %d, \"warm\": %u, \"lead\": %u, "
           "\"l2_mib\": %.2f, \"ways\": %u, \"tiles\": %" PRIu64 ", \"rounds\": %" PRIu64 ", \"rows_per_tile\": %u, \"panels\": %u, "
           "\"requests\": %" PRIu64 ", \"hits\": %" PRIu64 ", \"misses\": %" PRIu64 ", \"hit_rate\": %.4f, "
           "\"gather\": {\"req\": %" PRIu64 ", \"hit\": %" PRIu64 ", \"miss\": %" PRIu64 "}, \"stream\": {\"req\": %" PRIu64 ", \"miss\": %" PRIu64 "}, "
           "\"epilogue\": {\"rd_req\": %" PRIu64 ", \"rd_miss\": %" PRIu64 ", \"wr_req\": %" PRIu64 ", \"wr_miss\": %" PRIu64 "}, "
           "\"read_fill_bytes\": %" PRIu64 ", \"distinct_lines_filled\": %" PRIu64 ", \"fills\": %" PRIu64 ", \"pacing_stalls\": %" PRIu64 ", \"ticks\": %" PRIu64 "}\n",
           c.n, c.k, c.w, c.cus, c.xcds, c.pbits, L.slack, c.jitter, c.xcd_spans, c.nt, c.warm, c.lead, c.l2_mib, c.ways, L.n_tiles, L.rounds, L.rpw, L.n_panels, req, hit,
           req - hit, (double)hit / (double)req, s.gather_req, s.gather_hit, s.gather_req - s.gather_hit, s.stream_req, s.stream_req - s.stream_hit, s.epi_rd_req,
           s.epi_rd_req - s.epi_rd_hit, s.epi_wr_req, s.epi_wr_req - s.epi_wr_hit, rd_miss * LINE, distinct, fills, s.stalls, s.ticks);
    return 0;
}
