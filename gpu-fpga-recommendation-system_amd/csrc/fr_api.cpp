// Host side of the C-ABI (include/fleetrec.h): context = device + model + tables + weights,
// worker = stream + staging buffers, submit/sync = the hot-loop body of the reference's
// thread_consume() (GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c:101-503) with
// the FPGA embedding stage (FPGA/kernel/user_krnl/embedding_*_krnl) pulled in front of the FC chain.
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "fr_internal.h"

// ---- errors -------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void fr_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#ifdef FR_EXPERIMENTS
int fr_knob_env(const char *name, int dflt) {  // experiments build only: the product library reads no environment variable
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#endif

static thread_local char g_kernel[96] = "";
void fr_note_kernel(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
const char *fr_noted_kernel() { return g_kernel; }
static void keep_kernel(fr_worker *w) { snprintf(w->last_kernel, sizeof(w->last_kernel), "%s", g_kernel); }
extern "C" const char *fr_worker_last_kernel(const fr_worker *w) { return w ? w->last_kernel : ""; }

extern "C" const char *fr_last_error(void) { return g_err; }
extern "C" int fr_abi_version(void) { return FR_ABI_VERSION; }

extern "C" int fr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int fr_cpu_set_threads(int n) {
    if (n < 0) FR_FAIL(FR_ERR_INVALID, "fr_cpu_set_threads(%d): n >= 1, or 0 for every usable core", n);
    return frc_set_threads(n);
}

static int select_device(int device) {
    int n = fr_device_count();
    if (n <= 0) FR_FAIL(FR_ERR_NO_DEVICE, "no HIP device visible (device >= 0 never falls back to the CPU; device = -1 asks for the CPU back-end)");
    if (device < 0 || device >= n) FR_FAIL(FR_ERR_NO_DEVICE, "device %d not available (%d visible)", device, n);
    hipDeviceProp_t prop;
    FR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        FR_FAIL(FR_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code objects only", device, prop.gcnArchName);
    FR_HIP(hipSetDevice(device));
    return FR_OK;
}

// every entry point selects its context's device first; a CPU context (device = -1, fr_cpu.cpp) has none
#define FR_SET_DEVICE(ctx_)                                    \
    do {                                                       \
        if (!(ctx_)->cpu) FR_HIP(hipSetDevice((ctx_)->device)); \
    } while (0)

static unsigned long long *g_stamp_buffer = nullptr;  // diagnostics (tools/experiments): see fr_debug_set_stamp_buffer
extern "C" __attribute__((visibility("default"))) void fr_debug_set_stamp_buffer(void *dptr) { g_stamp_buffer = (unsigned long long *)dptr; }

static int fused_group_initial();
static bool fused_eligible(const fr_ctx *c);
static bool lp_image_applies_hs(const fr_ctx *c);
static int lp_ensure_image(fr_ctx *c, int prec);
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int round_up(int v, int a) { return (v + a - 1) / a * a; }

// ---- context --------------------------------------------------------------------------------------
static void ctx_free(fr_ctx *c) {
    if (!c) return;
    if (c->cpu) {   // the CPU back-end: host memory only
        frc_arena_free(c->table_arena, c->table_arena_bytes);
        for (int i = 0; i < 4; i++) free(c->d_w[i]);
        delete c;
        return;
    }
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->table_arena) (void)hipFree(c->table_arena);
    if (c->lp_arena) (void)hipFree(c->lp_arena);
    if (c->d_words_lp) (void)hipFree(c->d_words_lp);
    if (c->d_words) (void)hipFree(c->d_words);
    if (c->d_passes) (void)hipFree(c->d_passes);
    if (c->d_chunks) (void)hipFree(c->d_chunks);
    if (c->d_merged) (void)hipFree(c->d_merged);
    for (int i = 0; i < 4; i++) {
        if (c->d_w[i]) (void)hipFree(c->d_w[i]);
        if (i < 3 && c->d_wq[i]) (void)hipFree(c->d_wq[i]);
        if (c->d_w_bf16[i]) (void)hipFree(c->d_w_bf16[i]);
        if (i < 3 && c->d_w_fp8[i]) (void)hipFree(c->d_w_fp8[i]);
        if (i < 3 && c->d_w_fp8h[i]) (void)hipFree(c->d_w_fp8h[i]);
    }
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->setup_stream) (void)hipStreamDestroy(c->setup_stream);
    delete c;
}

// Contiguous, float-balanced split of the record's segments over n_shards (SURVEY section 8(e):
// "partition tables by table-ID ... balancing floats-per-item ... slices are whole segments").
void fr_shard_bounds(const fr_model_desc &m, int n_shards, std::vector<int> &seg_begin) {
    seg_begin.assign(n_shards + 1, m.n_segments);
    seg_begin[0] = 0;
    int s = 0;
    for (int g = 1; g < n_shards; g++) {
        const double target = (double)m.record_len * g / n_shards;
        // advance while the end of segment s is closer to (or before) the target than its start
        while (s < m.n_segments && m.segments[s].rec_offset + m.segments[s].len / 2.0 <= target) s++;
        // a COPY segment must stay with the table segment it duplicates only for locality, not correctness
        if (s < g) s = g;  // at least one segment per shard
        if (s > m.n_segments - (n_shards - g)) s = m.n_segments - (n_shards - g);
        seg_begin[g] = s;
    }
}

// XCD partition of the record words for gather_pack_xcd_kernel (FrGatherGroups): 8 contiguous runs of words cut on SOURCE-ROW
// boundaries, balanced by a cost per word run.  An "atom" is a maximal run of consecutive record words that one item reads from ONE
// source row -- a table row, a whole bank row in a bank-interleaved context (one contiguous 112-256-byte fetch), 128 bytes of the
// dense block.  Cutting inside an atom makes two XCDs fetch the same 128-byte line through two L2s: with Model-C's 992 words dealt
// 124 per group, 6 of the 7 cuts split a bank row -- 6 extra lines on 142 per item.  Cost of an atom = the bytes it writes + the
// 128-byte lines it fetches, weighted by where its table lives (FR_GATHER_COST = "write,l2,cache,hbm" weights; tables up to 2 MiB
// per XCD group are taken as L2-resident, up to 64 MiB as Infinity-Cache-resident); a linear-partition DP minimises the costliest
// group, then the sum of squares.  Groups wider than 256 words (the kernel's block) leave max_words = 0: the launcher then deals
// n_words / 8 words per group as before.
static void plan_gather_groups(fr_ctx *c) {
    FrGatherGroups &gg = c->gather_groups;
    gg = FrGatherGroups{};
    const int n = c->n_words;
    if (n < 64) return;
    double wt[4] = {1.0, 0.0, 0.0, 0.0};  // write, fetch from L2-class, cache-class, HBM-class tables
#ifdef FR_EXPERIMENTS
    if (const char *e = getenv("FR_GATHER_COST")) sscanf(e, "%lf,%lf,%lf,%lf", &wt[0], &wt[1], &wt[2], &wt[3]);
#endif
    struct Atom { int w0, nw; double cost; };
    std::vector<Atom> atoms;
    auto close_atom = [&](int w0, int w1) {
        const FrWordDesc &d = c->h_words[w0];
        const size_t bytes = (size_t)(w1 - w0) * 16;
        double fetch = 0.0;
        if (!(d.idx_col & FR_DESC_DENSE)) {
            size_t acc = 0;  // expected 128-byte lines per fetched row, over one period of the stride
            for (size_t r = 0; r < 128; r++) acc += ((d.src + r * d.stride) % 128 + bytes + 127) / 128;
            const double lines = (double)acc / 128.0;
            const double footprint = (double)d.rows * d.stride;
            fetch = lines * 128.0 * (footprint <= 2.0 * 1048576 ? wt[1] : footprint <= 64.0 * 1048576 ? wt[2] : wt[3]);
        }
        atoms.push_back(Atom{w0, w1 - w0, wt[0] * (double)bytes + fetch});
    };
    int a0 = 0;
    for (int w = 1; w <= n; w++) {
        bool same = false;
        if (w < n) {
            const FrWordDesc &p = c->h_words[w - 1], &q = c->h_words[w];
            const bool pd = (p.idx_col & FR_DESC_DENSE) != 0, qd = (q.idx_col & FR_DESC_DENSE) != 0;
            if (pd && qd) same = q.src == p.src + 16 && (w - a0) % 8 != 0;
            else if (!pd && !qd) same = p.idx_col == q.idx_col && p.stride == q.stride && (q.src > p.src ? q.src - p.src : p.src - q.src) < p.stride;
        }
        if (!same) {
            close_atom(a0, w);
            a0 = w;
        }
    }
    const int na = (int)atoms.size();
    if (na < 8) return;
    std::vector<double> pc(na + 1, 0.0);
    std::vector<int> pw(na + 1, 0);
    for (int i = 0; i < na; i++) {
        pc[i + 1] = pc[i] + atoms[i].cost;
        pw[i + 1] = pw[i] + atoms[i].nw;
    }
    const double INF = 1e300;
    struct Cell { double mx, sq; int from; };
    std::vector<std::vector<Cell>> dp(9, std::vector<Cell>(na + 1, Cell{INF, INF, -1}));
    dp[0][0] = Cell{0.0, 0.0, -1};
    for (int g = 1; g <= 8; g++)
        for (int i = g; i <= na - (8 - g); i++)
            for (int j = g - 1; j < i; j++) {
                if (dp[g - 1][j].mx >= INF || pw[i] - pw[j] > 256) continue;
                const double cst = pc[i] - pc[j];
                const double mx = dp[g - 1][j].mx > cst ? dp[g - 1][j].mx : cst, sq = dp[g - 1][j].sq + cst * cst;
                Cell &t = dp[g][i];
                if (mx < t.mx * (1.0 - 1e-12) || (mx <= t.mx * (1.0 + 1e-12) && sq < t.sq)) t = Cell{mx, sq, j};
            }
    if (dp[8][na].mx >= INF) return;
    int i = na;
    gg.start[8] = n;
    for (int g = 8; g >= 1; g--) {
        const int j = dp[g][i].from;
        gg.start[g - 1] = atoms[j].w0;
        const int width = gg.start[g] - gg.start[g - 1];
        if (width > gg.max_words) gg.max_words = width;
        i = j;
    }
}

static int build_words(fr_ctx *c) {
    const fr_model_desc &m = c->model;
    std::vector<int> seg_begin;
    int s0 = 0, s1 = m.n_segments;
    if (c->n_shards > 1) {
        fr_shard_bounds(m, c->n_shards, seg_begin);
        s0 = seg_begin[c->shard_rank];
        s1 = seg_begin[c->shard_rank + 1];
        int maxlen = 0;
        c->shard_offset.assign(c->n_shards, 0);
        c->shard_len.assign(c->n_shards, 0);
        for (int g = 0; g < c->n_shards; g++) {
            int b = seg_begin[g], e = seg_begin[g + 1];
            int len = (e > b) ? (m.segments[e - 1].rec_offset + m.segments[e - 1].len - m.segments[b].rec_offset) : 0;
            c->shard_offset[g] = (e > b) ? m.segments[b].rec_offset : 0;
            c->shard_len[g] = len;
            if (len > maxlen) maxlen = len;
        }
        c->slice_offset = m.segments[s0].rec_offset;
        c->slice_len = m.segments[s1 - 1].rec_offset + m.segments[s1 - 1].len - c->slice_offset;
        c->slice_padded = maxlen;
    } else {
        c->slice_offset = 0;
        c->slice_len = m.record_len;
        c->slice_padded = m.record_len;
        c->shard_offset.assign(1, 0);
        c->shard_len.assign(1, m.record_len);
    }
    // which tables are resident on this shard: those referenced by its segments
    for (auto &tm : c->table_mem) tm.resident = false;
    for (int s = s0; s < s1; s++)
        if (m.segments[s].kind != FR_SEG_DENSE) c->table_mem[m.segments[s].src].resident = true;
    // arena layout
    c->n_banks = fr_bank_map(m, c->bank_of_table, c->bank_rows);
    size_t off = 0;
    std::vector<char> placed(m.n_tables, 0);
    if (m.index_mode == FR_INDEX_PER_BANK) {
        // Bank-interleaved regions: the resident tables of one bank share "bank rows" -- row r of every table side by side -- for the
        // rows every one of them has (r < min rows = the valid range of the bank's index); one bank costs one contiguous fetch per
        // item.  The bank-row stride is padded when that lowers the expected number of 128-byte lines a row touches (112 -> 128,
        // 224 -> 256 bytes: a fetch beyond L2 costs whole lines, profiles/archive/r01_experiments.md).
        for (int b = 0; b < c->n_banks; b++) {
            std::vector<int> members;
            size_t payload = 0;
            for (int t = 0; t < m.n_tables; t++)
                if (c->bank_of_table[t] == b && c->table_mem[t].resident) {
                    members.push_back(t);
                    payload += (size_t)m.tables[t].dim * 4;
                }
            if (members.size() < 2) continue;  // a lone table is its own bank row already
            auto lines_x128 = [](size_t stride, size_t bytes) {  // expected 128-byte lines per row, x128 (exact over one period)
                size_t acc = 0;
                for (size_t r = 0; r < 128; r++) acc += ((r * stride) % 128 + bytes + 127) / 128;
                return acc;
            };
            size_t best = payload;
            for (size_t cand : {align_up(payload, 32), align_up(payload, 64), align_up(payload, 128)})
                if (lines_x128(cand, payload) < lines_x128(best, payload)) best = cand;
            const uint64_t il_rows = (uint64_t)c->bank_rows[b];
            off = align_up(off, 256);
            size_t col = 0;
            for (int t : members) {
                FrTableMem &tm = c->table_mem[t];
                tm.byte_offset = off + col;
                tm.row_stride = best;
                tm.il_rows = il_rows;
                col += (size_t)m.tables[t].dim * 4;
                placed[t] = 1;
            }
            off = align_up(off + (size_t)il_rows * best, 256);
            for (int t : members) {  // rows beyond the bank's common range: contiguous, reachable only by upload / download / fill
                FrTableMem &tm = c->table_mem[t];
                tm.tail_offset = off;
                off = align_up(off + ((size_t)m.tables[t].rows - (size_t)il_rows) * m.tables[t].dim * 4, 256);
            }
        }
    }
    for (int t = 0; t < m.n_tables; t++) {
        if (!c->table_mem[t].resident || placed[t]) continue;
        c->table_mem[t].byte_offset = off;
        c->table_mem[t].row_stride = (uint64_t)m.tables[t].dim * 4;
        c->table_mem[t].il_rows = 0;
        off = align_up(off + (size_t)m.tables[t].rows * m.tables[t].dim * 4, 256);
    }
    c->table_arena_bytes = off;
    if (off && c->cpu) {
        c->table_arena = (char *)frc_arena_alloc(off);
        if (!c->table_arena) return FR_ERR_OOM;
    } else if (off) {
        FR_HIP(hipMalloc((void **)&c->table_arena, off));
    }

    // source runs (for the BLOCKED layout)
    int src_start[3] = {0, 0, 0}, src_len[3] = {0, 0, 0};
    for (int s = 0; s < m.n_segments; s++) {
        const fr_segment &g = m.segments[s];
        if (src_len[g.source] == 0) src_start[g.source] = g.rec_offset;
        src_len[g.source] += g.len;
    }
    c->h_words.clear();
    c->h_word_table.clear();
    c->h_word_col.clear();
    for (int s = s0; s < s1; s++) {
        const fr_segment &g = m.segments[s];
        for (int j = 0; j < g.len / 4; j++) {
            FrWordDesc w{};
            c->h_word_table.push_back(g.kind == FR_SEG_DENSE ? -1 : g.src);
            c->h_word_col.push_back(g.src_col + 4 * j);
            if (g.kind == FR_SEG_DENSE) {
                w.src = (uint64_t)(g.src_col + 4 * j) * 4;
                w.stride = (uint32_t)m.dense_len * 4;
                w.idx_col = FR_DESC_DENSE;
                w.rows = 0xFFFFFFFFu;
            } else {
                const fr_table_desc &t = m.tables[g.src];
                w.src = (uint64_t)(uintptr_t)(c->table_arena + c->table_mem[g.src].byte_offset) + (uint64_t)(g.src_col + 4 * j) * 4;
                w.stride = (uint32_t)c->table_mem[g.src].row_stride;
                w.idx_col = m.index_mode == FR_INDEX_PER_TABLE ? (uint32_t)g.src : (m.index_mode == FR_INDEX_PER_BANK ? (uint32_t)c->bank_of_table[g.src] : 0u);
                // PER_BANK: the bank's index addresses every table of the bank, so it must stay below the smallest of them
                w.rows = m.index_mode == FR_INDEX_PER_BANK ? (uint32_t)c->bank_rows[c->bank_of_table[g.src]] : (uint32_t)t.rows;
            }
            const int rec_pos = g.rec_offset + 4 * j;
            if (c->n_shards > 1) {
                w.dst_off = (uint32_t)(rec_pos - c->slice_offset) / 4;
                w.dst_stride = (uint32_t)c->slice_padded / 4;
                w.dst_blk = 0;
            } else if (m.layout == FR_LAYOUT_BLOCKED) {
                w.dst_off = (uint32_t)(rec_pos - src_start[g.source]) / 4;
                w.dst_stride = (uint32_t)src_len[g.source] / 4;
                w.dst_blk = (uint32_t)src_start[g.source] / 4;
            } else {
                w.dst_off = (uint32_t)rec_pos / 4;
                w.dst_stride = (uint32_t)m.record_len / 4;
                w.dst_blk = 0;
            }
            c->h_words.push_back(w);
        }
    }
    c->n_words = (int)c->h_words.size();
    if (c->cpu) return FR_OK;   // the CPU back-end walks h_words
    plan_gather_groups(c);
    FR_HIP(hipMalloc((void **)&c->d_words, sizeof(FrWordDesc) * c->n_words));
    FR_HIP(hipMemcpy(c->d_words, c->h_words.data(), sizeof(FrWordDesc) * c->n_words, hipMemcpyHostToDevice));
    // Item-tile gather plan (gather_tile_kernel): every record segment is cut in power-of-two pieces of <= 16 words, a piece of w words
    // becomes w passes (64 / w items each), consecutive pieces are grouped in chunks of <= FR_TILE_WORDS words.  The destination must
    // be item-major with one stride (SEMANTIC layout or a shard's slice).
    c->n_chunks = 0;
    if (m.layout == FR_LAYOUT_SEMANTIC || c->n_shards > 1) {
        std::vector<FrPassDesc> passes;
        std::vector<FrChunkDesc> chunks;
        FrChunkDesc cur{0, 0, 0, 0};
        int w = 0;  // index into h_words: words are listed segment by segment, in destination order
        for (int sg = s0; sg < s1; sg++) {
            int left = m.segments[sg].len / 4;
            while (left > 0) {
                int pw = 16;
                while (pw > left) pw >>= 1;
                if (cur.n_words + pw > FR_TILE_WORDS) {
                    chunks.push_back(cur);
                    cur = FrChunkDesc{(int)passes.size(), (int)passes.size(), 0, 0};
                }
                const FrWordDesc &wd = c->h_words[w];
                if (cur.n_words == 0) cur.word0 = (int)wd.dst_off;
                int lg = 0;
                while ((1 << lg) < pw) lg++;
                for (int k = 0; k < pw; k++) {
                    FrPassDesc pd{};
                    pd.src = wd.src;
                    pd.stride = wd.stride;
                    pd.idx_col = wd.idx_col;
                    pd.rows = wd.rows;
                    pd.tile_word = (uint16_t)((int)wd.dst_off - cur.word0);
                    pd.log2_words = (uint8_t)lg;
                    pd.item0 = (uint8_t)(k * (FR_TILE_ITEMS / pw));
                    passes.push_back(pd);
                }
                cur.n_words += pw;
                cur.pass_end = (int)passes.size();
                w += pw;
                left -= pw;
            }
        }
        if (cur.n_words) chunks.push_back(cur);
        c->n_chunks = (int)chunks.size();
        if (c->n_chunks) {
            FR_HIP(hipMalloc((void **)&c->d_passes, sizeof(FrPassDesc) * passes.size()));
            FR_HIP(hipMemcpy(c->d_passes, passes.data(), sizeof(FrPassDesc) * passes.size(), hipMemcpyHostToDevice));
            FR_HIP(hipMalloc((void **)&c->d_chunks, sizeof(FrChunkDesc) * chunks.size()));
            FR_HIP(hipMemcpy(c->d_chunks, chunks.data(), sizeof(FrChunkDesc) * chunks.size(), hipMemcpyHostToDevice));
            FR_HIP(hipMalloc((void **)&c->d_merged, sizeof(unsigned long long)));
            FR_HIP(hipMemset(c->d_merged, 0, sizeof(unsigned long long)));
        }
    }
    return FR_OK;
}

extern "C" int fr_ctx_create_sharded(const fr_model_desc *m, int device, int shard_rank, int n_shards, fr_ctx **out) {
    if (!out) FR_FAIL(FR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int rc = fr_model_validate(m);
    if (rc) return rc;
    if (n_shards < 1 || shard_rank < 0 || shard_rank >= n_shards) FR_FAIL(FR_ERR_INVALID, "bad shard %d of %d", shard_rank, n_shards);
    if (n_shards > m->n_segments) FR_FAIL(FR_ERR_INVALID, "more shards (%d) than record segments (%d)", n_shards, m->n_segments);
    if (n_shards > 1 && m->layout != FR_LAYOUT_SEMANTIC) FR_FAIL(FR_ERR_INVALID, "table sharding requires the SEMANTIC layout");
    const bool cpu = device == -1;   // the CPU back-end (fr_cpu.cpp): no device is selected, no HIP call is made
    if (!cpu) {
        rc = select_device(device);
        if (rc) return rc;
    }
    fr_ctx *c = new (std::nothrow) fr_ctx();
    if (!c) FR_FAIL(FR_ERR_OOM, "out of host memory");
    c->device = device;
    c->cpu = cpu;
    c->tables.assign(m->tables, m->tables + m->n_tables);
    c->segments.assign(m->segments, m->segments + m->n_segments);
    c->model = *m;
    c->model.tables = c->tables.data();
    c->model.segments = c->segments.data();
    c->table_mem.assign(m->n_tables, FrTableMem{});
    c->shard_rank = shard_rank;
    c->n_shards = n_shards;
    c->stream_group.store(fused_group_initial(), std::memory_order_relaxed);
    hipError_t e = cpu ? hipSuccess : hipStreamCreateWithFlags(&c->setup_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        fr_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        ctx_free(c);
        return FR_ERR_HIP;
    }
    rc = build_words(c);
    if (rc) {
        ctx_free(c);
        return rc;
    }
    if (cpu) {   // fp32 master weights in the reference's column-major H x K layout: all the CPU chain reads
        for (int l = 0; l < 4; l++) {
            c->d_w[l] = (float *)malloc((size_t)m->fc[l] * m->fc[l + 1] * sizeof(float));
            if (!c->d_w[l]) {
                ctx_free(c);
                FR_FAIL(FR_ERR_OOM, "out of host memory (weights layer %d)", l);
            }
        }
        *out = c;
        return FR_OK;
    }
    {   // immutable after creation (read by every driver thread in fused_flush without synchronisation): the device's compute units and
        // whether the persistent K-outer fused kernel applies to this context's descriptors and FC shape
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) n_cu = 0;
        c->n_cu = n_cu > 0 ? n_cu : 256;
        c->hk_ok = frk_fused_hk_ok(m->fc[0], m->fc[1], m->fc[2], m->fc[3], c->h_words.data(), c->n_words) ? 1 : 0;
    }
    for (int l = 0; l < 4; l++) {
        size_t n = (size_t)m->fc[l] * m->fc[l + 1];
        e = hipMalloc((void **)&c->d_w[l], n * sizeof(float));
        if (e == hipSuccess && l < 3) e = hipMalloc((void **)&c->d_wq[l], n * sizeof(float));
        if (e != hipSuccess) {
            fr_set_error("hipMalloc(weights layer %d) failed: %s", l, hipGetErrorString(e));
            ctx_free(c);
            return FR_ERR_OOM;
        }
    }
    *out = c;
    return FR_OK;
}

// Pure host query: the shard plan fr_ctx_create_sharded would use (no device needed).
extern "C" int fr_model_shard_plan(const fr_model_desc *m, int n_shards, int *slice_offset, int *slice_len, int *slice_padded_len) {
    int rc = fr_model_validate(m);
    if (rc) return rc;
    if (n_shards < 1 || n_shards > m->n_segments) FR_FAIL(FR_ERR_INVALID, "n_shards %d outside [1, %d segments]", n_shards, m->n_segments);
    std::vector<int> sb;
    if (n_shards > 1) fr_shard_bounds(*m, n_shards, sb);
    else sb = {0, m->n_segments};
    int maxlen = 0;
    for (int g = 0; g < n_shards; g++) {
        const int b = sb[g], e = sb[g + 1];
        const int off = m->segments[b].rec_offset;
        const int len = m->segments[e - 1].rec_offset + m->segments[e - 1].len - off;
        if (slice_offset) slice_offset[g] = off;
        if (slice_len) slice_len[g] = len;
        if (len > maxlen) maxlen = len;
    }
    if (slice_padded_len) *slice_padded_len = maxlen;
    return FR_OK;
}

extern "C" int fr_ctx_create(const fr_model_desc *m, int device, fr_ctx **out) { return fr_ctx_create_sharded(m, device, 0, 1, out); }
void fr_ctx_ref(fr_ctx *c) {
    if (c) c->life.fetch_add(1, std::memory_order_relaxed);
}
void fr_ctx_unref(fr_ctx *c) {
    if (c && c->life.fetch_sub(1, std::memory_order_acq_rel) == 1) ctx_free(c);
}
// Workers and communicators that are still alive keep the context (its tables, weights, descriptors) until the last of them is destroyed.
extern "C" void fr_ctx_destroy(fr_ctx *ctx) { fr_ctx_unref(ctx); }
extern "C" const fr_model_desc *fr_ctx_model(const fr_ctx *ctx) { return ctx ? &ctx->model : nullptr; }

extern "C" int fr_ctx_shard_info(const fr_ctx *ctx, int *shard_rank, int *n_shards, int *slice_offset, int *slice_len,
                                 int *slice_padded_len) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (shard_rank) *shard_rank = ctx->shard_rank;
    if (n_shards) *n_shards = ctx->n_shards;
    if (slice_offset) *slice_offset = ctx->slice_offset;
    if (slice_len) *slice_len = ctx->slice_len;
    if (slice_padded_len) *slice_padded_len = ctx->slice_padded;
    return FR_OK;
}

extern "C" int fr_ctx_fill_tables(fr_ctx *ctx, int mode, uint32_t seed) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (mode < FR_FILL_EVEN_ODD || mode > FR_FILL_TAGGED) FR_FAIL(FR_ERR_INVALID, "bad fill mode %d", mode);
    if (!ctx->cpu) FR_SET_DEVICE(ctx);
    for (int t = 0; t < ctx->model.n_tables; t++) {
        if (!ctx->table_mem[t].resident) continue;
        const fr_table_desc &d = ctx->tables[t];
        const FrTableMem &tm = ctx->table_mem[t];
        const int64_t head = tm.il_rows ? (int64_t)tm.il_rows : d.rows;  // rows at byte_offset (all of them unless interleaved)
        if (ctx->cpu) {
            frc_fill_table((float *)(ctx->table_arena + tm.byte_offset), 0, head, d.dim, (int64_t)tm.row_stride, mode, seed, fr_table_uid(d));
            if (head < d.rows) frc_fill_table((float *)(ctx->table_arena + tm.tail_offset), head, d.rows - head, d.dim, (int64_t)d.dim * 4, mode, seed, fr_table_uid(d));
            continue;
        }
        int rc = frk_fill_table((float *)(ctx->table_arena + tm.byte_offset), 0, head, d.dim, (int64_t)tm.row_stride, mode, seed, fr_table_uid(d), ctx->setup_stream);
        if (rc) return rc;
        if (head < d.rows) {
            rc = frk_fill_table((float *)(ctx->table_arena + tm.tail_offset), head, d.rows - head, d.dim, (int64_t)d.dim * 4, mode, seed, fr_table_uid(d), ctx->setup_stream);
            if (rc) return rc;
        }
    }
    if (!ctx->cpu) FR_HIP(hipStreamSynchronize(ctx->setup_stream));
    ctx->tables_filled = true;
    ctx->tables_gen++;   // an operand-type bank image made from older contents is stale (lp_ensure_image)
    return FR_OK;
}

// Copies rows [row0, row0 + nrows) of a table between host (dense rows of dim floats) and the arena, whatever the table's layout:
// a plain table is one contiguous span; a bank-interleaved one is a strided 2-D copy for its rows < il_rows plus a contiguous tail.
static int table_copy(fr_ctx *ctx, int table, int64_t row0, int64_t nrows, float *host_rows, bool to_device) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (table < 0 || table >= ctx->model.n_tables) FR_FAIL(FR_ERR_INVALID, "table %d out of range", table);
    if (!ctx->table_mem[table].resident) FR_FAIL(FR_ERR_STATE, "table %d is not resident on shard %d", table, ctx->shard_rank);
    const fr_table_desc &d = ctx->tables[table];
    if (row0 < 0 || nrows < 0 || row0 + nrows > d.rows) FR_FAIL(FR_ERR_INVALID, "rows [%lld,+%lld) outside table %d", (long long)row0, (long long)nrows, table);
    if (!host_rows && nrows) FR_FAIL(FR_ERR_INVALID, "host_rows is NULL");
    if (!ctx->cpu) FR_SET_DEVICE(ctx);
    const FrTableMem &tm = ctx->table_mem[table];
    const size_t row_bytes = (size_t)d.dim * 4;
    const int64_t head_rows = tm.il_rows ? (int64_t)tm.il_rows : d.rows;
    const int64_t n_head = row0 < head_rows ? (row0 + nrows < head_rows ? nrows : head_rows - row0) : 0;  // rows of the request below head_rows
    if (ctx->cpu) {   // host arena: plain copies, row by row where the table is one column block of a bank row
        for (int64_t r = 0; r < nrows; r++) {
            char *a = r < n_head ? ctx->table_arena + tm.byte_offset + (size_t)(row0 + r) * tm.row_stride
                                 : ctx->table_arena + tm.tail_offset + (size_t)(row0 + r - head_rows) * row_bytes;
            float *h = host_rows + (size_t)r * d.dim;
            if (to_device) memcpy(a, h, row_bytes);
            else memcpy(h, a, row_bytes);
        }
        return FR_OK;
    }
    if (n_head > 0) {
        char *dev = ctx->table_arena + tm.byte_offset + (size_t)row0 * tm.row_stride;
        if (tm.row_stride == row_bytes) {
            if (to_device) FR_HIP(hipMemcpy(dev, host_rows, (size_t)n_head * row_bytes, hipMemcpyHostToDevice));
            else FR_HIP(hipMemcpy(host_rows, dev, (size_t)n_head * row_bytes, hipMemcpyDeviceToHost));
        } else if (to_device) {
            FR_HIP(hipMemcpy2D(dev, tm.row_stride, host_rows, row_bytes, row_bytes, (size_t)n_head, hipMemcpyHostToDevice));
        } else {
            FR_HIP(hipMemcpy2D(host_rows, row_bytes, dev, tm.row_stride, row_bytes, (size_t)n_head, hipMemcpyDeviceToHost));
        }
    }
    if (nrows > n_head) {
        const int64_t t0 = row0 + n_head - head_rows;  // first tail row of the request
        char *dev = ctx->table_arena + tm.tail_offset + (size_t)t0 * row_bytes;
        float *host = host_rows + (size_t)n_head * d.dim;
        if (to_device) FR_HIP(hipMemcpy(dev, host, (size_t)(nrows - n_head) * row_bytes, hipMemcpyHostToDevice));
        else FR_HIP(hipMemcpy(host, dev, (size_t)(nrows - n_head) * row_bytes, hipMemcpyDeviceToHost));
    }
    return FR_OK;
}

extern "C" int fr_ctx_upload_table(fr_ctx *ctx, int table, int64_t row0, int64_t nrows, const float *host_rows) {
    int rc = table_copy(ctx, table, row0, nrows, const_cast<float *>(host_rows), true);
    if (rc) return rc;
    ctx->tables_filled = true;
    ctx->tables_gen++;   // an operand-type bank image made from older contents is stale (lp_ensure_image)
    return FR_OK;
}

extern "C" int fr_ctx_download_table(fr_ctx *ctx, int table, int64_t row0, int64_t nrows, float *host_rows) {
    return table_copy(ctx, table, row0, nrows, host_rows, false);
}

// ---- weights ----------------------------------------------------------------------------------------
static int floor_log2f(float v) {  // floor(log2(v)) for v > 0
    int e;
    (void)std::frexp(v, &e);  // v = m * 2^e, m in [0.5, 1)
    return e - 1;
}

// fp8 activation exponents from an rms estimate (until a calibration batch replaces them): inputs are taken as U(-1,1)
// (rms 0.58, the FR_FILL_HASH tables), each layer multiplies the rms by ||W||_F / sqrt(N), and 8 rms must still fit 448.
static void f8_estimate_act_exponents(fr_ctx *ctx) {
    if (ctx->f8_calibrated) return;
    float rms = 0.58f;
    for (int l = 0; l < 4; l++) {
        const float absmax_est = 8.0f * (rms > 0.0f ? rms : 1.0f);
        ctx->f8_e_act[l] = floor_log2f(448.0f / absmax_est);
        if (l < 3) rms *= ctx->f8_w_rms_gain[l];
    }
}

static int refresh_fp8(fr_ctx *ctx, int layer) {
    if (layer >= 3) return FR_OK;  // the output layer stays in fp32
    const int K = ctx->model.fc[layer], H = ctx->model.fc[layer + 1];
    const int KP = (K + 63) / 64 * 64;
    if (!ctx->d_w_fp8[layer]) FR_HIP(hipMalloc(&ctx->d_w_fp8[layer], (size_t)KP * H));
    if (!ctx->d_stats) FR_HIP(hipMalloc((void **)&ctx->d_stats, 64));
    int rc = frk_stats(ctx->d_w[layer], (size_t)K * H, ctx->d_stats, ctx->setup_stream);
    if (rc) return rc;
    uint32_t st[2];
    FR_HIP(hipMemcpyAsync(st, ctx->d_stats, 8, hipMemcpyDeviceToHost, ctx->setup_stream));
    FR_HIP(hipStreamSynchronize(ctx->setup_stream));
    float absmax, sumsq;
    memcpy(&absmax, &st[0], 4);
    memcpy(&sumsq, &st[1], 4);
    ctx->f8_e_w[layer] = absmax > 0.0f ? floor_log2f(448.0f / absmax) : 0;  // max |W| * 2^e_w lands in (224, 448]
    ctx->f8_w_rms_gain[layer] = std::sqrt(sumsq / (float)H);
    f8_estimate_act_exponents(ctx);
    rc = frk_pack_weights_q16_fp8(ctx->d_w[layer], ctx->d_w_fp8[layer], K, H, ctx->f8_e_w[layer], ctx->setup_stream);
    if (rc) return rc;
#ifdef FR_EXPERIMENTS
    if (frk_fused_f8_ok(ctx->model.fc[0], ctx->model.fc[1], ctx->model.fc[2], ctx->model.fc[3])) {  // the persistent kernel's fp8 form (experiments build only)
        const int KP32 = (K + 31) / 32 * 32;
        if (!ctx->d_w_fp8h[layer]) FR_HIP(hipMalloc(&ctx->d_w_fp8h[layer], (size_t)KP32 * H));
        rc = frk_pack_weights_q16h_fp8(ctx->d_w[layer], ctx->d_w_fp8h[layer], K, H, ctx->f8_e_w[layer], ctx->setup_stream);
    }
#endif
    return rc;
}

// derived copies of one layer's weights: the q4 re-pack for the fp32 chain, the bf16 / fp8 casts for the low-precision chains
static int refresh_bf16(fr_ctx *ctx, int layer) {
    if (layer < 3) {
        int rc = frk_pack_weights_q4(ctx->d_w[layer], ctx->d_wq[layer], ctx->model.fc[layer], ctx->model.fc[layer + 1], ctx->setup_stream);
        if (rc) return rc;
    }
    if (ctx->fc_precision == FR_FC_FP8) return refresh_fp8(ctx, layer);
    if (ctx->fc_precision != FR_FC_BF16) return FR_OK;
    size_t n = (size_t)ctx->model.fc[layer] * ctx->model.fc[layer + 1];
    if (!ctx->d_w_bf16[layer]) FR_HIP(hipMalloc((void **)&ctx->d_w_bf16[layer], n * sizeof(uint16_t)));
    // Wh[k/8][h][k%8]; for the output layer (H == 1) this is simply the bf16 vector w[k]
    return frk_pack_weights_q8_bf16(ctx->d_w[layer], ctx->d_w_bf16[layer], ctx->model.fc[layer], ctx->model.fc[layer + 1], ctx->setup_stream);
}

extern "C" int fr_ctx_set_weights(fr_ctx *ctx, int layer, const float *w, size_t count) {
    if (!ctx || !w) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (layer < 0 || layer > 3) FR_FAIL(FR_ERR_INVALID, "layer %d out of range", layer);
    size_t n = (size_t)ctx->model.fc[layer] * ctx->model.fc[layer + 1];
    if (count != n) FR_FAIL(FR_ERR_INVALID, "layer %d expects %zu weights (H=%d x K=%d), got %zu", layer, n, ctx->model.fc[layer + 1], ctx->model.fc[layer], count);
    if (ctx->cpu) {
        memcpy(ctx->d_w[layer], w, n * sizeof(float));
        ctx->weights_set = true;
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    FR_HIP(hipMemcpy(ctx->d_w[layer], w, n * sizeof(float), hipMemcpyHostToDevice));
    int rc = refresh_bf16(ctx, layer);
    if (rc) return rc;
    FR_HIP(hipStreamSynchronize(ctx->setup_stream));
    ctx->weights_set = true;
    return FR_OK;
}

extern "C" int fr_ctx_fill_weights(fr_ctx *ctx, int mode, uint32_t seed) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (mode != FR_WEIGHTS_ONES && mode != FR_WEIGHTS_UNIFORM) FR_FAIL(FR_ERR_INVALID, "bad weight mode %d", mode);
    if (ctx->cpu) {
        for (int l = 0; l < 4; l++)
            frc_fill_weights(ctx->d_w[l], (size_t)ctx->model.fc[l] * ctx->model.fc[l + 1], mode, seed, (uint32_t)l, 1.0f / std::sqrt((float)ctx->model.fc[l]));
        ctx->weights_set = true;
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    for (int l = 0; l < 4; l++) {
        size_t n = (size_t)ctx->model.fc[l] * ctx->model.fc[l + 1];
        float scale = 1.0f / std::sqrt((float)ctx->model.fc[l]);
        int rc = frk_fill_weights(ctx->d_w[l], n, mode, seed, (uint32_t)l, scale, ctx->setup_stream);
        if (rc) return rc;
        rc = refresh_bf16(ctx, l);
        if (rc) return rc;
    }
    FR_HIP(hipStreamSynchronize(ctx->setup_stream));
    ctx->weights_set = true;
    return FR_OK;
}

extern "C" int fr_ctx_get_weights(fr_ctx *ctx, int layer, float *w, size_t count) {
    if (!ctx || !w) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (layer < 0 || layer > 3) FR_FAIL(FR_ERR_INVALID, "layer %d out of range", layer);
    size_t n = (size_t)ctx->model.fc[layer] * ctx->model.fc[layer + 1];
    if (count != n) FR_FAIL(FR_ERR_INVALID, "layer %d holds %zu weights, got %zu", layer, n, count);
    if (ctx->cpu) {
        memcpy(w, ctx->d_w[layer], n * sizeof(float));
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    FR_HIP(hipMemcpy(w, ctx->d_w[layer], n * sizeof(float), hipMemcpyDeviceToHost));
    return FR_OK;
}

extern "C" int fr_ctx_set_fc_precision(fr_ctx *ctx, int precision) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (precision != FR_FC_FP32 && precision != FR_FC_BF16 && precision != FR_FC_FP8) FR_FAIL(FR_ERR_INVALID, "bad precision %d", precision);
    if (ctx->cpu) {
        if (precision != FR_FC_FP32) FR_FAIL(FR_ERR_INVALID, "the CPU back-end computes the FC chain in fp32 only (the reference's own precision, cuda_server.c:211)");
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    if (precision == FR_FC_BF16) {
        for (int l = 0; l < 4; l++)
            if (ctx->model.fc[l] % 16) FR_FAIL(FR_ERR_INVALID, "bf16 chain needs fc[%d]=%d to be a multiple of 16", l, ctx->model.fc[l]);
    }
    if (precision == FR_FC_FP8) {  // the record is zero-padded to 64 k inside the q16 image; hidden widths are not
        for (int l = 1; l < 4; l++)
            if (ctx->model.fc[l] % 64) FR_FAIL(FR_ERR_INVALID, "fp8 chain needs fc[%d]=%d to be a multiple of 64", l, ctx->model.fc[l]);
    }
    ctx->fc_precision = precision;
    if (precision != FR_FC_FP32 && ctx->weights_set) {
        for (int l = 0; l < 4; l++) {
            int rc = refresh_bf16(ctx, l);
            if (rc) return rc;
        }
        FR_HIP(hipStreamSynchronize(ctx->setup_stream));
    }
    // fp8 on a model that streams through the fused item-tile kernels (Models A / B) is accepted and CORRECT, but it is not
    // where the e4m3 peak is: the chunked fr_fused_tile_f8_kernel gathers with the matrix pipes idle and then streams weights at the rate
    // the bf16 kernel does -- 0.19 of the fp8 peak, 14-19 % above the bf16 chain.  The caller is told instead of finding out from a profile:
    // the call succeeds and fr_last_error() carries the note (fleetrec.h, fr_fc_precision).
    if (precision == FR_FC_FP8 && fused_eligible(ctx))
        fr_set_error("note: FR_FC_FP8 on a fused-kernel model (record of %d floats) runs fr_fused_tile_f8_kernel at ~0.19 of the fp8 MFMA peak, 14-19 %% above FR_FC_BF16 "
                     "(the 64-item tile is bound by its gather phase and its weight stream, not by the matrix pipes); the scaled-MFMA chain that reaches 0.55 of the fp8 peak "
                     "is the GEMM path of chain models (Model-C shapes at batch >= 1024)", ctx->model.fc[0]);
    return FR_OK;
}

// ---- worker -----------------------------------------------------------------------------------------
extern "C" void fr_worker_destroy(fr_worker *w) {
    if (!w) return;
    if (w->ctx && w->ctx->cpu) {
        fr_comm_worker_release(w);   // a sharded step still on the worker's host stream: waited for (bounded), its communicator let go
        if (w->counted) w->ctx->n_workers.fetch_sub(1, std::memory_order_relaxed);
        void *host[] = {w->h_idx, w->h_dense, w->h_score, w->d_records, w->c_scratch, w->c_x, w->d_slice, w->d_gathered, w->d_score_part, w->d_score_all, w->h_sh_status};
        for (void *p : host) free(p);
        fr_ctx *held = w->counted ? w->ctx : nullptr;
        delete w;
        fr_ctx_unref(held);   // (the last worker of a context that was already destroyed releases it)
        return;
    }
    if (w->ctx) (void)hipSetDevice(w->ctx->device);
    fr_comm_worker_release(w);   // a sharded step still being issued by the worker's host stream (staged exchange) is waited for (bounded); the communicator is let go
    if (w->stream) (void)hipStreamSynchronize(w->stream);
    if (w->counted) w->ctx->n_workers.fetch_sub(1, std::memory_order_relaxed);
    if (w->h_idx) (void)hipHostFree(w->h_idx);
    if (w->h_dense) (void)hipHostFree(w->h_dense);
    if (w->h_score) (void)hipHostFree(w->h_score);
    if (w->h_err) (void)hipHostFree(w->h_err);
    if (w->h_sh_status) (void)hipHostFree(w->h_sh_status);
    if (w->h_stage_send) (void)hipHostFree(w->h_stage_send);
    if (w->h_stage_recv) (void)hipHostFree(w->h_stage_recv);
    void *dev[] = {w->d_idx, w->d_dense, w->d_records, w->d_act[0], w->d_act[1], w->d_score, w->d_slice, w->d_gathered, w->d_score_part, w->d_score_all};
    for (void *p : dev)
        if (p) (void)hipFree(p);
    if (w->hr.h_idx) (void)hipHostFree(w->hr.h_idx);
    if (w->hr.h_dense) (void)hipHostFree(w->hr.h_dense);
    if (w->hr.h_sc) (void)hipHostFree(w->hr.h_sc);
    void *hdev[] = {w->hr.d_idx, w->hr.d_dense, w->hr.d_sc};
    for (void *p : hdev)
        if (p) (void)hipFree(p);
    for (hipEvent_t e : w->hr.ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : w->hr.ev_in)
        if (e) (void)hipEventDestroy(e);
    if (w->hr.copy) {
        (void)hipStreamSynchronize(w->hr.copy);
        (void)hipStreamDestroy(w->hr.copy);
    }
    if (w->h_blist) (void)hipHostFree(w->h_blist);
    if (w->d_blist) (void)hipFree(w->d_blist);
    for (hipEvent_t e : w->ev_blist)
        if (e) (void)hipEventDestroy(e);
    if (w->aux) (void)hipStreamSynchronize(w->aux);
    for (int p = 0; p < 2; p++) {
        if (w->ev_x_ready[p]) (void)hipEventDestroy(w->ev_x_ready[p]);
        if (w->ev_x_free[p]) (void)hipEventDestroy(w->ev_x_free[p]);
    }
    if (w->aux) (void)hipStreamDestroy(w->aux);
    if (w->ev_start) (void)hipEventDestroy(w->ev_start);
    if (w->ev_stop) (void)hipEventDestroy(w->ev_stop);
    if (w->stream) (void)hipStreamDestroy(w->stream);
    fr_ctx *held = w->counted ? w->ctx : nullptr;
    delete w;
    fr_ctx_unref(held);   // (the last worker of a context that was already destroyed releases it)
}

static size_t idx_cols(const fr_ctx *c) {
    const int mode = c->model.index_mode;
    return mode == FR_INDEX_PER_TABLE ? (size_t)c->model.n_tables : (mode == FR_INDEX_PER_BANK ? (size_t)c->n_banks : 1);
}

#define W_HIP(call)                                                                       \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fr_set_error("%s failed: %s", #call, hipGetErrorString(e_));                  \
            fr_worker_destroy(w);                                                         \
            return (e_ == hipErrorOutOfMemory) ? FR_ERR_OOM : FR_ERR_HIP;                 \
        }                                                                                 \
    } while (0)

extern "C" int fr_worker_create(fr_ctx *ctx, int max_batch, fr_worker **out) {
    if (!ctx || !out) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (max_batch <= 0 || max_batch > (1 << 24)) FR_FAIL(FR_ERR_INVALID, "max_batch %d out of range", max_batch);
    {   // the FC kernels address every operand through a buffer resource: 32-bit byte offsets, so each tensor must stay below 4 GiB
        int widest = 0;
        for (int l = 0; l < 4; l++) widest = ctx->model.fc[l] > widest ? ctx->model.fc[l] : widest;
        if ((uint64_t)widest * (uint64_t)round_up(max_batch, 64) * 4ull >= (1ull << 32))
            FR_FAIL(FR_ERR_INVALID, "max_batch %d: an activation tensor of %d x batch floats would reach 4 GiB (32-bit buffer offsets)", max_batch, widest);
    }
    FR_SET_DEVICE(ctx);
    fr_worker *w = new (std::nothrow) fr_worker();
    if (!w) FR_FAIL(FR_ERR_OOM, "out of host memory");
    w->ctx = ctx;
    w->max_batch = max_batch;
    const fr_model_desc &m = ctx->model;
    const size_t B = (size_t)max_batch;
    if (ctx->cpu) {   // the CPU back-end: plain host buffers behind the same accessors (fr_worker_idx_ptr / dense_ptr / score_ptr)
        auto host = [](size_t bytes) { return aligned_alloc(64, align_up(bytes ? bytes : 64, 64)); };
        w->h_idx = (int32_t *)host(B * idx_cols(ctx) * sizeof(int32_t));
        if (m.dense_len) w->h_dense = (float *)host(B * m.dense_len * sizeof(float));
        w->h_score = (float *)host(B * sizeof(float));
        w->d_records = (float *)host(B * (size_t)ctx->slice_padded * sizeof(float));
        w->c_scratch = (float *)host(B * ((size_t)m.fc[1] + m.fc[2] + m.fc[3]) * sizeof(float));
        if (ctx->n_shards > 1) w->c_x = (float *)host(B * (size_t)m.fc[0] * sizeof(float));
        if (!w->h_idx || (m.dense_len && !w->h_dense) || !w->h_score || !w->d_records || !w->c_scratch || (ctx->n_shards > 1 && !w->c_x)) {
            fr_worker_destroy(w);
            FR_FAIL(FR_ERR_OOM, "out of host memory (worker buffers for batch %d)", max_batch);
        }
        w->h_err = w->d_err = &w->c_err;
        w->counted = true;
        fr_ctx_ref(ctx);
        ctx->n_workers.fetch_add(1, std::memory_order_relaxed);
        *out = w;
        return FR_OK;
    }
    {
        // A model that runs the stage pipeline (several launches per step: Model-C, sharded contexts) gives its workers hardware queues of
        // their own: the HIP runtime keeps one pool of hardware queues per stream priority, four equal-priority streams share two queues
        // (profiles/archive/r04_C4096_chain_trace_bf16.txt), and there a small launch of one worker waits behind the other's FC1.  The workers
        // alternate between the HIGHEST and the LOWEST priority, never the default one: those two pools are this library's alone, so the
        // first four workers get four consecutively created queues = one per compute pipe of the command processor (queue id mod 4;
        // two workers whose queues share a pipe run 10 % behind the others: profiles/archive/r04_stream_queues_after_other_contexts.txt -- with
        // the default priority in the rotation, a context served earlier in the process had that effect).  Model-C 4096, four workers:
        // bf16 41.9 -> 43.1 M inf/s, fp8 65.5 -> 68.9 M = what GPU_MAX_HW_QUEUES=8 buys (profiles/archive/r04_stream_priorities_ab.txt).
        // Fused-kernel models (one launch per group) gain nothing from it and keep the default streams.
        // (a model no fused kernel of any precision serves -- the decision must not depend on the precision the context happens to have now)
        const bool any_fused = frk_fused_ok(m.fc[0], m.fc[1], m.fc[2], m.fc[3]) || frk_fused_h_ok(m.fc[0], m.fc[1], m.fc[2], m.fc[3]) ||
                               frk_fused_f8_ok(m.fc[0], m.fc[1], m.fc[2], m.fc[3]) || ctx->hk_ok == 1;
        const bool chain = !(ctx->n_shards == 1 && m.layout == FR_LAYOUT_SEMANTIC && any_fused);
        const int spread = FR_KNOB_ONCE("STREAM_PRIO", -1);   // experiment knob: 0 = never, 1 = always, 2 = rotate over every level (default included)
        int lo = 0, hi = 0;   // numerically lower = higher priority
        if (spread < 0 ? chain : spread != 0) W_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // k = how many workers the context has EVER created (one fetch_add: two threads creating workers at once cannot draw the same k, and a
        // destroy + create keeps alternating -- ADVICE r04: the live count did neither)
        const int k = ctx->worker_seq.fetch_add(1, std::memory_order_relaxed);
        // (Round 6 tried to CHECK and REPAIR the outcome, and kept none of it -- profiles/r06_experiments.md section 4: after some stream churn in the
        // process one pair of a context's four workers takes turns (four Model-C chains 48 -> 37-45 M inf/s).  Handshake kernels on the two streams
        // see each other resident; a BURST of part-chip spin launches on both does expose the pair (0.8-1.2 kernels resident of 2 against 1.95-2.0)
        // -- but a stream that probes clean at creation pairs up again when the next queue is created, a pool of streams pairs up all the same,
        // and replacing streams after all workers exist does not converge.  What the library controls is what it does here.)
        if (lo > hi) W_HIP(hipStreamCreateWithPriority(&w->stream, hipStreamNonBlocking, spread == 2 ? hi + k % (lo - hi + 1) : (k % 2 ? lo : hi)));
        else W_HIP(hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking));
    }
    W_HIP(hipHostMalloc((void **)&w->h_idx, B * idx_cols(ctx) * sizeof(int32_t), hipHostMallocDefault));
    if (m.dense_len) W_HIP(hipHostMalloc((void **)&w->h_dense, B * m.dense_len * sizeof(float), hipHostMallocDefault));
    W_HIP(hipHostMalloc((void **)&w->h_score, B * sizeof(float), hipHostMallocDefault));
    // index-range flag: pinned, device-visible host word.  Kernels touch it only on the error path
    // (system-scope atomicOr), so fr_worker_sync() needs no D2H copy -- it just reads the word after the stream drained.
    W_HIP(hipHostMalloc((void **)&w->h_err, sizeof(int), hipHostMallocMapped));
    *w->h_err = 0;
    W_HIP(hipHostGetDevicePointer((void **)&w->d_err, w->h_err, 0));
    W_HIP(hipMalloc((void **)&w->d_idx, B * idx_cols(ctx) * sizeof(int32_t)));
    if (m.dense_len) W_HIP(hipMalloc((void **)&w->d_dense, B * m.dense_len * sizeof(float)));
    W_HIP(hipMalloc((void **)&w->d_records, B * (size_t)ctx->slice_padded * sizeof(float) * (ctx->n_shards > 1 ? 1 : 1)));
    if (ctx->n_shards == 1 && ctx->slice_padded != m.record_len) {
        fr_worker_destroy(w);
        FR_FAIL(FR_ERR_STATE, "internal: unsharded ctx with a partial slice");
    }
    w->ld_max = round_up(max_batch, 64);
    for (int p = 0; p < 2; p++)
        W_HIP(hipMalloc((void **)&w->d_act[p], (size_t)w->ld_max * ((size_t)m.fc[0] + 2 * ((size_t)m.fc[1] + m.fc[2] + m.fc[3])) * sizeof(float)));
    W_HIP(hipMalloc((void **)&w->d_score, B * sizeof(float)));
    W_HIP(hipEventCreate(&w->ev_start));
    W_HIP(hipEventCreate(&w->ev_stop));
    w->counted = true;
    fr_ctx_ref(ctx);
    const int live = ctx->n_workers.fetch_add(1, std::memory_order_relaxed) + 1;
    // A width frozen by an early launch is kept (scores in flight must not change), but the caller is told once per worker that outnumbers
    // it: the call succeeds and fr_last_error() carries the note.
    const int cw = ctx->chain_width.load(std::memory_order_relaxed);
    if (cw && cw < 4 && live > cw && ctx->chain_width_auto.load(std::memory_order_relaxed))
        fr_set_error("note: the context's chain width was frozen at %d by a launch made while %d worker(s) existed; %d workers are live now and share tiles sized for %d "
                     "side-by-side chains -- call fr_ctx_set_chain_width(ctx, %d) before streaming if that is not intended", cw, cw, live, cw, live > 4 ? 4 : live);
    *out = w;
    return FR_OK;
}

extern "C" int32_t *fr_worker_idx_ptr(fr_worker *w) { return w ? w->h_idx : nullptr; }
extern "C" float *fr_worker_dense_ptr(fr_worker *w) { return w ? w->h_dense : nullptr; }
extern "C" float *fr_worker_score_ptr(fr_worker *w) { return w ? w->h_score : nullptr; }
extern "C" void *fr_worker_stream(fr_worker *w) { return w ? (void *)w->stream : nullptr; }
extern "C" float *fr_worker_records_dptr(fr_worker *w) { return w ? w->d_records : nullptr; }
extern "C" float *fr_worker_features_dptr(fr_worker *w, int *ld_max) {
    if (!w) return nullptr;
    if (ld_max) *ld_max = w->ld_max;
    return w->d_act[w->last_x_parity];
}

static int check_ready(fr_worker *w, int batch, bool need_tables, bool need_weights) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (batch <= 0 || batch > w->max_batch) FR_FAIL(FR_ERR_INVALID, "batch %d outside (0, max_batch=%d]", batch, w->max_batch);
    if (need_tables && !w->ctx->tables_filled) FR_FAIL(FR_ERR_STATE, "tables have not been filled or uploaded");
    if (need_weights && !w->ctx->weights_set) FR_FAIL(FR_ERR_STATE, "FC weights have not been set");
    return FR_OK;
}

static int launch_gather(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, void *d_records, int transport = FR_FC_FP32) {
    fr_ctx *c = w->ctx;
    if (!d_idx) FR_FAIL(FR_ERR_INVALID, "d_idx is NULL");
    if (c->model.dense_len && !d_dense) {
        bool needs = false;
        for (const FrWordDesc &wd : c->h_words) needs |= (wd.idx_col & FR_DESC_DENSE) != 0;
        if (needs) FR_FAIL(FR_ERR_INVALID, "model has dense features but d_dense is NULL");
    }
    if (c->cpu) {
        if (transport != FR_FC_FP32) FR_FAIL(FR_ERR_STATE, "the CPU back-end gathers fp32 records only");
        return frc_gather(c->h_words.data(), c->n_words, d_idx, (int)idx_cols(c), d_dense, reinterpret_cast<float *>(d_records), batch, &w->c_err);
    }
    const int variant = c->gather_variant.load(std::memory_order_relaxed);
    const bool one_chunk = variant == FR_GATHER_WORD_MAJOR_ONE_CHUNK;
    if (variant != FR_GATHER_WORD_MAJOR && !one_chunk && transport == FR_FC_FP32 && c->n_chunks > 0)
        return frk_gather_tile(c->d_passes, c->d_chunks, c->n_chunks, d_idx, (int)idx_cols(c), d_dense, d_records, c->slice_padded / 4, batch, w->d_err,
                               variant != FR_GATHER_ITEM_TILE, variant == FR_GATHER_ITEM_TILE_DEDUP_COUNT ? c->d_merged : nullptr, w->stream);
    return frk_gather(c->d_words, c->n_words, c->gather_groups, d_idx, (int)idx_cols(c), d_dense, d_records, batch, w->d_err, transport, c->f8_e_act[0], w->stream,
                      c->slice_padded / 4, one_chunk);  // words per item of the destination: the record (any layout) or the shard's padded slice
}

extern "C" int fr_ctx_set_gather_variant(fr_ctx *ctx, int variant) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (variant < FR_GATHER_WORD_MAJOR || variant > FR_GATHER_WORD_MAJOR_ONE_CHUNK) FR_FAIL(FR_ERR_INVALID, "bad gather variant %d", variant);
    if (variant != FR_GATHER_WORD_MAJOR && variant != FR_GATHER_WORD_MAJOR_ONE_CHUNK && ctx->n_chunks == 0) FR_FAIL(FR_ERR_STATE, "the item-tile gather needs the SEMANTIC layout (or a shard slice)");
    ctx->gather_variant.store(variant, std::memory_order_relaxed);
    return FR_OK;
}

extern "C" int fr_ctx_gather_variant(const fr_ctx *ctx) { return ctx ? ctx->gather_variant.load(std::memory_order_relaxed) : FR_ERR_INVALID; }

extern "C" int fr_ctx_gather_groups(const fr_ctx *ctx, int starts[9]) {
    if (!ctx || !starts) FR_FAIL(FR_ERR_INVALID, "ctx or starts is NULL");
    if (ctx->gather_groups.max_words <= 0) FR_FAIL(FR_ERR_STATE, "no XCD partition for this record (%d words)", ctx->n_words);
    for (int g = 0; g <= 8; g++) starts[g] = ctx->gather_groups.start[g];
    return FR_OK;
}

extern "C" int fr_ctx_gather_merged_lookups(fr_ctx *ctx, uint64_t *merged, int reset) {
    if (!ctx || !merged) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    *merged = 0;
    if (!ctx->d_merged) return FR_OK;
    FR_SET_DEVICE(ctx);
    FR_HIP(hipDeviceSynchronize());
    unsigned long long v = 0;
    FR_HIP(hipMemcpy(&v, ctx->d_merged, sizeof(v), hipMemcpyDeviceToHost));
    if (reset) FR_HIP(hipMemset(ctx->d_merged, 0, sizeof(v)));
    *merged = (uint64_t)v;
    return FR_OK;
}

// ---- the stage pipeline ------------------------------------------------------------------------------

// K-split of one FC layer into 2 workgroups per output tile (partials summed by the next stage's loads):
// only when the layer has fewer tiles than CUs and each wave still gets >= 16 k-pairs.
static int pick_nsplit(int K, int N, int ldm) {
    const long tiles = (long)(N / 32) * (ldm / 32);
    const int groups = K / 8;  // groups of 8 k = one 16-byte operand load per lane
    return (tiles < 256 && groups % 16 == 0 && groups / 16 >= 4) ? 2 : 1;
}

struct ActSet {
    float *x, *r1, *r2, *r3;
    size_t p1, p2, p3;  // partial strides
};
static ActSet act_set(const fr_worker *w, int parity) {
    const int32_t *fc = w->ctx->model.fc;
    const size_t ld = (size_t)w->ld_max;
    ActSet a;
    a.x = w->d_act[parity];
    a.r1 = a.x + (size_t)fc[0] * ld;
    a.p1 = (size_t)fc[1] * ld;
    a.r2 = a.r1 + 2 * a.p1;
    a.p2 = (size_t)fc[2] * ld;
    a.r3 = a.r2 + 2 * a.p2;
    a.p3 = (size_t)fc[3] * ld;
    return a;
}

// The context's chain width: frozen by the first low-precision GEMM-layer launch of a submit / push path at min(live workers, 4) unless
// fr_ctx_set_chain_width (or a driver) decided it before; from then on only that call changes it.  Diagnostic single-layer launches
// (freeze = false) and calibration batches (fp32 stages: they never get here) read the width they would run at and freeze nothing
// ("create one worker, calibrate or warm up, then create the other three" must not pin W = 1 by accident).
static int chain_width(fr_ctx *c, bool freeze = true) {
    int w = c->chain_width.load(std::memory_order_relaxed);
    if (w) return w;
    const int live = c->n_workers.load(std::memory_order_relaxed);
    const int want = live < 1 ? 1 : (live > 4 ? 4 : live);
    if (!freeze) return want;
    int expected = 0;
    if (!c->chain_width.compare_exchange_strong(expected, want, std::memory_order_relaxed)) return expected;
    c->chain_width_auto.store(true, std::memory_order_relaxed);
    return want;
}

extern "C" int fr_ctx_set_chain_width(fr_ctx *ctx, int width) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (width < 0 || width > 4) FR_FAIL(FR_ERR_INVALID, "chain width %d outside [0, 4] (0 = undecided again: the next low-precision GEMM-layer launch freezes it)", width);
    ctx->chain_width.store(width, std::memory_order_relaxed);
    ctx->chain_width_auto.store(false, std::memory_order_relaxed);
    return FR_OK;
}

#ifdef FR_EXPERIMENTS
int frk_spin_probe(hipStream_t s, int wgs, int lds_bytes, unsigned ticks);
// experiments build only (no header): n spin launches on wa's stream alone, then n on each of wa's and wb's streams interleaved -> microseconds per launch
extern "C" __attribute__((visibility("default"))) int fr_exp_burst_probe(fr_worker *wa, fr_worker *wb, int n, int wgs, int lds_bytes, int ticks, float *us_alone, float *us_pair) {
    if (!wa || !wb || !us_alone || !us_pair) FR_FAIL(FR_ERR_INVALID, "bad argument");
    FR_SET_DEVICE(wa->ctx);
    hipEvent_t e[4];
    for (auto &x : e) FR_HIP(hipEventCreate(&x));
    for (int i = 0; i < 3; i++) {   // warm both
        int rc = frk_spin_probe(wa->stream, wgs, lds_bytes, (unsigned)ticks);
        if (!rc) rc = frk_spin_probe(wb->stream, wgs, lds_bytes, (unsigned)ticks);
        if (rc) return rc;
    }
    FR_HIP(hipStreamSynchronize(wa->stream));
    FR_HIP(hipStreamSynchronize(wb->stream));
    FR_HIP(hipEventRecord(e[0], wa->stream));
    for (int i = 0; i < n; i++) {
        int rc = frk_spin_probe(wa->stream, wgs, lds_bytes, (unsigned)ticks);
        if (rc) return rc;
    }
    FR_HIP(hipEventRecord(e[1], wa->stream));
    FR_HIP(hipStreamSynchronize(wa->stream));
    float ms = 0;
    FR_HIP(hipEventElapsedTime(&ms, e[0], e[1]));
    *us_alone = 1e3f * ms / n;
    FR_HIP(hipEventRecord(e[0], wa->stream));
    FR_HIP(hipEventRecord(e[2], wb->stream));
    for (int i = 0; i < n; i++) {
        int rc = frk_spin_probe(wa->stream, wgs, lds_bytes, (unsigned)ticks);
        if (!rc) rc = frk_spin_probe(wb->stream, wgs, lds_bytes, (unsigned)ticks);
        if (rc) return rc;
    }
    FR_HIP(hipEventRecord(e[1], wa->stream));
    FR_HIP(hipEventRecord(e[3], wb->stream));
    FR_HIP(hipStreamSynchronize(wa->stream));
    FR_HIP(hipStreamSynchronize(wb->stream));
    float ma = 0, mb = 0;
    FR_HIP(hipEventElapsedTime(&ma, e[0], e[1]));
    FR_HIP(hipEventElapsedTime(&mb, e[2], e[3]));
    *us_pair = 1e3f * (ma > mb ? ma : mb) / n;
    for (auto &x : e) (void)hipEventDestroy(x);
    return FR_OK;
}
#endif

extern "C" int fr_ctx_chain_width(const fr_ctx *ctx) { return ctx ? ctx->chain_width.load(std::memory_order_relaxed) : FR_ERR_INVALID; }


// ---- operand-type bank image (fr_internal.h, fr_ctx::lp_arena) ------------------------------------------------------------------------
// Which launches read one.  (a) The large-batch gather of a chain model (Model-C) on a FR_INDEX_PER_BANK context, bf16 / fp8 chain: fewer lines
// per item.  (Per-table chain contexts gain nothing there: a row is one 128-byte line whatever its element type -- that gather is bound by
// the fabric's REQUEST rate)  (b) EXPERIMENTS build only (FR_FUSED_LP_ROWS=1): the persistent bf16 fused kernel
// (fr_fused_tile_hs_kernel, any index mode) -- 8-byte row words are twice the row sets in flight in its producers' registers; measured
// slower (profiles/r06_experiments.md section 3), so frk_fused_hk_takes_lp_rows() is false in the product.
static bool lp_image_applies_hs(const fr_ctx *c) {
    return !c->cpu && c->fc_precision == FR_FC_BF16 && c->n_words > 0 && c->lp_image_on.load(std::memory_order_relaxed) != 0;
}
static bool lp_image_applies(const fr_ctx *c, int prec) {
    return !c->cpu && c->model.index_mode == FR_INDEX_PER_BANK && (prec == FR_FC_BF16 || prec == FR_FC_FP8) && c->n_words > 0 &&
           c->lp_image_on.load(std::memory_order_relaxed) != 0;
}

extern "C" int fr_ctx_set_lp_bank_image(fr_ctx *ctx, int on) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    ctx->lp_image_on.store(on ? 1 : 0, std::memory_order_relaxed);
    return FR_OK;
}
extern "C" size_t fr_ctx_lp_bank_image_bytes(const fr_ctx *ctx) { return (ctx && ctx->lp_prec) ? ctx->lp_arena_bytes : 0; }

// Make the image current for (prec, X exponent, table contents).  Called by the launch path right before a gather that reads it; the
// rebuild runs on the context's set-up stream and is waited for (tens of milliseconds for Model-C, once per change of precision /
// calibration / table contents -- set-up calls, which by contract do not run beside a stream in flight).
static int lp_ensure_image(fr_ctx *c, int prec) {
    const int e_x = prec == FR_FC_FP8 ? c->f8_e_act[0] : 0;
    std::lock_guard<std::mutex> lk(c->lp_mutex);
    if (c->lp_prec == prec && c->lp_e_x == e_x && c->lp_tables_gen == c->tables_gen && c->d_words_lp) return FR_OK;
    const fr_model_desc &m = c->model;
    const size_t esz = prec == FR_FC_BF16 ? 2 : 1;
    // a "bank" of the image = the tables one index addresses: a memory bank of a FR_INDEX_PER_BANK context (rows: the bank's common range),
    // a single table otherwise (rows: its own)
    const bool per_bank = m.index_mode == FR_INDEX_PER_BANK;
    struct Bank { std::vector<int> members; size_t payload_floats = 0, lp_off = 0, lp_stride = 0; uint64_t rows = 0; };
    const int n_groups = per_bank ? c->n_banks : m.n_tables;
    std::vector<int> group_of(m.n_tables, 0);
    std::vector<Bank> banks(n_groups);
    for (int t = 0; t < m.n_tables; t++) {
        group_of[t] = per_bank ? c->bank_of_table[t] : t;
        if (c->table_mem[t].resident) {
            Bank &b = banks[group_of[t]];
            b.members.push_back(t);
            b.payload_floats += (size_t)m.tables[t].dim;
            b.rows = per_bank ? (uint64_t)c->bank_rows[group_of[t]] : (uint64_t)m.tables[t].rows;
        }
    }
    auto lines_x128 = [](size_t stride, size_t bytes) {  // expected 128-byte lines per row, x128 (exact over one period) -- as for the fp32 bank rows
        size_t acc = 0;
        for (size_t r = 0; r < 128; r++) acc += ((r * stride) % 128 + bytes + 127) / 128;
        return acc;
    };
    size_t off = 0;
    for (int bi = 0; bi < n_groups; bi++) {
        Bank &b = banks[bi];
        if (b.members.empty()) continue;
        const size_t payload = b.payload_floats * esz;
        size_t best = align_up(payload, 8);
        for (size_t cand : {align_up(payload, 16), align_up(payload, 32), align_up(payload, 64), align_up(payload, 128)})
            if (lines_x128(cand, payload) < lines_x128(best, payload)) best = cand;
        b.lp_stride = best;
        b.lp_off = align_up(off, 256);
        off = b.lp_off + (size_t)b.rows * best;
    }
    off = align_up(off, 256);
    FR_SET_DEVICE(c);
    if (off != c->lp_arena_bytes) {
        if (c->lp_arena) (void)hipFree(c->lp_arena);
        c->lp_arena = nullptr;
        c->lp_arena_bytes = 0;
        c->lp_prec = 0;
        if (hipMalloc((void **)&c->lp_arena, off) != hipSuccess) {
            (void)hipGetLastError();
            FR_FAIL(FR_ERR_OOM, "operand-type bank image: hipMalloc(%zu bytes) failed", off);
        }
        c->lp_arena_bytes = off;
    }
    std::vector<size_t> lp_base(m.n_tables, 0);   // offset of (row 0, column 0) of every resident table inside lp_arena
    for (Bank &b : banks) {
        size_t col = 0;
        for (int t : b.members) {
            const FrTableMem &tm = c->table_mem[t];
            lp_base[t] = b.lp_off + col * esz;
            // a table's reachable rows: the bank-interleaved region (il_rows = the bank's rows), or a lone table's own rows
            int rc = frk_convert_rows_lp(prec, c->table_arena + tm.byte_offset, (size_t)tm.row_stride, c->lp_arena + lp_base[t], b.lp_stride, (int64_t)b.rows, m.tables[t].dim, e_x,
                                         c->setup_stream);
            if (rc) return rc;
            col += (size_t)m.tables[t].dim;
        }
    }
    std::vector<FrWordDesc> lw(c->h_words);
    for (int i = 0; i < c->n_words; i++) {
        const int t = c->h_word_table[i];
        if (t < 0) continue;   // dense words come from the request, in fp32
        lw[i].src = (uint64_t)(uintptr_t)(c->lp_arena + lp_base[t]) + (uint64_t)c->h_word_col[i] * esz;
        lw[i].stride = (uint32_t)banks[group_of[t]].lp_stride;
    }
    if (!c->d_words_lp) FR_HIP(hipMalloc((void **)&c->d_words_lp, sizeof(FrWordDesc) * c->n_words));
    FR_HIP(hipMemcpyAsync(c->d_words_lp, lw.data(), sizeof(FrWordDesc) * c->n_words, hipMemcpyHostToDevice, c->setup_stream));
    FR_HIP(hipStreamSynchronize(c->setup_stream));
    c->lp_prec = prec;
    c->lp_e_x = e_x;
    c->lp_tables_gen = c->tables_gen;
    return FR_OK;
}

// Issue ONE pipeline launch: every in-flight batch advances by one stage; `fresh` (may be NULL) enters at stage 0.
// only_stage >= 0: debugging/roofline -- run just that stage of the (single) in-flight batch as its own kernel.
static int pipeline_step(fr_worker *w) {
    fr_ctx *c = w->ctx;
    const int32_t *fc = c->model.fc;
    const uint64_t L = w->launch_no;
    const int par = (int)(L & 1);
    const ActSet wr = act_set(w, par), rd = act_set(w, par ^ 1);
    const int prec = w->calibrating ? (int)FR_FC_FP32 : c->fc_precision;  // a calibration batch runs the fp32 chain
    FrPipeArgs a{};
    a.words = c->d_words;
    a.n_words = c->n_words;
    a.idx_stride = (int)idx_cols(c);
    a.err_flag = w->d_err;
    a.stamps = g_stamp_buffer;
    int blocks = 0, n_stages = 0, only = -1;
    bool split0 = false;   // this step's gather (a large-batch transposing gather) leaves on the aux stream
    int blocks0 = 0;
    // Large batches in the bf16 / fp8 chains: this step's gather (batch L) rides INSIDE the FC1 launch of batch L - 1 (fc_gemm_gather_kernel:
    // producer waves beside the GEMM's consumer waves) whenever both stages are present and FC1 takes the 128 x 256 tile.
    bool fuse01 = false;
    {
        const fr_worker::Slot &s1 = w->ring[(L - 1) % 8], &s0 = w->ring[L % 8];
        const bool has1 = L >= 1 && s1.active && s1.launch0 == L - 1 && s1.first_stage <= 1;
        const bool has0 = s0.active && s0.launch0 == L && s0.first_stage == 0;
        if (has0 && has1 && !w->calibrating && frk_fc_gemm_gather_ok(prec, fc[0], fc[1], s1.ldm)) {
            const int trv = frk_gather_tr_variant(s0.batch, (int)idx_cols(c));
            fuse01 = trv == 2 && frk_gather_tr_blocks(c->n_words, s0.ldm, trv) > 0 && (c->n_words & 1) == 0;
        }
    }
    for (int s = 0; s < FR_N_STAGES; s++) {
        FrStageArgs &st = a.st[s];
        st.block_begin = blocks;
        if (L < (uint64_t)s) continue;
        fr_worker::Slot &sl = w->ring[(L - s) % 8];
        if (!sl.active || sl.launch0 != L - s || s < sl.first_stage) continue;
        st.batch = sl.batch;
        st.ldm = sl.ldm;
        const int ldm = sl.ldm;
        // K-split plan of the three FC layers of this batch
        const bool bf16 = prec == FR_FC_BF16, fp8 = prec == FR_FC_FP8;  // no K-split partials in the low-precision chains
        const bool one = bf16 || fp8 || w->calibrating;                // (nor while calibrating: whole activations are measured)
        const int ns1 = one ? 1 : pick_nsplit(fc[0], fc[1], ldm), ns2 = one ? 1 : pick_nsplit(fc[1], fc[2], ldm),
                  ns3 = one ? 1 : pick_nsplit(fc[2], fc[3], ldm);
        const float *wl[4];
        for (int l = 0; l < 4; l++) {
            if (bf16) wl[l] = reinterpret_cast<const float *>(c->d_w_bf16[l]);
            else if (fp8 && l < 3) wl[l] = reinterpret_cast<const float *>(c->d_w_fp8[l]);
            else wl[l] = (l < 3 && !fp8) ? c->d_wq[l] : c->d_w[3];
        }
        if (fp8) {  // power-of-two quantisation exponents of this stage's operands / result
            st.e_w = (s >= 1 && s <= 3) ? c->f8_e_w[s - 1] : 0;
            st.e_in = s >= 1 ? c->f8_e_act[s - 1] : 0;
            st.e_out = s <= 3 ? c->f8_e_act[s] : 0;
        }
        switch (s) {
            case 0: {
                a.idx = sl.d_idx;
                a.dense = sl.d_dense;
                st.out = wr.x;
                st.K = fc[0];
                w->last_x_parity = par;
                if (fuse01) {   // gathered by the producer waves of this step's FC1 launch (below)
                    st.batch = 0;
                    continue;
                }
                const int trv = frk_gather_tr_variant(sl.batch, a.idx_stride);
                const int trb = frk_gather_tr_blocks(c->n_words, ldm, trv);
                if (trb > 0) {  // large batch: LDS-transposing gather
                    st.variant = trv;
                    if (trv == 2 && lp_image_applies(c, prec)) {   // ... of bank rows that are already in the chain's operand type
                        const int irc = lp_ensure_image(c, prec);
                        if (irc == FR_ERR_OOM) {
                            // no HBM left for the image: the gather reads the fp32 rows and converts them itself -- the same scores, bit for bit
                            c->lp_image_on.store(0, std::memory_order_relaxed);
                        } else if (irc) {
                            return irc;
                        } else {
                            a.words = c->d_words_lp;
                            a.src_lp = 1;
                        }
                    }
                    blocks += (trb + 7) / 8 * 8;
                    n_stages++;
                    only = s;
                    // ... on the worker's second stream when the chain's layers are GEMM launches of their own (see below): beside FC1, not before it
                    split0 = trv == 2 && !w->calibrating && FR_KNOB_ONCE("GATHER_AUX", 0) != 0 && frk_fc_lp_gemm_ok(prec, fc[0], fc[1], ldm);
                    blocks0 = blocks;
                    continue;
                }
                if (fp8) {  // the q16 gather has its own grid shape
                    blocks += frk_stage_blocks_f8_gather(fc[0], ldm);
                    n_stages++;
                    only = s;
                    continue;
                }
                break;
            }
            case 1:
                st.K = fc[0]; st.N = fc[1]; st.nsplit = ns1; st.nparts_in = 1;
                st.in = rd.x; st.in_part_stride = 0; st.out = wr.r1; st.part_stride = (int)wr.p1; st.w = wl[0];
                break;
            case 2:
                st.K = fc[1]; st.N = fc[2]; st.nsplit = ns2; st.nparts_in = ns1;
                st.in = rd.r1; st.in_part_stride = (int)rd.p1; st.out = wr.r2; st.part_stride = (int)wr.p2; st.w = wl[1];
                break;
            case 3:
                st.K = fc[2]; st.N = fc[3]; st.nsplit = ns3; st.nparts_in = ns2;
                st.in = rd.r2; st.in_part_stride = (int)rd.p2; st.out = wr.r3; st.part_stride = (int)wr.p3; st.w = wl[2];
                break;
            case 4:
                st.K = fc[3]; st.N = 1; st.nsplit = 1; st.nparts_in = ns3;
                st.in = rd.r3; st.in_part_stride = (int)rd.p3; st.out = sl.d_scores; st.w = wl[3];
                break;
        }
        if (s == 3 && st.nparts_in == 1 && st.nsplit == 1 && !w->calibrating && frk_fc_lp_gemm_ok(prec, st.K, st.N, ldm) && frk_fc_tail_ok(prec, st.K, st.N, ldm)) {
            // large batch, bf16 / fp8: FC3 and the output layer of this batch in ONE launch (fc_tail_kernel); the batch leaves the pipeline here
            int rc = frk_fc_tail(prec, st.w, st.in, wl[3], sl.d_scores, st.K, st.N, ldm, sl.batch, st.e_w, st.e_in, st.e_out, w->stream);
            if (rc) return rc;
            st.batch = 0;
            sl.active = false;
            w->n_active--;
            continue;
        }
        if (s == 1 && fuse01) {
            const fr_worker::Slot &g0 = w->ring[L % 8];   // the batch pushed in this step: its operand image goes to this step's WRITE set
            int rc = frk_fc_gemm_gather(prec, st.w, st.in, st.out, st.K, st.N, ldm, st.e_w, st.e_in, st.e_out, c->d_words, c->n_words, (int)idx_cols(c), g0.d_idx, g0.d_dense,
                                        g0.batch, g0.ldm, fc[0], wr.x, prec == FR_FC_FP8 ? c->f8_e_act[0] : 0, w->d_err, w->stream);
            if (rc) return rc;
            w->last_x_parity = par;
            st.batch = 0;
            continue;
        }
        if (s >= 1 && s <= 3 && st.nparts_in == 1 && st.nsplit == 1 && frk_fc_lp_gemm_ok(prec, st.K, st.N, ldm)) {
            // a layer big enough to fill the chip alone runs as its own LDS-tiled GEMM launch (same stream, same step)
            if (s == 1 && w->x_ready_set[par ^ 1]) {  // its X was gathered on the aux stream (previous step)
                FR_HIP(hipStreamWaitEvent(w->stream, w->ev_x_ready[par ^ 1], 0));
                w->x_ready_set[par ^ 1] = false;
            }
            if (s == 1 && w->aux && FR_KNOB_ONCE("GATHER_AUX", 0) == 2) {   // experiment: the gather of this step starts WITH this FC1, not before
                FR_HIP(hipEventRecord(w->ev_x_free[par], w->stream));
                w->x_free_set[par] = true;
            }
            // a MINOR layer (at most half the work of the chain's heaviest) may sit on fewer CUs than its share: see lp_gemm_mu
            long heaviest = 0;
            for (int l = 0; l < 3; l++) heaviest = std::max(heaviest, (long)fc[l] * fc[l + 1]);
            const bool minor = 2 * (long)st.K * st.N <= heaviest;
            int rc = frk_fc_lp_gemm(prec, st.w, st.in, st.out, st.K, st.N, ldm, st.e_w, st.e_in, st.e_out, prec == FR_FC_FP32 ? 1 : chain_width(c, !w->diag_launch), minor, w->stream);
            if (rc) return rc;
            if (s == 1 && w->aux && FR_KNOB_ONCE("GATHER_AUX", 0) != 2) {  // X[par ^ 1] may be overwritten by the gather of the NEXT step once this launch has finished
                FR_HIP(hipEventRecord(w->ev_x_free[par ^ 1], w->stream));
                w->x_free_set[par ^ 1] = true;
            }
            st.batch = 0;  // inactive in the combined launch
            continue;
        }
        blocks += frk_stage_blocks(s, c->n_words, st.K, st.N, ldm, st.nsplit);
        n_stages++;
        only = s;
        if (s == FR_N_STAGES - 1) {  // the batch leaves the pipeline with this launch
            sl.active = false;
            w->n_active--;
        }
    }
    a.n_blocks = blocks;
    w->launch_no = L + 1;
    if (n_stages == 0) return FR_OK;
    if (split0) {
        // The gather of batch L on the aux stream, enqueued AFTER this step's FC1 launch (batch L - 1, main stream): FC1's 256 workgroups
        // (one per CU: 120 KiB of LDS) take their CUs first and one 56-register gather workgroup fits beside each of them, so the two
        // kernels share every CU for the length of the gather instead of two gathers (or two FC1s) of different workers meeting each other
        // (kernel trace of round 3's chain, profiles/archive/r04_experiments.md section 1).
        if (!w->aux) {
            if (FR_KNOB_ONCE("GATHER_AUX_PRIO", 0)) {   // experiment: the gather's stream at the lowest priority
                int lo = 0, hi = 0;
                FR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
                FR_HIP(hipStreamCreateWithPriority(&w->aux, hipStreamNonBlocking, lo));
            } else
            FR_HIP(hipStreamCreateWithFlags(&w->aux, hipStreamNonBlocking));
            for (int p = 0; p < 2; p++) {
                FR_HIP(hipEventCreateWithFlags(&w->ev_x_ready[p], hipEventDisableTiming));
                FR_HIP(hipEventCreateWithFlags(&w->ev_x_free[p], hipEventDisableTiming));
            }
        }
        if (w->x_free_set[par]) {  // the FC1 that read X[par] (previous step) must be through before X[par] is overwritten
            FR_HIP(hipStreamWaitEvent(w->aux, w->ev_x_free[par], 0));
            w->x_free_set[par] = false;
        } else {  // no event yet (first steps, or a step whose FC1 was not a GEMM launch): order behind everything the main stream holds
            FR_HIP(hipEventRecord(w->ev_x_free[par], w->stream));
            FR_HIP(hipStreamWaitEvent(w->aux, w->ev_x_free[par], 0));
        }
        FrPipeArgs g = a;
        for (int s = 1; s < FR_N_STAGES; s++) {
            g.st[s].block_begin = blocks0;
            g.st[s].batch = 0;
        }
        g.n_blocks = blocks0;
        int rc = frk_pipeline_launch(g, -1, prec, w->aux);  // (the light gather | out kernel with an empty out stage)
        if (rc) return rc;
        FR_HIP(hipEventRecord(w->ev_x_ready[par], w->aux));
        w->x_ready_set[par] = true;
        for (int s = 1; s < FR_N_STAGES; s++) a.st[s].block_begin -= blocks0;
        a.st[0].batch = 0;
        a.n_blocks = blocks - blocks0;
        n_stages--;
        if (n_stages == 0) return FR_OK;
        only = -1;
        for (int s = 1; s < FR_N_STAGES; s++)
            if (a.st[s].batch > 0 && (s == FR_N_STAGES - 1 ? a.n_blocks - a.st[s].block_begin : a.st[s + 1].block_begin - a.st[s].block_begin) > 0) only = s;
        if (w->x_ready_set[par ^ 1] && a.st[1].batch > 0) {  // FC1 inside the combined launch reads an X the aux stream gathered
            FR_HIP(hipStreamWaitEvent(w->stream, w->ev_x_ready[par ^ 1], 0));
            w->x_ready_set[par ^ 1] = false;
        }
    } else if (w->aux) {
        // a step that gathers on the main stream (or not at all) while aux-stream gathers may be outstanding: keep both orders
        if (a.st[0].batch > 0 && blocks > 0) {
            FR_HIP(hipEventRecord(w->ev_x_ready[par], w->aux));   // everything the aux stream holds ...
            FR_HIP(hipStreamWaitEvent(w->stream, w->ev_x_ready[par], 0));  // ... before this step's launch writes X[par]
            w->x_ready_set[par] = false;
        }
        if (w->x_ready_set[par ^ 1] && a.st[1].batch > 0) {
            FR_HIP(hipStreamWaitEvent(w->stream, w->ev_x_ready[par ^ 1], 0));
            w->x_ready_set[par ^ 1] = false;
        }
    }
    if (n_stages == 1) {  // a lone stage runs as its own, separately named kernel (rocprof attribution)
        const int begin = a.st[only].block_begin;
        for (int s = 0; s < FR_N_STAGES; s++) a.st[s].block_begin -= (s >= only) ? begin : 0;
        a.st[only].block_begin = 0;
        return frk_pipeline_launch(a, only, prec, w->stream);
    }
    return frk_pipeline_launch(a, -1, prec, w->stream);
}

static int pipeline_push(fr_worker *w, int batch, int first_stage, const int32_t *d_idx, const float *d_dense, float *d_scores) {
    const uint64_t L0 = w->launch_no - (uint64_t)first_stage;  // launch number its (virtual) stage 0 had
    fr_worker::Slot &sl = w->ring[L0 % 8];
    if (sl.active) FR_FAIL(FR_ERR_STATE, "internal: pipeline ring slot still busy");
    sl.active = true;
    sl.launch0 = L0;
    sl.first_stage = first_stage;
    sl.batch = batch;
    sl.ldm = round_up(batch, 32);
    sl.d_idx = d_idx;
    sl.d_dense = d_dense;
    sl.d_scores = d_scores;
    w->n_active++;
    return pipeline_step(w);
}

static int pipeline_flush(fr_worker *w) {
    while (w->n_active > 0) {
        int rc = pipeline_step(w);
        if (rc) return rc;
    }
    return FR_OK;
}

// ---- fused item-tile path (fr_fused_tile_kernel): used by the streaming push for models whose activations fit in LDS --
// Batches per fused launch: a per-CONTEXT knob (fr_ctx_set_stream_group; env FR_FUSED_GROUP sets the initial value of every new
// context).  64 batches of 256 = one 64-item workgroup per CU, so ONE stream's launch fills the chip; lower it to trade throughput
// for latency.  Atomic: driver threads read it while a control thread may change it.
static int fused_group_initial() {
    const int g = FR_KNOB_ONCE("FUSED_GROUP", FR_FUSED_DEFAULT_BATCHES);
    return g < 1 ? 1 : (g > FR_FUSED_MAX_QUEUE ? FR_FUSED_MAX_QUEUE : g);
}
static int fused_group(const fr_ctx *c) { return c->stream_group.load(std::memory_order_relaxed); }

static bool fused_eligible(const fr_ctx *c) {
    const int enabled = FR_KNOB_ONCE("FUSED", 1);  // experiment knob
    const int32_t *fc = c->model.fc;
    if (!enabled || c->n_shards != 1 || c->model.layout != FR_LAYOUT_SEMANTIC) return false;
    if (c->fc_precision == FR_FC_FP8) return frk_fused_f8_ok(fc[0], fc[1], fc[2], fc[3]);
    // bf16: the chunked kernel exists for the two reference records (352 / 880 floats); any other record the K-outer persistent kernel
    // accepts (64 .. 880 floats in whole k-groups, frk_fused_hk_ok) streams through that kernel at every launch size
    if (c->fc_precision == FR_FC_BF16) return frk_fused_h_ok(fc[0], fc[1], fc[2], fc[3]) || c->hk_ok == 1;
    return frk_fused_ok(fc[0], fc[1], fc[2], fc[3]);
}

// Batches one streaming launch carries on this context: 1 for the stage pipeline, the fused kernel's group otherwise.
extern "C" int fr_ctx_set_stream_group(fr_ctx *ctx, int batches_per_launch) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (batches_per_launch < 1 || batches_per_launch > FR_FUSED_MAX_QUEUE)
        FR_FAIL(FR_ERR_INVALID, "batches_per_launch %d outside [1, %d]", batches_per_launch, FR_FUSED_MAX_QUEUE);
    ctx->stream_group.store(batches_per_launch, std::memory_order_relaxed);  // queues already holding more are launched by their next push / sync
    return FR_OK;
}

extern "C" int fr_ctx_set_small_block(fr_ctx *ctx, int max_batches) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (max_batches < 0 || max_batches > 8) FR_FAIL(FR_ERR_INVALID, "max_batches %d outside [0, 8]", max_batches);
    ctx->small_block.store(max_batches, std::memory_order_relaxed);
    return FR_OK;
}

extern "C" int fr_ctx_stream_group(const fr_ctx *ctx) { return (ctx && !ctx->cpu && fused_eligible(ctx)) ? fused_group(ctx) : 1; }   // (a CPU context computes every push at once)

// One launch of the kernarg-fed fused kernels: batches [first, first + n) of the worker's queue (n <= FR_FUSED_MAX_BATCHES).
static int fused_launch_slice(fr_worker *w, FrFusedArgs &a, int first, int n) {
    fr_ctx *c = w->ctx;
    const bool bf16 = c->fc_precision == FR_FC_BF16, fp8 = c->fc_precision == FR_FC_FP8;
    // fp32: the 64-item kernel needs 64 queued batches to cover the chip; smaller groups keep the 32-item kernel (experiments: FR_FUSED_M2=0/1 forces)
    const int m2_forced = FR_KNOB_ONCE("FUSED_M2", -1);
    // ... and a PARTIAL launch (fr_worker_sync with a few batches queued) that would put 64-item workgroups on at most half of the CUs
    // takes the 32-item kernel as well: twice the workgroups, 133 instead of 236 us each, bit-identical scores
    int tiles64 = 0;
    for (int i = 0; i < n; i++) tiles64 += (w->pending[first + i].batch + 63) / 64;
    const bool m2 = !bf16 && c->fc_precision == FR_FC_FP32 && frk_fused_m2_ok(c->model.fc[0], c->model.fc[1], c->model.fc[2], c->model.fc[3]) &&
                    (m2_forced == 1 || (m2_forced != 0 && fused_group(c) >= 64 && tiles64 > 128));
    const int per_wg = (bf16 || fp8) ? frk_fused_h_items_per_wg() : (m2 ? 64 : 32);  // items per workgroup
    int max_tiles = 0;
    for (int i = 0; i < n; i++) {
        a.b[i] = w->pending[first + i];
        const int tiles = (a.b[i].batch + per_wg - 1) / per_wg;
        if (tiles > max_tiles) max_tiles = tiles;
    }
    a.n_batches = n;
    a.tiles_per_batch = max_tiles;
    a.blist = nullptr;
    if (fp8) return frk_fused_f8_launch(a, w->stream);
    if (m2) return frk_fused_m2_launch(a, w->stream);
    if (bf16) return frk_fused_h_launch(a, w->stream);
    return frk_fused_launch(a, w->stream);
}

static int fused_flush(fr_worker *w) {
    if (w->n_pending == 0) return FR_OK;
    fr_ctx *c = w->ctx;
    FrFusedArgs a{};
    const bool bf16 = c->fc_precision == FR_FC_BF16, fp8 = c->fc_precision == FR_FC_FP8;
    a.words = c->d_words;
    a.n_words = c->n_words;
    a.idx_stride = (int)idx_cols(c);
    a.err_flag = w->d_err;
    if (fp8) {  // e4m3 q16 copies of FC1..FC3, fp32 output weights, the quantisation exponents
        a.w1q = reinterpret_cast<const float4 *>(c->d_w_fp8[0]);
        a.w2q = reinterpret_cast<const float4 *>(c->d_w_fp8[1]);
        a.w3q = reinterpret_cast<const float4 *>(c->d_w_fp8[2]);
        a.wout = c->d_w[3];
        for (int l = 0; l < 3; l++) a.e_w[l] = c->f8_e_w[l];
        for (int l = 0; l < 4; l++) a.e_act[l] = c->f8_e_act[l];
    } else if (bf16) {  // the bf16 kernels read the q8-packed bf16 copies through the same argument slots
        a.w1q = reinterpret_cast<const float4 *>(c->d_w_bf16[0]);
        a.w2q = reinterpret_cast<const float4 *>(c->d_w_bf16[1]);
        a.w3q = reinterpret_cast<const float4 *>(c->d_w_bf16[2]);
        a.wout = reinterpret_cast<const float *>(c->d_w_bf16[3]);
    } else {
        a.w1q = reinterpret_cast<const float4 *>(c->d_wq[0]);
        a.w2q = reinterpret_cast<const float4 *>(c->d_wq[1]);
        a.w3q = reinterpret_cast<const float4 *>(c->d_wq[2]);
        a.wout = c->d_w[3];
    }
    a.K = c->model.fc[0];
    a.H1 = c->model.fc[1];
    a.H2 = c->model.fc[2];
    a.H3 = c->model.fc[3];
    a.stamps = g_stamp_buffer;
    const int n_all = w->n_pending;
    w->n_pending = 0;
    w->pending_items = 0;
    struct Keep {  // whichever launcher returns below, the worker remembers the kernel it enqueued
        fr_worker *w;
        ~Keep() { keep_kernel(w); }
    } keep{w};
    // (fp8: the persistent kernel's fp8 form exists in the experiments build only and is opt-in there, FR_FUSED_HK=1 -- it is slower than the chunked fp8 kernel)
    if (bf16 || (fp8 && FR_KNOB_ONCE("FUSED_HK", -1) == 1 && c->d_w_fp8h[0] && c->d_w_fp8h[1] && c->d_w_fp8h[2])) {
        // The K-outer, persistent, wave-specialised kernel (fr_fused_ko.hip; bf16 and fp8 forms) whenever the context's descriptors fit its packed form AND the
        // launch gives every workgroup at least two tiles: the first tile of a workgroup pays the whole dependent chain index -> row -> LDS
        // (13-14 us, all workgroups at once), which only a second tile amortises.  Smaller launches (64 batches of 256 items = one tile per
        // compute unit, partial groups at fr_worker_sync) keep the chunked kernel.  Its batch list travels through device memory, so one
        // launch carries up to FR_FUSED_MAX_QUEUE batches (fr_ctx_set_stream_group above 64).
        int tiles = 0, max_tiles = 0;
        for (int i = 0; i < n_all; i++) {
            const int t_ = (w->pending[i].batch + 63) / 64;
            tiles += t_;
            if (t_ > max_tiles) max_tiles = t_;
        }
        const int hk = FR_KNOB_ONCE("FUSED_HK", -1);  // experiments build: 0 = never, 1 = whenever it applies
        // Where the persistent kernel starts to pay (re-measured in round 4, after its scratch was removed: profiles/archive/r04_fused_hs_threshold.txt):
        // K <= 352 (Model-A) level with the chunked kernel at ONE tile per compute unit (451 vs 452 M inf/s, the kernel alone 46.4 vs 49.0 us)
        // and 10-30 % ahead from 1.25 on; K = 880 (Model-B) 5 % behind at one tile (305 vs 320 M), 5 % ahead at 1.25, level at 1.5, 6 % ahead at 2.
        const int hs_from = a.K <= 352 ? c->n_cu : c->n_cu + c->n_cu / 4;
        const bool only_hs = bf16 && !frk_fused_h_ok(a.K, a.H1, a.H2, a.H3);   // a record the chunked kernel has no instantiation for
        if (c->hk_ok == 1 && (only_hs || (hk != 0 && (hk == 1 || tiles >= hs_from)))) {
            if (!w->h_blist) {
                FR_HIP(hipHostMalloc((void **)&w->h_blist, sizeof(FrFusedBatch) * FR_FUSED_MAX_QUEUE * FR_BLIST_RING, hipHostMallocDefault));
                FR_HIP(hipMalloc((void **)&w->d_blist, sizeof(FrFusedBatch) * FR_FUSED_MAX_QUEUE * FR_BLIST_RING));
                for (int k = 0; k < FR_BLIST_RING; k++) FR_HIP(hipEventCreateWithFlags(&w->ev_blist[k], hipEventDisableTiming));
            }
            const int k = w->blist_cur;
            w->blist_cur = (k + 1) % FR_BLIST_RING;
            if (w->blist_busy[k]) FR_HIP(hipEventSynchronize(w->ev_blist[k]));  // the copy that read this host block four launches ago has executed
            FrFusedBatch *hb = w->h_blist + (size_t)k * FR_FUSED_MAX_QUEUE, *db = w->d_blist + (size_t)k * FR_FUSED_MAX_QUEUE;
            memcpy(hb, w->pending, sizeof(FrFusedBatch) * n_all);
            FR_HIP(hipMemcpyAsync(db, hb, sizeof(FrFusedBatch) * n_all, hipMemcpyHostToDevice, w->stream));
            FR_HIP(hipEventRecord(w->ev_blist[k], w->stream));
            w->blist_busy[k] = true;
            a.blist = db;
            a.n_batches = n_all;
            a.tiles_per_batch = max_tiles;
            if (bf16 && frk_fused_hk_takes_lp_rows(a.K) && lp_image_applies_hs(c)) {   // rows already bf16: twice the row sets in flight in the producers' registers
                const int irc = lp_ensure_image(c, FR_FC_BF16);
                if (irc) return irc;
                a.words = c->d_words_lp;
                a.src_lp = 1;
            }
            if (bf16 && FR_KNOB_ONCE("FUSED_HS_ABLATE", 0)) a.e_act[3] = -776 - FR_KNOB_ONCE("FUSED_HS_ABLATE", 0);   // experiments build: the kernel's timing ablations (1: no row loads, 2: every row load reads row 0)
            if (fp8) {  // the "q16h" copies of the weights (the non-scaled fp8 MFMA's operand layout)
                a.w1q = reinterpret_cast<const float4 *>(c->d_w_fp8h[0]);
                a.w2q = reinterpret_cast<const float4 *>(c->d_w_fp8h[1]);
                a.w3q = reinterpret_cast<const float4 *>(c->d_w_fp8h[2]);
            }
            return frk_fused_hk_launch(a, c->n_cu, c->fc_precision, w->stream);
        }
    }
    for (int first = 0; first < n_all; first += FR_FUSED_MAX_BATCHES) {  // the kernarg-fed kernels: slices of at most 64 batches
        const int n = n_all - first < FR_FUSED_MAX_BATCHES ? n_all - first : FR_FUSED_MAX_BATCHES;
        int rc = fused_launch_slice(w, a, first, n);
        if (rc) return rc;
    }
    return FR_OK;
}

static int check_gather_args(fr_worker *w, const int32_t *d_idx, const float *d_dense) {
    fr_ctx *c = w->ctx;
    if (!d_idx) FR_FAIL(FR_ERR_INVALID, "d_idx is NULL");
    if (c->model.dense_len && !d_dense) FR_FAIL(FR_ERR_INVALID, "model has dense features but d_dense is NULL");
    return FR_OK;
}

// fc_only diagnostic: item-major records in the model's layout -> scores (pipeline must be idle)
static int launch_fc(fr_worker *w, int batch, const float *d_records, float *d_scores) {
    fr_ctx *c = w->ctx;
    if (c->n_shards > 1) FR_FAIL(FR_ERR_STATE, "fc on a sharded ctx needs the all-gathered records (use the sharded driver)");
    if (c->cpu) {
        frc_fc_chain(c->model.fc, c->d_w, d_records, batch, w->c_scratch, d_scores);
        return FR_OK;
    }
    if (w->n_active) FR_FAIL(FR_ERR_STATE, "pipeline busy: call fr_worker_sync first");
    if (w->launch_no == 0) w->launch_no = 1;  // stage 1 of the next launch reads the set the (virtual) previous launch wrote
    const int ldm = round_up(batch, 32);
    const int par_prev = (int)((w->launch_no - 1) & 1);
    int rc;
    if (c->fc_precision == FR_FC_FP8) rc = frk_records_to_q16_fp8(d_records, act_set(w, par_prev).x, batch, c->model.fc[0], ldm, c->f8_e_act[0], w->stream);
    else if (c->fc_precision == FR_FC_BF16) rc = frk_records_to_q8_bf16(d_records, act_set(w, par_prev).x, batch, c->model.fc[0], ldm, w->stream);
    else rc = frk_transpose_records(d_records, act_set(w, par_prev).x, batch, c->model.fc[0], ldm, w->stream);
    if (rc) return rc;
    rc = pipeline_push(w, batch, 1, nullptr, nullptr, d_scores);
    if (rc) return rc;
    return pipeline_flush(w);
}

// Sharded mode, after the all-gather: d_gathered = [n_shards][batch_total][slice_padded] floats (every shard's padded slice,
// item-major).  Runs the FC chain for items [item0, item0 + n_items) of the batch; d_scores receives n_items floats.
// Shared body of fr_worker_fc_from_slices / fr_worker_calibrate_fp8_slices.  Returns the launch number the batch's stage 1 ran in.
static int fc_from_slices_impl(fr_worker *w, int batch_total, int item0, int n_items, const float *d_gathered, float *d_scores, uint64_t *stage1_launch) {
    int rc = check_ready(w, n_items, false, true);
    if (rc) return rc;
    fr_ctx *c = w->ctx;
    if (!d_gathered || !d_scores) FR_FAIL(FR_ERR_INVALID, "NULL device pointer");
    if (batch_total < 1 || item0 < 0 || item0 + n_items > batch_total) FR_FAIL(FR_ERR_INVALID, "items [%d,+%d) outside batch %d", item0, n_items, batch_total);
    if (w->n_active || w->n_pending) FR_FAIL(FR_ERR_STATE, "pipeline busy: call fr_worker_sync first");
    if (c->cpu) {
        if (c->n_shards < 2) FR_FAIL(FR_ERR_STATE, "fc_from_slices needs a sharded context");
        frc_slices_to_records(d_gathered, c->n_shards, batch_total, c->slice_padded, c->shard_offset.data(), c->shard_len.data(), item0, n_items, w->c_x, c->model.fc[0]);
        frc_fc_chain(c->model.fc, c->d_w, w->c_x, n_items, w->c_scratch, d_scores);
        return FR_OK;
    }
    FR_SET_DEVICE(c);
    if (w->launch_no == 0) w->launch_no = 1;
    const int ldm = round_up(n_items, 32);
    const int par_prev = (int)((w->launch_no - 1) & 1);
    const int prec = w->calibrating ? (int)FR_FC_FP32 : c->fc_precision;
    // the slices arrive as fp32: transpose them into q4 elements, then (low-precision chains) re-pack that image; the other
    // activation set's X region is free scratch while the pipeline is idle
    float *xq = prec == FR_FC_FP32 ? act_set(w, par_prev).x : act_set(w, par_prev ^ 1).x;
    rc = frk_transpose_slices(d_gathered, c->n_shards, batch_total, c->slice_padded, c->shard_offset.data(), c->shard_len.data(), item0, n_items, xq, ldm,
                              w->stream);
    if (rc) return rc;
    if (prec != FR_FC_FP32) {
        rc = frk_q4_to_lp(prec, xq, act_set(w, par_prev).x, c->model.fc[0], ldm, c->f8_e_act[0], w->stream);
        if (rc) return rc;
    }
    if (stage1_launch) *stage1_launch = w->launch_no;
    rc = pipeline_push(w, n_items, 1, nullptr, nullptr, d_scores);
    if (rc) return rc;
    return pipeline_flush(w);
}

// Sharded mode, after the all-gather: d_gathered = [n_shards][batch_total][slice_padded] floats (every shard's padded slice,
// item-major).  Runs the FC chain (in the context's precision) for items [item0, item0 + n_items) of the batch; d_scores
// receives n_items floats.
extern "C" int fr_worker_fc_from_slices(fr_worker *w, int batch_total, int item0, int n_items, const float *d_gathered, float *d_scores) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    int rc = fc_from_slices_impl(w, batch_total, item0, n_items, d_gathered, d_scores, nullptr);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

// whole hot path for ONE batch, unpipelined: index rows -> scores (five dependent launches)
static int launch_pipeline(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, float *d_scores) {
    fr_ctx *c = w->ctx;
    if (c->n_shards > 1) FR_FAIL(FR_ERR_STATE, "submit on a sharded ctx: use the sharded driver (gather_only + all-gather + fc_only)");
    if (c->cpu) {   // the record in the model's layout (SEMANTIC, or the 3-node buffer read as B x K item-major), then the chain
        int rc = launch_gather(w, batch, d_idx, d_dense, w->d_records);
        if (rc) return rc;
        return launch_fc(w, batch, w->d_records, d_scores);
    }
    if (c->model.layout != FR_LAYOUT_SEMANTIC) {
        // literal 3-node buffer arithmetic (F8): materialise the blocked records, then read them as B x K item-major
        int rc = launch_gather(w, batch, d_idx, d_dense, w->d_records);
        if (rc) return rc;
        return launch_fc(w, batch, w->d_records, d_scores);
    }
    int rc = check_gather_args(w, d_idx, d_dense);
    if (rc) return rc;
    rc = pipeline_push(w, batch, 0, d_idx, d_dense, d_scores);
    if (rc) return rc;
    return pipeline_flush(w);
}

extern "C" int fr_worker_gather_only(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, float *d_records) {
    int rc = check_ready(w, batch, true, false);
    if (rc) return rc;
    if (!d_records) FR_FAIL(FR_ERR_INVALID, "d_records is NULL");
    FR_SET_DEVICE(w->ctx);
    rc = launch_gather(w, batch, d_idx, d_dense, d_records);
    keep_kernel(w);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

// Sharded mode, low-precision transport: the shard's slice [batch][slice_padded] as bf16 or e4m3 (x 2^e of the context's X exponent)
// instead of fp32 -- what travels through the all-gather.  transport = FR_FC_FP32 is fr_worker_gather_only.
extern "C" int fr_worker_gather_slices(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, void *d_slice, int transport) {
    int rc = check_ready(w, batch, true, false);
    if (rc) return rc;
    if (!d_slice) FR_FAIL(FR_ERR_INVALID, "d_slice is NULL");
    if (transport != FR_FC_FP32 && transport != FR_FC_BF16 && transport != FR_FC_FP8) FR_FAIL(FR_ERR_INVALID, "bad transport %d", transport);
    if (transport != FR_FC_FP32 && w->ctx->model.layout != FR_LAYOUT_SEMANTIC) FR_FAIL(FR_ERR_STATE, "low-precision transport: SEMANTIC layout only");
    FR_SET_DEVICE(w->ctx);
    rc = launch_gather(w, batch, d_idx, d_dense, d_slice, transport);
    keep_kernel(w);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

// ... and the receiving side: d_gathered = [n_shards][batch_total][slice_padded] elements of the transport type.  The transport must
// be the context's FC precision (the slices become the chain's operand image without another rounding).
extern "C" int fr_worker_fc_from_slices_lp(fr_worker *w, int batch_total, int item0, int n_items, const void *d_gathered, int transport, float *d_scores) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (transport == FR_FC_FP32) return fr_worker_fc_from_slices(w, batch_total, item0, n_items, reinterpret_cast<const float *>(d_gathered), d_scores);
    int rc = check_ready(w, n_items, false, true);
    if (rc) return rc;
    fr_ctx *c = w->ctx;
    if (transport != c->fc_precision) FR_FAIL(FR_ERR_STATE, "slice transport %d does not match the context's FC precision %d", transport, c->fc_precision);
    if (!d_gathered || !d_scores) FR_FAIL(FR_ERR_INVALID, "NULL device pointer");
    if (batch_total < 1 || item0 < 0 || item0 + n_items > batch_total) FR_FAIL(FR_ERR_INVALID, "items [%d,+%d) outside batch %d", item0, n_items, batch_total);
    if (w->n_active || w->n_pending) FR_FAIL(FR_ERR_STATE, "pipeline busy: call fr_worker_sync first");
    FR_SET_DEVICE(c);
    if (w->launch_no == 0) w->launch_no = 1;
    const int ldm = round_up(n_items, 32);
    const int par_prev = (int)((w->launch_no - 1) & 1);
    rc = frk_transpose_slices_lp(transport, d_gathered, c->n_shards, batch_total, c->slice_padded, c->shard_offset.data(), c->shard_len.data(), item0, n_items,
                                 act_set(w, par_prev).x, c->model.fc[0], ldm, w->stream);
    if (rc) return rc;
    rc = pipeline_push(w, n_items, 1, nullptr, nullptr, d_scores);
    if (rc) return rc;
    rc = pipeline_flush(w);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

extern "C" int fr_worker_fc_only(fr_worker *w, int batch, const float *d_records, float *d_scores) {
    int rc = check_ready(w, batch, false, true);
    if (rc) return rc;
    if (!d_records || !d_scores) FR_FAIL(FR_ERR_INVALID, "NULL device pointer");
    FR_SET_DEVICE(w->ctx);
    rc = launch_fc(w, batch, d_records, d_scores);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

// Roofline hook: launch ONE stage of the chain (1..3 = FC1..FC3, 4 = output layer; 0 is not available here) on the
// worker's resident activations, exactly as an unpipelined submit launches it.  API layer index = stage - 1.
extern "C" int fr_worker_fc_layer_only(fr_worker *w, int batch, int layer) {
    int rc = check_ready(w, batch, false, true);
    if (rc) return rc;
    if (layer < 0 || layer > 3) FR_FAIL(FR_ERR_INVALID, "layer %d out of range", layer);
    FR_NOT_ON_CPU(w->ctx, "fr_worker_fc_layer_only");
    if (w->n_active) FR_FAIL(FR_ERR_STATE, "pipeline busy: call fr_worker_sync first");
    fr_ctx *c = w->ctx;
    FR_SET_DEVICE(c);
    // place a virtual batch so that the next launch runs exactly stage layer+1 of it, then drop it again
    const int stage = layer + 1;
    if (w->launch_no < (uint64_t)stage) w->launch_no = stage;
    const uint64_t L0 = w->launch_no - (uint64_t)stage;
    fr_worker::Slot &sl = w->ring[L0 % 8];
    sl.active = true;
    sl.launch0 = L0;
    sl.first_stage = stage;
    sl.batch = batch;
    sl.ldm = round_up(batch, 32);
    sl.d_idx = nullptr;
    sl.d_dense = nullptr;
    sl.d_scores = w->d_score;
    w->n_active++;
    w->diag_launch = true;   // a diagnostic launch never freezes the context's chain width
    rc = pipeline_step(w);
    w->diag_launch = false;
    keep_kernel(w);  // the layer's kernel, as launched (fr_worker_last_kernel)
    if (sl.active) {  // stages 1..3 leave the batch in flight: retire it by hand
        sl.active = false;
        w->n_active--;
    }
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

extern "C" int fr_worker_fc_layer_repeat(fr_worker *w, int batch, int layer, int n) {
    if (n < 0) FR_FAIL(FR_ERR_INVALID, "n = %d", n);
    for (int i = 0; i < n; i++) {
        const int rc = fr_worker_fc_layer_only(w, batch, layer);
        if (rc) return rc;
    }
    return FR_OK;
}

// Streaming form of the hot loop body: enqueue batch after batch without synchronising (cuda_server.c:406-497 does
// exactly that).  Each call issues ONE launch in which this batch is gathered while the previous four batches of
// this worker advance through FC1, FC2, FC3 and the output layer.  d_scores of a pushed batch are complete after
// four more pushes have executed, or after fr_worker_sync() (which drains the pipeline).  The caller keeps
// d_idx / d_dense / d_scores valid and distinct for every batch still in the pipeline (up to 5).
extern "C" int fr_worker_push_device(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, float *d_scores) {
    int rc = check_ready(w, batch, true, true);
    if (rc) return rc;
    if (!d_scores) FR_FAIL(FR_ERR_INVALID, "d_scores is NULL");
    fr_ctx *c = w->ctx;
    if (c->n_shards > 1 || c->model.layout != FR_LAYOUT_SEMANTIC)
        FR_FAIL(FR_ERR_STATE, "push_device needs an unsharded SEMANTIC-layout context");
    rc = check_gather_args(w, d_idx, d_dense);
    if (rc) return rc;
    if (c->cpu) {   // the CPU back-end has nothing to queue behind: the batch is computed before the call returns
        rc = launch_pipeline(w, batch, d_idx, d_dense, d_scores);
        if (rc) return rc;
        w->in_flight = true;
        return FR_OK;
    }
    FR_SET_DEVICE(c);
    // Launch groups below FR_FUSED_MIN_GROUP ride the stage pipeline even on a fused-eligible context: a fused launch of g batches takes
    // one item tile's time (~130 us for Model-A) whatever g is, so small groups give 7 M (g = 1) .. 35 M inferences/s (g = 8) at 145 us,
    // where the pipelined stage launches give 43 M at 36 us (profiles/archive/r02_launch_group_paths.txt).
    // the fused kernels read the index rows through a buffer resource with 32-bit offsets (out-of-range items come back as 0 without a
    // branch): an index buffer of 4000 MiB or more rides the stage pipeline, whose gather stage has a 64-bit fallback
    const bool idx_fits = (size_t)batch * idx_cols(c) * sizeof(int32_t) < ((size_t)4000 << 20);
    const bool fused = fused_eligible(c) && fused_group(c) >= FR_FUSED_MIN_GROUP && idx_fits;
    if (fused && w->n_active > 0) {          // the group was raised while batches were riding the stage pipeline: drain them first
        rc = pipeline_flush(w);
        if (rc) return rc;
    }
    if (!fused && w->n_pending > 0) {        // ... or lowered with batches queued for a fused launch
        rc = fused_flush(w);
        if (rc) return rc;
    }
    if (fused) {
        // whole-path-per-item-tile kernel: queue the batch, launch when a group is full (fr_worker_sync launches the rest)
        FrFusedBatch &fb = w->pending[w->n_pending++];
        fb.idx = d_idx;
        fb.dense = d_dense;
        fb.scores = d_scores;
        fb.batch = batch;
        w->pending_items += batch;
        w->in_flight = true;
        // a launch is due when the group is full or when the queue already covers the chip (256 CUs x 64 items): large batches
        // need fewer of them per launch
        // a launch carries at most 16384 items (one 64-item tile per compute unit) -- in the bf16 chain, whose persistent kernel's workgroups
        // overlap the gather of their next tile with the FC phases of the current one, as many as the launch group allows (up to 256 batches
        // of 1024 items): every launch exposes its first tile's gather once (13-14 us), so more tiles per workgroup amortise it -- Model-B
        // 1024: 330 M inf/s at 4 tiles per workgroup (group 64), 340 M at 8 (group 128), 342 M at 16 (profiles/archive/r03_fused_hs_items_ab.txt)
        const bool big = c->fc_precision == FR_FC_BF16 || (c->fc_precision == FR_FC_FP8 && FR_KNOB_ONCE("FUSED_HK", -1) == 1);
        const int64_t max_items = FR_KNOB_ONCE("FUSED_ITEMS", 0) ? FR_KNOB_ONCE("FUSED_ITEMS", 0) : (big ? 4096 * 64 : 256 * 64);
        return (w->n_pending >= fused_group(c) || w->pending_items >= max_items || w->n_pending >= FR_FUSED_MAX_QUEUE) ? fused_flush(w) : FR_OK;
    }
    rc = pipeline_push(w, batch, 0, d_idx, d_dense, d_scores);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

extern "C" int fr_worker_push_device_list(fr_worker *w, int n, const int *batch, const int32_t *const *d_idx, const float *const *d_dense,
                                          float *const *d_scores) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (n < 0 || (n > 0 && (!batch || !d_idx || !d_scores))) FR_FAIL(FR_ERR_INVALID, "push_device_list: n = %d needs batch, d_idx and d_scores arrays", n);
    for (int i = 0; i < n; i++) {
        const int rc = fr_worker_push_device(w, batch[i], d_idx[i], d_dense ? d_dense[i] : nullptr, d_scores[i]);
        if (rc) return rc;
    }
    return FR_OK;
}

extern "C" int fr_worker_submit_device(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, float *d_scores) {
    int rc = check_ready(w, batch, true, true);
    if (rc) return rc;
    if (!d_scores) FR_FAIL(FR_ERR_INVALID, "d_scores is NULL");
    FR_SET_DEVICE(w->ctx);
    if (w->n_active || w->n_pending) FR_FAIL(FR_ERR_STATE, "pipeline busy (push_device in flight): call fr_worker_sync first");
    rc = launch_pipeline(w, batch, d_idx, d_dense, d_scores);
    if (rc) return rc;
    w->in_flight = true;
    return FR_OK;
}

extern "C" int fr_worker_submit(fr_worker *w, int batch) {
    int rc = check_ready(w, batch, true, true);
    if (rc) return rc;
    if (w->in_flight) FR_FAIL(FR_ERR_STATE, "a batch is already in flight on this worker: call fr_worker_sync first");
    fr_ctx *c = w->ctx;
    if (c->cpu) {
        rc = launch_pipeline(w, batch, w->h_idx, c->model.dense_len ? w->h_dense : nullptr, w->h_score);
        if (rc) return rc;
        w->in_flight = true;
        return FR_OK;
    }
    FR_SET_DEVICE(c);
    // Default: no copy commands on the stream -- the gather stage reads the index rows (and dense features) straight from the worker's
    // pinned host buffers over PCIe and the output layer writes the scores straight into the pinned score buffer.  Against the
    // reference's H2D / D2H commands (cuda_server.c:460-461,494-495; FR_SUBMIT_ZEROCOPY=0 keeps them) that takes 4-8 us off a submit +
    // sync at every batch size (batch 256: 46.1 -> 38.4 us p50) and costs 5 % of the rate of 16 workers submitting at once
    // (profiles/archive/r02_submit_latency.txt): this entry point is the latency path, the streaming entry points are the throughput path.
    const int zero_copy = FR_KNOB_ONCE("SUBMIT_ZEROCOPY", 1);
    if (zero_copy) {
        rc = launch_pipeline(w, batch, w->h_idx, c->model.dense_len ? w->h_dense : nullptr, w->h_score);
        if (rc) return rc;
        w->in_flight = true;
        return FR_OK;
    }
    // input H2D (cuda_server.c:460-461) -- indices instead of the already-gathered features
    FR_HIP(hipMemcpyAsync(w->d_idx, w->h_idx, (size_t)batch * idx_cols(c) * sizeof(int32_t), hipMemcpyHostToDevice, w->stream));
    if (c->model.dense_len)
        FR_HIP(hipMemcpyAsync(w->d_dense, w->h_dense, (size_t)batch * c->model.dense_len * sizeof(float), hipMemcpyHostToDevice, w->stream));
    rc = launch_pipeline(w, batch, w->d_idx, w->d_dense, w->d_score);
    if (rc) return rc;
    // output D2H (cuda_server.c:494-495)
    FR_HIP(hipMemcpyAsync(w->h_score, w->d_score, (size_t)batch * sizeof(float), hipMemcpyDeviceToHost, w->stream));
    w->in_flight = true;
    return FR_OK;
}

// ---- host-fed streaming ------------------------------------------------------------------------------
// The reference's loop reads a batch from the socket into pinned memory and enqueues H2D + GEMMs + D2H for it
// (cuda_server.c:425-495).  Here batches that arrive in host memory are staged in pinned blocks of `g` batches; a full block
// travels as one H2D copy, one fused launch and one D2H copy, and FR_HOST_BLOCKS blocks rotate so that copies, kernels and the
// host's staging of the next block overlap.
static int host_ring_init(fr_worker *w) {
    fr_worker::HostRing &r = w->hr;
    if (r.g) return FR_OK;
    fr_ctx *c = w->ctx;
    int g = 16384 / (w->max_batch > 0 ? w->max_batch : 1);  // the queue launches once it holds 256 x 64 items
    if (g < 1) g = 1;
    if (g > fused_group(c)) g = fused_group(c);
    if (g > FR_FUSED_MAX_BATCHES) g = FR_FUSED_MAX_BATCHES;   // the host-fed blocks stay within one kernarg-fed launch
    r.idx_slot = (size_t)w->max_batch * idx_cols(c);
    r.dense_slot = (size_t)w->max_batch * c->model.dense_len;
    r.score_slot = (size_t)w->max_batch;
    const size_t slots = (size_t)g * FR_HOST_BLOCKS;
    FR_HIP(hipHostMalloc((void **)&r.h_idx, slots * r.idx_slot * sizeof(int32_t), hipHostMallocDefault));
    FR_HIP(hipMalloc((void **)&r.d_idx, slots * r.idx_slot * sizeof(int32_t)));
    if (r.dense_slot) {
        FR_HIP(hipHostMalloc((void **)&r.h_dense, slots * r.dense_slot * sizeof(float), hipHostMallocDefault));
        FR_HIP(hipMalloc((void **)&r.d_dense, slots * r.dense_slot * sizeof(float)));
    }
    FR_HIP(hipHostMalloc((void **)&r.h_sc, slots * r.score_slot * sizeof(float), hipHostMallocDefault));
    FR_HIP(hipMalloc((void **)&r.d_sc, slots * r.score_slot * sizeof(float)));
    for (int b = 0; b < FR_HOST_BLOCKS; b++) FR_HIP(hipEventCreateWithFlags(&r.ev[b], hipEventDisableTiming));
    {   // The copy stream lives in the LOWEST stream priority's pool of hardware queues: the HIP runtime pools queues per priority, and in the
        // default pool the marker packet behind a block's copy (its event) would sit in a hardware queue that some worker's launches share --
        // that worker's next launch then waits out another worker's 65 us copy (seen as a bimodal 97 % / 99 % of the HBM-resident rate,
        // by which queue the runtime happened to hand out: profiles/r05_experiments.md section 1).  Fused-kernel models keep their worker
        // streams on the default priority, so the low pool is the copy streams' alone.
        int lo = 0, hi = 0;
        FR_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        if (lo > hi) FR_HIP(hipStreamCreateWithPriority(&r.copy, hipStreamNonBlocking, lo));
        else FR_HIP(hipStreamCreateWithFlags(&r.copy, hipStreamNonBlocking));
    }
    for (int b = 0; b < FR_HOST_BLOCKS; b++) FR_HIP(hipEventCreateWithFlags(&r.ev_in[b], hipEventDisableTiming));
    r.g = g;
    return FR_OK;
}

// scores of a finished block -> the callers' buffers
static int host_block_deliver(fr_worker *w, int b) {
    fr_worker::HostRing &r = w->hr;
    if (!r.inflight[b]) return FR_OK;
    FR_HIP(hipEventSynchronize(r.ev[b]));
    for (int i = 0; i < r.count[b]; i++)
        memcpy(r.dst[b][i], r.h_sc + ((size_t)b * r.g + i) * r.score_slot, (size_t)r.bsz[b][i] * sizeof(float));
    r.delivered += r.count[b];
    r.inflight[b] = false;
    r.count[b] = 0;
    return FR_OK;
}

// the block being filled leaves: H2D, fused launch, D2H, event
static int host_block_launch(fr_worker *w) {
    fr_worker::HostRing &r = w->hr;
    const int b = r.cur, n = r.count[b];
    if (n == 0) return FR_OK;
    int rc = fused_flush(w);  // batches queued by fr_worker_push_device go first
    if (rc) return rc;
    const size_t s0 = (size_t)b * r.g;
    if (n <= w->ctx->small_block.load(std::memory_order_relaxed) && w->n_active == 0) {
        // A block of very few batches (fr_worker_flush on a nearly idle server): one batch through the fused kernel is 8 workgroups working
        // through the whole chain for 133 us; the stage launches of fr_worker_submit spread the same batch over the chip in ~25 us.  The
        // kernels read the staged rows and write the scores in the pinned staging itself (no copy commands), exactly as fr_worker_submit
        // does; the scores are fr_worker_submit's, bit for bit (not the fused kernel's: another fp32 summation order, equal to ~1e-6).
        // The block's batches follow each other through the stage pipeline (launch L = gather of batch L | FC1 of L-1 | ... | out of L-4) and
        // four more launches drain it: n + 4 launches for the block, not 5 n.
        const int serial = FR_KNOB_ONCE("SMALL_BLOCK_SERIAL", 0);  // experiment knob: 1 = five launches per batch
        for (int i = 0; i < n; i++) {
            rc = pipeline_push(w, r.bsz[b][i], 0, r.h_idx + (s0 + i) * r.idx_slot, r.dense_slot ? r.h_dense + (s0 + i) * r.dense_slot : nullptr,
                               r.h_sc + (s0 + i) * r.score_slot);
            if (rc == FR_OK && serial) rc = pipeline_flush(w);
            if (rc) return rc;
        }
        rc = pipeline_flush(w);
        if (rc) return rc;
        FR_HIP(hipEventRecord(r.ev[b], w->stream));
        r.inflight[b] = true;
        r.cur = (b + 1) % FR_HOST_BLOCKS;
        return FR_OK;
    }
    // The two PCIe hops of the reference's loop (cuda_server.c:460-461,494-495), arranged so that the worker's stream carries NOTHING but its
    // kernels: the block's index rows travel as one H2D copy on the worker's COPY stream -- issued now, while the previous block's kernel
    // still runs, and joined by an event the launch waits for (long satisfied when the launch reaches the head of its hardware queue) --
    // and the output layer writes the scores straight into the pinned staging (no D2H command).  With both copies as commands of the one
    // stream (round 4) a block's H2D could not start before the previous block's D2H, its 130 us sat between two launches of the stream,
    // and the barrier packet in front of the launch held back every other stream that shares the hardware queue: 95-96 % of the
    // HBM-resident rate with 8 streams on 4 queues; this form: 98-99 % (profiles/r05_host_fed_timeline.txt).
    // Experiments build: FR_HOST_ZEROCOPY 0 = round 4's commands, 1 = no H2D either (the kernel reads the rows over PCIe: slower), 2 = H2D on the worker's stream.
    const int zc = FR_KNOB_ONCE("HOST_ZEROCOPY", 3);
    hipStream_t cs = zc == 3 ? r.copy : w->stream;
    const int pending0 = w->n_pending;
    auto issue = [&]() -> int {
        if (zc != 1) {
            FR_HIP(hipMemcpyAsync(r.d_idx + s0 * r.idx_slot, r.h_idx + s0 * r.idx_slot, (size_t)n * r.idx_slot * sizeof(int32_t), hipMemcpyHostToDevice, cs));
            if (r.dense_slot)
                FR_HIP(hipMemcpyAsync(r.d_dense + s0 * r.dense_slot, r.h_dense + s0 * r.dense_slot, (size_t)n * r.dense_slot * sizeof(float), hipMemcpyHostToDevice, cs));
            if (zc == 3) {   // (the device block is free: host_slot_prepare delivered its previous use -- its kernel has finished -- before refilling the staging)
                FR_HIP(hipEventRecord(r.ev_in[b], r.copy));
                FR_HIP(hipStreamWaitEvent(w->stream, r.ev_in[b], 0));
            }
        }
        for (int i = 0; i < n; i++) {
            FrFusedBatch &fb = w->pending[w->n_pending++];
            fb.idx = (zc == 1 ? r.h_idx : r.d_idx) + (s0 + i) * r.idx_slot;
            fb.dense = r.dense_slot ? (zc == 1 ? r.h_dense : r.d_dense) + (s0 + i) * r.dense_slot : nullptr;
            fb.scores = (zc ? r.h_sc : r.d_sc) + (s0 + i) * r.score_slot;
            fb.batch = r.bsz[b][i];
        }
        int rc2 = fused_flush(w);
        if (rc2) return rc2;
        if (!zc) FR_HIP(hipMemcpyAsync(r.h_sc + s0 * r.score_slot, r.d_sc + s0 * r.score_slot, (size_t)n * r.score_slot * sizeof(float), hipMemcpyDeviceToHost, w->stream));
        FR_HIP(hipEventRecord(r.ev[b], w->stream));
        return FR_OK;
    };
    rc = issue();
    if (rc) {
        // the block's H2D copy may already be queued while the block is NOT marked in flight -- the caller could refill the
        // pinned staging of block b under the pending copy.  Drain both streams before handing the error back (the message of the failing
        // call stays in fr_last_error), and forget the batches that were queued for the launch that did not happen.
        (void)hipStreamSynchronize(cs);
        (void)hipStreamSynchronize(w->stream);
        if (w->n_pending > pending0) w->n_pending = pending0;
        return rc;
    }
    r.inflight[b] = true;
    r.cur = (b + 1) % FR_HOST_BLOCKS;
    return FR_OK;
}

// The staging slot of the next pushed batch: the oldest block is delivered first if its staging memory is about to be reused.
static int host_slot_prepare(fr_worker *w, int batch, int32_t **idx_slot, float **dense_slot) {
    int rc = check_ready(w, batch, true, true);
    if (rc) return rc;
    fr_ctx *c = w->ctx;
    FR_NOT_ON_CPU(c, "host-fed streaming (fr_worker_push_host / fr_worker_stage_acquire)");
    if (!fused_eligible(c)) FR_FAIL(FR_ERR_STATE, "host-fed streaming needs a model that streams through the fused item-tile kernel (use fr_worker_submit)");
    FR_SET_DEVICE(c);
    rc = host_ring_init(w);
    if (rc) return rc;
    fr_worker::HostRing &r = w->hr;
    const int b = r.cur;
    if (r.inflight[b]) {  // the oldest block: wait for its scores and hand them out before its staging memory is reused
        rc = host_block_deliver(w, b);
        if (rc) return rc;
    }
    const int i = r.count[b];
    *idx_slot = r.h_idx + ((size_t)b * r.g + i) * r.idx_slot;
    *dense_slot = r.dense_slot ? r.h_dense + ((size_t)b * r.g + i) * r.dense_slot : nullptr;
    return FR_OK;
}

// The batch written into the current slot joins its block; a full block leaves (H2D, fused launch, D2H).
static int host_slot_commit(fr_worker *w, int batch, float *h_scores) {
    fr_worker::HostRing &r = w->hr;
    const int b = r.cur, i = r.count[b];
    r.dst[b][i] = h_scores;
    r.bsz[b][i] = batch;
    r.count[b] = i + 1;
    w->in_flight = true;
    return (r.count[b] >= r.g) ? host_block_launch(w) : FR_OK;
}

extern "C" int fr_worker_push_host(fr_worker *w, int batch, const int32_t *h_idx, const float *h_dense, float *h_scores) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (!h_idx || !h_scores) FR_FAIL(FR_ERR_INVALID, "NULL host pointer");
    if (w->ctx->model.dense_len && !h_dense) FR_FAIL(FR_ERR_INVALID, "model has dense features but h_dense is NULL");
    if (w->hr.staged) FR_FAIL(FR_ERR_STATE, "a staging slot is acquired: push it with fr_worker_push_staged first");
    int32_t *si = nullptr;
    float *sd = nullptr;
    int rc = host_slot_prepare(w, batch, &si, &sd);
    if (rc) return rc;
    memcpy(si, h_idx, (size_t)batch * idx_cols(w->ctx) * sizeof(int32_t));
    if (sd) memcpy(sd, h_dense, (size_t)batch * w->ctx->model.dense_len * sizeof(float));
    return host_slot_commit(w, batch, h_scores);
}

extern "C" int fr_worker_stage_acquire(fr_worker *w, int batch, int32_t **h_idx, float **h_dense) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (!h_idx) FR_FAIL(FR_ERR_INVALID, "h_idx is NULL");
    if (w->ctx->model.dense_len && !h_dense) FR_FAIL(FR_ERR_INVALID, "model has dense features but h_dense is NULL");
    if (w->hr.staged) FR_FAIL(FR_ERR_STATE, "a staging slot is already acquired");
    int32_t *si = nullptr;
    float *sd = nullptr;
    int rc = host_slot_prepare(w, batch, &si, &sd);
    if (rc) return rc;
    *h_idx = si;
    if (h_dense) *h_dense = sd;
    w->hr.staged = batch;
    return FR_OK;
}

extern "C" int fr_worker_push_staged(fr_worker *w, int batch, float *h_scores) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (!h_scores) FR_FAIL(FR_ERR_INVALID, "h_scores is NULL");
    if (!w->hr.staged) FR_FAIL(FR_ERR_STATE, "no staging slot acquired (fr_worker_stage_acquire)");
    if (batch < 1 || batch > w->hr.staged) FR_FAIL(FR_ERR_INVALID, "batch %d outside (0, %d] acquired", batch, w->hr.staged);
    w->hr.staged = 0;
    return host_slot_commit(w, batch, h_scores);
}

static int host_ring_drain(fr_worker *w) {
    fr_worker::HostRing &r = w->hr;
    if (!r.g) return FR_OK;
    r.staged = 0;                   // an acquired slot that was never pushed is dropped
    int rc = host_block_launch(w);  // partial block
    if (rc) return rc;
    for (int k = 0; k < FR_HOST_BLOCKS; k++) {
        rc = host_block_deliver(w, (r.cur + k) % FR_HOST_BLOCKS);  // oldest first
        if (rc) return rc;
    }
    return FR_OK;
}

extern "C" int fr_ctx_get_fp8_exponents(const fr_ctx *ctx, int act_exp[4], int w_exp[3]) {
    if (!ctx || !act_exp || !w_exp) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    for (int l = 0; l < 4; l++) act_exp[l] = ctx->f8_e_act[l];
    for (int l = 0; l < 3; l++) w_exp[l] = ctx->f8_e_w[l];
    return FR_OK;
}

extern "C" int fr_ctx_set_fp8_act_exponents(fr_ctx *ctx, const int act_exp[4]) {
    if (!ctx || !act_exp) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    for (int l = 0; l < 4; l++)
        if (act_exp[l] < -100 || act_exp[l] > 100) FR_FAIL(FR_ERR_INVALID, "act_exp[%d]=%d outside [-100, 100]", l, act_exp[l]);
    for (int l = 0; l < 4; l++) ctx->f8_e_act[l] = act_exp[l];
    ctx->f8_calibrated = true;
    return FR_OK;
}

// max |.| of X, R1, R2, R3 of the batch whose (virtual) stage 0 ran in launch L0 -> the context's activation exponents
static int f8_calibrate_finish(fr_worker *w, uint64_t L0, int batch) {
    fr_ctx *c = w->ctx;
    const int32_t *fc = c->model.fc;
    const size_t ldm = (size_t)round_up(batch, 32);
    const float *reg[4] = {act_set(w, (int)(L0 & 1)).x, act_set(w, (int)((L0 + 1) & 1)).r1, act_set(w, (int)((L0 + 2) & 1)).r2,
                           act_set(w, (int)((L0 + 3) & 1)).r3};
    for (int l = 0; l < 4; l++) {
        int rc = frk_stats(reg[l], (size_t)fc[l] * ldm, c->d_stats + 2 * l, w->stream);
        if (rc) return rc;
    }
    uint32_t st[8];
    FR_HIP(hipMemcpyAsync(st, c->d_stats, sizeof(st), hipMemcpyDeviceToHost, w->stream));
    FR_HIP(hipStreamSynchronize(w->stream));
    if (__atomic_load_n(w->h_err, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(w->h_err, 0, __ATOMIC_RELEASE);
        FR_FAIL(FR_ERR_INDEX_RANGE, "a lookup index of the calibration batch was outside its table");
    }
    for (int l = 0; l < 4; l++) {
        float absmax;
        memcpy(&absmax, &st[2 * l], 4);
        if (absmax > 0.0f && std::isfinite(absmax)) c->f8_e_act[l] = floor_log2f(448.0f / (2.0f * absmax));  // one binade of headroom
    }
    c->f8_calibrated = true;
    return FR_OK;
}

extern "C" int fr_worker_calibrate_fp8(fr_worker *w, int batch) {
    int rc = check_ready(w, batch, true, true);
    if (rc) return rc;
    fr_ctx *c = w->ctx;
    FR_NOT_ON_CPU(c, "fr_worker_calibrate_fp8");
    if (c->n_shards > 1 || c->model.layout != FR_LAYOUT_SEMANTIC) FR_FAIL(FR_ERR_STATE, "fp8 calibration from index rows: unsharded SEMANTIC contexts only (sharded: fr_worker_calibrate_fp8_slices)");
    if (w->in_flight || w->n_active || w->n_pending) FR_FAIL(FR_ERR_STATE, "worker busy: call fr_worker_sync first");
    FR_SET_DEVICE(c);
    if (!c->d_stats) FR_HIP(hipMalloc((void **)&c->d_stats, 64));
    FR_HIP(hipMemcpyAsync(w->d_idx, w->h_idx, (size_t)batch * idx_cols(c) * sizeof(int32_t), hipMemcpyHostToDevice, w->stream));
    if (c->model.dense_len)
        FR_HIP(hipMemcpyAsync(w->d_dense, w->h_dense, (size_t)batch * c->model.dense_len * sizeof(float), hipMemcpyHostToDevice, w->stream));
    rc = check_gather_args(w, w->d_idx, w->d_dense);
    if (rc) return rc;
    const uint64_t L0 = w->launch_no;
    w->calibrating = true;  // fp32 stages, no K-split partials: every activation buffer holds whole sums
    rc = pipeline_push(w, batch, 0, w->d_idx, w->d_dense, w->d_score);
    if (!rc) rc = pipeline_flush(w);
    w->calibrating = false;
    if (rc) return rc;
    return f8_calibrate_finish(w, L0, batch);
}

// The sharded form: calibrate on all-gathered slices (same arguments as fr_worker_fc_from_slices, no scores returned).
extern "C" int fr_worker_calibrate_fp8_slices(fr_worker *w, int batch_total, int item0, int n_items, const float *d_gathered) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    fr_ctx *c = w->ctx;
    FR_NOT_ON_CPU(c, "fr_worker_calibrate_fp8_slices");
    if (w->in_flight) FR_FAIL(FR_ERR_STATE, "worker busy: call fr_worker_sync first");
    FR_SET_DEVICE(c);
    if (!c->d_stats) FR_HIP(hipMalloc((void **)&c->d_stats, 64));
    uint64_t L1 = 0;
    w->calibrating = true;
    int rc = fc_from_slices_impl(w, batch_total, item0, n_items, d_gathered, w->d_score, &L1);
    w->calibrating = false;
    if (rc) return rc;
    return f8_calibrate_finish(w, L1 - 1, n_items);
}

// Launch what is queued -- the partially filled host block, the batches queued by fr_worker_push_device -- without waiting for it.
extern "C" int fr_worker_flush(fr_worker *w) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (w->hr.staged) FR_FAIL(FR_ERR_STATE, "a staging slot is acquired: push it with fr_worker_push_staged first");
    FR_SET_DEVICE(w->ctx);
    if (w->hr.g) {
        int rc = host_block_launch(w);
        if (rc) return rc;
    }
    return fused_flush(w);
}

// Hand out the scores of the host-fed blocks that have finished, oldest first, WITHOUT waiting for the others; *delivered = host-fed
// batches delivered since the worker was created (they are delivered in push order).
extern "C" int fr_worker_host_poll(fr_worker *w, long long *delivered) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    fr_worker::HostRing &r = w->hr;
    if (r.g) {
        FR_SET_DEVICE(w->ctx);
        for (int k = 0; k < FR_HOST_BLOCKS; k++) {
            const int b = (r.cur + k) % FR_HOST_BLOCKS;  // oldest first
            if (!r.inflight[b]) continue;
            const hipError_t q = hipEventQuery(r.ev[b]);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) FR_FAIL(FR_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
            int rc = host_block_deliver(w, b);
            if (rc) return rc;
        }
    }
    if (delivered) *delivered = r.delivered;
    return FR_OK;
}

// Host-fed batches that are queued in the block being filled (not launched yet) and that are launched but not delivered yet.
extern "C" int fr_worker_host_pending(const fr_worker *w, int *queued, int *in_flight, int *blocks_in_flight) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    const fr_worker::HostRing &r = w->hr;
    int q = 0, f = 0, nb = 0;
    if (r.g)
        for (int b = 0; b < FR_HOST_BLOCKS; b++) {
            (r.inflight[b] ? f : q) += r.count[b];
            nb += r.inflight[b] ? 1 : 0;
        }
    if (queued) *queued = q;
    if (in_flight) *in_flight = f;
    if (blocks_in_flight) *blocks_in_flight = nb;
    return FR_OK;
}

extern "C" int fr_worker_sync(fr_worker *w) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (w->ctx->cpu) {   // everything was computed inside the calls -- but for a table-sharded step, which runs on the worker's host stream
        const int crc = fr_comm_wait(w);   // (bounded wait + the ranks' status words, fr_comm.cpp; FR_OK at once when no such step is in flight)
        w->in_flight = false;
        if (crc) return crc;
        if (__atomic_exchange_n(&w->c_err, 0, __ATOMIC_ACQ_REL)) FR_FAIL(FR_ERR_INDEX_RANGE, "a lookup index was outside its table (row 0 was read instead)");
        return FR_OK;
    }
    FR_SET_DEVICE(w->ctx);
    if (w->sh_comm && w->sh_host_stream) {
        // a sharded step of a STAGED exchange is issued by the worker's host stream (it blocks in rendezvous): nothing below may touch the worker's
        // pipeline state while that thread still runs the FC chain's launches -- the bounded wait comes first
        const int crc0 = fr_comm_wait(w);
        if (crc0) {
            w->in_flight = false;
            return crc0;
        }
    }
    int frc = host_ring_drain(w);  // host-fed blocks: launch the partial one, deliver every block's scores
    if (!frc) frc = fused_flush(w);  // launch batches still queued by fr_worker_push_device
    if (!frc) frc = pipeline_flush(w);  // drain the stage pipeline
    if (frc) {
        fr_comm_forget(w);   // a sharded step in flight keeps no hold on its communicator past a failed sync
        return frc;
    }
    const int crc = fr_comm_wait(w);  // a table-sharded step in flight: bounded wait + the ranks' status words (fr_comm.cpp)
    if (crc) {
        w->in_flight = false;
        return crc;
    }
    FR_HIP(hipStreamSynchronize(w->stream));
    if (w->aux) FR_HIP(hipStreamSynchronize(w->aux));  // (its gathers were consumed by main-stream launches that have finished: returns at once)
    w->in_flight = false;
    if (__atomic_load_n(w->h_err, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(w->h_err, 0, __ATOMIC_RELEASE);
        FR_FAIL(FR_ERR_INDEX_RANGE, "a lookup index was outside its table (row 0 was read instead)");
    }
    return FR_OK;
}

extern "C" int fr_worker_timer_start(fr_worker *w) {
    if (!w) FR_FAIL(FR_ERR_INVALID, "worker is NULL");
    if (w->ctx->cpu) {
        w->c_t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        return FR_OK;
    }
    FR_SET_DEVICE(w->ctx);
    FR_HIP(hipEventRecord(w->ev_start, w->stream));
    return FR_OK;
}

extern "C" int fr_worker_timer_stop_ms(fr_worker *w, float *ms) {
    if (!w || !ms) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (w->ctx->cpu) {
        *ms = (float)(1e3 * (std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - w->c_t0));
        return FR_OK;
    }
    FR_SET_DEVICE(w->ctx);
    FR_HIP(hipEventRecord(w->ev_stop, w->stream));
    FR_HIP(hipEventSynchronize(w->ev_stop));
    FR_HIP(hipEventElapsedTime(ms, w->ev_start, w->ev_stop));
    return FR_OK;
}

// ---- device memory helpers ------------------------------------------------------------------------------
extern "C" int fr_device_malloc(fr_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (ctx->cpu) {   // "device memory" of the CPU back-end is host memory: the same calling code runs against either
        *dptr = aligned_alloc(64, align_up(bytes ? bytes : 64, 64));
        if (!*dptr) FR_FAIL(FR_ERR_OOM, "out of host memory (%zu bytes)", bytes);
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    FR_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return FR_OK;
}
extern "C" int fr_device_free(fr_ctx *ctx, void *dptr) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (ctx->cpu) {
        free(dptr);
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    if (dptr) FR_HIP(hipFree(dptr));
    return FR_OK;
}
extern "C" int fr_memcpy_h2d(fr_ctx *ctx, void *dptr, const void *host, size_t bytes) {
    if (!ctx || (bytes && (!dptr || !host))) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (ctx->cpu) {
        if (bytes) memcpy(dptr, host, bytes);
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    if (bytes) FR_HIP(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
    return FR_OK;
}
extern "C" int fr_memcpy_d2h(fr_ctx *ctx, void *host, const void *dptr, size_t bytes) {
    if (!ctx || (bytes && (!dptr || !host))) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    if (ctx->cpu) {
        if (bytes) memcpy(host, dptr, bytes);
        return FR_OK;
    }
    FR_SET_DEVICE(ctx);
    if (bytes) FR_HIP(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
    return FR_OK;
}
extern "C" int fr_device_synchronize(fr_ctx *ctx) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (ctx->cpu) return FR_OK;
    FR_SET_DEVICE(ctx);
    FR_HIP(hipDeviceSynchronize());
    return FR_OK;
}
