// Internal declarations shared by the host-side C-ABI (fr_api.cpp, fr_registry.cpp) and the
// gfx950 kernels (fr_fill / fr_gather / fr_pipeline / fr_gemm / fr_fused .hip).  Not installed; the public surface is include/fleetrec.h (+ fleetrec_serving.h, fleetrec_diag.h).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdint>
#include <string>
#include <vector>

#include "fleetrec.h"
#include "fleetrec_diag.h"
#include "fleetrec_serving.h"

// ---- error plumbing ---------------------------------------------------------------------------
void fr_set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define FR_FAIL(code, ...)         \
    do {                           \
        fr_set_error(__VA_ARGS__); \
        return (code);             \
    } while (0)

#define FR_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            fr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (e_ == hipErrorOutOfMemory) ? FR_ERR_OOM : FR_ERR_HIP;                            \
        }                                                                                            \
    } while (0)

// ---- experiment knobs ---------------------------------------------------------------------------
// The product library reads NO environment variable: FR_KNOB folds to its default and the variable's name is not even in the
// binary.  `make exp` builds libfleetrec_exp.so with -DFR_EXPERIMENTS, where FR_KNOB("X", d) reads the environment variable FR_X
// (FR_KNOB: at every call -- in-process sweeps; FR_KNOB_ONCE: once per process); A/B runs load that build through FR_LIB.
#ifdef FR_EXPERIMENTS
int fr_knob_env(const char *name, int dflt);
#define FR_KNOB(name, dflt) fr_knob_env("FR_" name, (dflt))
#define FR_KNOB_ONCE(name, dflt) ([]() -> int { static const int v_ = fr_knob_env("FR_" name, (dflt)); return v_; }())
#else
#define FR_KNOB(name, dflt) (dflt)
#define FR_KNOB_ONCE(name, dflt) (dflt)
#endif

// ---- which kernel ran ---------------------------------------------------------------------------
// Every launcher names the kernel it just enqueued (instantiation included, as rocprofv3 prints it) in a thread-local note; the entry
// points of the C-ABI copy the note of their dominant launch into the worker (fr_worker_last_kernel), so that bench.py's roofline
// objects name the kernel that actually ran instead of the one the author expected.
void fr_note_kernel(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *fr_noted_kernel();

// ---- device-side descriptors --------------------------------------------------------------------
// One per 16-byte word of the per-item record (the unit the reference's packers move:
// `typedef ap_uint<128> axi_t`, constants.hpp:4).  32 bytes = two dwordx4 loads, read once per thread.
struct FrWordDesc {
    uint64_t src;        // TABLE/COPY: device address of (table base + 16*word_in_row); DENSE: byte offset in the item's dense row
    uint32_t stride;     // bytes between consecutive rows of the source (dim*4; DENSE: dense_len*4)
    uint32_t idx_col;    // column of the item's index row feeding this word (0 in PER_ITEM mode); bit 31 set = DENSE
    uint32_t rows;       // row count for the range check (DENSE: unused)
    uint32_t dst_off;    // word offset inside the item's destination row
    uint32_t dst_stride; // words per item of the destination block (SEMANTIC: K/4; BLOCKED: source_len/4)
    uint32_t dst_blk;    // BLOCKED: words-per-item of all preceding source blocks (x batch at run time); SEMANTIC: 0
};
static_assert(sizeof(FrWordDesc) == 32, "FrWordDesc must be 32 bytes");
#define FR_DESC_DENSE 0x80000000u

// How gather_pack_xcd_kernel deals the record words to the 8 XCD groups: group g owns words [start[g], start[g + 1]).  The cuts
// sit on SOURCE-ROW boundaries (a table row, or a whole bank row of a bank-interleaved context): a row cut in two would be fetched
// by two XCDs -- the same 128-byte line through two L2s.  max_words = the widest group (the kernel's block size).
struct FrGatherGroups {
    int start[9];
    int max_words;
};

// One wave-instruction of the item-tile gather (gather_tile_kernel, fr_gather.hip): 64 / n_words items x n_words consecutive
// 16-byte words of ONE source row each (n_words = 1, 2, 4, 8 or 16: a whole table row, or a power-of-two piece of one).  A record
// segment of w words appears as w passes (item0 = 0, 64/w, 2*64/w, ...) so that the passes of a 64-item tile cover every item.
struct FrPassDesc {
    uint64_t src;        // TABLE/COPY: device address of the piece inside row 0; DENSE: byte offset inside the item's dense row
    uint32_t stride;     // bytes between rows of the source
    uint32_t idx_col;    // index column (bit 31: DENSE)
    uint32_t rows;       // range check (DENSE: unused)
    uint16_t tile_word;  // first word of the piece inside the chunk's LDS tile row
    uint8_t log2_words;  // log2(n_words)
    uint8_t item0;       // first item of the tile this pass serves
    uint32_t pad_;
};
static_assert(sizeof(FrPassDesc) == 32, "FrPassDesc must be 32 bytes");
constexpr int FR_TILE_ITEMS = 64, FR_TILE_WORDS = 64;  // gather_tile_kernel: items per workgroup, record words per chunk
struct FrChunkDesc {
    int pass_begin, pass_end;  // passes of the chunk inside ctx->d_passes
    int word0, n_words;        // the chunk's word range inside the item's destination row
};

// ---- pipelined FC chain: launch arguments (see fr_pipeline.hip) ------------------------------------
constexpr int FR_N_STAGES = 5;  // gather | FC1 | FC2 | FC3 | out

struct FrStageArgs {
    int block_begin;  // first workgroup of this stage inside the launch; == next stage's begin when inactive
    int batch;        // real items
    int ldm;          // padded items (multiple of 32) = leading dimension of every activation matrix of this batch
    int K, N;         // FC stages: reduction length / outputs; out stage: K = H3
    int nsplit;       // FC stages: workgroups along K per output tile (partials written to out + p*part_stride);
                      // K % (8 * nsplit) == 0 is required (groups of 8 k)
    int nparts_in;    // FC/out stages: partial inputs to add while loading (1 or 2)
    int variant;      // gather stage: LDS-transposing forms for large batches, 1 = gather_tr_body, 2 = gather_tr_stream_body
    int part_stride;  // floats between the partial OUTPUT buffers
    int in_part_stride;  // floats between the partial INPUT buffers
    int e_w, e_in, e_out;  // fp8 chain: power-of-two quantisation exponents of the weights, the input and the output activations
    const float *in;  // activations in (feature-major)
    float *out;       // activations out / scores
    const float *w;   // weights Wt[K][N] (FC) or w[H] (out)
};

struct FrPipeArgs {
    FrStageArgs st[FR_N_STAGES];
    int n_blocks;
    // gather stage
    const struct FrWordDesc *words;
    int src_lp;   // 1: `words` address the operand-type bank image (rows already bf16 / e4m3: 8 / 4 bytes per record word); dense words stay fp32
    int n_words;
    int idx_stride;
    const int32_t *idx;
    const float *dense;
    int *err_flag;
    // diagnostic build aid (NULL in normal operation): per workgroup {s_memrealtime at entry, at exit, stage, hw id}
    unsigned long long *stamps;
};

// ---- fused item-tile kernel: launch arguments (see fr_fused.hip) -------------------------------------
constexpr int FR_FUSED_MAX_BATCHES = 64;   // kernel-argument array size (64 x 32 B): what one launch of the kernarg-fed kernels carries
constexpr int FR_FUSED_MAX_QUEUE = 256;    // batches a worker queues / the persistent bf16 kernel carries per launch (its list lives in device memory)
constexpr int FR_BLIST_RING = 4;           // staging blocks of that list per worker
constexpr int FR_FUSED_DEFAULT_BATCHES = 64;
constexpr int FR_FUSED_MIN_GROUP = 12;       // fr_worker_push_device: smaller launch groups ride the stage pipeline (one launch per push)
constexpr int FR_HOST_BLOCKS = 4;  // 64 batches of 256 = one 64-item workgroup per CU (fr_fused_tile_m2_kernel)
struct FrFusedBatch {
    const int32_t *idx;
    const float *dense;
    float *scores;
    int batch;
    int pad_;
};
struct FrFusedArgs {
    FrFusedBatch b[FR_FUSED_MAX_BATCHES];
    int n_batches;
    int tiles_per_batch;  // max over the batches of ceil(batch / 32)
    const struct FrWordDesc *words;
    int src_lp;   // 1: `words` address the operand-type bank image (rows already bf16 / e4m3: 8 / 4 bytes per record word); dense words stay fp32
    int n_words;
    int idx_stride;
    int *err_flag;
    const FrFusedBatch *blist;      // fr_fused_tile_hs_kernel: the launch's batches in device memory (up to FR_FUSED_MAX_QUEUE; b[] is unused then)
    const float4 *w1q, *w2q, *w3q;  // q4-packed weights
    const float *wout;
    int K, H1, H2, H3;
    int e_w[3], e_act[4];        // fp8 kernel only: quantisation exponents (weights W1..W3; activations X, R1, R2, R3)
    unsigned long long *stamps;  // diagnostics only (NULL normally): 16 s_memrealtime stamps per workgroup
};

// ---- host objects -------------------------------------------------------------------------------
// Where a table's rows live inside ctx->table_arena.  A table stored on its own: row r at byte_offset + r * dim * 4 (il_rows == 0).
// FR_INDEX_PER_BANK contexts interleave the tables of one memory bank: rows [0, il_rows) of every table of the bank share one
// "bank row" of row_stride bytes (row r of this table at byte_offset + r * row_stride, byte_offset already including the table's
// column offset inside the bank row); rows [il_rows, rows) -- unreachable through a per-bank index, kept so that upload / download /
// fill see the whole table -- sit contiguously at tail_offset.
struct FrTableMem {
    uint64_t byte_offset = 0;  // inside ctx->table_arena: row 0
    uint64_t row_stride = 0;   // bytes between rows [0, il_rows) when interleaved; dim * 4 otherwise
    uint64_t il_rows = 0;      // rows held in the bank-interleaved region (0 = none)
    uint64_t tail_offset = 0;  // inside ctx->table_arena: row il_rows (interleaved tables only)
    bool resident = false;     // false when the table belongs to another shard
};

struct fr_ctx {
    int device = -1;
    bool cpu = false;                      // the CPU back-end (fr_ctx_create with device = -1, fr_cpu.cpp): every pointer below is host memory, no HIP call is made
    fr_model_desc model{};                 // deep copy (tables/segments point into the vectors below)
    std::vector<fr_table_desc> tables;
    std::vector<fr_segment> segments;
    std::vector<FrTableMem> table_mem;
    std::vector<int> bank_of_table;        // index column of every table in FR_INDEX_PER_BANK mode (banks numbered by first appearance)
    std::vector<int64_t> bank_rows;        // per bank: min rows over its tables = the valid range of the bank's index
    int n_banks = 0;
    char *table_arena = nullptr;           // one allocation holding every resident table, 256-B aligned starts
    size_t table_arena_bytes = 0;
    bool tables_filled = false;
    // record geometry
    int n_words = 0;                       // words this ctx gathers per item (whole record, or the shard's slice)
    FrWordDesc *d_words = nullptr;         // [n_words]
    std::vector<FrWordDesc> h_words;
    FrGatherGroups gather_groups{};        // XCD partition of the record words (gather_pack_xcd_kernel); max_words == 0: none
    // item-tile gather plan (SEMANTIC / sharded layouts): passes grouped in chunks of <= FR_TILE_WORDS record words
    FrPassDesc *d_passes = nullptr;
    FrChunkDesc *d_chunks = nullptr;
    int n_chunks = 0;
    // Operand-type bank image (round 6; FR_INDEX_PER_BANK contexts, bf16 / fp8 chains): the reachable rows [0, bank_rows[b]) of every bank
    // once more, already in the chain's operand type -- bf16, or e4m3 at the X exponent -- so that the in-chain gather of a large batch
    // fetches 82 lines per Model-C item instead of 142 and converts nothing.  Made from the fp32 arena by the SAME rounding functions the
    // gather applies (RNE at fill = RNE at gather: bit-identical scores); rebuilt lazily (lp_ensure_image) when the precision, the X exponent
    // or the table contents have changed.  The fp32 arena stays the master (gather_only, fp32 chain, calibration, upload / download).
    std::vector<int> h_word_table, h_word_col;   // per h_words entry: source table (-1: dense) and the float column inside its row
    char *lp_arena = nullptr;
    size_t lp_arena_bytes = 0;
    FrWordDesc *d_words_lp = nullptr;      // h_words with src / stride pointing into lp_arena (dense words unchanged)
    int lp_prec = 0, lp_e_x = 0;           // what the image holds (FR_FC_BF16 / FR_FC_FP8 and, for fp8, the X exponent); 0 = nothing
    uint64_t lp_tables_gen = 0;            // tables_gen the image was made from
    uint64_t tables_gen = 1;               // bumped by every fill / upload
    std::mutex lp_mutex;
    std::atomic<int> lp_image_on{1};       // fr_ctx_set_lp_bank_image (fleetrec_diag.h): 0 = the in-chain gather reads the fp32 rows and converts them itself
    unsigned long long *d_merged = nullptr;  // lookups merged by the dedup gather (FR_GATHER_ITEM_TILE_DEDUP_COUNT)
    std::atomic<int> gather_variant{0};    // fr_gather_variant (fr_ctx_set_gather_variant)
    // FC weights: fp32 master copies in the reference's column-major H x K layout
    // (= K-major "Wt[k][h]", cuda_server.c:215) and optional bf16 copies.
    float *d_w[4] = {nullptr, nullptr, nullptr, nullptr};
    float *d_wq[3] = {nullptr, nullptr, nullptr};  // FC1..FC3 re-packed as Wq[k/4][h][k%4] for the 16-byte operand loads
    uint16_t *d_w_bf16[4] = {nullptr, nullptr, nullptr, nullptr};
    // fp8 chain: e4m3 q16 copies of FC1..FC3 and the power-of-two exponents (weights per layer; activations X, R1, R2, R3)
    void *d_w_fp8[3] = {nullptr, nullptr, nullptr};
    void *d_w_fp8h[3] = {nullptr, nullptr, nullptr};  // the same weights in the "q16h" layout of the non-scaled fp8 MFMA (fr_fused_tile_hs_kernel<2, ...>); fused-eligible models only
    int f8_e_w[3] = {0, 0, 0};
    int f8_e_act[4] = {0, 0, 0, 0};
    float f8_w_rms_gain[3] = {1.0f, 1.0f, 1.0f};  // ||W_l||_F / sqrt(N_l): rms growth of layer l under uncorrelated inputs
    bool f8_calibrated = false;
    uint32_t *d_stats = nullptr;  // 16 words of reduction scratch (max |x| bits, sum x^2 per tensor)
    bool weights_set = false;
    int fc_precision = FR_FC_FP32;
    // sharding
    int shard_rank = 0, n_shards = 1;
    int slice_offset = 0, slice_len = 0, slice_padded = 0;
    std::vector<int> shard_offset, shard_len;  // record range of EVERY shard (floats), for the all-gathered layout
    hipStream_t setup_stream = nullptr;
    // batches one fused streaming launch carries (fr_ctx_set_stream_group): per context; read by every driver thread
    std::atomic<int> stream_group{FR_FUSED_DEFAULT_BATCHES};
    // host-fed blocks of at most this many batches take the stage pipeline instead of the fused kernel (fr_ctx_set_small_block; 0: never)
    std::atomic<int> small_block{0};
    int n_cu = 0;                          // compute units of the device (grid of the persistent fused kernel); set at creation
    int hk_ok = 0;                         // the persistent K-outer fused kernel applies to this context's descriptors; set at creation
    // Lifetime (round 6): one reference for the handle fr_ctx_create gave out, one per live worker and per communicator handle.  fr_ctx_destroy
    // drops the handle's; the context is released by whoever drops the last -- a worker or communicator destroyed AFTER its context no longer
    // walks freed memory (a failed test's teardown did exactly that: "corrupted double-linked list").
    std::atomic<int> life{1};
    std::atomic<int> n_workers{0};         // live workers of the context
    std::atomic<int> worker_seq{0};        // workers ever created on the context: a chain model's k-th worker takes the highest (k even) / lowest (k odd) stream priority
    // chain width W (fr_ctx_set_chain_width): the bf16 / fp8 GEMM layers of a chain model take tiles that cover 1 / W of the chip (lp_gemm_mu).
    // 0 = not decided yet: the first low-precision GEMM-layer launch on the context freezes it at min(live workers, 4); from then on only an
    // explicit fr_ctx_set_chain_width changes it -- never a worker coming or going
    std::atomic<int> chain_width{0};
    std::atomic<bool> chain_width_auto{false};   // the width was frozen by a launch (not by fr_ctx_set_chain_width / a driver): fr_worker_create leaves a note when workers outnumber it
};

struct fr_worker {
    fr_ctx *ctx = nullptr;
    int max_batch = 0;
    // CPU back-end: a call computes before it returns; an index-range error stays in c_err until fr_worker_sync reports it
    float *c_scratch = nullptr;  // [max_batch][H1 + H2 + H3]
    float *c_x = nullptr;        // records of the sharded FC entry points, [max_batch][K]
    int c_err = 0;
    double c_t0 = 0.0;           // fr_worker_timer_start (steady clock, seconds)
    hipStream_t stream = nullptr;
    // pinned host staging (cudaMallocHost in the reference, cuda_server.c:136-160)
    int32_t *h_idx = nullptr;
    float *h_dense = nullptr;
    float *h_score = nullptr;
    // device buffers (cudaMalloc in the reference, cuda_server.c:170-183)
    int32_t *d_idx = nullptr;
    float *d_dense = nullptr;
    float *d_records = nullptr; // [max_batch][K]
    // feature-major activations, two sets alternating by launch parity; per set:
    //   Xt[K][ld] | R1t[2][H1][ld] | R2t[2][H2][ld] | R3t[2][H3][ld]   ([2] = K-split partials)
    float *d_act[2] = {nullptr, nullptr};
    int ld_max = 0;             // round_up(max_batch, 64)
    // software pipeline over consecutive batches: slot of the batch whose gather ran in launch L is ring[L % 8]
    struct Slot {
        bool active = false;
        uint64_t launch0 = 0;   // launch number of its stage 0
        int first_stage = 0;    // 0 normally; 1 for fc_only (activations supplied by a transpose)
        int batch = 0, ldm = 0;
        const int32_t *d_idx = nullptr;
        const float *d_dense = nullptr;
        float *d_scores = nullptr;
    } ring[8];
    // Large batches (transposing gather + GEMM-kernel FC layers: Model-C at batch 4096): the gather of batch L runs on a SECOND stream of the
    // worker, beside the FC1 GEMM of batch L - 1 on the main one, ordered by events per activation-set parity: x_ready[p] = the gather into
    // X[p] has finished (aux -> main), x_free[p] = the FC1 that read X[p] has finished (main -> aux).
    hipStream_t aux = nullptr;
    hipEvent_t ev_x_ready[2] = {nullptr, nullptr}, ev_x_free[2] = {nullptr, nullptr};
    bool x_ready_set[2] = {false, false}, x_free_set[2] = {false, false};
    uint64_t launch_no = 0;     // number of pipeline launches issued so far
    int n_active = 0;
    bool counted = false;       // in fr_ctx::n_workers
    bool diag_launch = false;   // fr_worker_fc_layer_only: a single-layer diagnostic launch reads the chain width it WOULD run at and freezes nothing
    bool calibrating = false;   // fr_worker_calibrate_fp8: the pushed batch runs the fp32 stages without K-split partials
    int last_x_parity = 0;      // which activation set holds Xt of the most recently pushed batch (debug hook)
    // fused item-tile path: batches queued by fr_worker_push_device until a launch group is full
    FrFusedBatch pending[FR_FUSED_MAX_QUEUE];
    int n_pending = 0;
    // batch lists of the persistent kernel's launches: FR_BLIST_RING pinned host blocks + device copies, one H2D copy per launch on the
    // worker's stream; a block is rewritten only after the copy that read it has executed (its event)
    FrFusedBatch *h_blist = nullptr, *d_blist = nullptr;
    hipEvent_t ev_blist[FR_BLIST_RING] = {nullptr, nullptr, nullptr, nullptr};
    bool blist_busy[FR_BLIST_RING] = {false, false, false, false};
    int blist_cur = 0;
    int64_t pending_items = 0;
    float *d_score = nullptr;
    // host-fed streaming (fr_worker_push_host): FR_HOST_BLOCKS staging blocks of `g` batches each; a block is filled by CPU copies
    // into pinned memory, then moves as ONE H2D copy + ONE fused launch + ONE D2H copy; its event says when the scores are home
    struct HostRing {
        int g = 0;                 // batches per block (fixed when the ring is created)
        size_t idx_slot = 0, dense_slot = 0, score_slot = 0;  // elements per batch slot
        int32_t *h_idx = nullptr, *d_idx = nullptr;
        float *h_dense = nullptr, *d_dense = nullptr, *h_sc = nullptr, *d_sc = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
        hipStream_t copy = nullptr;   // the blocks' H2D copies run here, ahead of the launches on the worker's stream (ev_in: block b's rows have landed)
        hipEvent_t ev_in[4] = {nullptr, nullptr, nullptr, nullptr};
        bool inflight[4] = {false, false, false, false};
        int count[4] = {0, 0, 0, 0};          // batches in the block
        float *dst[4][FR_FUSED_MAX_BATCHES];  // where each batch's scores go
        int bsz[4][FR_FUSED_MAX_BATCHES];
        int cur = 0;               // block being filled
        int staged = 0;            // batch size handed out by fr_worker_stage_acquire and not yet pushed (0: none)
        long long delivered = 0;   // host-fed batches whose scores have been copied to their h_scores so far (fr_worker_host_poll)
    } hr;
    // table-sharded exchange (fr_worker_submit_sharded, fr_comm.cpp): this shard's slice, the all-gathered slices, score chunks
    void *d_slice = nullptr, *d_gathered = nullptr;
    float *d_score_part = nullptr, *d_score_all = nullptr;  // [chunk + 1], [G][chunk + 1]: every rank's chunk ends with its status word
    float *h_sh_status = nullptr;  // pinned: [0] this rank's status word as sent, [1 .. G] the status words of all ranks as received
    struct fr_comm *sh_comm = nullptr;  // a sharded step is in flight through this communicator (fr_worker_sync -> fr_comm_wait)
    int sh_ranks = 0;
    std::atomic<int> sh_inject_fc_fail{0};   // fr_worker_inject_fc_failure (fleetrec_diag.h): sharded steps left whose FC chain is reported as failed
    void *h_stage_send = nullptr, *h_stage_recv = nullptr;   // pinned staging of the STAGED host exchange (GPU shard contexts that share a device)
    void *sh_host_stream = nullptr;  // CPU workers: the host stream their sharded steps run on (fr_comm.cpp HostStream), made on first use
    int *h_err = nullptr;  // sticky index-range flag: pinned host word ...
    int *d_err = nullptr;  // ... and its device-side alias
    bool in_flight = false;
    char last_kernel[96] = "";  // fr_worker_last_kernel: the dominant kernel of the most recent launch this worker enqueued
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};

void fr_ctx_ref(fr_ctx *c);     // + 1 reference (a worker, a communicator handle)
void fr_ctx_unref(fr_ctx *c);   // - 1; the last one releases the context

// ---- table-sharded exchange (fr_comm.cpp) ------------------------------------------------------------
int fr_comm_wait(fr_worker *w);  // bounded wait for the sharded step in flight + the ranks' status words; FR_OK when none is in flight
bool fr_comm_step_on_host_stream(const fr_worker *w);   // a sharded step of this worker is being issued by its host stream right now
void fr_comm_worker_release(fr_worker *w);  // fr_worker_destroy of a CPU worker: stop its host stream (bounded), then fr_comm_forget
void fr_comm_forget(fr_worker *w);  // drop the step's hold on its communicator without waiting (fr_worker_sync leaving early)

// ---- CPU back-end (fr_cpu.cpp): device = -1 -----------------------------------------------------------
int frc_set_threads(int n);   // 0: all usable cores; -> the pool's size
int frc_threads();
void *frc_arena_alloc(size_t bytes);   // nullptr (+ fr_last_error) when the host cannot hold it
void frc_arena_free(void *p, size_t bytes);
int frc_fill_table(float *base, int64_t row0, int64_t rows, int dim, int64_t row_stride_bytes, int mode, uint32_t seed, uint32_t uid);
int frc_fill_weights(float *w, size_t count, int mode, uint32_t seed, uint32_t layer, float scale);
int frc_gather(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, float *out, int batch, int *err_flag);
void frc_fc_chain(const int32_t fc[5], const float *const w[4], const float *X, int batch, float *scratch, float *scores);
void frc_slices_to_records(const float *gathered, int n_shards, int batch_total, int slice_padded, const int *offs, const int *lens, int item0, int n_items, float *X, int K);
#define FR_NOT_ON_CPU(ctx_, what)                                                                                                  \
    do {                                                                                                                           \
        if ((ctx_)->cpu) FR_FAIL(FR_ERR_STATE, "%s is not available on the CPU back-end (device = -1): fp32 submit / sync / gather_only / fc_only only", what); \
    } while (0)

// ---- registry (fr_registry.cpp) -------------------------------------------------------------------
int fr_model_validate(const fr_model_desc *m);

// ---- kernel launchers (fr_*.hip) -------------------------------------------------------------------
// uid used by the procedural fills: source*1024 + class*256 + table_id
static inline uint32_t fr_table_uid(const fr_table_desc &t) {
    return (uint32_t)t.source * 1024u + (uint32_t)t.mem_class * 256u + (uint32_t)t.table_id;
}
int frk_fill_table(float *base, int64_t row0, int64_t rows, int dim, int64_t row_stride_bytes, int mode, uint32_t seed, uint32_t uid, hipStream_t s);
int frk_fill_weights(float *w, size_t count, int mode, uint32_t seed, uint32_t layer, float scale, hipStream_t s);
// rows [0, rows) of an fp32 region (row r at src + r * src_stride, `floats` floats wide) -> the same rows in the operand type of `precision`
// (bf16: RNE; fp8: x 2^e_x, saturated, e4m3) at dst + r * dst_stride: the rounding of the gather kernels themselves (fr_device.h)
int frk_convert_rows_lp(int precision, const void *src, size_t src_stride, void *dst, size_t dst_stride, int64_t rows, int floats, int e_x, hipStream_t s);
int frk_gather(const FrWordDesc *words, int n_words, const FrGatherGroups &groups, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch, int *err_flag,
               int transport, int e_x, hipStream_t s, int out_words, bool one_chunk = false);
int frk_gather_tile(const FrPassDesc *passes, const FrChunkDesc *chunks, int n_chunks, const int32_t *idx, int idx_stride, const float *dense, void *out,
                    int out_stride_words, int batch, int *err_flag, bool dedup, unsigned long long *dup_counter, hipStream_t s);
int frk_transpose_slices_lp(int precision, const void *gathered, int n_shards, int batch_total, int slice_padded, const int *h_offsets, const int *h_lens,
                            int item0, int n_items, void *X, int K, int ldm, hipStream_t s);
// feature-major FC chain, stage-pipelined across batches (see fr_pipeline.hip)
int frk_pipeline_launch(const FrPipeArgs &a, int single_stage, int precision, hipStream_t s);
int frk_pack_weights_q8_bf16(const float *W, uint16_t *Wh, int K, int H, hipStream_t s);
bool frk_fc_lp_gemm_ok(int precision, int K, int N, int ldm);
// width: the context's chain width (fr_ctx::chain_width) -> the larger tile already when it covers 1 / width of the chip (lp_gemm_mu)
int frk_fc_lp_gemm(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, int width, bool minor, hipStream_t s);
bool frk_fused_hk_takes_lp_rows(int K);   // the persistent bf16 fused kernel has an instantiation that reads operand-type (bf16) rows for this record length
bool frk_fc_gemm_gather_ok(int precision, int K, int N, int ldm);   // FC1 of batch L - 1 + the gather of batch L in one launch (fc_gemm_gather_kernel)
int frk_fc_gemm_gather(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, const FrWordDesc *words, int n_words,
                       int idx_stride, const int32_t *idx, const float *dense, int g_batch, int g_ldm, int g_K, void *g_out, int g_e_x, int *err_flag, hipStream_t s);
bool frk_fc_tail_ok(int precision, int K, int N, int ldm);   // FC3 + output layer as one launch (fc_tail_kernel, fr_gemm.hip)
int frk_fc_tail(int precision, const void *W3, const void *R2, const void *wout, float *scores, int K, int N, int ldm, int batch, int e_w, int e_in, int e_r3, hipStream_t s);
int frk_records_to_q8_bf16(const float *X, void *Xh, int batch, int K, int ldm, hipStream_t s);
int frk_pack_weights_q16_fp8(const float *W, void *Wf, int K, int H, int e_w, hipStream_t s);
int frk_pack_weights_q16h_fp8(const float *W, void *Wg, int K, int H, int e_w, hipStream_t s);  // the persistent fp8 kernel's operand layout
int frk_records_to_q16_fp8(const float *X, void *Xf, int batch, int K, int ldm, int e_x, hipStream_t s);
int frk_stats(const float *p, size_t n, void *d_out, hipStream_t s);
int frk_q4_to_lp(int precision, const float *Xq, void *Xo, int K, int ldm, int e_x, hipStream_t s);
int frk_stage_blocks_f8_gather(int K, int ldm);
int frk_pack_weights_q4(const float *W, float *Wq, int K, int H, hipStream_t s);
int frk_stage_blocks(int stage, int n_words, int K, int N, int ldm, int nsplit);
int frk_gather_tr_blocks(int n_words, int ldm, int variant);
int frk_gather_tr_variant(int batch, int idx_stride);
bool frk_fused_ok(int K, int H1, int H2, int H3);
int frk_fused_launch(const FrFusedArgs &a, hipStream_t s);
bool frk_fused_m2_ok(int K, int H1, int H2, int H3);
int frk_fused_m2_launch(const FrFusedArgs &a, hipStream_t s);
bool frk_fused_h_ok(int K, int H1, int H2, int H3);
bool frk_fused_f8_ok(int K, int H1, int H2, int H3);
int frk_fused_f8_launch(const FrFusedArgs &a, hipStream_t s);
int frk_fused_h_items_per_wg();
int frk_fused_h_launch(const FrFusedArgs &a, hipStream_t s);
bool frk_fused_hk_ok(int K, int H1, int H2, int H3, const FrWordDesc *h_words, int n_words);  // the K-outer persistent kernel (fr_fused_ko.hip), bf16 and fp8 forms
int frk_fused_hk_launch(const FrFusedArgs &a, int n_cu, int precision, hipStream_t s);  // 0 when the transposing gather does not apply
int frk_transpose_records(const float *X, float *Xt, int batch, int K, int ldm, hipStream_t s);
int frk_transpose_slices(const float *gathered, int n_shards, int batch_total, int slice_padded, const int *h_offsets, const int *h_lens,
                         int item0, int n_items, float *Xq, int ldm, hipStream_t s);
void fr_shard_bounds(const fr_model_desc &m, int n_shards, std::vector<int> &seg_begin);
// memory banks of the model: bank key = (source, mem_class, bank); numbered by first appearance in the table list
int fr_bank_map(const fr_model_desc &m, std::vector<int> &bank_of_table, std::vector<int64_t> &bank_min_rows);
