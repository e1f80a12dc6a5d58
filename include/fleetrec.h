/*
 * fleetrec.h -- C-ABI of the MI355X-native FleetRec inference hot path.
 *
 * The reference (fpgasystems/GPU-FPGA-Recommendation-System) has no plugin/FFI interface: its GPU
 * server calls CUDA/cuBLASLt inline from thread_consume(), and the embedding lookup runs on an FPGA
 * behind a TCP hop.  This header is the boundary a maintainer of the reference would bind to in
 * order to replace BOTH stages (FPGA user_krnl gather/concat + GPU cublasLt FC chain) by one MI355X.
 * Every entry point cites the reference span it replaces.  Citations are path:line under the
 * reference tree; "cuda_server.c" = GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c,
 * "3-node" = the same file under GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/,
 * "embedding_N_krnl.cpp" = FPGA/kernel/user_krnl/embedding_N_krnl/src/hls/embedding_N_krnl.cpp,
 * "host.cpp" = FPGA/host/embedding_47_krnl/host.cpp.
 *
 * Plain C: opaque handles, plain pointers and sizes, no C++/torch types.  All functions return
 * FR_OK (0) or a negative fr_status; fr_last_error() gives a thread-local message.  Nothing here
 * ever exit()s or prints-and-continues (contrast checkCudaStatus, cuda_server.c:27-32).
 *
 * There is no SILENT CPU fallback: a context created on device >= 0 needs a gfx950 device and fails with FR_ERR_NO_DEVICE otherwise.
 * device = -1 asks for the CPU back-end explicitly (BASELINE configs[0], the reference's `make check
 * TARGET=sw_emu` plumbing run, FPGA/Makefile:154-158): the same symbols, fp32 only, the library's own host code (csrc/fr_cpu.cpp) --
 * see fr_ctx_create.
 */
#ifndef FLEETREC_H
#define FLEETREC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden */
#endif

/* 6 (round 6): fr_comm_init_all over CPU shard contexts = the in-process host exchange (the table-sharded step with G > 1 ranks without G GPUs);
 * fleetrec_diag.h gains fr_worker_inject_fc_failure, fr_ctx_set_lp_bank_image / fr_ctx_lp_bank_image_bytes (the operand-type bank image);
 * fr_ctx_set_fc_precision / fr_worker_create may succeed with a "note:" in fr_last_error().
 * 5 (round 5): fr_ctx_create(device = -1) = the CPU back-end, fr_cpu_set_threads; fr_ctx_set_chain_width / fr_ctx_chain_width (the GEMM tile
 * shape of a chain model no longer follows the number of live workers).
 * 4 (round 4): the surface is three headers -- this one (the three spans of thread_consume() that SURVEY section 8(b) cuts, the request
 * driver core and the table-sharded mode), fleetrec_serving.h (host-fed streaming / serving extensions) and fleetrec_diag.h (measurement
 * and parity hooks); fr_worker_submit_sharded all-gathers a status word behind every score chunk and fr_worker_sync bounds its wait for
 * the collectives.  3 (round 3): fr_ctx_set_stream_group is per context (1..256), groups below 12 ride the stage pipeline;
 * FR_INDEX_PER_BANK stores bank-interleaved tables; the library reads no environment variable.  A binding must refuse a library whose
 * fr_abi_version() differs from the header it was written against (the Python binding does, also for a build loaded through FR_LIB). */
#define FR_ABI_VERSION 6

typedef enum fr_status {
    FR_OK = 0,
    FR_ERR_INVALID = -1,     /* bad argument / malformed model description */
    FR_ERR_NO_DEVICE = -2,   /* no usable gfx950 device (device >= 0 never falls back to the CPU; device = -1 asks for it) */
    FR_ERR_OOM = -3,         /* device or pinned-host allocation failed */
    FR_ERR_HIP = -4,         /* HIP runtime error; see fr_last_error() */
    FR_ERR_INDEX_RANGE = -5, /* a lookup index was >= the table's row count (reference: silent OOB,
                                embedding_47_krnl.cpp:927-933) */
    FR_ERR_STATE = -6,       /* call sequence error (e.g. sync without submit, tables not filled) */
    FR_ERR_COMM = -7         /* exchange (RCCL / in-process host exchange) error in the table-sharded mode (fr_comm_*, fr_worker_submit_sharded) */
} fr_status;

/* Memory class a table lived in on the FPGA card.  Purely descriptive on MI355X (everything is in
 * HBM3E; small tables end up L2/Infinity-Cache resident), kept so reference table names map 1:1.
 * constants.hpp:30-38 (47), :29-41 (98), :31-43 (377). */
typedef enum fr_mem_class { FR_MEM_HBM = 0, FR_MEM_DDR = 1, FR_MEM_PLRAM = 2 } fr_mem_class;

/* One embedding table.  Row r of the table is `dim` consecutive fp32 (dim in {4,8,16,32}: whole
 * 128-bit AXI words, constants.hpp:4 `typedef ap_uint<128> axi_t`).  DATA_SIZE/PADDED_SIZE/
 * AXI_PADDED_SIZE/TABLE_SIZE/ADDR_AXI of the reference's generated constants.hpp. */
typedef struct fr_table_desc {
    int32_t mem_class; /* fr_mem_class */
    int32_t table_id;  /* N of DATA_SIZE_<CLASS>_N */
    int32_t source;    /* 0 for Models A/B; 0/1 = FPGA0/FPGA1 for Model C (3-node constant.h:25-27) */
    int32_t dim;       /* floats per row = 4 * AXI_PADDED_SIZE */
    int64_t rows;      /* TABLE_SIZE */
    int32_t bank;      /* memory bank on the card the table sat in */
    int32_t round;     /* position inside that bank's load_single_embedding_K_tables call */
    int64_t addr_axi;  /* ADDR_AXI_*: start address inside the bank, in 128-bit words */
} fr_table_desc;

typedef enum fr_segment_kind {
    FR_SEG_TABLE = 0, /* one table row: record[rec_offset .. +len) = table[src].row(idx)             */
    FR_SEG_COPY = 1,  /* pad: record[rec_offset .. +len) = table[src].row(idx)[src_col .. +len)
                         (Model-B floats [220,224) = first AXI word of PLRAM16's row,
                         embedding_98_krnl.cpp:1078,1099)                                            */
    FR_SEG_DENSE = 2  /* dense features from the request: record[..] = dense[item][src_col .. +len)
                         (Model-C: 64 floats from the CPU node, 3-node constant.h:27)                */
} fr_segment_kind;

/* One contiguous run of the per-item record ("wire order" = what gather_embeddings emits,
 * embedding_47_krnl.cpp:1097-1217, embedding_98_krnl.cpp:1331-1605, embedding_377_krnl.cpp:1665-1873). */
typedef struct fr_segment {
    int32_t kind;       /* fr_segment_kind */
    int32_t src;        /* table index (TABLE/COPY); unused for DENSE */
    int32_t src_col;    /* first float inside the source row / dense vector */
    int32_t rec_offset; /* first float inside the item's record */
    int32_t len;        /* floats, multiple of 4 */
    int32_t source;     /* which sender the segment came from: 0 = FPGA0 (A/B: the only one),
                           1 = FPGA1, 2 = CPU dense node */
} fr_segment;

/* Record layout switch.
 *  SEMANTIC: item b's record is K contiguous floats at records[b*K]; Model-C = [dense64|half0|half1].
 *  BLOCKED : the 3-node server's literal receive buffer (3-node cuda_server.c:515,541,566):
 *            [CPU: B x 64][FPGA0: B x 1952][FPGA1: B x 1952], then read by the GEMM as if it were
 *            B x 3968 item-major (3-node cuda_server.c:216-217).  Only differs for multi-source models. */
typedef enum fr_layout { FR_LAYOUT_SEMANTIC = 0, FR_LAYOUT_BLOCKED = 1 } fr_layout;

/* Index mode.
 *  PER_TABLE: idx[B][n_tables], column t = table t of the model (tables are listed in wire order).  A generalisation: the
 *             reference cannot address the tables of a bank independently.
 *  PER_ITEM : idx[B], one index per item reused for every table -- what the reference's built-in index source does
 *             (load_access_idx feeds the same stream to every bank: embedding_47_krnl.cpp:580-628,899-914).
 *  PER_BANK : idx[B][n_banks], ONE index per memory bank per item, reused for every table ("round") of that bank -- the
 *             kernel's real contract: load_single_embedding_{2,4,5}_tables reads s_idx_buffer once per item and addresses each
 *             round's table with it (embedding_98_krnl.cpp:1026-1040, embedding_377_krnl.cpp:1261-1290).  Banks are numbered by
 *             first appearance in the model's table list (fr_model_bank_map); bank key = (source, mem_class, bank).  A bank's
 *             index must be < the row count of EVERY table of the bank (the reference reads the next table instead,
 *             embedding_47_krnl.cpp:927-933; here FR_ERR_INDEX_RANGE).  In this mode the context stores the tables of a bank
 *             row-interleaved in HBM (row r of all its tables in one contiguous 112..256-byte "bank row", padded to whole
 *             128-byte lines where that saves a line), so a bank costs one contiguous fetch per item instead of one scattered
 *             row per table: 82 fetches instead of 376 rows for Model-C. */
typedef enum fr_index_mode { FR_INDEX_PER_TABLE = 0, FR_INDEX_PER_ITEM = 1, FR_INDEX_PER_BANK = 2 } fr_index_mode;

typedef struct fr_model_desc {
    char name[32];
    int32_t n_tables;
    int32_t n_segments;
    const fr_table_desc *tables; /* in wire order (order of first appearance in the record) */
    const fr_segment *segments;  /* in record order; must tile [0, record_len) exactly */
    int32_t record_len;          /* K: floats per item = FC input length (352 / 880 / 3968) */
    int32_t dense_len;           /* floats per item supplied by the request (0 / 0 / 64) */
    int32_t fc[5];               /* K, H1, H2, H3, OUT (constant.h:21-27; 3-node constant.h:25-33) */
    int32_t layout;              /* fr_layout */
    int32_t index_mode;          /* fr_index_mode */
} fr_model_desc;

typedef enum fr_builtin_model {
    FR_MODEL_A = 0, /* embedding_47_krnl : 47 tables -> 352 floats,  FC 352-1024-512-256-1            */
    FR_MODEL_B = 1, /* embedding_98_krnl : 98 tables -> 876(+4 pad) floats, FC 880-1024-512-256-1     */
    FR_MODEL_C = 2  /* 2 x embedding_377_krnl (188 tables, 1952 floats each) + 64 dense floats,
                       FC 3968-2048-512-256-1                                                        */
} fr_builtin_model;

/* Table / weight contents. */
typedef enum fr_fill_mode {
    FR_FILL_EVEN_ODD = 0, /* reference pattern: even rows 1.0f, odd rows 0.0f
                             (host.cpp:66-88 init_vectors; embedding_47_krnl.cpp:869-897 init_plram_t_1_table).
                             The reference only initialises the first 200 rows of HBM/DDR tables
                             (`#define DEBUG`, host.cpp:75-80); here every row follows the pattern. */
    FR_FILL_HASH = 1,     /* v = hash32(seed, table, row, col) mapped to [-1,1): the roofline workload  */
    FR_FILL_TAGGED = 2    /* u32 bit pattern (source<<31 | class<<29 | table_id<<21 | (row&0xffff)<<5 | (col&31)):
                             every float identifies where it came from -- pins the wire order          */
} fr_fill_mode;

typedef enum fr_weight_mode {
    FR_WEIGHTS_ONES = 0,   /* init_array(w, n, 1.0f), cuda_server.c:152-160 */
    FR_WEIGHTS_UNIFORM = 1 /* U(-1,1)/sqrt(K_layer) by hash32(seed, layer, element) */
} fr_weight_mode;

typedef enum fr_fc_precision {
    FR_FC_FP32 = 0, /* fp32 in, fp32 accumulate: CUBLAS_COMPUTE_32F + CUDA_R_32F (cuda_server.c:211) */
    FR_FC_BF16 = 1, /* bf16 operands on MFMA, fp32 accumulate (BASELINE configs 3/4) */
    FR_FC_FP8 = 2   /* OCP e4m3 operands on the CDNA4 scaled MFMA (v_mfma_scale_f32_32x32x64_f8f6f4), per-tensor
                       power-of-two scales, fp32 accumulate; output layer in fp32 (BASELINE configs[4]).  Where it pays: the GEMM
                       chain of Model-C shapes (0.55 of the fp8 peak per layer).  On the fused-kernel models (A, B) it is correct
                       but runs the chunked 64-item-tile kernel at ~0.19 of the fp8 peak, 14-19 % above FR_FC_BF16:
                       fr_ctx_set_fc_precision succeeds there and leaves a "note:" saying so in fr_last_error(). */
} fr_fc_precision;

typedef struct fr_ctx fr_ctx;       /* device + model + tables + weights; shared by all workers,
                                       like the single ltHandle (cuda_server.c:547-551) */
typedef struct fr_worker fr_worker; /* one stream + pinned/device buffers = one thread_consume()
                                       (cuda_server.c:101-354) */

/* ---- library ------------------------------------------------------------------------------- */
int fr_abi_version(void);
const char *fr_last_error(void); /* thread-local, never NULL */
/* Number of visible HIP devices (0 when none; never fails).  cuda_server.c:508-522 device probe. */
int fr_device_count(void);
/* Host threads the CPU back-end (device = -1) spreads a call over: n >= 1 sets it, n = 0 = every usable core (the default); returns the
 * number in use.  Process-wide; the back-end runs one parallel region at a time.  (No environment variable: the library reads none.) */
int fr_cpu_set_threads(int n);

/* ---- model descriptions ---------------------------------------------------------------------- */
/* Built-in reference models; returned pointer is static, never freed. */
const fr_model_desc *fr_model_builtin(int which /* fr_builtin_model */);
/* Heap copy of `src` with every table's row count replaced by
 *   rows' = clamp(round(rows * row_scale), min_rows, max_rows)   (max_rows <= 0: no upper clamp)
 * Used to shrink models for tests and to inflate them past 288 GB (BASELINE config 5). */
int fr_model_clone_scaled(const fr_model_desc *src, double row_scale, int64_t min_rows, int64_t max_rows,
                          fr_model_desc **out);
void fr_model_free(fr_model_desc *m); /* only for fr_model_clone_scaled results */
/* Σ rows*dim*4 over the model's tables. */
int64_t fr_model_table_bytes(const fr_model_desc *m);
/* int32 columns of one item's index row in the model's index_mode: n_tables, 1 or the number of banks (<= 0: invalid model). */
int fr_model_index_cols(const fr_model_desc *m);
/* Memory banks of the model, numbered by first appearance in the table list.  Returns the number of banks (negative fr_status on
 * error); bank_of_table (n_tables ints, may be NULL) receives each table's bank = its index column in FR_INDEX_PER_BANK mode;
 * bank_rows (one int64 per bank, may be NULL; size it n_tables) the smallest row count among the bank's tables = the valid range of
 * that bank's index. */
int fr_model_bank_map(const fr_model_desc *m, int32_t *bank_of_table, int64_t *bank_rows);

/* ---- context: replaces main()'s device probe + cublasLtCreate (cuda_server.c:508-522,547-551)
 *      plus the FPGA host's table set-up (host.cpp:264-423,691-731) ----------------------------- */
/* Validates `m`, selects `device`, allocates every table and the FC weights in HBM.
 * Shard [shard_rank, n_shards): n_shards == 1 keeps all tables; n_shards > 1 keeps only the tables of
 * segments assigned to this shard (table-ID sharding); see fr_ctx_shard_info.
 * device = -1: the CPU back-end.  Tables and weights live in host memory (refused with FR_ERR_OOM when they cannot fit the host's RAM),
 * the gather walks the same record-word descriptors the gfx950 kernels walk, the FC chain is a k-ordered fp32 multiply-add chain over
 * the column-major weights (CUBLAS_COMPUTE_32F, cuda_server.c:211).  Available on such a context: everything in the "context" and
 * "worker" sections, fr_worker_submit / submit_device / push_device (each computes before it returns) / sync, fr_worker_gather_only,
 * fr_worker_fc_only, the fp32 forms of fr_worker_gather_slices / fr_worker_fc_from_slices(_lp), the fr_driver_* loops except the
 * host-fed streaming one, the fr_device_* helpers ("device memory" is host memory there, so the same calling code runs against either
 * back-end).  Not available (FR_ERR_STATE / FR_ERR_INVALID): bf16 / fp8 chains, host-fed streaming, RCCL, the diagnostics of
 * fleetrec_diag.h that name kernels.  Records are bit-identical to the device's; fp32 scores agree to ~1e-6 (another summation order). */
int fr_ctx_create(const fr_model_desc *m, int device, fr_ctx **out);
int fr_ctx_create_sharded(const fr_model_desc *m, int device, int shard_rank, int n_shards, fr_ctx **out);
/* Workers, drivers and communicator handles that are still alive keep the context (tables, weights) until the last of them is destroyed:
 * destroying a context first is allowed and leaves nothing dangling. */
void fr_ctx_destroy(fr_ctx *ctx);
const fr_model_desc *fr_ctx_model(const fr_ctx *ctx);

/* Fill every resident table on the device (no host staging: Model-C is 63 GB). */
int fr_ctx_fill_tables(fr_ctx *ctx, int mode /* fr_fill_mode */, uint32_t seed);
/* Upload / read back rows [row0, row0+nrows) of table t (host arrays of nrows*dim floats).
 * host.cpp:739-749 enqueueMigrateMemObjects. */
int fr_ctx_upload_table(fr_ctx *ctx, int table, int64_t row0, int64_t nrows, const float *host_rows);
int fr_ctx_download_table(fr_ctx *ctx, int table, int64_t row0, int64_t nrows, float *host_rows);

/* FC weights.  layer 0..3 = W1, W2, W3, Wout; `w` is column-major H x K with ld = H
 * (element (h,k) at w[h + k*H]) exactly as cuda_server.c:215 hands it to cuBLASLt.  No bias and no
 * activation exist on the reference path (cuda_server.c:403,468-491: alpha=1, beta=0). */
int fr_ctx_set_weights(fr_ctx *ctx, int layer, const float *w_colmajor, size_t count);
int fr_ctx_fill_weights(fr_ctx *ctx, int mode /* fr_weight_mode */, uint32_t seed);
int fr_ctx_get_weights(fr_ctx *ctx, int layer, float *w_colmajor, size_t count);
int fr_ctx_set_fc_precision(fr_ctx *ctx, int precision /* fr_fc_precision */);
/* fp8 chain: quantisation exponents.  Tensor T is stored as e4m3(saturate(T * 2^e)): w_exp[l] for W1..W3 (chosen from max|W|
 * when the weights are packed), act_exp[l] for X, R1, R2, R3 (an rms estimate until fr_worker_calibrate_fp8 or
 * fr_ctx_set_fp8_act_exponents replaces it).  No counterpart in the reference (its chain is fp32 only, cuda_server.c:211). */
int fr_ctx_get_fp8_exponents(const fr_ctx *ctx, int act_exp[4], int w_exp[3]);
int fr_ctx_set_fp8_act_exponents(fr_ctx *ctx, const int act_exp[4]);

/* ---- worker: replaces thread_consume()'s set-up (cuda_server.c:110-354) --------------------- */
/* One worker = one stream + its buffers (the reference's per-thread cudaStream_t, pinned and device buffers).  Workers of one context
 * share the chip: a model that runs as a chain of launches per batch (Model-C; any sharded context) gets each worker a hardware queue of
 * its own -- the k-th worker EVER created on the context takes the highest (k even) or the lowest (k odd) stream priority, never the
 * default one.  Side effect across contexts of one process: a lowest-priority worker can be held back by default-priority streams of
 * other contexts or of a host framework, a highest-priority one runs ahead of them (measured on this chip: the priority only selects the
 * queue pool, profiles/archive/r04_experiments.md section 6.7); fused-kernel models (A, B) keep default-priority streams. */
int fr_worker_create(fr_ctx *ctx, int max_batch, fr_worker **out);
/* Chain width W (1..4) of a chain model: its bf16 / fp8 GEMM layers use tiles that cover 1 / W of the chip, so that W workers' chains run
 * side by side on part-chip tiles (fewer operand bytes per output) instead of time-sharing every compute unit: Model-C batch 4096, W = 4:
 * bf16 +16 %, fp8 +10 %; a lone busy worker at W = 4 loses 15-18 % (profiles/archive/r04_C4096_half_chip_tiles_ab.txt).  CONTRACT: the width is
 * decided ONCE per context -- by this call, or else by the context's first low-precision GEMM-layer launch of a submit / push path, which
 * freezes it at min(workers alive at that moment, 4) (calibration batches and fleetrec_diag.h's single-layer launches freeze nothing; a
 * fr_worker_create that makes the workers outnumber a width frozen that way succeeds and leaves a note in fr_last_error()) -- and never
 * follows workers coming or going afterwards: scores of a stream in flight cannot
 * change because an unrelated worker was created or destroyed.  Call it again only on purpose: it takes effect from the next launch on,
 * and in bf16 the tile shape fixes the summation order (<= 1e-2 relative between widths; fp8 and fp32 scores are bit-identical for every
 * width; width = 0 makes the context undecided again).  fr_ctx_chain_width: the current value, 0 while undecided.  Set W = 1 if a runtime update should ever stop giving the
 * workers hardware queues of their own (tests/test_gpu_chain.py::test_chain_workers_run_their_layers_side_by_side guards that). */
int fr_ctx_set_chain_width(fr_ctx *ctx, int width);
int fr_ctx_chain_width(const fr_ctx *ctx);
void fr_worker_destroy(fr_worker *w);
/* Pinned host staging buffers the driver's socket read() lands in directly (cuda_server.c:437 reads
 * into pinned input_feature): int32 idx[max_batch][fr_model_index_cols(model)],
 * float dense[max_batch][dense_len] (NULL when dense_len == 0), float score[max_batch]. */
int32_t *fr_worker_idx_ptr(fr_worker *w);
float *fr_worker_dense_ptr(fr_worker *w);
float *fr_worker_score_ptr(fr_worker *w);
/* The worker's HIP stream (a hipStream_t), so that a host framework can order its own streams against the worker's without a
 * host synchronisation (e.g. torch.cuda.ExternalStream for the RCCL exchange of the sharded mode). */
void *fr_worker_stream(fr_worker *w);

/* ---- hot loop body (cuda_server.c:460-495) --------------------------------------------------- */
/* Asynchronous: idx(+dense) host -> device, gather+pack, 4-GEMM FC chain, scores device -> host, all on the worker's
 * stream.  Exactly one batch may be in flight per worker: fr_worker_sync() must be called before the
 * pinned buffers are touched again (this fixes the reference's unsynchronised reuse, cuda_server.c:406-497).
 * The two PCIe hops are not copy commands by default: the gather stage reads the pinned index rows and the output layer writes the
 * pinned score buffer directly (4-8 us less per submit + sync than the reference's H2D / D2H commands; the experiments build of the
 * library -- `make -C csrc exp` -- keeps them behind FR_SUBMIT_ZEROCOPY=0 for A/B timing). */
int fr_worker_submit(fr_worker *w, int batch);
/* Same, with inputs/outputs already resident in HBM (device pointers; no PCIe traffic):
 * d_idx int32 [batch][fr_model_index_cols(model)]; d_dense float [batch][dense_len] or NULL;
 * d_scores float [batch]. */
int fr_worker_submit_device(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense,
                            float *d_scores);
/* Streaming form of the hot loop: enqueue batch after batch WITHOUT synchronising, as the reference's loop does
 * (cuda_server.c:406-497).  Any FC precision; needs an unsharded, SEMANTIC-layout context.  Two execution forms, chosen per context:
 *  - models whose activations fit in LDS (A, B) stream through the fused item-tile kernel: pushed batches are QUEUED on the worker and
 *    one launch carries fr_ctx_stream_group(ctx) of them (or fewer once 16384 items are queued; 262144 in the bf16 chain); nothing runs before the group is full
 *    or fr_worker_sync() is called;
 *  - every other model rides the stage pipeline: each push issues one launch in which this batch is gathered while the previous four
 *    batches of the worker advance through FC1, FC2, FC3 and the output layer.
 * BUFFER LIFETIME (both forms): d_idx / d_dense / d_scores of a pushed batch must stay valid, distinct and untouched until
 * fr_worker_sync(w) returns -- the only completion point this API defines.  A caller that wants to bound its buffer count rotates
 * R >= 2 * max(fr_ctx_stream_group(ctx), 5) buffer sets and calls fr_worker_sync once per trip round the ring (what
 * fr_driver_run_resident does with R = 512). */
int fr_worker_push_device(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, float *d_scores);
/* How many pushed batches one streaming launch carries on this context: 1 when fr_worker_push_device rides the stage pipeline
 * (models that do not fit the fused kernel), G (1..256, default 64) when
 * the model streams through the fused item-tile kernel -- the value fr_ctx_set_stream_group set, also when it is below 12 and the
 * pushes ride the stage pipeline. */
int fr_ctx_stream_group(const fr_ctx *ctx);
/* Throughput/latency knob of the fused streaming path, PER CONTEXT: batches per launch, 1..256 (default 64).  Groups above 64 exist for
 * the bf16 chain: its persistent kernel (fr_fused_tile_hs_kernel: one workgroup per compute unit walks several 64-item tiles, the gather
 * of the next tile under the FC phases of the current one) takes a launch with at least 1.25 tiles per compute unit (records of 352 floats or fewer: one) and up to 256 batches
 * (Model-A's batches of 256 items need a group of 128+ for that; batches of 1024 reach it at the default); every other kernel carries at
 * most 64 batches per launch and a larger group is launched in slices of 64.  64 batches of 256 items = one 64-item
 * workgroup per CU (fp32: fr_fused_tile_m2_kernel, used only for a group of 64 AND a launch of more than 128 such tiles); smaller groups
 * take the 32-item kernel, which halves the queueing latency of a pushed batch and
 * leaves CUs to other streams (a partial launch -- fr_worker_sync with few batches queued -- whose 64-item tiles would cover at most half
 * of the CUs takes the 32-item kernel too).  Scores are bit-identical for every group size from 12 up.  Groups below 12 make
 * fr_worker_push_device ride the stage pipeline instead (one launch per push; the batch's scores are complete four pushes later or at
 * fr_worker_sync): a fused launch costs one item tile's time however few batches it carries, so small groups are both slower and later
 * than the pipelined stage launches (Model-A batch 256: group 8 = 35 M inferences/s at 148 us, stage pipeline = 43 M at 36 us); the
 * pipeline's fp32 sums run in another order (scores agree to 1e-5 relative, not to the bit).  May be called while workers are pushing
 * (atomic); a worker's queue that already holds >= the new size launches at its next push or sync, and a worker whose path changes
 * drains the other path first. */
int fr_ctx_set_stream_group(fr_ctx *ctx, int batches_per_launch);
/* Launches whatever is still queued, drains the pipeline and waits for everything enqueued on the worker; returns
 * FR_ERR_INDEX_RANGE if any index was out of range. */
int fr_worker_sync(fr_worker *w);
/* fp8 chain: run `batch` items (the worker's pinned idx/dense buffers, as for fr_worker_submit) through the fp32 chain, take
 * max|.| of X, R1, R2, R3 and set the context's activation exponents so that twice that maximum still fits e4m3's 448.
 * Synchronous. */
int fr_worker_calibrate_fp8(fr_worker *w, int batch);
/* The same for table-sharded contexts: calibrate on all-gathered slices (arguments as fr_worker_fc_from_slices). */
int fr_worker_calibrate_fp8_slices(fr_worker *w, int batch_total, int item0, int n_items, const float *d_gathered);

/* Diagnostic / roofline entry points (same kernels as submit, run alone).
 * gather_only: d_records receives batch*record_len floats in the model's layout (device pointer).
 * fc_only    : d_records is read in the model's layout; d_scores receives batch floats.
 * Both are asynchronous on the worker's stream; follow with fr_worker_sync(). */
int fr_worker_gather_only(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense,
                          float *d_records);
int fr_worker_fc_only(fr_worker *w, int batch, const float *d_records, float *d_scores);
/* ---- request-driver core: main() + the thread_consume() batch loop without sockets
 *      (cuda_server.c:23-25,406-497,554-560) ------------------------------------------------------- */
typedef struct fr_driver fr_driver;
/* n_threads host threads (THREAD_NUM, constant.h:42), each owning `depth` workers/streams.  A context whose chain width is still undecided
 * takes min(n_threads * depth, 4) from its first driver (fr_ctx_set_chain_width). */
int fr_driver_create(fr_ctx *ctx, int n_threads, int depth, int max_batch, fr_driver **out);
void fr_driver_destroy(fr_driver *d);
/* Processes `total_batches` batches of `batch` items: threads draw batch ids from a mutex-guarded global
 * counter (cuda_server.c:408-417); batch id i reads the HBM-resident index rows d_idx_pool[i % n_pool]
 * (and d_dense_pool[i % n_pool] when the model has dense features; may be NULL otherwise).
 * Returns wall time from first submit to last completion (device drained on both sides). */
int fr_driver_run_resident(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *d_idx_pool,
                           const float *const *d_dense_pool, int n_pool, double *elapsed_s);
/* Host-buffer form: batch id i copies h_idx_pool[i % n_pool] (pageable host memory; stands in for the socket read()) into a
 * worker's pinned buffers, then fr_worker_submit + fr_worker_sync -- the reference's per-batch sequence including the PCIe
 * transfers (cuda_server.c:425-495).  Returns wall time. */
int fr_driver_run_host(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *h_idx_pool,
                       const float *const *h_dense_pool, int n_pool, double *elapsed_s);
fr_worker *fr_driver_worker(fr_driver *d, int thread, int slot);
/* Device pointer of the score ring fr_driver_run_resident writes for worker (thread, slot): *ring_len buffers of max_batch
 * floats; the k-th batch pushed to that worker lands in buffer k % *ring_len. */
const float *fr_driver_score_ring(fr_driver *d, int thread, int slot, int *ring_len);

/* ---- device memory helpers (so hosts/tests need no other GPU runtime binding) ---------------- */
int fr_device_malloc(fr_ctx *ctx, size_t bytes, void **dptr);
int fr_device_free(fr_ctx *ctx, void *dptr);
int fr_memcpy_h2d(fr_ctx *ctx, void *dptr, const void *host, size_t bytes);
int fr_memcpy_d2h(fr_ctx *ctx, void *host, const void *dptr, size_t bytes);
int fr_device_synchronize(fr_ctx *ctx);

/* ---- table-sharded mode (BASELINE configs 4/5) --------------------------- */
/* For a sharded ctx: first record float and float count of this shard's slice, and the padded
 * per-shard slice length F (equal on all shards; slices are whole segments). */
int fr_ctx_shard_info(const fr_ctx *ctx, int *shard_rank, int *n_shards, int *slice_offset,
                      int *slice_len, int *slice_padded_len);
/* Pure host query (no device): the plan fr_ctx_create_sharded uses -- per shard the first record float and the float
 * count of its slice (arrays of n_shards ints), and the common padded slice length F. */
int fr_model_shard_plan(const fr_model_desc *m, int n_shards, int *slice_offset, int *slice_len, int *slice_padded_len);
/* After the all-gather: d_gathered = [n_shards][batch_total][F] floats (every shard's padded slice, as
 * fr_worker_gather_only wrote it on that shard, concatenated in shard order = what ncclAllGather delivers).
 * Runs the FC chain for items [item0, item0+n_items) of the batch on this GPU (FC weights are replicated);
 * d_scores receives n_items floats.  Asynchronous on the worker's stream; follow with fr_worker_sync(). */
int fr_worker_fc_from_slices(fr_worker *w, int batch_total, int item0, int n_items, const float *d_gathered, float *d_scores);
/* Low-precision slice TRANSPORT for the sharded mode: the shard's slice as bf16 (2 bytes per float) or e4m3 (1 byte, scaled by the
 * context's X exponent) -- half / a quarter of the all-gather bytes -- and the FC entry point that consumes [n_shards][batch_total]
 * [slice_padded] elements of that type.  transport = fr_fc_precision; it must equal the context's FC precision (FR_FC_FP32 falls
 * back to fr_worker_gather_only / fr_worker_fc_from_slices).  Scores are bit-identical to fp32 transport in the same precision. */
int fr_worker_gather_slices(fr_worker *w, int batch, const int32_t *d_idx, const float *d_dense, void *d_slice, int transport);
int fr_worker_fc_from_slices_lp(fr_worker *w, int batch_total, int item0, int n_items, const void *d_gathered, int transport, float *d_scores);

/* ---- the exchange step on RCCL: the sharded hot-loop body behind the ABI ------------------------------------------------------
 * The 3-node reference server receives every batch in three parts from three senders before its GEMMs (3-node cuda_server.c:513-591).
 * Here the parts are the table-ID shards held by G GPUs of one node; librccl.so is loaded on first use of these entry points.
 * One process per GPU: rank 0 calls fr_comm_unique_id, the host ships the 128 bytes to the other ranks (socket, file, launcher),
 * every rank calls fr_comm_init_rank with its sharded context (rank = shard_rank, size = n_shards; collective).
 * One process driving G GPUs: fr_comm_init_all(ctxs, G, comms) with ctxs[r] = shard r on its own device (ncclCommInitAll); the
 * collective calls below must then come from G different threads.
 * CPU contexts (device = -1; round 6): fr_comm_init_all over the G CPU shard contexts of one process sets up the IN-PROCESS HOST EXCHANGE
 * instead -- the same step, status words, reference counts and bounded wait above another transport (a rendezvous of the ranks' host
 * streams; librccl.so is not touched).  A CPU worker's step runs on a host stream of its own behind fr_worker_submit_sharded, so one thread
 * may submit on all G workers and then synchronise them, or G threads may drive one rank each.  fp32 only; fr_comm_init_rank /
 * fr_comm_unique_id (ranks in different processes) remain RCCL-only.
 * GPU shard contexts of which two SHARE a device (RCCL wants one device per rank): fr_comm_init_all gives them the same host exchange, STAGED
 * (every all-gather = D2H into pinned staging, the rendezvous, H2D of all G parts; the step is issued by a host stream per worker, the
 * kernels run on the worker's HIP stream) -- every precision, every entry point; PCIe-bound, meant for rehearsing a G-rank job on fewer
 * GPUs.  fr_worker_calibrate_fp8_sharded stays a synchronous collective there: one thread per rank. */
typedef struct fr_comm fr_comm;
int fr_comm_unique_id(void *id128);
int fr_comm_init_rank(fr_ctx *ctx, const void *id128, fr_comm **out);
int fr_comm_init_all(fr_ctx *const *ctxs, int n, fr_comm **out /* n handles */);
void fr_comm_destroy(fr_comm *c);
/* How long fr_worker_sync lets the collectives of a sharded step take on this rank before it gives the peers up (default 60000 ms): the
 * stream is polled together with ncclCommGetAsyncError; past the bound the rank aborts its communicator and returns FR_ERR_COMM (from
 * then on for every call).  The reference blocks in read() for ever when a sender dies (3-node cuda_server.c:513-591). */
int fr_comm_set_wait_ms(fr_comm *c, int wait_ms);
/* COLLECTIVE, asynchronous on the worker's stream: every rank holds the whole request batch in its worker's pinned idx / dense
 * buffers (as for fr_worker_submit).  Per rank: H2D -> gather of this shard's slice (in the chain's operand type: fp32, bf16 or e4m3)
 * -> ncclAllGather of the [batch x F] slices over xGMI -> FC chain on this rank's batch / G items -> ncclAllGather of the score
 * chunks -> D2H.  After fr_worker_sync() EVERY rank's fr_worker_score_ptr() holds all `batch` scores.  RCCL failures: FR_ERR_COMM.
 * Failure protocol: argument / state errors are returned before anything is enqueued and leave the communicator usable (the ranks of a job
 * are driven with the same arguments); a rank whose FC chain fails still takes part in both collectives, its score chunk travels as NaN
 * and its status word (one float all-gathered behind every chunk) makes EVERY rank's fr_worker_sync return FR_ERR_COMM naming it; a
 * device / RCCL failure aborts the rank's communicator, and the peers' waits are bounded (fr_comm_set_wait_ms).  (Host exchange: the step
 * runs behind the call, so a rank whose FC chain fails learns it from its own fr_worker_sync like its peers; an abort releases every rank
 * of the in-process group at once.)
 * A call that returns an error AFTER its first collective (the local FC chain failed) has still enqueued the step: the caller must call
 * fr_worker_sync(w) before the worker's next submit (it returns FR_ERR_COMM naming the rank).  fr_comm_destroy between a submit and its
 * fr_worker_sync is safe: the worker keeps the communicator alive until it has synchronised. */
int fr_worker_submit_sharded(fr_worker *w, fr_comm *comm, int batch);
/* COLLECTIVE, synchronous: fp8 activation exponents of a sharded context from this batch -- every rank calibrates on the same
 * all-gathered fp32 slices, so all ranks end with identical exponents (a slice encoded by one rank is decoded by the others). */
int fr_worker_calibrate_fp8_sharded(fr_worker *w, fr_comm *comm, int batch);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* FLEETREC_H */
