/*
 * fleetrec_diag.h -- measurement and parity hooks (bench.py's roofline legs, tests/): single-layer launches, gather-kernel variants,
 * which kernel ran, HIP-event timing on the worker's stream, views of internal buffers.  No counterpart in the reference.
 *
 * Part of the C-ABI of the MI355X-native FleetRec hot path; include/fleetrec.h is the boundary proper (the three spans of
 * thread_consume(), cuda_server.c:110-354,460-495, that SURVEY section 8(b) cuts).  Same conventions: plain C, opaque handles,
 * FR_OK or a negative fr_status, fr_last_error() for the text.  Citations are path:line under the reference tree (see fleetrec.h).
 */
#ifndef FLEETREC_DIAG_H
#define FLEETREC_DIAG_H

#include "fleetrec.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* The kernel (instantiation included, as rocprofv3 prints it) of the most recent launch this worker enqueued through
 * fr_worker_push_device / fr_worker_sync (the fused item-tile kernel that carried the group), fr_worker_fc_layer_only (that layer's
 * kernel) or fr_worker_gather_only / fr_worker_gather_slices (the gather kernel).  "" before the first such launch.  The pointer
 * stays valid for the worker's lifetime; measurement code uses it so that a roofline figure names the kernel that actually ran. */
const char *fr_worker_last_kernel(const fr_worker *w);
/* Roofline hook: launch ONE layer of the FC chain (0..2 = FC1..FC3, 3 = output layer) on the worker's resident
 * activations, exactly as submit() launches it. */
int fr_worker_fc_layer_only(fr_worker *w, int batch, int layer);
/* The same launch n times back to back from ONE native call: a roofline leg that prices a 57 us kernel on two streams side by side must not
 * be paced by its host loop (under rocprofv3 a Python-issued launch costs more than that kernel takes). */
int fr_worker_fc_layer_repeat(fr_worker *w, int batch, int layer, int n);
/* Which kernel serves the record-producing gather (fr_worker_gather_only; fp32 records, SEMANTIC layout or a shard slice) -- a tuning
 * knob with no counterpart in the reference (its gather is 28-47 independent HLS pipelines, embedding_47_krnl.cpp:645-740).
 *  WORD_MAJOR (default): one thread per 16-byte record word, lanes along the record (gather_pack_kernel): a wave never holds two
 *                        lookups of the same table, duplicate rows of a batch are merged by L1 / L2;
 *  ITEM_TILE           : 64 items x 64 record words per workgroup, lanes along (item, word-of-row), rows staged in LDS and written
 *                        out as whole 1 KiB record pieces (gather_tile_kernel<false>);
 *  ITEM_TILE_DEDUP     : the same with a wave-level merge of duplicate lookups (LDS hash + __shfl: only one lane per distinct
 *                        index loads the row) -- BASELINE.json north_star's "ballot/shuffle index dedup";
 *  ..._DEDUP_COUNT     : DEDUP + a __ballot count of the merged lookups (diagnostic; fr_ctx_gather_merged_lookups);
 *  WORD_MAJOR_ONE_CHUNK: the word-major mapping with one chunk of items per workgroup and no software pipeline
 *                        (gather_pack_xcd_kernel) -- what WORD_MAJOR itself falls back to when the records or the index buffer
 *                        of a launch reach 4000 MiB; selectable so that the fallback is parity-tested at ordinary sizes.
 * All variants produce bit-identical records.  DESIGN.md section 3.1 holds the measured A/B. */
typedef enum fr_gather_variant {
    FR_GATHER_WORD_MAJOR = 0, FR_GATHER_ITEM_TILE = 1, FR_GATHER_ITEM_TILE_DEDUP = 2, FR_GATHER_ITEM_TILE_DEDUP_COUNT = 3,
    FR_GATHER_WORD_MAJOR_ONE_CHUNK = 4
} fr_gather_variant;
int fr_ctx_set_gather_variant(fr_ctx *ctx, int variant);
int fr_ctx_gather_variant(const fr_ctx *ctx);
/* Lookups (rows) the DEDUP_COUNT variant did NOT load because another lane of the wave loaded the same row, summed since the last
 * reset.  Synchronises the device. */
int fr_ctx_gather_merged_lookups(fr_ctx *ctx, uint64_t *merged, int reset);
/* Diagnostic: how the word-major gather deals the record's 16-byte words to the chip's 8 XCDs at large batches -- group g owns words
 * [starts[g], starts[g + 1]) of the record (the shard's slice).  The cuts sit on source-row boundaries (a table row; a whole bank row
 * of an FR_INDEX_PER_BANK context), so that no row is fetched through two L2s.  Returns FR_ERR_STATE when the context has no such plan
 * (short records: every group would be narrower than a wave). */
int fr_ctx_gather_groups(const fr_ctx *ctx, int starts[9]);
/* Device pointer of the worker's own record buffer ([max_batch][record_len] floats). */
float *fr_worker_records_dptr(fr_worker *w);
/* Debug/parity hook: device pointer of the worker's feature-major activation buffer written by the gather stage
 * of the LAST submitted/pushed batch, in the chain's q4 layout: feature k of item m at
 * xq[((k/4)*ld + m)*4 + k%4] with ld = round_up(batch, 32).  Lets tests check the pipeline's own gather bit-exactly. */
float *fr_worker_features_dptr(fr_worker *w, int *ld_max);

/* Operand-type bank image: a FR_INDEX_PER_BANK context running the bf16 / fp8 chain keeps its reachable bank rows a
 * second time in the chain's operand type (bf16; e4m3 at the calibrated X exponent), made from the fp32 tables with the gather's own
 * rounding and rebuilt by itself whenever precision, calibration or table contents change; the in-chain gather of a large batch reads THAT
 * (82 lines per Model-C item instead of 142, nothing to convert).  Scores are bit-identical either way.  on = 0 makes the gather read the
 * fp32 rows and convert them itself (the A/B and parity hook; default 1; a context that finds no HBM for the image switches to 0 by itself: same
 * scores).  fr_ctx_lp_bank_image_bytes: HBM the image holds now (0: none). */
int fr_ctx_set_lp_bank_image(fr_ctx *ctx, int on);
size_t fr_ctx_lp_bank_image_bytes(const fr_ctx *ctx);

/* Failure-injection hook for the table-sharded step's failure protocol (fleetrec.h, kind (2)): the FC chains of this worker's next `steps`
 * fr_worker_submit_sharded calls are reported as failed (FR_ERR_STATE) AFTER they ran -- the rank still joins both collectives, its score
 * chunk travels as NaN and its status word makes every rank's fr_worker_sync return FR_ERR_COMM naming it.  steps = 0 disarms.  A test hook:
 * it lets the cross-rank protocol be exercised on one rank of G, in process, in the product build (tests/test_cpu_backend.py, also under
 * ThreadSanitizer); nothing else reads the counter. */
int fr_worker_inject_fc_failure(fr_worker *w, int steps);

/* HIP-event timing on the worker's stream (the stream the kernels are launched on). */
int fr_worker_timer_start(fr_worker *w);
int fr_worker_timer_stop_ms(fr_worker *w, float *ms); /* records stop, synchronises, returns elapsed */


#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
