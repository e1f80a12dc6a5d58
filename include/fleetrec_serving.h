/*
 * fleetrec_serving.h -- serving extensions: host-fed streaming with pinned staging blocks, replies, adaptive batching.
 *
 * None of this is needed to replace the reference's loop (fr_worker_submit + fr_worker_sync per batch does that, fleetrec.h); it is what
 * a server that feeds the GPU from sockets at the GPU's own rate uses (host/fleetrec_server.cpp --stream [--reply]).
 *
 * Part of the C-ABI of the MI355X-native FleetRec hot path; include/fleetrec.h is the boundary proper (the three spans of
 * thread_consume(), cuda_server.c:110-354,460-495, that SURVEY section 8(b) cuts).  Same conventions: plain C, opaque handles,
 * FR_OK or a negative fr_status, fr_last_error() for the text.  Citations are path:line under the reference tree (see fleetrec.h).
 */
#ifndef FLEETREC_SERVING_H
#define FLEETREC_SERVING_H

#include "fleetrec.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* n consecutive fr_worker_push_device calls in one: batch[i], d_idx[i], d_dense[i] (the array or any entry may be NULL for a model
 * without dense features), d_scores[i] for i = 0 .. n-1, in that order, stopping at the first error (its status is returned; the
 * batches before it stay pushed).  For callers whose per-call cost is comparable to a batch's share of a launch -- a language
 * binding feeding 256 batches of a 0.5 us share each -- so that the stream, not the caller, sets the pace (bench.py's one-stream
 * roofline legs).  Same buffer-lifetime rule as fr_worker_push_device.  The reference's loop is the n = 1 case (cuda_server.c:406-497). */
int fr_worker_push_device_list(fr_worker *w, int n, const int *batch, const int32_t *const *d_idx, const float *const *d_dense,
                               float *const *d_scores);
/* Host-fed streaming: like fr_worker_push_device for a batch that sits in (any) host memory.  The rows are copied into the worker's
 * pinned staging before the call returns (h_idx / h_dense may be reused at once); batches travel in blocks: ONE H2D copy on the worker's
 * copy stream (issued when the block is full, while the previous block's kernel still runs) + ONE launch whose output layer writes the
 * scores straight into pinned memory -- the worker's own stream carries nothing but kernels (round 5; rounds 2-4 issued H2D, launch and
 * D2H as three commands of that stream and lost 5 % to it: profiles/r05_host_fed_timeline.txt).  h_scores[0..batch) is valid after
 * fr_worker_sync (earlier deliveries happen -- a block's scores are copied out before its staging is reused, i.e. at the latest 4 blocks
 * later -- but fr_worker_sync is the only completion point the API defines).  Feed it from at most four streaming workers per context
 * (one worker stream per hardware queue).  The streaming counterpart of the per-batch recv -> H2D -> GEMMs -> D2H sequence of
 * cuda_server.c:425-495. */
int fr_worker_push_host(fr_worker *w, int batch, const int32_t *h_idx, const float *h_dense, float *h_scores);
/* The same without the copy into staging -- the reference's read() lands in pinned memory (cuda_server.c:136-160,437):
 * fr_worker_stage_acquire hands out where the NEXT pushed batch of this worker has to be written (*h_idx: batch x index_cols int32,
 * *h_dense: batch x dense_len floats or NULL; pinned, owned by the worker; it may first wait for the oldest block's scores and deliver
 * them, exactly as fr_worker_push_host does), the caller fills it (e.g. reads the socket into it), fr_worker_push_staged queues it
 * (batch <= the acquired size; h_scores as for fr_worker_push_host).  One slot at a time per worker; fr_worker_push_host between the
 * two calls is FR_ERR_STATE; fr_worker_sync drops a slot that was acquired and never pushed. */
int fr_worker_stage_acquire(fr_worker *w, int batch, int32_t **h_idx, float **h_dense);
int fr_worker_push_staged(fr_worker *w, int batch, float *h_scores);
/* Serving with replies (fleetrec_server --stream --reply): fr_worker_flush launches what is queued on the worker right now -- a
 * partially filled host block, queued device pushes -- without waiting (the latency knob under light load: call it when the request
 * source runs dry); fr_worker_host_poll delivers the scores of the host-fed blocks that have FINISHED (oldest first, no waiting) and
 * reports how many host-fed batches have been delivered since the worker was created: batches are delivered in push order, so the
 * caller knows exactly which h_scores buffers are valid. */
int fr_worker_flush(fr_worker *w);
/* Latency of nearly empty host-fed blocks, PER CONTEXT: a block that leaves with at most max_batches batches (0..8; default 0 = never) rides
 * the stage pipeline of fr_worker_submit -- its n batches follow each other through the five stage launches, n + 4 launches of ~10 us,
 * the whole chip per layer -- instead of the fused item-tile kernel (133 us for any number of batches up to a chip-full).  Such batches
 * get fr_worker_submit's scores bit for bit (the fused kernel sums in another order: equal to ~1e-6, not bit for bit).
 * fleetrec_server --stream --reply sets 8 (4 requests in flight per connection: 18 M inferences/s at 185 us request -> reply, against
 * 14 M at 245 us with five launches per batch; profiles/archive/r02_tcp_reply_small_blocks.txt). */
int fr_ctx_set_small_block(fr_ctx *ctx, int max_batches);
int fr_worker_host_poll(fr_worker *w, long long *delivered);
/* Host-fed batches queued in the block being filled (not launched yet) / launched and not delivered yet, and the number of launched
 * blocks not delivered yet (at most 4) -- what an adaptive batcher needs: flush when the request source is dry AND at most one block is
 * still in flight; while more are running, let the next block fill (any output pointer may be NULL). */
int fr_worker_host_pending(const fr_worker *w, int *queued, int *in_flight, int *blocks_in_flight);
/* Host-buffer STREAMING form: the same host-resident request stream handed to fr_worker_push_host (pinned staging blocks, one
 * H2D + one fused launch per block, scores written to pinned memory by the kernel, no per-batch synchronisation).  Only for models that stream through the fused
 * item-tile kernel.  Scores land in per-worker host rings (fr_driver_host_score_ring, same indexing as fr_driver_score_ring). */
int fr_driver_run_host_streaming(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *h_idx_pool,
                                 const float *const *h_dense_pool, int n_pool, double *elapsed_s);
const float *fr_driver_host_score_ring(fr_driver *d, int thread, int slot, int *ring_len);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
