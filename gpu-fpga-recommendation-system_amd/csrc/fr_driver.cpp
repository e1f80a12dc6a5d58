// Request-driver core: the batch loop of the reference server without its sockets.
//
// Reference shape (GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c):
//   main() spawns THREAD_NUM pthreads running thread_consume() (:554-556); each owns a stream and its
//   buffers (:136-187) and loops { lock; id = global_batch_count++; unlock (:408-417); receive the
//   batch (:425-450); H2D; 4 GEMMs; D2H (:460-495) } until TOTAL_BATCH_NUM batches are done.
// Here the "receive" step is replaced by picking a device-resident index buffer from a pool (the
// synthetic request stream of bench.py); the TCP front-end in host/ feeds the same loop through
// the pinned fr_worker buffers instead.  Each thread keeps `depth` workers in flight so that its
// stream(s) never drain while the host prepares the next submit.
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "fr_internal.h"

// the device drained on both sides of a timed run (nothing to drain on a CPU context: its calls compute before they return)
#define FR_DRAIN_DEVICE(ctx_)                          \
    do {                                               \
        if (!(ctx_)->cpu) {                            \
            FR_HIP(hipSetDevice((ctx_)->device));      \
            FR_HIP(hipDeviceSynchronize());            \
        }                                              \
    } while (0)

constexpr int FR_SCORE_RING = 512;  // eight launches of 64 batches (two of 256: the largest group) per worker between two syncs

struct fr_driver {
    fr_ctx *ctx = nullptr;
    int n_threads = 0, depth = 0, max_batch = 0;
    std::vector<fr_worker *> workers;  // [n_threads * depth]
    std::vector<std::vector<float>> host_rings;  // per worker: FR_SCORE_RING x max_batch floats in host memory (run_host_streaming)
    std::vector<float *> score_rings;  // per worker: FR_SCORE_RING x max_batch floats; the loop syncs a worker every
                                       // FR_SCORE_RING pushes, so no two batches in flight ever share a score buffer
};

extern "C" void fr_driver_destroy(fr_driver *d) {
    if (!d) return;
    for (fr_worker *w : d->workers) fr_worker_destroy(w);
    if (d->ctx && !d->ctx->cpu) (void)hipSetDevice(d->ctx->device);
    for (float *p : d->score_rings) {
        if (p && d->ctx->cpu) free(p);
        else if (p) (void)hipFree(p);
    }
    fr_ctx *held = d->ctx;
    delete d;
    fr_ctx_unref(held);   // the driver's own reference (fr_driver_create): a driver destroyed after its context releases it last
}

extern "C" int fr_driver_create(fr_ctx *ctx, int n_threads, int depth, int max_batch, fr_driver **out) {
    if (!ctx || !out) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_threads < 1 || n_threads > 256 || depth < 1 || depth > 64) FR_FAIL(FR_ERR_INVALID, "n_threads %d / depth %d out of range", n_threads, depth);
    fr_driver *d = new (std::nothrow) fr_driver();
    if (!d) FR_FAIL(FR_ERR_OOM, "out of host memory");
    d->ctx = ctx;
    fr_ctx_ref(ctx);   // (dropped by fr_driver_destroy, also on the failure paths below, which all go through it)
    d->n_threads = n_threads;
    d->depth = depth;
    d->max_batch = max_batch;
    for (int i = 0; i < n_threads * depth; i++) {
        fr_worker *w = nullptr;
        int rc = fr_worker_create(ctx, max_batch, &w);
        if (rc) {
            fr_driver_destroy(d);
            return rc;
        }
        d->workers.push_back(w);
        float *ring = nullptr;
        if (ctx->cpu) {   // the CPU back-end's "device" buffers are host memory
            ring = (float *)calloc((size_t)FR_SCORE_RING * max_batch, sizeof(float));
            if (!ring) {
                fr_driver_destroy(d);
                FR_FAIL(FR_ERR_OOM, "out of host memory (score ring)");
            }
            d->score_rings.push_back(ring);
            continue;
        }
        if (hipMalloc((void **)&ring, (size_t)FR_SCORE_RING * max_batch * sizeof(float)) != hipSuccess) {
            fr_driver_destroy(d);
            FR_FAIL(FR_ERR_OOM, "hipMalloc(score ring) failed");
        }
        (void)hipMemset(ring, 0, (size_t)FR_SCORE_RING * max_batch * sizeof(float));
        d->score_rings.push_back(ring);
    }
    // a driver knows how many chains it will run side by side: an undecided context takes its chain width from it (fr_ctx_set_chain_width)
    {
        int expected = 0;
        const int w = n_threads * depth > 4 ? 4 : n_threads * depth;
        if (ctx->chain_width.compare_exchange_strong(expected, w, std::memory_order_relaxed)) ctx->chain_width_auto.store(false, std::memory_order_relaxed);
    }
    *out = d;
    return FR_OK;
}

extern "C" int fr_driver_run_resident(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *d_idx_pool,
                                      const float *const *d_dense_pool, int n_pool, double *elapsed_s) {
    if (!d || !d_idx_pool || n_pool < 1 || !elapsed_s) FR_FAIL(FR_ERR_INVALID, "bad argument");
    if (batch < 1 || batch > d->max_batch || total_batches < 0) FR_FAIL(FR_ERR_INVALID, "batch %d / total %lld out of range", batch, (long long)total_batches);
    std::mutex mtx;               // pthread_mutex_t mtx (cuda_server.c:25)
    int64_t global_batch_count = 0;  // cuda_server.c:23
    std::vector<int> status(d->n_threads, FR_OK);
    std::vector<std::string> messages(d->n_threads);
    std::vector<std::thread> threads;
    FR_DRAIN_DEVICE(d->ctx);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < d->n_threads; t++) {
        threads.emplace_back([&, t]() {
            fr_worker **wk = &d->workers[(size_t)t * d->depth];
            float **rings = &d->score_rings[(size_t)t * d->depth];
            int64_t local = 0;
            int rc = FR_OK;
#ifdef FR_EXPERIMENTS
            double t_push = 0.0, t_sync = 0.0, t_push_max = 0.0;
#endif
            while (rc == FR_OK) {
                int64_t id;
                {
                    std::lock_guard<std::mutex> g(mtx);
                    if (global_batch_count >= total_batches) break;
                    id = global_batch_count++;
                }
                // enqueue without synchronising, like the reference's loop body (cuda_server.c:460-495)
                const int slot = (int)(local % d->depth);
                float *scores = rings[slot] + (size_t)((local / d->depth) % FR_SCORE_RING) * d->max_batch;
                local++;
                const int p = (int)(id % n_pool);
#ifdef FR_EXPERIMENTS
                const auto tp0 = std::chrono::steady_clock::now();
#endif
                rc = fr_worker_push_device(wk[slot], batch, d_idx_pool[p], d_dense_pool ? d_dense_pool[p] : nullptr, scores);
#ifdef FR_EXPERIMENTS
                const auto tp1 = std::chrono::steady_clock::now();
                t_push += std::chrono::duration<double>(tp1 - tp0).count();
                const double dt = std::chrono::duration<double>(tp1 - tp0).count();
                if (dt > t_push_max) t_push_max = dt;
#endif
                // bound the host's run-ahead (and keep score buffers unique): a worker is synchronised once per trip round its
                // ring, the workers of a thread at staggered points so that one of them always has launches queued
                const int64_t mine = (local - 1) / d->depth + 1;  // pushes this worker has received
                if (rc == FR_OK && (mine + (int64_t)slot * (FR_SCORE_RING / d->depth)) % FR_SCORE_RING == 0) rc = fr_worker_sync(wk[slot]);
#ifdef FR_EXPERIMENTS
                t_sync += std::chrono::duration<double>(std::chrono::steady_clock::now() - tp1).count();
#endif
            }
#ifdef FR_EXPERIMENTS
            if (FR_KNOB_ONCE("DRIVER_TIMING", 0))
                fprintf(stderr, "driver thread %d: %lld pushes, in push %.3f s (max %.1f us), in sync %.3f s\n", t, (long long)local, t_push, 1e6 * t_push_max, t_sync);
#endif
            for (int s = 0; s < d->depth; s++) {
                int r2 = fr_worker_sync(wk[s]);
                if (rc == FR_OK) rc = r2;
            }
            status[t] = rc;
            if (rc) messages[t] = fr_last_error();
        });
    }
    for (auto &th : threads) th.join();
    FR_DRAIN_DEVICE(d->ctx);
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int t = 0; t < d->n_threads; t++)
        if (status[t]) FR_FAIL(status[t], "driver thread %d: %s", t, messages[t].c_str());
    return FR_OK;
}

// Host-buffer form of the loop: every batch's index rows (and dense features) are copied from pageable host memory into
// the worker's pinned buffers (standing in for the socket read(), cuda_server.c:425-450), then submit + sync per batch --
// exactly the reference's per-batch sequence including the PCIe transfers (cuda_server.c:460-495).
extern "C" int fr_driver_run_host(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *h_idx_pool,
                                  const float *const *h_dense_pool, int n_pool, double *elapsed_s) {
    if (!d || !h_idx_pool || n_pool < 1 || !elapsed_s) FR_FAIL(FR_ERR_INVALID, "bad argument");
    if (batch < 1 || batch > d->max_batch || total_batches < 0) FR_FAIL(FR_ERR_INVALID, "batch %d / total %lld out of range", batch, (long long)total_batches);
    const fr_model_desc &m = d->ctx->model;
    const size_t idx_bytes = (size_t)batch * (size_t)fr_model_index_cols(&m) * sizeof(int32_t);
    const size_t dense_bytes = (size_t)batch * m.dense_len * sizeof(float);
    if (dense_bytes && !h_dense_pool) FR_FAIL(FR_ERR_INVALID, "model has dense features but h_dense_pool is NULL");
    std::mutex mtx;
    int64_t global_batch_count = 0;
    std::vector<int> status(d->n_threads, FR_OK);
    std::vector<std::string> messages(d->n_threads);
    std::vector<std::thread> threads;
    FR_DRAIN_DEVICE(d->ctx);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < d->n_threads; t++) {
        threads.emplace_back([&, t]() {
            fr_worker **wk = &d->workers[(size_t)t * d->depth];
            std::vector<char> busy(d->depth, 0);
            int64_t local = 0;
            int rc = FR_OK;
            while (rc == FR_OK) {
                int64_t id;
                {
                    std::lock_guard<std::mutex> g(mtx);
                    if (global_batch_count >= total_batches) break;
                    id = global_batch_count++;
                }
                const int slot = (int)(local++ % d->depth);
                if (busy[slot]) {  // one batch in flight per worker: its pinned buffers are reused
                    rc = fr_worker_sync(wk[slot]);
                    busy[slot] = 0;
                    if (rc) break;
                }
                const int p = (int)(id % n_pool);
                memcpy(fr_worker_idx_ptr(wk[slot]), h_idx_pool[p], idx_bytes);
                if (dense_bytes) memcpy(fr_worker_dense_ptr(wk[slot]), h_dense_pool[p], dense_bytes);
                rc = fr_worker_submit(wk[slot], batch);
                if (rc == FR_OK) busy[slot] = 1;
            }
            for (int s = 0; s < d->depth; s++)
                if (busy[s]) {
                    int r2 = fr_worker_sync(wk[s]);
                    if (rc == FR_OK) rc = r2;
                }
            status[t] = rc;
            if (rc) messages[t] = fr_last_error();
        });
    }
    for (auto &th : threads) th.join();
    FR_DRAIN_DEVICE(d->ctx);
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int t = 0; t < d->n_threads; t++)
        if (status[t]) FR_FAIL(status[t], "driver thread %d: %s", t, messages[t].c_str());
    return FR_OK;
}

// Host-fed STREAMING form: the same host-resident request stream, but batches are handed to fr_worker_push_host -- staged in
// pinned blocks and moved as one H2D copy + one fused launch + one D2H copy per block, no per-batch synchronisation.  Scores land
// in per-worker host rings (fr_driver_host_score_ring).  Models that do not stream through the fused kernel are refused.
extern "C" int fr_driver_run_host_streaming(fr_driver *d, int batch, int64_t total_batches, const int32_t *const *h_idx_pool,
                                            const float *const *h_dense_pool, int n_pool, double *elapsed_s) {
    if (!d || !h_idx_pool || n_pool < 1 || !elapsed_s) FR_FAIL(FR_ERR_INVALID, "bad argument");
    if (batch < 1 || batch > d->max_batch || total_batches < 0) FR_FAIL(FR_ERR_INVALID, "batch %d / total %lld out of range", batch, (long long)total_batches);
    if (d->ctx->model.dense_len && !h_dense_pool) FR_FAIL(FR_ERR_INVALID, "model has dense features but h_dense_pool is NULL");
    FR_NOT_ON_CPU(d->ctx, "fr_driver_run_host_streaming");
    if (d->host_rings.empty()) {
        d->host_rings.resize(d->workers.size());
        for (auto &r : d->host_rings) r.assign((size_t)FR_SCORE_RING * d->max_batch, 0.0f);
    }
    std::mutex mtx;
    int64_t global_batch_count = 0;
    std::vector<int> status(d->n_threads, FR_OK);
    std::vector<std::string> messages(d->n_threads);
    std::vector<std::thread> threads;
    FR_DRAIN_DEVICE(d->ctx);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < d->n_threads; t++) {
        threads.emplace_back([&, t]() {
            fr_worker **wk = &d->workers[(size_t)t * d->depth];
            int64_t local = 0;
            int rc = FR_OK;
            while (rc == FR_OK) {
                int64_t id;
                {
                    std::lock_guard<std::mutex> g(mtx);
                    if (global_batch_count >= total_batches) break;
                    id = global_batch_count++;
                }
                const int slot = (int)(local % d->depth);
                // at most 4 blocks x 64 batches of a worker are in flight, and a block is delivered before its staging is refilled:
                // a ring of FR_SCORE_RING = 512 destinations per worker is never overwritten before delivery
                float *scores = d->host_rings[(size_t)t * d->depth + slot].data() + (size_t)((local / d->depth) % FR_SCORE_RING) * d->max_batch;
                local++;
                const int p = (int)(id % n_pool);
                rc = fr_worker_push_host(wk[slot], batch, h_idx_pool[p], h_dense_pool ? h_dense_pool[p] : nullptr, scores);
            }
            for (int s = 0; s < d->depth; s++) {
                int r2 = fr_worker_sync(wk[s]);
                if (rc == FR_OK) rc = r2;
            }
            status[t] = rc;
            if (rc) messages[t] = fr_last_error();
        });
    }
    for (auto &th : threads) th.join();
    FR_DRAIN_DEVICE(d->ctx);
    *elapsed_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (int t = 0; t < d->n_threads; t++)
        if (status[t]) FR_FAIL(status[t], "driver thread %d: %s", t, messages[t].c_str());
    return FR_OK;
}

extern "C" const float *fr_driver_host_score_ring(fr_driver *d, int thread, int slot, int *ring_len) {
    if (!d || thread < 0 || thread >= d->n_threads || slot < 0 || slot >= d->depth || d->host_rings.empty()) return nullptr;
    if (ring_len) *ring_len = FR_SCORE_RING;
    return d->host_rings[(size_t)thread * d->depth + slot].data();
}

extern "C" const float *fr_driver_score_ring(fr_driver *d, int thread, int slot, int *ring_len) {
    if (!d || thread < 0 || thread >= d->n_threads || slot < 0 || slot >= d->depth) return nullptr;
    if (ring_len) *ring_len = FR_SCORE_RING;
    return d->score_rings[(size_t)thread * d->depth + slot];
}

extern "C" fr_worker *fr_driver_worker(fr_driver *d, int thread, int slot) {
    if (!d || thread < 0 || thread >= d->n_threads || slot < 0 || slot >= d->depth) return nullptr;
    return d->workers[(size_t)thread * d->depth + slot];
}
