// fleetrec_server -- the reference GPU server's request/batch shell on top of libfleetrec.so.
//
// Counterpart of GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c (and the 3-node variant):
//   main()            : device/model set-up once, THREAD_NUM threads, join, latency statistics   (cuda_server.c:505-595)
//   thread_consume()  : listen/accept on PORT+i (:359-399), then the hot loop (:406-497):
//                       { lock; id = global_batch_count++; unlock } -> blocking read() of ONE fixed-size block into
//                       the pinned request buffer -> enqueue -> ... ; first five outputs printed at the end (:499-502)
// What changed on the wire: the lookup moved behind the boundary, so a block is B x T int32 indices
// (+ B x 64 dense floats for Model-C) instead of B x K gathered floats; framing is the reference's
// (no header, fixed-size little-endian blocks, constant.h:37-38).  Optionally the B scores are written back
// (--reply), which the reference never did (its "Finish receiving." message is defined but never sent, :363).
//
// Usage: fleetrec_server --model A|B|C [--batch 256] [--threads 4] [--port 8080] [--total 1024] [--device 0]
//        [--stream [--reply [--flush-us 50] [--flush-min 32]]]: streaming with score replies and adaptive batching -- see thread_consume
//                        [--tables evenodd|hash] [--weights ones|uniform] [--per-item | --per-bank] [--reply] [--row-cap N]
//                        [--shards G [--one-device] [--precision f32|bf16|fp8]]
// --shards G: BASELINE configs[3]/[4] -- the tables are sharded by table-ID over GPUs device .. device + G - 1 of this node, or with
// --device -1 over G CPU shard contexts of this process exchanging through the library's in-process host exchange (one
// context and one worker per shard, fr_comm_init_all); every batch goes through fr_worker_submit_sharded on all shards (slices
// all-gathered over RCCL, FC on batch / G items per GPU, scores all-gathered).  The counterpart of the 3-node server, whose batch
// arrives in three parts from three senders (3-node cuda_server.c:513-591).
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fleetrec.h"
#include "fleetrec_serving.h"

struct Options {
    int model = FR_MODEL_A, batch = 256, threads = 4, port = 8080, device = 0;
    long total = 1024;
    int tables = FR_FILL_EVEN_ODD, weights = FR_WEIGHTS_ONES;
    bool per_item = false, per_bank = false, reply = false;
    int shards = 0;            // > 0: table-sharded over `shards` GPUs
    bool one_device = false;   // --shards G --one-device: all G shard contexts on GPU `device` (the library's staged host exchange: a G-rank job rehearsed on one GPU)
    int precision = FR_FC_FP32;
    bool stream = false;   // throughput mode: fr_worker_push_host (blocks of batches per launch) instead of submit + sync per batch
    long flush_us = 50;    // --stream --reply: how long the socket must stay dry before a partial block is launched
    int flush_min = 32;    // ... while earlier blocks are still in flight: only once this many requests are queued (with nothing in flight: any number)
    int small_block = 8;   // --stream --reply: blocks of at most this many batches ride the stage pipeline, n + 4 launches (fr_ctx_set_small_block)
    bool latency = false;  // latency-measurement mode: per-batch recv -> enqueued -> scores times (measure_network_cuda_cp_latency_*/cuda_server.c)
    long row_cap = 0;
};

static bool read_exact(int fd, void *buf, size_t n) {  // the recv loop of cuda_server.c:425-450
    char *p = (char *)buf;
    while (n) {
        ssize_t r = read(fd, p, n);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}

static bool write_exact(int fd, const void *buf, size_t n) {
    const char *p = (const char *)buf;
    while (n) {
        ssize_t r = send(fd, p, n, MSG_NOSIGNAL);
        if (r <= 0) return false;
        p += r;
        n -= (size_t)r;
    }
    return true;
}

static std::mutex g_mtx;            // pthread_mutex_t mtx            (cuda_server.c:25)
static long g_global_batch_count = 0;  // int global_batch_count      (cuda_server.c:23)
static std::atomic<long long> g_first_connection_ns{0};  // steady-clock time of the first accepted connection (throughput without the wait for the sender)

// The sharded engine: G persistent threads, one per shard (RCCL's collectives of one communicator must come from one thread per
// rank).  A connection thread that holds a complete batch takes the engine, every shard thread copies the request into its worker's
// pinned buffers and runs fr_worker_submit_sharded + fr_worker_sync, shard 0's pinned score buffer then holds all scores.
struct ShardedEngine {
    int G = 0, batch = 0;
    std::vector<fr_ctx *> ctxs;
    std::vector<fr_comm *> comms;
    std::vector<fr_worker *> workers;
    std::vector<std::thread> threads;
    std::mutex mtx, user;  // mtx guards the hand-over state; user serialises the connection threads
    std::condition_variable cv;
    long generation = 0;
    int pending = 0;
    bool stop = false;
    const int32_t *req_idx = nullptr;
    const float *req_dense = nullptr;
    size_t idx_bytes = 0, dense_bytes = 0;
    int status = 0;
    std::string error;

    void shard_loop(int r) {
        long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mtx);
            cv.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            lk.unlock();
            memcpy(fr_worker_idx_ptr(workers[r]), req_idx, idx_bytes);
            if (dense_bytes) memcpy(fr_worker_dense_ptr(workers[r]), req_dense, dense_bytes);
            int rc = fr_worker_submit_sharded(workers[r], comms[r], batch);
            if (rc == FR_OK) rc = fr_worker_sync(workers[r]);
            lk.lock();
            if (rc != FR_OK && status == 0) {
                status = rc;
                error = fr_last_error();
            }
            if (--pending == 0) cv.notify_all();
        }
    }
    // -> 0 and scores in fr_worker_score_ptr(workers[0]); called under `user`
    int run(const int32_t *idx, const float *dense) {
        std::unique_lock<std::mutex> lk(mtx);
        req_idx = idx;
        req_dense = dense;
        pending = G;
        generation++;
        cv.notify_all();
        cv.wait(lk, [&] { return pending == 0; });
        return status;
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(mtx);
            stop = true;
        }
        cv.notify_all();
        for (auto &t : threads) t.join();
        for (auto *w : workers) fr_worker_destroy(w);
        for (auto *c : comms) fr_comm_destroy(c);
        for (auto *c : ctxs) fr_ctx_destroy(c);
    }
};
static ShardedEngine *g_engine = nullptr;

struct ThreadInfo {  // struct CUDA_thread_info (cuda_server.c:91-98)
    int port;
    fr_ctx *ctx;
    std::vector<double> recv_to_submit_us;
    std::vector<double> recv_to_scores_us;  // latency mode: batch fully received -> its scores are in host memory
    std::vector<float> first_scores;
    long batches = 0;
    int status = 0;
    std::string error;
};

static void thread_consume(ThreadInfo *t, const Options &o) {
    const fr_model_desc *m = fr_ctx_model(t->ctx);
    fr_worker *wk = nullptr;
    if (fr_worker_create(t->ctx, o.batch, &wk) != FR_OK) {
        t->status = -1;
        t->error = fr_last_error();
        return;
    }
    const size_t idx_cols = (size_t)fr_model_index_cols(m);
    const size_t idx_bytes = (size_t)o.batch * idx_cols * sizeof(int32_t);
    const size_t dense_bytes = (size_t)o.batch * m->dense_len * sizeof(float);
    int server_fd = socket(AF_INET, SOCK_STREAM, 0), opt = 1;
    setsockopt(server_fd, SOL_SOCKET, SO_REUSEADDR, &opt, sizeof(opt));
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = INADDR_ANY;
    addr.sin_port = htons((uint16_t)t->port);
    int sock = -1;
    if (server_fd < 0 || bind(server_fd, (sockaddr *)&addr, sizeof(addr)) < 0 || listen(server_fd, 3) < 0) {
        t->status = -2;
        t->error = std::string("socket/bind/listen on port ") + std::to_string(t->port) + ": " + strerror(errno);
    } else {
        socklen_t len = sizeof(addr);
        sock = accept(server_fd, (sockaddr *)&addr, &len);
        if (sock < 0) {
            t->status = -3;
            t->error = std::string("accept: ") + strerror(errno);
        }
    }
    if (t->status == 0) {
        setsockopt(sock, IPPROTO_TCP, TCP_NODELAY, &opt, sizeof(opt));
        printf("Successfully built connection on port %d.\n", t->port);
        {
            long long expect = 0;
            g_first_connection_ns.compare_exchange_strong(expect, std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count());
        }
        // --stream: every batch is read into the worker's staging (fr_worker_stage_acquire / fr_worker_push_staged); scores come back
        // in blocks, the last batch's are available after the final sync
        constexpr long long kStreamRing = 512;  // score slots per connection: at most 4 blocks x 64 batches are undelivered at any time
        std::vector<float> stream_scores(o.stream ? (size_t)kStreamRing * o.batch : 0);
        std::vector<int32_t> sh_idx(g_engine ? idx_bytes / sizeof(int32_t) : 0);
        std::vector<float> sh_dense(g_engine ? dense_bytes / sizeof(float) : 0), sh_scores(g_engine ? (size_t)o.batch : 0);
        while (g_engine) {   // table-sharded mode: the whole batch is received here, then every shard works on it
            {
                std::lock_guard<std::mutex> g(g_mtx);
                if (g_global_batch_count >= o.total) break;
                g_global_batch_count++;
            }
            if (!read_exact(sock, sh_idx.data(), idx_bytes) || (dense_bytes && !read_exact(sock, sh_dense.data(), dense_bytes))) {
                t->status = -4;
                t->error = "Receiving data UNSUCCESSFUL (peer closed before the batch was complete)";
                break;
            }
            const auto t_recv = std::chrono::steady_clock::now();
            {
                std::lock_guard<std::mutex> u(g_engine->user);
                if (g_engine->run(sh_idx.data(), dense_bytes ? sh_dense.data() : nullptr) != 0) {
                    t->status = -5;
                    t->error = g_engine->error;
                    break;
                }
                memcpy(sh_scores.data(), fr_worker_score_ptr(g_engine->workers[0]), (size_t)o.batch * sizeof(float));
            }
            t->recv_to_scores_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_recv).count());
            if (o.reply && !write_exact(sock, sh_scores.data(), (size_t)o.batch * sizeof(float))) {
                t->status = -7;
                t->error = "sending scores failed";
                break;
            }
            t->batches++;
        }
        if (g_engine)
            for (int j = 0; j < 5 && j < o.batch && t->batches > 0; j++) t->first_scores.push_back(sh_scores[j]);
        // --stream --reply: scores go back over the socket, block by block, in request order.  fr_worker_host_poll says how many of the
        // pushed batches have their scores; when the socket has nothing to read and requests are still queued, fr_worker_flush launches the
        // partial block instead of waiting for it to fill (light load: latency of one launch; heavy load: full blocks).
        long long replied = 0;
        auto send_ready = [&]() -> bool {
            long long delivered = 0;
            if (fr_worker_host_poll(wk, &delivered) != FR_OK) {
                t->status = -5;
                t->error = fr_last_error();
                return false;
            }
            for (; replied < delivered; replied++)
                if (!write_exact(sock, stream_scores.data() + (size_t)(replied % kStreamRing) * o.batch, (size_t)o.batch * sizeof(float))) {
                    t->status = -7;
                    t->error = "sending scores failed";
                    return false;
                }
            return true;
        };
        while (o.stream && !g_engine) {
            {   // (a full counter ends the loop before anything is read; the slot itself is taken below, once a request is known to be coming)
                std::lock_guard<std::mutex> g(g_mtx);
                if (g_global_batch_count >= o.total) break;
            }
            if (o.reply) {  // wait for the next request without letting queued ones sit
                // Requests of a burst arrive microseconds apart: the socket must stay dry for --flush-us (default 50 us) before the partial
                // block is launched, or every request of the burst would become its own launch.
                bool flushed = false;
                const auto t_dry = std::chrono::steady_clock::now();
                for (;;) {
                    pollfd pfd{sock, POLLIN, 0};
                    const bool waiting = t->batches > replied;
                    const int pr = poll(&pfd, 1, waiting ? 0 : -1);
                    if (pr != 0) break;  // data, EOF or an error: the read below sorts it out
                    if (!flushed) {
                        int queued0 = 0, blocks0 = 0;
                        fr_worker_host_pending(wk, &queued0, nullptr, &blocks0);
                        // an idle worker with only a few requests queued waits a fifth of --flush-us (they take the stage launches, there is
                        // little to gain from collecting more); otherwise the socket must stay dry for --flush-us first
                        const bool few_and_idle = blocks0 == 0 && queued0 > 0 && queued0 <= o.small_block;
                        const double grace = few_and_idle ? (double)o.flush_us / 5.0 : (double)o.flush_us;  // 10 us is enough to see the rest of a burst coming
                        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_dry).count() < grace) continue;  // spin
                        // adaptive batching: a launch costs the worker's stream >= 150 us whatever it carries.  With nothing in flight the
                        // queued requests leave at once (latency); with blocks in flight they leave once they are worth a launch of their own
                        // (--flush-min, default half a block) -- fewer would only queue behind the running ones, so they keep collecting
                        int queued = 0, blocks = 0;
                        fr_worker_host_pending(wk, &queued, nullptr, &blocks);
                        if (blocks == 0 || queued >= o.flush_min || queued == 0) {
                            if (fr_worker_flush(wk) != FR_OK) {
                                t->status = -5;
                                t->error = fr_last_error();
                                break;
                            }
                            flushed = true;
                        }
                    }
                    if (!send_ready()) break;
                    if (t->batches > replied) {  // scores still on their way and nothing to read: spin for the first 200 us (a small block is
                        pollfd p2{sock, POLLIN, 0};  // back in < 100 us and a sleep's granularity is ~60 us), then yield
                        if (poll(&p2, 1, 0) == 0 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_dry).count() > 200.0) usleep(20);
                    }
                }
                if (t->status) break;
            }
            // End of stream?  A sender that is done half-closes its side (fleetrec_sender --window W: shutdown(SHUT_WR)) and still expects
            // the replies of everything it sent.  EOF with ZERO bytes of the next request is therefore a clean end: no counter slot is
            // taken, no staging slot is acquired, the requests already accepted are synchronised and answered below, and the thread ends
            // with status 0.  (-4 stays for a request that is cut off part-way.)  ADVICE r02: the slot used to be taken before the read, so a
            // fast connection that hit EOF dropped up to W pending replies and left a slower connection one request short.
            {
                char probe;
                ssize_t pk;
                do pk = recv(sock, &probe, 1, MSG_PEEK); while (pk < 0 && errno == EINTR);
                if (pk == 0) break;   // orderly shutdown, nothing pending on the wire
                if (pk < 0) {
                    t->status = -4;
                    t->error = "Receiving data UNSUCCESSFUL (socket error)";
                    break;
                }
            }
            {
                std::lock_guard<std::mutex> g(g_mtx);
                if (g_global_batch_count >= o.total) break;   // another connection took the last slot meanwhile: its request stays unread
                g_global_batch_count++;
            }
            // the socket is read straight into the worker's pinned staging slot (the reference reads into its pinned input_feature,
            // cuda_server.c:437), then the slot is queued: no copy between the socket buffer and the H2D source
            int32_t *slot_idx = nullptr;
            float *slot_dense = nullptr;
            if (fr_worker_stage_acquire(wk, o.batch, &slot_idx, &slot_dense) != FR_OK) {
                t->status = -5;
                t->error = fr_last_error();
                break;
            }
            if (!read_exact(sock, slot_idx, idx_bytes) || (dense_bytes && !read_exact(sock, slot_dense, dense_bytes))) {
                t->status = -4;
                t->error = "Receiving data UNSUCCESSFUL (peer closed before the batch was complete)";
                break;
            }
            if (fr_worker_push_staged(wk, o.batch, stream_scores.data() + (size_t)(t->batches % kStreamRing) * o.batch) != FR_OK) {
                t->status = -5;
                t->error = fr_last_error();
                break;
            }
            t->batches++;
            if (o.reply && (t->batches & 15) == 0 && !send_ready()) break;  // (the dry-socket path above polls too: nothing waits on this one)
        }
        if (o.stream && !g_engine) {
            if (t->status == 0 && fr_worker_sync(wk) != FR_OK) {
                t->status = -6;
                t->error = fr_last_error();
            }
            if (t->status == 0 && o.reply) send_ready();  // everything is delivered after the sync: the rest of the replies
            if (t->batches > 0)
                for (int j = 0; j < 5 && j < o.batch; j++) t->first_scores.push_back(stream_scores[(size_t)((t->batches - 1) % kStreamRing) * o.batch + j]);
        }
        while (!o.stream && !g_engine) {
            {
                std::lock_guard<std::mutex> g(g_mtx);
                if (g_global_batch_count >= o.total) break;
                g_global_batch_count++;
            }
            if (!read_exact(sock, fr_worker_idx_ptr(wk), idx_bytes) ||
                (dense_bytes && !read_exact(sock, fr_worker_dense_ptr(wk), dense_bytes))) {
                t->status = -4;
                t->error = "Receiving data UNSUCCESSFUL (peer closed before the batch was complete)";
                break;
            }
            const auto t_recv = std::chrono::steady_clock::now();  // network_time / cuda_time pair of cuda_server.c:429,462
            if (fr_worker_submit(wk, o.batch) != FR_OK) {
                t->status = -5;
                t->error = fr_last_error();
                break;
            }
            t->recv_to_submit_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_recv).count());
            if (fr_worker_sync(wk) != FR_OK) {
                t->status = -6;
                t->error = fr_last_error();
                break;
            }
            if (o.latency) t->recv_to_scores_us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_recv).count());
            if (o.reply && !write_exact(sock, fr_worker_score_ptr(wk), (size_t)o.batch * sizeof(float))) {
                t->status = -7;
                t->error = "sending scores failed";
                break;
            }
            t->batches++;
        }
        if (!o.stream && !g_engine) {
            const float *sc = fr_worker_score_ptr(wk);
            for (int j = 0; j < 5 && j < o.batch; j++) t->first_scores.push_back(sc[j]);  // cuda_server.c:499-502
        }
    }
    if (sock >= 0) close(sock);
    if (server_fd >= 0) close(server_fd);
    fr_worker_destroy(wk);
}

int main(int argc, char **argv) {
    Options o;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--model") { std::string v = next(); o.model = v == "A" ? FR_MODEL_A : v == "B" ? FR_MODEL_B : FR_MODEL_C; }
        else if (a == "--batch") o.batch = atoi(next());
        else if (a == "--threads") o.threads = atoi(next());
        else if (a == "--port") o.port = atoi(next());
        else if (a == "--total") o.total = atol(next());
        else if (a == "--device") o.device = atoi(next());
        else if (a == "--tables") o.tables = std::string(next()) == "hash" ? FR_FILL_HASH : FR_FILL_EVEN_ODD;
        else if (a == "--weights") o.weights = std::string(next()) == "uniform" ? FR_WEIGHTS_UNIFORM : FR_WEIGHTS_ONES;
        else if (a == "--per-item") o.per_item = true;
        else if (a == "--per-bank") o.per_bank = true;
        else if (a == "--shards") o.shards = atoi(next());
        else if (a == "--one-device") o.one_device = true;
        else if (a == "--precision") { std::string v = next(); o.precision = v == "bf16" ? FR_FC_BF16 : v == "fp8" ? FR_FC_FP8 : FR_FC_FP32; }
        else if (a == "--reply") o.reply = true;
        else if (a == "--latency") o.latency = true;
        else if (a == "--stream") o.stream = true;
        else if (a == "--flush-us") o.flush_us = atol(next());
        else if (a == "--flush-min") o.flush_min = atoi(next());
        else if (a == "--small-block") o.small_block = atoi(next());
        else if (a == "--row-cap") o.row_cap = atol(next());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    printf("HIP devices visible: %d\n", fr_device_count());  // device probe of cuda_server.c:508-522
    fr_model_desc *model = nullptr;
    if (fr_model_clone_scaled(fr_model_builtin(o.model), 1.0, 1, o.row_cap, &model) != FR_OK) { fprintf(stderr, "%s\n", fr_last_error()); return 1; }
    if (o.per_item) model->index_mode = FR_INDEX_PER_ITEM;
    if (o.per_bank) model->index_mode = FR_INDEX_PER_BANK;
    fr_ctx *ctx = nullptr;
    ShardedEngine engine;
    if (o.shards > 0) {  // one context + worker + communicator handle per shard, devices device .. device + G - 1
        engine.G = o.shards;
        engine.batch = o.batch;
        engine.ctxs.assign(o.shards, nullptr);
        engine.comms.assign(o.shards, nullptr);
        engine.workers.assign(o.shards, nullptr);
        for (int r = 0; r < o.shards; r++) {
            // GPUs device .. device + G - 1, or (--device -1) G CPU shard contexts exchanging in process (fr_comm_init_all picks the transport)
            if (fr_ctx_create_sharded(model, o.device < 0 ? -1 : (o.one_device ? o.device : o.device + r), r, o.shards, &engine.ctxs[r]) != FR_OK || fr_ctx_fill_tables(engine.ctxs[r], o.tables, 0xF1EE7) != FR_OK ||
                fr_ctx_fill_weights(engine.ctxs[r], o.weights, 99) != FR_OK || fr_ctx_set_fc_precision(engine.ctxs[r], o.precision) != FR_OK ||
                fr_worker_create(engine.ctxs[r], o.batch, &engine.workers[r]) != FR_OK) {
                fprintf(stderr, "shard %d set-up failed: %s\n", r, fr_last_error());
                return 1;
            }
        }
        if (fr_comm_init_all(engine.ctxs.data(), o.shards, engine.comms.data()) != FR_OK) {
            fprintf(stderr, "exchange set-up failed: %s\n", fr_last_error());
            return 1;
        }
        engine.idx_bytes = (size_t)o.batch * (size_t)fr_model_index_cols(model) * sizeof(int32_t);
        engine.dense_bytes = (size_t)o.batch * model->dense_len * sizeof(float);
        for (int r = 0; r < o.shards; r++) engine.threads.emplace_back(&ShardedEngine::shard_loop, &engine, r);
        g_engine = &engine;
        ctx = engine.ctxs[0];
        if (o.device < 0) printf("table-sharded over %d CPU shard contexts (in-process host exchange of the looked-up slices)\n", o.shards);
        else if (o.one_device) printf("table-sharded over %d shard contexts on GPU %d (staged host exchange of the looked-up slices)\n", o.shards, o.device);
        else printf("table-sharded over %d GPUs (RCCL all-gather of the looked-up slices)\n", o.shards);
    } else if (fr_ctx_create(model, o.device, &ctx) != FR_OK || fr_ctx_fill_tables(ctx, o.tables, 0xF1EE7) != FR_OK ||
               fr_ctx_fill_weights(ctx, o.weights, 99) != FR_OK || fr_ctx_set_fc_precision(ctx, o.precision) != FR_OK) {
        fprintf(stderr, "set-up failed: %s\n", fr_last_error());
        return 1;
    }
    if (o.stream && o.reply && !g_engine && fr_ctx_set_small_block(ctx, o.small_block) != FR_OK) {
        fprintf(stderr, "set-up failed: %s\n", fr_last_error());
        return 1;
    }
    printf("model %s: %d tables, %.3f GB, record %d floats; batch %d, %d threads, ports %d..%d, %ld batches\n", model->name, model->n_tables,
           fr_model_table_bytes(model) / 1e9, model->record_len, o.batch, o.threads, o.port, o.port + o.threads - 1, o.total);
    fflush(stdout);
    std::vector<ThreadInfo> info(o.threads);
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < o.threads; i++) {
        info[i].port = o.port + i;  // PORT + i (cuda_server.c:541)
        info[i].ctx = ctx;
        th.emplace_back(thread_consume, &info[i], std::cref(o));
    }
    for (auto &t : th) t.join();
    const auto t_end = std::chrono::steady_clock::now();
    const double secs = std::chrono::duration<double>(t_end - t0).count();
    int rc = 0;
    long done = 0;
    double lat = 0;
    size_t nlat = 0;
    for (int i = 0; i < o.threads; i++) {
        if (info[i].status) {
            fprintf(stderr, "thread %d: %s\n", i, info[i].error.c_str());
            rc = 1;
        }
        done += info[i].batches;
        for (double v : info[i].recv_to_submit_us) lat += v, nlat++;
        if (info[i].batches > 0) {  // a thread whose connection came up after the last batch was taken has nothing to show
            printf("thread %d scores:", i);
            for (float v : info[i].first_scores) printf(" %f", v);
            printf("\n");
        } else {
            printf("thread %d took no batch\n", i);
        }
    }
    printf("processed %ld batches (%ld inferences) in %.3f s incl. connection set-up\n", done, done * o.batch, secs);
    if (const long long f = g_first_connection_ns.load()) {  // throughput without the wait for the sender to connect
        const double run_s = (std::chrono::duration_cast<std::chrono::nanoseconds>(t_end.time_since_epoch()).count() - f) * 1e-9;
        if (run_s > 0) printf("first connection -> last scores: %.3f s = %.2f M inferences/s over TCP\n", run_s, done * o.batch / run_s / 1e6);
    }
    if (nlat) printf("Average time from batch received to enqueued: %.3f us\n", lat / nlat);  // the reference's memcpy-time statistic (:565-591)
    if (o.latency) {
        // The reference's latency experiment (measure_network_cuda_cp_latency_single_node/cuda_server.c:1-15,227,548,728-737): the
        // sender is rate-limited so the server is never the bottleneck, and every batch's "received" and "on the device"
        // timestamps are differenced; here the second stamp also exists for "scores back in host memory".
        auto report = [&](const char *what, std::vector<double> v) {
            if (v.empty()) return;
            std::sort(v.begin(), v.end());
            double sum = 0;
            for (double x : v) sum += x;
            auto pct = [&](double p) { return v[(size_t)std::min<double>(v.size() - 1, p * v.size())]; };
            printf("latency %-34s n=%zu avg %.1f us  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f\n", what, v.size(), sum / v.size(), pct(0.50), pct(0.90), pct(0.99),
                   v.back());
        };
        std::vector<double> a, b;
        for (int i = 0; i < o.threads; i++) {
            const size_t skip = std::min<size_t>(info[i].recv_to_submit_us.size(), 8);  // first batches include kernel/code-object warm-up
            a.insert(a.end(), info[i].recv_to_submit_us.begin() + skip, info[i].recv_to_submit_us.end());
            const size_t skip2 = std::min<size_t>(info[i].recv_to_scores_us.size(), 8);
            b.insert(b.end(), info[i].recv_to_scores_us.begin() + skip2, info[i].recv_to_scores_us.end());
        }
        for (size_t i = 0; i < info[0].recv_to_scores_us.size() && i < 10; i++)  // the reference prints every batch (:732); first ten here
            printf("i = %zu recv->enqueued = %.0f ns, recv->scores = %.0f ns\n", i, info[0].recv_to_submit_us[i] * 1e3, info[0].recv_to_scores_us[i] * 1e3);
        report("batch received -> enqueued", a);
        report("batch received -> scores on host", b);
    }
    if (g_engine) engine.shutdown();
    else fr_ctx_destroy(ctx);
    fr_model_free(model);
    return rc;
}
