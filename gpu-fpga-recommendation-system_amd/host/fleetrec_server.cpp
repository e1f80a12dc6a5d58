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
//                        [--tables evenodd|hash] [--weights ones|uniform] [--per-item] [--reply] [--row-cap N]
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fleetrec.h"

struct Options {
    int model = FR_MODEL_A, batch = 256, threads = 4, port = 8080, device = 0;
    long total = 1024;
    int tables = FR_FILL_EVEN_ODD, weights = FR_WEIGHTS_ONES;
    bool per_item = false, reply = false;
    bool stream = false;   // throughput mode: fr_worker_push_host (blocks of batches per launch) instead of submit + sync per batch
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
    const size_t idx_cols = m->index_mode == FR_INDEX_PER_ITEM ? 1 : (size_t)m->n_tables;
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
        // --stream: the socket is read into ordinary host memory and every batch is handed to fr_worker_push_host; scores come
        // back in blocks, the last batch's are available after the final sync
        std::vector<float> stream_scores(o.stream ? (size_t)256 * o.batch : 0);
        std::vector<int32_t> stream_idx(o.stream ? idx_bytes / sizeof(int32_t) : 0);
        std::vector<float> stream_dense(o.stream ? dense_bytes / sizeof(float) : 0);
        while (o.stream) {
            {
                std::lock_guard<std::mutex> g(g_mtx);
                if (g_global_batch_count >= o.total) break;
                g_global_batch_count++;
            }
            if (!read_exact(sock, stream_idx.data(), idx_bytes) || (dense_bytes && !read_exact(sock, stream_dense.data(), dense_bytes))) {
                t->status = -4;
                t->error = "Receiving data UNSUCCESSFUL (peer closed before the batch was complete)";
                break;
            }
            if (fr_worker_push_host(wk, o.batch, stream_idx.data(), dense_bytes ? stream_dense.data() : nullptr,
                                    stream_scores.data() + (size_t)(t->batches % 256) * o.batch) != FR_OK) {
                t->status = -5;
                t->error = fr_last_error();
                break;
            }
            t->batches++;
        }
        if (o.stream) {
            if (t->status == 0 && fr_worker_sync(wk) != FR_OK) {
                t->status = -6;
                t->error = fr_last_error();
            }
            if (t->batches > 0)
                for (int j = 0; j < 5 && j < o.batch; j++) t->first_scores.push_back(stream_scores[(size_t)((t->batches - 1) % 256) * o.batch + j]);
        }
        while (!o.stream) {
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
        if (!o.stream) {
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
        else if (a == "--reply") o.reply = true;
        else if (a == "--latency") o.latency = true;
        else if (a == "--stream") o.stream = true;
        else if (a == "--row-cap") o.row_cap = atol(next());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    printf("HIP devices visible: %d\n", fr_device_count());  // device probe of cuda_server.c:508-522
    fr_model_desc *model = nullptr;
    if (fr_model_clone_scaled(fr_model_builtin(o.model), 1.0, 1, o.row_cap, &model) != FR_OK) { fprintf(stderr, "%s\n", fr_last_error()); return 1; }
    if (o.per_item) model->index_mode = FR_INDEX_PER_ITEM;
    fr_ctx *ctx = nullptr;
    if (fr_ctx_create(model, o.device, &ctx) != FR_OK || fr_ctx_fill_tables(ctx, o.tables, 0xF1EE7) != FR_OK ||
        fr_ctx_fill_weights(ctx, o.weights, 99) != FR_OK) {
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
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
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
    fr_ctx_destroy(ctx);
    fr_model_free(model);
    return rc;
}
