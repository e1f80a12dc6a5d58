// fleetrec_sender -- synthetic request stream; counterpart of the reference's FPGA/CPU simulators
// (GPU/final_network_cublasLt_1_node_no_FIFO_scatter/multiple_connections_network_client_sender.c): THREAD_NUM threads,
// each connect()s to SERVER:PORT+i and sends fixed-size blocks back to back.  The reference oversends 2x because sender
// and receiver threads progress unevenly (README.md:23-25); here every thread simply sends until the server closes
// the connection.  A block is B x T int32 indices (+ B x 64 dense floats for Model-C).
//
// Usage: fleetrec_sender --model A|B|C [--batch 256] [--threads 4] [--port 8080] [--host 127.0.0.1]
//                        [--indices reference|uniform] [--per-item | --per-bank] [--row-cap N] [--max-blocks N] [--reply]
//                        [--window W] (with --reply: up to W requests in flight per connection -- a reader thread takes the score replies and
//                                      times request -> reply; W = 1 waits for every reply before the next request, the default)
//                        [--pool N]   (uniform indices: N distinct blocks per connection are generated up front and sent in rotation --
//                                      drawing 12 k random indices per block is slower than the server; default 32, 0 = draw every block)
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "fleetrec.h"

// load_access_idx's 32 fixed indices (embedding_47_krnl.cpp:899-914; identical in the 98/377 kernels)
static const int kIdxRandom[32] = {3, 99, 38, 72, 29, 57, 1, 72, 36, 76, 35, 50, 37, 57, 13, 66,
                                   26, 70, 41, 93, 48, 82, 44, 78, 25, 52, 3, 92, 36, 56, 46, 88};

int main(int argc, char **argv) {
    int which = FR_MODEL_A, batch = 256, threads = 4, port = 8080;
    long row_cap = 0, max_blocks = 1L << 40, interval_us = 0, pool = 32, window = 1;
    std::string host = "127.0.0.1", indices = "reference";
    bool per_item = false, per_bank = false, reply = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--model") { std::string v = next(); which = v == "A" ? FR_MODEL_A : v == "B" ? FR_MODEL_B : FR_MODEL_C; }
        else if (a == "--batch") batch = atoi(next());
        else if (a == "--threads") threads = atoi(next());
        else if (a == "--port") port = atoi(next());
        else if (a == "--host") host = next();
        else if (a == "--indices") indices = next();
        else if (a == "--per-item") per_item = true;
        else if (a == "--per-bank") per_bank = true;
        else if (a == "--reply") reply = true;
        else if (a == "--row-cap") row_cap = atol(next());
        else if (a == "--max-blocks") max_blocks = atol(next());
        else if (a == "--interval-us") interval_us = atol(next());
        else if (a == "--pool") pool = atol(next());
        else if (a == "--window") window = atol(next());
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    fr_model_desc *m = nullptr;
    if (fr_model_clone_scaled(fr_model_builtin(which), 1.0, 1, row_cap, &m) != FR_OK) { fprintf(stderr, "%s\n", fr_last_error()); return 1; }
    if (per_item) m->index_mode = FR_INDEX_PER_ITEM;
    if (per_bank) m->index_mode = FR_INDEX_PER_BANK;   // one index per memory bank per item: the reference kernel's own contract
    const size_t cols = (size_t)fr_model_index_cols(m);
    std::vector<int64_t> col_range(cols, 100);          // exclusive upper bound of every index column
    if (per_bank) {
        std::vector<int64_t> rows(m->n_tables);
        fr_model_bank_map(m, nullptr, rows.data());
        for (size_t c = 0; c < cols; c++) col_range[c] = rows[c];
    } else if (!per_item) {
        for (size_t c = 0; c < cols; c++) col_range[c] = m->tables[c].rows;
    }
    std::vector<std::thread> th;
    std::vector<long> sent(threads, 0);
    std::vector<std::vector<double>> lat_us(threads);  // --reply --window W: request sent -> reply received, per request
    for (int t = 0; t < threads; t++) {
        th.emplace_back([&, t]() {
            std::vector<int32_t> idx((size_t)batch * cols);
            std::vector<float> dense((size_t)batch * m->dense_len), scores(batch);
            std::mt19937_64 rng(1234 + t);
            int sock = socket(AF_INET, SOCK_STREAM, 0), one = 1;
            sockaddr_in addr{};
            addr.sin_family = AF_INET;
            addr.sin_port = htons((uint16_t)(port + t));
            inet_pton(AF_INET, host.c_str(), &addr.sin_addr);
            int tries = 0;
            while (connect(sock, (sockaddr *)&addr, sizeof(addr)) < 0) {  // the server may still be filling 63 GB of tables
                if (++tries > 600) { fprintf(stderr, "connect to %s:%d failed\n", host.c_str(), port + t); return; }
                usleep(100000);
            }
            setsockopt(sock, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
            auto draw = [&](int32_t *ix, float *dn) {
                for (int b = 0; b < batch; b++) {
                    for (size_t c = 0; c < cols; c++) {
                        int32_t v;
                        if (indices == "reference") v = kIdxRandom[b % 32];  // same index for every table of the item (F4)
                        else v = (int32_t)(rng() % (uint64_t)col_range[c]);
                        ix[(size_t)b * cols + c] = v;
                    }
                    for (int d = 0; d < m->dense_len; d++)
                        dn[(size_t)b * m->dense_len + d] = (indices == "reference") ? ((kIdxRandom[b % 32] % 2 == 0) ? 1.0f : 0.0f)
                                                                                     : (float)((rng() >> 11) * (2.0 / 9007199254740992.0) - 1.0);
                }
            };
            // the reference pattern is one block; uniform indices: `pool` blocks drawn up front and sent in rotation
            const long n_pool = indices == "reference" ? 1 : pool;
            std::vector<int32_t> pidx((size_t)(n_pool > 0 ? n_pool : 0) * idx.size());
            std::vector<float> pdense((size_t)(n_pool > 0 ? n_pool : 0) * dense.size());
            for (long q = 0; q < n_pool; q++) draw(pidx.data() + (size_t)q * idx.size(), pdense.data() + (size_t)q * dense.size());
            const bool async_reply = reply && window > 1;
            std::atomic<long> n_sent{0}, n_recv{0};
            std::atomic<bool> reader_done{false};
            std::vector<std::chrono::steady_clock::time_point> t_send((size_t)(window > 1 ? window : 1));
            std::thread reader;
            if (async_reply)
                reader = std::thread([&]() {  // score replies arrive in request order
                    std::vector<float> sc(batch);
                    for (;;) {
                        size_t got = 0;
                        while (got < sc.size() * 4) {
                            ssize_t r = read(sock, (char *)sc.data() + got, sc.size() * 4 - got);
                            if (r <= 0) break;
                            got += (size_t)r;
                        }
                        if (got < sc.size() * 4) break;
                        const long k = n_recv.load(std::memory_order_relaxed);
                        lat_us[t].push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_send[(size_t)(k % window)]).count());
                        n_recv.store(k + 1, std::memory_order_release);
                    }
                    reader_done.store(true);
                });
            for (long blk = 0; blk < max_blocks; blk++) {
                const int32_t *bi = idx.data();
                const float *bd = dense.data();
                if (n_pool > 0) {
                    bi = pidx.data() + (size_t)(blk % n_pool) * idx.size();
                    bd = pdense.data() + (size_t)(blk % n_pool) * dense.size();
                } else {
                    draw(idx.data(), dense.data());
                }
                if (async_reply) {  // at most `window` requests without a reply
                    while (n_sent.load(std::memory_order_relaxed) - n_recv.load(std::memory_order_acquire) >= window && !reader_done.load()) usleep(5);
                    if (reader_done.load()) break;
                    t_send[(size_t)(n_sent.load(std::memory_order_relaxed) % window)] = std::chrono::steady_clock::now();
                }
                const auto t_req = std::chrono::steady_clock::now();
                if (send(sock, bi, idx.size() * 4, MSG_NOSIGNAL) <= 0) break;
                if (!dense.empty() && send(sock, bd, dense.size() * 4, MSG_NOSIGNAL) <= 0) break;
                sent[t]++;
                n_sent.fetch_add(1, std::memory_order_release);
                if (interval_us > 0 && !(reply && !async_reply)) usleep((useconds_t)interval_us);  // rate limit of the latency experiment (reference sender: usleep(useconds))
                if (reply && !async_reply) {
                    size_t got = 0;
                    while (got < scores.size() * 4) {
                        ssize_t r = read(sock, (char *)scores.data() + got, scores.size() * 4 - got);
                        if (r <= 0) break;
                        got += (size_t)r;
                    }
                    if (got < scores.size() * 4) break;
                    lat_us[t].push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_req).count());
                    if (interval_us > 0) usleep((useconds_t)interval_us);  // (the wait for the reply comes first: it is what is timed)
                }
            }
            if (async_reply) {
                shutdown(sock, SHUT_WR);  // no more requests; the reader runs until the server has answered what it took and closes
                reader.join();
            }
            close(sock);
        });
    }
    for (auto &x : th) x.join();
    long tot = 0;
    for (long s : sent) tot += s;
    printf("sender: %ld blocks sent over %d connections\n", tot, threads);
    std::vector<double> all;
    for (auto &v : lat_us) all.insert(all.end(), v.begin() + (long)(v.size() / 20), v.end());  // the first 5 % of every connection are warm-up
    if (!all.empty()) {
        std::sort(all.begin(), all.end());
        double sum = 0;
        for (double v : all) sum += v;
        auto pct = [&](double q) { return all[(size_t)(q * (all.size() - 1))]; };
        printf("latency request sent -> scores received  n=%zu avg %.1f us  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f (window %ld per connection)\n", all.size(),
               sum / all.size(), pct(0.50), pct(0.90), pct(0.99), all.back(), window);
    }
    fr_model_free(m);
    return 0;
}
