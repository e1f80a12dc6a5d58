// The CPU back-end behind the same symbols: fr_ctx_create(model, device = -1, ...) ("device /* -1 = CPU backend */";
// BASELINE configs[0]: Model-A batch 1 on the host CPU, no accelerator -- the reference's `make check TARGET=sw_emu` plumbing run,
// FPGA/Makefile:154-158).  Plain C++17 and std::thread; no HIP call, no GPU needed.  It is the product's own code: it walks the SAME
// FrWordDesc list the gfx950 gather kernels walk (fr_api.cpp build_words) and runs the FC chain as a k-ordered fp32 multiply-add chain
// (CUBLAS_COMPUTE_32F, cuda_server.c:211) over the fp32 master weights in the reference's column-major H x K layout (cuda_server.c:215).
// Scope (fleetrec.h): context set-up, fr_worker_submit / submit_device / push_device / sync, gather_only, fc_only, the fp32 forms of the
// table-sharded entry points, the request-driver core.  fp32 only; everything else returns FR_ERR_STATE on a CPU context.
#include <pthread.h>
#include <sys/mman.h>
#include <sched.h>
#include <unistd.h>

#include <cstdio>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "fr_content.h"
#include "fr_internal.h"

// ---- a small persistent thread pool (one parallel region at a time; the caller takes part) -------------------------------------------
namespace {
class Pool {
  public:
    static Pool &get() {
        // Never destroyed: helper threads may outlive static destructors.  A fork()ed child (Python's multiprocessing) inherits the object but
        // none of its threads -- and possibly its mutexes in a locked state: the child's atfork handler drops the pointer, the child's first
        // call builds a pool of its own (the parent's object is leaked there).
        Pool *p = inst_.load(std::memory_order_acquire);
        if (p) return *p;
        Pool *n = new Pool(first_size_.load(std::memory_order_relaxed));
        if (inst_.compare_exchange_strong(p, n, std::memory_order_acq_rel)) return *n;
        n->stop_helpers();   // another thread was first
        delete n;
        return *p;
    }
    // fr_cpu_set_threads(n) BEFORE the first parallel region: the pool is built with n threads straight away (it used to
    // start one helper per usable core first and shrink afterwards)
    static bool exists() { return inst_.load(std::memory_order_acquire) != nullptr; }
    static void set_first_size(int n) { first_size_.store(n, std::memory_order_relaxed); }
    int size() {
        std::lock_guard<std::mutex> g(run_m_);
        return n_;
    }
    int resize(int n) {
        std::lock_guard<std::mutex> g(run_m_);
        if (n <= 0) n = usable_cpus();
        if (n > 1024) n = 1024;
        if (n == n_) return n_;
        stop_helpers();
        n_ = n;
        start_helpers();
        return n_;
    }
    // fn(u) for every u in [0, units); returns when all of them have run
    void run(int units, const std::function<void(int)> &fn) {
        if (units <= 0) return;
        std::lock_guard<std::mutex> region(run_m_);
        if (n_ <= 1 || units == 1) {
            for (int u = 0; u < units; u++) fn(u);
            return;
        }
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = &fn;
            units_ = units;
            next_.store(0, std::memory_order_relaxed);
            pending_.store((int)helpers_.size(), std::memory_order_relaxed);
            generation_.fetch_add(1, std::memory_order_release);
        }
        cv_work_.notify_all();
        work();
        for (int spin = 0; spin < kSpin && pending_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
        if (pending_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> lk(m_);
            cv_done_.wait(lk, [&] { return pending_.load(std::memory_order_acquire) == 0; });
        }
        job_ = nullptr;
    }

  private:
    explicit Pool(int n) : n_(n > 0 ? (n > 1024 ? 1024 : n) : usable_cpus()) {
        static const int once = pthread_atfork(nullptr, nullptr, [] { inst_.store(nullptr, std::memory_order_release); });
        (void)once;
        start_helpers();
    }
    static std::atomic<Pool *> inst_;
    static std::atomic<int> first_size_;
    static int usable_cpus() {
        int n = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) {
            const int a = CPU_COUNT(&set);
            if (a > 0 && (n <= 0 || a < n)) n = a;
        }
        return n > 0 ? n : 1;
    }
    void work() {
        for (;;) {
            const int u = next_.fetch_add(1, std::memory_order_relaxed);
            if (u >= units_) return;
            (*job_)(u);
        }
    }
    void start_helpers() {
        stop_.store(false, std::memory_order_relaxed);
        for (int i = 1; i < n_; i++)
            helpers_.emplace_back([this] {
                uint64_t seen = 0;
                for (;;) {
                    // a request is several parallel regions a few hundred microseconds apart (gather, four layers): a helper that went to sleep
                    // between them costs a futex wake-up per region (measured here: 11 ms instead of 2.4 ms for a batch of 256) -- spin briefly first
                    for (int spin = 0; spin < kSpin && generation_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed); spin++) cpu_relax();
                    if (generation_.load(std::memory_order_acquire) == seen) {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_work_.wait(lk, [&] { return stop_.load(std::memory_order_relaxed) || generation_.load(std::memory_order_acquire) != seen; });
                    }
                    if (stop_.load(std::memory_order_relaxed)) return;
                    seen = generation_.load(std::memory_order_acquire);
                    work();
                    if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                        std::lock_guard<std::mutex> lk(m_);
                        cv_done_.notify_one();
                    }
                }
            });
    }
    void stop_helpers() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true, std::memory_order_relaxed);
        }
        cv_work_.notify_all();
        for (auto &t : helpers_) t.join();
        helpers_.clear();
    }
    std::mutex run_m_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> helpers_;
    const std::function<void(int)> *job_ = nullptr;
    static constexpr int kSpin = 40000;   // ~100-200 us of pause instructions before a thread blocks
#if defined(__x86_64__)
    static void cpu_relax() { __builtin_ia32_pause(); }
#else
    static void cpu_relax() { std::atomic_signal_fence(std::memory_order_seq_cst); }   // portable: a compiler barrier, no pause hint
#endif
    std::atomic<int> next_{0}, pending_{0};
    int units_ = 0, n_ = 1;
    std::atomic<uint64_t> generation_{0};
    std::atomic<bool> stop_{false};
};
std::atomic<Pool *> Pool::inst_{nullptr};
std::atomic<int> Pool::first_size_{0};
}  // namespace

int frc_set_threads(int n) {
    if (!Pool::exists()) Pool::set_first_size(n);
    return Pool::get().resize(n);
}
int frc_threads() { return Pool::get().size(); }

// ---- memory ------------------------------------------------------------------------------------------------------------------------
// Table arena: anonymous zero pages, committed when first written (a table that is never filled costs nothing).  Refused up front when
// it could not fit the host's physical memory: the kernel would otherwise kill the process half-way through the fill.
void *frc_arena_alloc(size_t bytes) {
    if (bytes == 0) bytes = 4096;
    const long pages = sysconf(_SC_PHYS_PAGES), psz = sysconf(_SC_PAGESIZE);
    double limit = pages > 0 && psz > 0 ? (double)pages * (double)psz : 0.0;
    for (const char *path : {"/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"}) {   // a container's own limit, where one is set
        if (FILE *f = fopen(path, "r")) {
            double v = 0.0;
            if (fscanf(f, "%lf", &v) == 1 && v > 0.0 && (limit == 0.0 || v < limit)) limit = v;
            fclose(f);
        }
    }
    if (limit > 0.0 && (double)bytes > 0.85 * limit) {
        fr_set_error("CPU back-end: %.1f GB of tables do not fit this host's %.1f GB of memory", bytes / 1e9, limit / 1e9);
        return nullptr;
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) {
        fr_set_error("CPU back-end: mmap of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}
void frc_arena_free(void *p, size_t bytes) {
    if (p) munmap(p, bytes ? bytes : 4096);
}

// ---- procedural contents (fr_content.h: the same functions the gfx950 fill kernels run) -------------------------------------------------
int frc_fill_table(float *base, int64_t row0, int64_t rows, int dim, int64_t row_stride_bytes, int mode, uint32_t seed, uint32_t uid) {
    const uint32_t h0 = fr_table_hash_seed(seed, uid);
    const int64_t chunk = 4096;
    const int units = (int)((rows + chunk - 1) / chunk);
    char *b = reinterpret_cast<char *>(base);
    Pool::get().run(units, [&](int u) {
        const int64_t r0 = (int64_t)u * chunk, r1 = r0 + chunk < rows ? r0 + chunk : rows;
        for (int64_t lr = r0; lr < r1; lr++) {
            uint32_t *dst = reinterpret_cast<uint32_t *>(b + (size_t)lr * (size_t)row_stride_bytes);
            for (int c = 0; c < dim; c++) dst[c] = fr_content_bits(mode, h0, uid, (uint64_t)(row0 + lr), (uint32_t)c);
        }
    });
    return FR_OK;
}

int frc_fill_weights(float *w, size_t count, int mode, uint32_t seed, uint32_t layer, float scale) {
    const uint32_t h0 = fr_weight_hash_seed(seed, layer);
    const size_t chunk = 1 << 16;
    Pool::get().run((int)((count + chunk - 1) / chunk), [&](int u) {
        const size_t i0 = (size_t)u * chunk, i1 = i0 + chunk < count ? i0 + chunk : count;
        for (size_t i = i0; i < i1; i++) w[i] = fr_weight_value(mode, h0, (uint64_t)i, scale);
    });
    return FR_OK;
}

// ---- the gather: load_single_embedding_* + the packers (embedding_47_krnl.cpp:916-1217), as the word descriptors state them -------------
// Record word w of item b = the 16 bytes at words[w].src + idx[b][words[w].idx_col] * words[w].stride, stored at word
// dst_blk * batch + b * dst_stride + dst_off of `out` -- exactly what gather_pack_kernel (fr_gather.hip) does per thread.  An index
// outside its table reads row 0 and raises *err_flag (reference: silent out-of-bounds read, embedding_47_krnl.cpp:927-933).
int frc_gather(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, float *out, int batch, int *err_flag) {
    const int chunk = 32;
    std::atomic<int> bad{0};
    char *o = reinterpret_cast<char *>(out);
    Pool::get().run((batch + chunk - 1) / chunk, [&](int u) {
        const int b0 = u * chunk, b1 = b0 + chunk < batch ? b0 + chunk : batch;
        int local_bad = 0;
        for (int b = b0; b < b1; b++) {
            const int32_t *row = idx + (size_t)b * idx_stride;
            for (int w = 0; w < n_words; w++) {
                const FrWordDesc &d = words[w];
                const char *src;
                if (d.idx_col & FR_DESC_DENSE) {
                    src = reinterpret_cast<const char *>(dense) + d.src + (size_t)b * d.stride;
                } else {
                    uint32_t r = (uint32_t)row[d.idx_col];
                    if (r >= d.rows) {
                        local_bad = 1;
                        r = 0;
                    }
                    src = reinterpret_cast<const char *>(d.src) + (uint64_t)r * d.stride;
                }
                memcpy(o + ((size_t)d.dst_blk * (size_t)batch + (size_t)b * d.dst_stride + d.dst_off) * 16, src, 16);
            }
        }
        if (local_bad) bad.store(1, std::memory_order_relaxed);
    });
    if (bad.load() && err_flag) __atomic_store_n(err_flag, 1, __ATOMIC_RELEASE);
    return FR_OK;
}

// ---- the FC chain: 4 x cublasLtMatmul, alpha = 1, beta = 0, no bias, no activation (cuda_server.c:211-217,468-491) ---------------------
// Y[b][h] = sum over k, IN k ORDER, of W[h + k * H] * X[b][k], each step one fused multiply-add in fp32.  A tile of MB items x HT outputs
// keeps its sums in registers while k runs; W is read in its own (column-major: h contiguous) order.
namespace {
template <int MB, int HT>
static inline __attribute__((always_inline)) void fc_tile(const float *W, int H, int K, const float *const x[MB], float *const y[MB], int h0) {
    float acc[MB][HT];
    for (int b = 0; b < MB; b++)
        for (int j = 0; j < HT; j++) acc[b][j] = 0.0f;
    for (int k = 0; k < K; k++) {
        const float *wr = W + (size_t)k * H + h0;
        for (int b = 0; b < MB; b++) {
            const float xv = x[b][k];
            for (int j = 0; j < HT; j++) acc[b][j] = __builtin_fmaf(wr[j], xv, acc[b][j]);
        }
    }
    for (int b = 0; b < MB; b++)
        if (y[b])
            for (int j = 0; j < HT; j++) y[b][h0 + j] = acc[b][j];
}

// one unit of a layer: items [b0, b0 + MB) x outputs [h0, h0 + HT) (ragged item blocks compute item b0 again and do not store it)
template <int MB, int HT>
static inline __attribute__((always_inline)) void fc_unit(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch, int b0, int h0) {
    const float *x[MB];
    float *y[MB];
    for (int b = 0; b < MB; b++) {
        const bool in = b0 + b < batch;
        x[b] = X + (size_t)(in ? b0 + b : b0) * ldx;
        y[b] = in ? Y + (size_t)(b0 + b) * ldy : nullptr;
    }
    fc_tile<MB, HT>(W, H, K, x, y, h0);
}

constexpr int FC_MB = 4, FC_HT = 32;   // H is a multiple of 32 (fr_model_validate)
// a unit = FC_MB items x FC_HT outputs; the last, ragged item block of a batch runs item by item (a request of one item must not pay for four)
template <int MB>
static inline __attribute__((always_inline)) void fc_unit_any(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch, int b0, int h0) {
    if (b0 + MB <= batch) {
        fc_unit<MB, FC_HT>(W, H, K, X, ldx, Y, ldy, batch, b0, h0);
        return;
    }
    for (int b = b0; b < batch; b++) fc_unit<1, FC_HT>(W, H, K, X, ldx, Y, ldy, batch, b, h0);
}
#if defined(__x86_64__)
// Per-function ISA clones, each naming EXACTLY the features pick_fc_unit() checks (target("arch=x86-64-v3") also licenses
// BMI / LZCNT / MOVBE / F16C, which a VM may mask while it shows AVX2 + FMA).
__attribute__((target("avx512f,avx512vl,avx512bw,avx512dq,avx2,fma"))) void fc_unit_v4(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch, int b0, int h0) {
    fc_unit_any<FC_MB>(W, H, K, X, ldx, Y, ldy, batch, b0, h0);
}
__attribute__((target("avx2,fma"))) void fc_unit_v3(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch, int b0, int h0) {
    fc_unit_any<FC_MB>(W, H, K, X, ldx, Y, ldy, batch, b0, h0);
}
#endif
void fc_unit_base(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch, int b0, int h0) {   // no FMA unit: fmaf() in software, still one rounding per step
    fc_unit_any<FC_MB>(W, H, K, X, ldx, Y, ldy, batch, b0, h0);
}
using FcUnitFn = void (*)(const float *, int, int, const float *, int, float *, int, int, int, int);
FcUnitFn pick_fc_unit() {
#if defined(__x86_64__)
    __builtin_cpu_init();
    const bool v3 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    if (v3 && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq")) return fc_unit_v4;
    if (v3) return fc_unit_v3;
#endif
    return fc_unit_base;   // any other host: the compiler's baseline (fmaf() keeps one rounding per step everywhere)
}

void fc_layer(const float *W, int H, int K, const float *X, int ldx, float *Y, int ldy, int batch) {
    static const FcUnitFn unit = pick_fc_unit();
    if (H % FC_HT) {   // the output layer (H = 1) and any narrow layer: one k-ordered chain per (item, output)
        Pool::get().run(batch, [&](int b) {
            const float *x = X + (size_t)b * ldx;
            for (int h = 0; h < H; h++) {
                float acc = 0.0f;
                for (int k = 0; k < K; k++) acc = __builtin_fmaf(W[h + (size_t)k * H], x[k], acc);
                Y[(size_t)b * ldy + h] = acc;
            }
        });
        return;
    }
    const int nb = (batch + FC_MB - 1) / FC_MB, nh = H / FC_HT;
    if ((double)batch * H * K < 4e6) {   // a request of a few items: a parallel region's hand-over would cost more than the work
        for (int u = 0; u < nb * nh; u++) unit(W, H, K, X, ldx, Y, ldy, batch, (u / nh) * FC_MB, (u % nh) * FC_HT);
        return;
    }
    // a work unit = ONE 32-output slice of W (K x 128 bytes: it stays in the core's L1 / L2) against a run of FC_IB item blocks: the weights are
    // streamed once per run instead of once per item block (batch 16384, Model-A FC1: 0.7 GB of X re-reads instead of 5.9 GB of W re-reads)
    constexpr int FC_IB = 16;
    const int nr = (nb + FC_IB - 1) / FC_IB;
    Pool::get().run(nr * nh, [&](int u) {
        const int h0 = (u % nh) * FC_HT, ib0 = (u / nh) * FC_IB, ib1 = ib0 + FC_IB < nb ? ib0 + FC_IB : nb;
        for (int ib = ib0; ib < ib1; ib++) unit(W, H, K, X, ldx, Y, ldy, batch, ib * FC_MB, h0);
    });
}
}  // namespace

// X: item-major records [batch][K] (for the BLOCKED layout: the 3-node buffer read "as if it were B x K item-major", 3-node
// cuda_server.c:216-217).  scratch: batch * (H1 + H2 + H3) floats.  w[l]: column-major H x K (fr_ctx_set_weights).
void frc_fc_chain(const int32_t fc[5], const float *const w[4], const float *X, int batch, float *scratch, float *scores) {
    float *r1 = scratch, *r2 = r1 + (size_t)batch * fc[1], *r3 = r2 + (size_t)batch * fc[2];
    fc_layer(w[0], fc[1], fc[0], X, fc[0], r1, fc[1], batch);   // R1 = W1 * X        cuda_server.c:468-472
    fc_layer(w[1], fc[2], fc[1], r1, fc[1], r2, fc[2], batch);  // R2 = W2 * R1       :474-478
    fc_layer(w[2], fc[3], fc[2], r2, fc[2], r3, fc[3], batch);  // R3 = W3 * R2       :480-484
    fc_layer(w[3], fc[4], fc[3], r3, fc[3], scores, fc[4], batch);  // out = Wout * R3 :486-491
}

// Table-sharded mode after the exchange: gathered = [n_shards][batch_total][slice_padded] floats -> item-major records [n_items][K] of
// items [item0, item0 + n_items) (the 3-node server's three-part receive buffer generalised to G parts, 3-node cuda_server.c:513-591).
void frc_slices_to_records(const float *gathered, int n_shards, int batch_total, int slice_padded, const int *offs, const int *lens, int item0, int n_items, float *X, int K) {
    Pool::get().run(n_items, [&](int i) {
        for (int g = 0; g < n_shards; g++)
            memcpy(X + (size_t)i * K + offs[g], gathered + ((size_t)g * batch_total + item0 + i) * slice_padded, (size_t)lens[g] * sizeof(float));
    });
}
