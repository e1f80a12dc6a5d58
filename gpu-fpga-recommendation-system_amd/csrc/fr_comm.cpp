// Table-sharded mode behind the C-ABI: the exchange step on RCCL (include/fleetrec.h "table-sharded mode").
//
// Reference shape: the 3-node GPU server receives the three parts of every batch from three senders (CPU0, FPGA0, FPGA1) over
// three sockets before its GEMMs (GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/cuda_server.c:513-591).  Here the "senders" are
// the G GPUs that hold the table-ID shards: each gathers its slice of the record, ONE ncclAllGather over xGMI delivers all slices to
// everybody, each GPU runs the FC chain on its B/G items, and a second (tiny) all-gather hands every rank all B scores.
// librccl.so is opened on first use (dlopen), so unsharded users never load it; every RCCL failure surfaces as FR_ERR_COMM.
//
// TWO TRANSPORTS under ONE protocol (round 6).  The step -- the uneven item split, the status word behind every score chunk, the
// communicator's reference counts, the bounded wait of fr_worker_sync, the three kinds of failure -- is written once (sharded_step,
// comm_wait_step).  What differs sits below it:
//   * RCCL: G GPU contexts; ncclAllGather on the worker's HIP stream; the step is ENQUEUED by fr_worker_submit_sharded.
//   * the in-process host exchange: G CPU contexts (device = -1) of one process driven by G host threads; the all-gather is a rendezvous of
//     those threads (HostGroup: publish, barrier, copy, barrier); the step RUNS on the worker's host stream (HostStream: one thread per
//     worker that executes the posted step in order -- what a HIP stream is to a GPU worker), so fr_worker_submit_sharded returns at once and
//     fr_worker_sync polls that stream with the same bounded loop.  Like an RCCL collective, the rendezvous itself waits without a bound and is
//     released by an abort; the bound lives in fr_worker_sync.
//   * the same host exchange, STAGED, for GPU shard contexts that share a device (fr_comm_init_all over contexts of which two sit on one GPU --
//     RCCL wants one device per rank): every all-gather is D2H into the worker's pinned staging, the rendezvous, H2D of all G parts; the step runs
//     on a host stream as a CPU worker's does (the rendezvous blocks), the kernels and copies on the worker's HIP stream.  It is how the WHOLE
//     step -- slice gathers in the chain's operand type, the uneven split on the device, NaN poisoning, status words through hipMemcpy2DAsync,
//     the bf16 / fp8 chains on all-gathered slices, the sharded fp8 calibration -- runs with G = 2 .. 8 ranks on a one-GPU box.
// The host exchange exists so that the protocol meets ranks 1 .. G-1 on machines without G GPUs (tests/test_cpu_backend.py, TSan build;
// tests/test_gpu_sharded.py on one GPU) and as the exchange of `fleetrec_server --shards G --device -1`.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the functions are resolved through dlsym

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>
#include <mutex>
#include <new>
#include <thread>

#include "fr_internal.h"

namespace {
struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: a rank that cannot take part in a collective aborts the communicator
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional: polled by the bounded wait of fr_comm_wait
};
Rccl g_rccl;
std::once_flag g_rccl_once;
bool g_rccl_ok = false;
char g_rccl_why[256] = "symbol missing";

int rccl_load() {
    std::call_once(g_rccl_once, [] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_rccl.handle) break;
            const char *e = dlerror();  // dlerror() clears itself: read it once
            if (e) snprintf(g_rccl_why, sizeof(g_rccl_why), "%s", e);
        }
        if (!g_rccl.handle) return;
        snprintf(g_rccl_why, sizeof(g_rccl_why), "symbol missing");
#define FR_SYM(field, name)                                              \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
    if (!g_rccl.field) return;
        FR_SYM(GetUniqueId, "ncclGetUniqueId")
        FR_SYM(CommInitRank, "ncclCommInitRank")
        FR_SYM(CommInitAll, "ncclCommInitAll")
        FR_SYM(CommDestroy, "ncclCommDestroy")
        FR_SYM(AllGather, "ncclAllGather")
        FR_SYM(GetErrorString, "ncclGetErrorString")
#undef FR_SYM
        g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(g_rccl.handle, "ncclCommAbort"));
        g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(g_rccl.handle, "ncclCommGetAsyncError"));
        g_rccl_ok = true;
    });
    if (!g_rccl_ok) FR_FAIL(FR_ERR_COMM, "librccl.so could not be loaded (dlopen / dlsym): %s", g_rccl_why);
    return FR_OK;
}
}  // namespace


// ---- transport 2: the in-process host exchange --------------------------------------------------------------------------------------
namespace {
constexpr int FR_HOST_MAX_RANKS = 64;
// One per fr_comm_init_all over CPU contexts (or over GPU contexts that share a device: the staged form), shared by its G handles.  all_gather: every rank publishes its send buffer and waits for the
// others (barrier 1), copies all G buffers into its own receive buffer, and waits until everybody has copied (barrier 2: only then may a
// send buffer be reused).  Barrier 1 ends early when the group is aborted; barrier 2 never does -- every rank that left barrier 1 is copying
// from its peers' buffers and will count itself in, and no buffer may go away under a peer's memcpy.
struct HostGroup {
    const int n;
    std::mutex m;
    std::condition_variable cv;
    const void *send[FR_HOST_MAX_RANKS] = {};
    size_t bytes[FR_HOST_MAX_RANKS] = {};
    int arrived = 0, copied = 0;
    uint64_t gen_arrive = 0, gen_copy = 0;
    bool broken = false;            // under m
    std::atomic<bool> broken_flag{false};   // the same, readable without the lock (the bounded wait's "asynchronous error" poll)
    std::atomic<int> refs;
    explicit HostGroup(int n_) : n(n_), refs(n_) {}
    void abort() {
        {
            std::lock_guard<std::mutex> lk(m);
            broken = true;
            broken_flag.store(true, std::memory_order_release);
        }
        cv.notify_all();
    }
    // -> FR_OK, or FR_ERR_COMM when the group was aborted before everybody had arrived / the ranks disagree about the size
    int all_gather(int rank, const void *sendbuf, void *recvbuf, size_t nbytes) {
        const void *src[FR_HOST_MAX_RANKS];
        bool same = true;
        {
            std::unique_lock<std::mutex> lk(m);
            if (broken) return FR_ERR_COMM;
            send[rank] = sendbuf;
            bytes[rank] = nbytes;
            const uint64_t g = gen_arrive;
            if (++arrived == n) {
                arrived = 0;
                gen_arrive++;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return gen_arrive != g || broken; });
                if (gen_arrive == g) {   // aborted while waiting: this rank's arrival is withdrawn
                    arrived--;
                    return FR_ERR_COMM;
                }
            }
            for (int q = 0; q < n; q++) {
                src[q] = send[q];
                same &= bytes[q] == nbytes;
            }
            if (!same) {   // every rank reads the same table and takes the same decision: nobody copies, everybody still passes barrier 2
                broken = true;
                broken_flag.store(true, std::memory_order_release);
            }
        }
        if (same)
            for (int q = 0; q < n; q++) memcpy(static_cast<char *>(recvbuf) + (size_t)q * nbytes, src[q], nbytes);
        {
            std::unique_lock<std::mutex> lk(m);
            const uint64_t g = gen_copy;
            if (++copied == n) {
                copied = 0;
                gen_copy++;
                cv.notify_all();
            } else {
                cv.wait(lk, [&] { return gen_copy != g; });
            }
        }
        return same ? FR_OK : FR_ERR_COMM;
    }
};

// The host stream of a worker whose exchange is the host group (CPU workers; GPU workers of the staged form): one thread, one posted step at a
// time (a worker has at most one sharded step in flight).
struct HostStream {
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, stop = false;
    std::atomic<bool> busy{false};
    int rc = FR_OK;           // of the last step; read after busy has gone false
    char text[512] = "";      // fr_last_error() of the stream thread after a failed step
    std::thread th;
    HostStream() : th([this] { loop(); }) {}
    ~HostStream() {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
        }
        cv.notify_all();
        th.join();
    }
    void post(std::function<int()> f) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(f);
            has_job = true;
            busy.store(true, std::memory_order_release);
        }
        cv.notify_all();
    }
    bool done() const { return !busy.load(std::memory_order_acquire); }
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !busy.load(std::memory_order_acquire); });
    }
    void loop() {
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_job || stop; });
                if (!has_job) return;
                f = std::move(job);
                has_job = false;
            }
            const int r = f();
            {
                std::lock_guard<std::mutex> lk(m);
                rc = r;
                snprintf(text, sizeof(text), "%s", r ? fr_last_error() : "");
                busy.store(false, std::memory_order_release);
            }
            cv.notify_all();
        }
    }
};
HostStream *host_stream(fr_worker *w) {
    if (!w->sh_host_stream) w->sh_host_stream = new HostStream();
    return static_cast<HostStream *>(w->sh_host_stream);
}
}  // namespace

#define FR_NCCL(call)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) FR_FAIL(FR_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

struct fr_comm {
    ncclComm_t comm = nullptr;   // transport 1: RCCL (GPU contexts)
    HostGroup *grp = nullptr;    // transport 2: the in-process host exchange (CPU contexts; staged: GPU contexts sharing a device); exactly one of the two is set
    int rank = 0, n_ranks = 1;
    fr_ctx *ctx = nullptr;
    // a collective step failed on this rank: the communicator was aborted, every later call returns FR_ERR_COMM (atomic: a CPU worker's step
    // runs on its host stream while fr_worker_sync polls)
    std::atomic<bool> broken{false};
    std::atomic<int> wait_ms{60000};  // bound of fr_comm_wait: how long fr_worker_sync lets a step's collectives take before it gives the peers up
    // one reference for the handle fr_comm_init_* gave out + one per worker with a sharded step in flight through this communicator
    // (fr_worker::sh_comm): fr_comm_destroy between a submit and its sync only drops the handle's reference, the communicator itself goes
    // when the last worker has synchronised (fr_comm_wait used to dereference a freed object in that sequence)
    std::atomic<int> refs{1};
};
constexpr int FR_COMM_MAX_RANKS = 64;   // fr_worker::h_sh_status holds 1 + FR_COMM_MAX_RANKS status words
static_assert(FR_HOST_MAX_RANKS == FR_COMM_MAX_RANKS, "one limit for both transports");

// Failure protocol of a collective step.  Three kinds of failure, three answers:
//  (1) argument / state errors found BEFORE anything was enqueued (worker busy, batch too large, tables not filled ...): returned as they
//      are, the communicator stays usable.  The ranks of a job are driven with the same arguments, so such an error is the same on every
//      rank; a caller that lets ONE rank skip a step has broken the contract and its peers run into (3).
//  (2) the local FC chain fails after the first collective: both collectives are still issued (the peers are on their way into them), the
//      rank's score chunk is poisoned with NaN and its STATUS WORD -- one float all-gathered behind every score chunk -- carries the code:
//      every rank's fr_worker_sync reads all G status words and returns FR_ERR_COMM naming the rank; nobody gets stale scores silently.
//      (RCCL: the failing rank's submit returns the chain's error too; host exchange: the step runs behind the submit, so the failing rank
//      learns it from its own fr_worker_sync like everybody else.)
//  (3) a real failure of the device / the transport on this rank (a HIP error, a collective that cannot be enqueued), or peers that do not
//      arrive: the rank ABORTS its communicator (comm_fail) and answers FR_ERR_COMM from then on; the peers' waits are BOUNDED (fr_comm_wait:
//      the step's stream is polled together with the transport's asynchronous-error state for at most wait_ms, then the waiting rank aborts
//      its own communicator and returns FR_ERR_COMM) -- a local ncclCommAbort does not by itself release a peer in another process; an abort
//      of the in-process host group does release every rank of it at once.
// Executed with G = 2, 3 and 8 ranks on the host exchange (tests/test_cpu_backend.py, also under ThreadSanitizer); over RCCL with one-rank
// communicators only -- the test boxes have one GPU.
static int comm_fail(fr_comm *comm, int rc) {
    if (comm && !comm->broken.exchange(true, std::memory_order_acq_rel)) {
        if (comm->grp) comm->grp->abort();
        if (comm->comm && g_rccl.CommAbort) {
            (void)g_rccl.CommAbort(comm->comm);
            comm->comm = nullptr;  // aborted communicators are already destroyed
        }
    }
    return rc;
}
static bool comm_is_broken(const fr_comm *comm) {
    return comm->broken.load(std::memory_order_acquire) || (comm->grp && comm->grp->broken_flag.load(std::memory_order_acquire));
}
#define FR_BROKEN_TEXT "the communicator was aborted by an earlier failure (on this rank, or on a peer of the in-process host exchange)"

static_assert(sizeof(ncclUniqueId) == 128, "fr_comm_unique_id hands out 128 bytes");

extern "C" int fr_comm_unique_id(void *id128) {
    if (!id128) FR_FAIL(FR_ERR_INVALID, "id128 is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId id;
    FR_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, sizeof(id));
    return FR_OK;
}

static int comm_check_ctx(const fr_ctx *ctx) {
    if (!ctx) FR_FAIL(FR_ERR_INVALID, "ctx is NULL");
    if (ctx->n_shards > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "%d shards: the exchange serves at most %d ranks", ctx->n_shards, FR_COMM_MAX_RANKS);
    if (ctx->model.layout != FR_LAYOUT_SEMANTIC) FR_FAIL(FR_ERR_STATE, "the sharded exchange needs the SEMANTIC layout");
    return FR_OK;
}

extern "C" int fr_comm_init_rank(fr_ctx *ctx, const void *id128, fr_comm **out) {
    if (!out || !id128) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = comm_check_ctx(ctx);
    if (rc) return rc;
    if (ctx->cpu) FR_FAIL(FR_ERR_STATE, "CPU contexts (device = -1) exchange in-process only: set the G shard contexts of one process up with fr_comm_init_all");
    rc = rccl_load();
    if (rc) return rc;
    FR_HIP(hipSetDevice(ctx->device));
    fr_comm *c = new (std::nothrow) fr_comm();
    if (!c) FR_FAIL(FR_ERR_OOM, "out of host memory");
    c->rank = ctx->shard_rank;
    c->n_ranks = ctx->n_shards;
    c->ctx = ctx;
    fr_ctx_ref(ctx);   // (dropped by comm_release: a communicator destroyed after its context touches no freed memory)
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, c->n_ranks, id, c->rank);
    if (r != ncclSuccess) {
        delete c;
        fr_ctx_unref(ctx);
        FR_FAIL(FR_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", ctx->shard_rank, ctx->n_shards, g_rccl.GetErrorString(r));
    }
    *out = c;
    return FR_OK;
}

extern "C" int fr_comm_init_all(fr_ctx *const *ctxs, int n, fr_comm **out) {
    if (!ctxs || !out || n < 1 || n > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "bad argument");
    for (int r = 0; r < n; r++) out[r] = nullptr;
    int rc;
    int devs[64];
    bool shared_device = false;
    for (int r = 0; r < n; r++) {
        rc = comm_check_ctx(ctxs[r]);
        if (rc) return rc;
        if (ctxs[r]->n_shards != n || ctxs[r]->shard_rank != r) FR_FAIL(FR_ERR_INVALID, "ctxs[%d] is shard %d of %d, expected %d of %d", r, ctxs[r]->shard_rank, ctxs[r]->n_shards, r, n);
        if (ctxs[r]->cpu != ctxs[0]->cpu) FR_FAIL(FR_ERR_INVALID, "ctxs[%d]: CPU and GPU shard contexts cannot share a communicator", r);
        devs[r] = ctxs[r]->device;
        if (!ctxs[r]->cpu)
            for (int q = 0; q < r; q++) shared_device |= devs[q] == devs[r];
    }
    // CPU contexts: the in-process host exchange.  GPU contexts of which two share a device (RCCL needs one device per rank): the same
    // exchange, staged through pinned host memory.  One device per rank: RCCL.
    const bool host = ctxs[0]->cpu || shared_device;
    ncclComm_t comms[64];
    HostGroup *grp = nullptr;
    if (host) {   // the in-process host exchange: nothing of RCCL is touched (or loaded)
        grp = new (std::nothrow) HostGroup(n);
        if (!grp) FR_FAIL(FR_ERR_OOM, "out of host memory");
    } else {
        rc = rccl_load();
        if (rc) return rc;
        FR_NCCL(g_rccl.CommInitAll(comms, n, devs));
    }
    for (int r = 0; r < n; r++) {
        fr_comm *c = new (std::nothrow) fr_comm();
        if (!c) {
            for (int q = 0; q < r; q++) {
                fr_comm_destroy(out[q]);
                out[q] = nullptr;
            }
            if (host) {
                if (grp->refs.fetch_sub(n - r, std::memory_order_acq_rel) == n - r) delete grp;   // the handles never made
            } else {
                for (int q = r; q < n; q++) (void)g_rccl.CommDestroy(comms[q]);
            }
            FR_FAIL(FR_ERR_OOM, "out of host memory");
        }
        if (host) c->grp = grp;
        else c->comm = comms[r];
        c->rank = r;
        c->n_ranks = n;
        c->ctx = ctxs[r];
        fr_ctx_ref(ctxs[r]);
        out[r] = c;
    }
    return FR_OK;
}

extern "C" int fr_comm_set_wait_ms(fr_comm *c, int wait_ms) {
    if (!c || wait_ms < 1) FR_FAIL(FR_ERR_INVALID, "bad argument");
    c->wait_ms.store(wait_ms, std::memory_order_relaxed);
    return FR_OK;
}

static void comm_release(fr_comm *c) {
    if (c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    if (c->comm && g_rccl_ok) {
        if (c->ctx) (void)hipSetDevice(c->ctx->device);
        (void)g_rccl.CommDestroy(c->comm);
    }
    if (c->grp && c->grp->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) delete c->grp;   // the last of the G handles takes the group along
    fr_ctx *held = c->ctx;
    delete c;
    fr_ctx_unref(held);
}

extern "C" void fr_comm_destroy(fr_comm *c) {
    if (c) comm_release(c);
}

// fr_worker_sync leaves early (another path of the worker failed first): the step's reference goes without a wait
void fr_comm_forget(fr_worker *w) {
    fr_comm *comm = w->sh_comm;
    w->sh_comm = nullptr;
    if (comm) comm_release(comm);
}

bool fr_comm_step_on_host_stream(const fr_worker *w) { return w->sh_host_stream && !static_cast<const HostStream *>(w->sh_host_stream)->done(); }

// fr_worker_destroy of a worker with a host stream (CPU workers; GPU workers of a staged exchange): a step still running on the host stream is given its communicator's bound, then the exchange is
// aborted (which releases a step parked in a rendezvous); the stream thread is joined, the step's reference dropped.
void fr_comm_worker_release(fr_worker *w) {
    if (w->sh_host_stream) {
        HostStream *hs = static_cast<HostStream *>(w->sh_host_stream);
        fr_comm *comm = w->sh_comm;
        if (!hs->done() && comm) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            while (!hs->done()) {
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > comm->wait_ms.load(std::memory_order_relaxed)) {
                    (void)comm_fail(comm, FR_ERR_COMM);
                    break;
                }
                struct timespec nap = {0, 200000};
                nanosleep(&nap, nullptr);
            }
        }
        hs->wait();
        delete hs;
        w->sh_host_stream = nullptr;
    }
    fr_comm_forget(w);
}

// ---- the few device operations of a step, on either back-end: a command on the worker's HIP stream, or the same thing done at once by the
// thread that runs the step (a CPU worker's step runs on its host stream: "enqueued" there means "done, in order") --------------------------
static void *step_alloc(fr_worker *w, size_t bytes) {
    void *p = nullptr;
    if (w->ctx->cpu) return aligned_alloc(64, (bytes + 63) / 64 * 64);
    return hipMalloc(&p, bytes) == hipSuccess ? p : nullptr;
}
static void step_free(fr_worker *w, void *p) {
    if (!p) return;
    if (w->ctx->cpu) free(p);
    else (void)hipFree(p);
}
static int step_copy(fr_worker *w, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    if (w->ctx->cpu) {
        memcpy(dst, src, bytes);
        return FR_OK;
    }
    FR_HIP(hipMemcpyAsync(dst, src, bytes, kind, w->stream));
    return FR_OK;
}
// the status word behind each of the G score chunks of d_score_all ([G][chunk + 1]) -> h_sh_status[1 .. G]: one strided copy
static int step_copy_status(fr_worker *w, int G, int chunk) {
    if (w->ctx->cpu) {
        for (int q = 0; q < G; q++) w->h_sh_status[1 + q] = w->d_score_all[(size_t)q * (chunk + 1) + chunk];
        return FR_OK;
    }
    FR_HIP(hipMemcpy2DAsync(w->h_sh_status + 1, sizeof(float), w->d_score_all + chunk, ((size_t)chunk + 1) * sizeof(float), sizeof(float), (size_t)G, hipMemcpyDeviceToHost, w->stream));
    return FR_OK;
}
static int step_fill_ff(fr_worker *w, void *dst, size_t bytes) {   // NaN in every float
    if (w->ctx->cpu) {
        memset(dst, 0xFF, bytes);
        return FR_OK;
    }
    FR_HIP(hipMemsetAsync(dst, 0xFF, bytes, w->stream));
    return FR_OK;
}
// ONE all-gather of `bytes` per rank through the communicator's transport
static int step_all_gather(fr_worker *w, fr_comm *comm, const void *send, void *recv, size_t bytes) {
    if (comm->grp) {
        const void *h_send = send;
        void *h_recv = recv;
        if (!w->ctx->cpu) {   // staged: this rank's part comes off the device first (everything the step enqueued so far has then run)
            FR_HIP(hipMemcpyAsync(w->h_stage_send, send, bytes, hipMemcpyDeviceToHost, w->stream));
            FR_HIP(hipStreamSynchronize(w->stream));
            h_send = w->h_stage_send;
            h_recv = w->h_stage_recv;
        }
        if (comm->grp->all_gather(comm->rank, h_send, h_recv, bytes) != FR_OK)
            FR_FAIL(FR_ERR_COMM, "host exchange: the all-gather of %zu bytes per rank was aborted (a peer failed, never arrived, or sent another size)", bytes);
        if (!w->ctx->cpu) FR_HIP(hipMemcpyAsync(recv, w->h_stage_recv, bytes * (size_t)comm->n_ranks, hipMemcpyHostToDevice, w->stream));
        return FR_OK;
    }
    ncclResult_t r = g_rccl.AllGather(send, recv, bytes, ncclChar, comm->comm, w->stream);
    if (r != ncclSuccess) FR_FAIL(FR_ERR_COMM, "ncclAllGather(%zu bytes per rank) failed: %s", bytes, g_rccl.GetErrorString(r));
    return FR_OK;
}
// -> 0 the step's stream is idle, 1 still working, < 0 the stream itself failed
static int step_query(fr_worker *w) {
    if (w->sh_host_stream && !static_cast<HostStream *>(w->sh_host_stream)->done()) return 1;   // the step is still being issued (host exchange, plain or staged)
    if (w->ctx->cpu) return 0;
    hipError_t q = hipStreamQuery(w->stream);
    if (q == hipSuccess) return 0;
    if (q == hipErrorNotReady) return 1;
    fr_set_error("sharded step: stream failed: %s", hipGetErrorString(q));
    return FR_ERR_HIP;
}
// the transport's asynchronous-error state (polled by the bounded wait): true = the exchange cannot complete any more
static bool step_async_error(fr_comm *comm) {
    if (comm->grp) {
        if (!comm->grp->broken_flag.load(std::memory_order_acquire)) return false;
        fr_set_error("sharded step: the in-process host exchange was aborted (by a peer rank, or by this rank's own step)");
        return true;
    }
    ncclResult_t ar = ncclSuccess;
    if (g_rccl.CommGetAsyncError && comm->comm && g_rccl.CommGetAsyncError(comm->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
        fr_set_error("sharded step: RCCL reported an asynchronous error: %s", g_rccl.GetErrorString(ar));
        return true;
    }
    return false;
}

// exchange buffers of a worker: slice [max_batch][F] and gathered [G][max_batch][F] sized for fp32 elements, score chunks
static int shard_buffers(fr_worker *w, int G, bool staged) {
    if (w->sh_ranks == G && w->d_slice && (!staged || w->ctx->cpu || w->h_stage_send)) return FR_OK;
    fr_ctx *c = w->ctx;
    const size_t slice = (size_t)w->max_batch * (size_t)c->slice_padded * sizeof(float);
    const size_t chunk = ((size_t)w->max_batch + G - 1) / G + 1;  // + the rank's status word (failure protocol, kind (2))
    if (G > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "%d ranks: the status words of a sharded step hold %d", G, FR_COMM_MAX_RANKS);
    if (!w->h_sh_status) {   // [0] sent, [1 .. G] received
        if (c->cpu) w->h_sh_status = static_cast<float *>(aligned_alloc(64, (sizeof(float) * (1 + FR_COMM_MAX_RANKS) + 63) / 64 * 64));
        else FR_HIP(hipHostMalloc((void **)&w->h_sh_status, sizeof(float) * (1 + FR_COMM_MAX_RANKS), hipHostMallocDefault));
        if (!w->h_sh_status) FR_FAIL(FR_ERR_OOM, "out of host memory (status words)");
    }
    void **bufs[] = {&w->d_slice, &w->d_gathered, (void **)&w->d_score_part, (void **)&w->d_score_all};
    for (void **b : bufs) {
        step_free(w, *b);
        *b = nullptr;
    }
    w->d_slice = step_alloc(w, slice);
    w->d_gathered = step_alloc(w, slice * G);
    w->d_score_part = static_cast<float *>(step_alloc(w, chunk * sizeof(float)));
    w->d_score_all = static_cast<float *>(step_alloc(w, chunk * G * sizeof(float)));
    if (!w->d_slice || !w->d_gathered || !w->d_score_part || !w->d_score_all) FR_FAIL(FR_ERR_OOM, "exchange buffers for %d ranks x batch %d: out of memory", G, w->max_batch);
    if (staged && !c->cpu) {   // the staged exchange of GPU contexts that share a device: one part out, G parts in, through pinned host memory
        if (w->h_stage_send) (void)hipHostFree(w->h_stage_send);
        if (w->h_stage_recv) (void)hipHostFree(w->h_stage_recv);
        w->h_stage_send = w->h_stage_recv = nullptr;
        const size_t part = slice > chunk * sizeof(float) ? slice : chunk * sizeof(float);
        FR_HIP(hipHostMalloc(&w->h_stage_send, part, hipHostMallocDefault));
        FR_HIP(hipHostMalloc(&w->h_stage_recv, part * G, hipHostMallocDefault));
    }
    w->sh_ranks = G;
    return FR_OK;
}

// kind (1) of the failure protocol: nothing has been enqueued, the communicator is untouched
static int sharded_check_args(fr_worker *w, fr_comm *comm, int batch) {
    if (!w || !comm) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    fr_ctx *c = w->ctx;
    if (comm->ctx != c) FR_FAIL(FR_ERR_INVALID, "the communicator belongs to another context");
    if (batch <= 0 || batch > w->max_batch) FR_FAIL(FR_ERR_INVALID, "batch %d outside (0, max_batch=%d]", batch, w->max_batch);
    if (!c->tables_filled) FR_FAIL(FR_ERR_STATE, "tables have not been filled or uploaded");
    if (!c->weights_set) FR_FAIL(FR_ERR_STATE, "FC weights have not been set");
    if (w->in_flight || w->sh_comm) FR_FAIL(FR_ERR_STATE, "a batch is already in flight on this worker: call fr_worker_sync first");
    return FR_OK;
}

// buffers + the request's H2D copies: a failure here is a device failure (kind (3)).  A CPU worker's "device" rows are its host rows.
static int sharded_prologue(fr_worker *w, fr_comm *comm, int batch, const int32_t **idx, const float **dense) {
    fr_ctx *c = w->ctx;
    if (!c->cpu) FR_HIP(hipSetDevice(c->device));
    int rc = shard_buffers(w, comm->n_ranks, comm->grp != nullptr);
    if (rc) return rc;
    if (c->cpu) {
        *idx = w->h_idx;
        *dense = w->h_dense;
        return FR_OK;
    }
    const size_t icols = c->model.index_mode == FR_INDEX_PER_TABLE ? (size_t)c->model.n_tables : (c->model.index_mode == FR_INDEX_PER_BANK ? (size_t)c->n_banks : 1);
    rc = step_copy(w, w->d_idx, w->h_idx, (size_t)batch * icols * sizeof(int32_t), hipMemcpyHostToDevice);
    if (!rc && c->model.dense_len) rc = step_copy(w, w->d_dense, w->h_dense, (size_t)batch * c->model.dense_len * sizeof(float), hipMemcpyHostToDevice);
    *idx = w->d_idx;
    *dense = w->d_dense;
    return rc;
}

// THE STEP, on either transport: slice gather -> all-gather of the slices -> FC chain on this rank's items -> all-gather of the score
// chunks (each followed by its rank's status word) -> every rank's pinned score buffer.  A GPU worker's fr_worker_submit_sharded calls it
// inline (every line enqueues on the worker's stream); a host-exchange worker's (CPU, or staged GPU) posts it to the worker's host stream.  *enqueued is set once the
// second collective is on its way: from then on fr_worker_sync has something to wait for, whatever this function returns.
static int sharded_step(fr_worker *w, fr_comm *comm, int batch, bool *enqueued) {
    *enqueued = false;
    const int32_t *idx = nullptr;
    const float *dense = nullptr;
    int rc = sharded_prologue(w, comm, batch, &idx, &dense);        // from here on: whatever fails aborts the communicator (kind (3))
    if (rc) return comm_fail(comm, rc);
    fr_ctx *c = w->ctx;
    const int G = comm->n_ranks, r = comm->rank;
    const int transport = c->fc_precision;  // slices travel in the chain's own operand type: fp32, bf16 (half) or e4m3 (a quarter of the bytes)
    const size_t esz = transport == FR_FC_FP32 ? 4 : (transport == FR_FC_BF16 ? 2 : 1);
    rc = fr_worker_gather_slices(w, batch, idx, dense, w->d_slice, transport);
    if (rc) return comm_fail(comm, rc);
    rc = step_all_gather(w, comm, w->d_slice, w->d_gathered, (size_t)batch * c->slice_padded * esz);
    if (rc) return comm_fail(comm, rc);
    const int base = batch / G, rem = batch % G, chunk = base + (rem ? 1 : 0);
    const int lo = r * base + (r < rem ? r : rem), n_mine = base + (r < rem ? 1 : 0);
    int fc_rc = FR_OK;
    if (n_mine > 0) {
        w->in_flight = false;  // fr_worker_fc_from_slices_lp is a public entry point with its own state checks
        fc_rc = fr_worker_fc_from_slices_lp(w, batch, lo, n_mine, w->d_gathered, transport, w->d_score_part);
        const int inject = FR_KNOB("SHARDED_INJECT_FC_FAIL", 0);   // experiments build only: the test of kind (2).  1 = every rank, 2 + q = rank q only
        if (!fc_rc && (inject == 1 || inject == 2 + r)) {
            fr_set_error("injected FC failure (FR_SHARDED_INJECT_FC_FAIL)");
            fc_rc = FR_ERR_STATE;
        }
        if (!fc_rc && w->sh_inject_fc_fail.load(std::memory_order_relaxed) > 0) {   // the product build's test hook (fleetrec_diag.h)
            w->sh_inject_fc_fail.fetch_sub(1, std::memory_order_relaxed);
            fr_set_error("injected FC failure (fr_worker_inject_fc_failure)");
            fc_rc = FR_ERR_STATE;
        }
    }
    // kind (2): the second collective is issued whatever the local FC chain returned -- the peers are already on their way into it -- but a
    // failed chain's chunk travels as NaN and its status word says so on every rank
    char fc_text[200] = "";
    if (fc_rc) {
        snprintf(fc_text, sizeof(fc_text), "%s", fr_last_error());
        if (step_fill_ff(w, w->d_score_part, (size_t)chunk * sizeof(float)) != FR_OK) return comm_fail(comm, FR_ERR_HIP);
    }
    w->h_sh_status[0] = (float)fc_rc;
    if (step_copy(w, w->d_score_part + chunk, w->h_sh_status, sizeof(float), hipMemcpyHostToDevice) != FR_OK) {
        fr_set_error("status word H2D failed");
        return comm_fail(comm, FR_ERR_HIP);
    }
    rc = step_all_gather(w, comm, w->d_score_part, w->d_score_all, ((size_t)chunk + 1) * sizeof(float));
    if (rc) return comm_fail(comm, rc);
    *enqueued = true;
    rc = FR_OK;
    rc = step_copy_status(w, G, chunk);           // all G status words ...
    for (int q = 0; q < G && rc == FR_OK; q++) {  // ... and every rank ends up with all B scores in its pinned score buffer
        const int qlo = q * base + (q < rem ? q : rem), qn = base + (q < rem ? 1 : 0);
        if (qn > 0) rc = step_copy(w, w->h_score + qlo, w->d_score_all + (size_t)q * (chunk + 1), (size_t)qn * sizeof(float), hipMemcpyDeviceToHost);
    }
    if (rc != FR_OK) return comm_fail(comm, FR_ERR_HIP);
    w->in_flight = true;
    if (fc_rc) FR_FAIL(fc_rc, "FC chain failed on this rank (its peers learn it from the status word): %s", fc_text);
    return FR_OK;
}

extern "C" int fr_worker_submit_sharded(fr_worker *w, fr_comm *comm, int batch) {
    if (comm && comm_is_broken(comm)) FR_FAIL(FR_ERR_COMM, FR_BROKEN_TEXT);
    int rc = sharded_check_args(w, comm, batch);  // kind (1): returned as it is
    if (rc) return rc;
    w->sh_comm = comm;  // fr_worker_sync waits through fr_comm_wait and reads the G status words
    comm->refs.fetch_add(1, std::memory_order_relaxed);
    if (comm->grp) {   // the host exchange (CPU worker, or staged GPU worker): the step runs behind this call on the worker's host stream -- its rendezvous blocks
        host_stream(w)->post([w, comm, batch] {
            bool enqueued = false;
            return sharded_step(w, comm, batch, &enqueued);
        });
        return FR_OK;
    }
    bool enqueued = false;
    rc = sharded_step(w, comm, batch, &enqueued);
    if (!enqueued) fr_comm_forget(w);   // kind (3) before the second collective: the communicator is aborted, there is no step to wait for
    return rc;
}

extern "C" int fr_worker_inject_fc_failure(fr_worker *w, int steps) {
    if (!w || steps < 0) FR_FAIL(FR_ERR_INVALID, "bad argument");
    w->sh_inject_fc_fail.store(steps, std::memory_order_relaxed);
    return FR_OK;
}

// fr_worker_sync of a worker with a sharded step in flight: a BOUNDED wait (failure protocol, kind (3)), then the status words (kind (2)).
static int comm_wait_step(fr_worker *w, fr_comm *comm);
int fr_comm_wait(fr_worker *w) {
    fr_comm *comm = w->sh_comm;
    if (!comm) return FR_OK;
    const int rc = comm_wait_step(w, comm);
    w->sh_comm = nullptr;
    comm_release(comm);   // the step's reference; the last one destroys a communicator that fr_comm_destroy has already let go
    return rc;
}

// a host-exchange worker's step has left its host stream: what it returned (kind (3) failures surface here; a kind (2) return is told by the status words)
static int host_step_result(fr_worker *w, fr_comm *comm) {
    HostStream *hs = static_cast<HostStream *>(w->sh_host_stream);
    if (!hs) return FR_OK;
    hs->wait();
    if (hs->rc != FR_OK && comm->broken.load(std::memory_order_acquire)) {
        fr_set_error("%s", hs->text);
        return hs->rc;
    }
    return FR_OK;
}

static int comm_wait_step(fr_worker *w, fr_comm *comm) {
    const int G = comm->n_ranks;
    const bool host = comm->grp != nullptr;
    if (!host && comm->broken.load(std::memory_order_acquire)) FR_FAIL(FR_ERR_COMM, FR_BROKEN_TEXT);
    // poll instead of a blocking wait: a peer that never arrives must not hold this rank for ever
    const int wait_ms = comm->wait_ms.load(std::memory_order_relaxed);
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spin = 0;; spin++) {
        const int q = step_query(w);
        if (q == 0) break;
        if (q < 0) return comm_fail(comm, q);
        if ((spin & 63) == 63) {
            bool give_up = step_async_error(comm);
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
            if (!give_up && ms > wait_ms) {
                fr_set_error("sharded step: the collectives did not complete within %d ms (a peer rank is missing): communicator aborted", wait_ms);
                give_up = true;
            }
            if (give_up) {
                char text[400];
                snprintf(text, sizeof(text), "%s", fr_last_error());
                (void)comm_fail(comm, FR_ERR_COMM);
                if (host) static_cast<HostStream *>(w->sh_host_stream)->wait();   // the abort has released the step: it leaves the stream before the worker's buffers may go
                fr_set_error("%s", text);
                return FR_ERR_COMM;
            }
            if (ms > 2.0) {  // long waits sleep between polls
                struct timespec nap = {0, 50000};
                nanosleep(&nap, nullptr);
            }
        }
    }
    if (host) {
        const int rc = host_step_result(w, comm);
        if (rc) return rc;
        if (comm->broken.load(std::memory_order_acquire)) FR_FAIL(FR_ERR_COMM, FR_BROKEN_TEXT);
    }
    for (int q = 0; q < G; q++)
        if (w->h_sh_status[1 + q] != 0.0f)
            FR_FAIL(FR_ERR_COMM, "shard rank %d reported a failed FC chain (status %d): the scores of its items are NaN", q, (int)w->h_sh_status[1 + q]);
    return FR_OK;
}

// fp8 chain on sharded contexts: every rank calibrates on the SAME all-gathered fp32 slices of the whole batch with the same
// (replicated) weights, so all ranks arrive at identical activation exponents without a further reduction.
extern "C" int fr_worker_calibrate_fp8_sharded(fr_worker *w, fr_comm *comm, int batch) {
    if (w && w->ctx) FR_NOT_ON_CPU(w->ctx, "fr_worker_calibrate_fp8_sharded");
    if (comm && comm_is_broken(comm)) FR_FAIL(FR_ERR_COMM, FR_BROKEN_TEXT);
    int rc = sharded_check_args(w, comm, batch);
    if (rc) return rc;
    const int32_t *idx = nullptr;
    const float *dense = nullptr;
    rc = sharded_prologue(w, comm, batch, &idx, &dense);
    if (rc) return comm_fail(comm, rc);
    fr_ctx *c = w->ctx;
    rc = fr_worker_gather_only(w, batch, idx, dense, reinterpret_cast<float *>(w->d_slice));
    if (rc) return comm_fail(comm, rc);
    rc = step_all_gather(w, comm, w->d_slice, w->d_gathered, (size_t)batch * c->slice_padded * sizeof(float));
    if (rc) return comm_fail(comm, rc);
    w->in_flight = false;
    return fr_worker_calibrate_fp8_slices(w, batch, 0, batch, reinterpret_cast<const float *>(w->d_gathered));
}
