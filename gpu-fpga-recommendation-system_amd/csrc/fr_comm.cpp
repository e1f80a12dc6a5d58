// Table-sharded mode behind the C-ABI: the exchange step on RCCL (include/fleetrec.h "table-sharded mode").
//
// Reference shape: the 3-node GPU server receives the three parts of every batch from three senders (CPU0, FPGA0, FPGA1) over
// three sockets before its GEMMs (GPU/final_network_cublasLt_3_nodes_no_FIFO_scatter/cuda_server.c:513-591).  Here the "senders" are
// the G GPUs that hold the table-ID shards: each gathers its slice of the record, ONE ncclAllGather over xGMI delivers all slices to
// everybody, each GPU runs the FC chain on its B/G items, and a second (tiny) all-gather hands every rank all B scores.
// librccl.so is opened on first use (dlopen), so unsharded users never load it; every RCCL failure surfaces as FR_ERR_COMM.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the functions are resolved through dlsym

#include <atomic>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <mutex>
#include <new>

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

#define FR_NCCL(call)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess) FR_FAIL(FR_ERR_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(r_)); \
    } while (0)

struct fr_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    fr_ctx *ctx = nullptr;
    bool broken = false;  // a collective step failed on this rank: the communicator was aborted, every later call returns FR_ERR_COMM
    int wait_ms = 60000;  // bound of fr_comm_wait: how long fr_worker_sync lets a step's collectives take before it gives the peers up
    // one reference for the handle fr_comm_init_* gave out + one per worker with a sharded step in flight through this communicator
    // (fr_worker::sh_comm): fr_comm_destroy between a submit and its sync only drops the handle's reference, the communicator itself goes
    // when the last worker has synchronised (fr_comm_wait used to dereference a freed object in that sequence)
    std::atomic<int> refs{1};
};
constexpr int FR_COMM_MAX_RANKS = 64;   // fr_worker::h_sh_status holds 1 + FR_COMM_MAX_RANKS status words

// Failure protocol of a collective step.  Three kinds of failure, three answers:
//  (1) argument / state errors found BEFORE anything was enqueued (worker busy, batch too large, tables not filled ...): returned as they
//      are, the communicator stays usable.  The ranks of a job are driven with the same arguments, so such an error is the same on every
//      rank; a caller that lets ONE rank skip a step has broken the contract and its peers run into (3).
//  (2) the local FC chain fails after the first collective: both collectives are still issued (the peers are on their way into them), the
//      rank's score chunk is poisoned with NaN and its STATUS WORD -- one float all-gathered behind every score chunk -- carries the code:
//      every rank's fr_worker_sync reads all G status words and returns FR_ERR_COMM naming the rank; nobody gets stale scores silently.
//  (3) a real failure of the device / RCCL on this rank (a HIP error, a collective that cannot be enqueued), or peers that do not arrive:
//      the rank ABORTS its communicator (comm_fail) and answers FR_ERR_COMM from then on; the peers' waits are BOUNDED (fr_comm_wait:
//      the step's stream is polled together with ncclCommGetAsyncError for at most wait_ms, then the waiting rank aborts its own
//      communicator and returns FR_ERR_COMM) -- a local ncclCommAbort does not by itself release a peer in another process.
// Verified on hardware with one-rank communicators only (the test boxes have one GPU): the cross-rank behaviour of (2) and (3) follows
// from the code, it has not run with G > 1.
static int comm_fail(fr_comm *comm, int rc) {
    if (comm && !comm->broken) {
        comm->broken = true;
        if (comm->comm && g_rccl.CommAbort) {
            (void)g_rccl.CommAbort(comm->comm);
            comm->comm = nullptr;  // aborted communicators are already destroyed
        }
    }
    return rc;
}
#define FR_NCCL_OR_ABORT(comm_, call)                                                          \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            fr_set_error("%s failed: %s", #call, g_rccl.GetErrorString(r_));                   \
            return comm_fail(comm_, FR_ERR_COMM);                                              \
        }                                                                                      \
    } while (0)

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
    FR_NOT_ON_CPU(ctx, "the RCCL exchange (fr_comm_*)");
    if (ctx->n_shards > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "%d shards: the RCCL exchange serves at most %d ranks", ctx->n_shards, FR_COMM_MAX_RANKS);
    if (ctx->model.layout != FR_LAYOUT_SEMANTIC) FR_FAIL(FR_ERR_STATE, "the sharded exchange needs the SEMANTIC layout");
    return FR_OK;
}

extern "C" int fr_comm_init_rank(fr_ctx *ctx, const void *id128, fr_comm **out) {
    if (!out || !id128) FR_FAIL(FR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    int rc = comm_check_ctx(ctx);
    if (rc) return rc;
    rc = rccl_load();
    if (rc) return rc;
    FR_HIP(hipSetDevice(ctx->device));
    fr_comm *c = new (std::nothrow) fr_comm();
    if (!c) FR_FAIL(FR_ERR_OOM, "out of host memory");
    c->rank = ctx->shard_rank;
    c->n_ranks = ctx->n_shards;
    c->ctx = ctx;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, c->n_ranks, id, c->rank);
    if (r != ncclSuccess) {
        delete c;
        FR_FAIL(FR_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", ctx->shard_rank, ctx->n_shards, g_rccl.GetErrorString(r));
    }
    *out = c;
    return FR_OK;
}

extern "C" int fr_comm_init_all(fr_ctx *const *ctxs, int n, fr_comm **out) {
    if (!ctxs || !out || n < 1 || n > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "bad argument");
    for (int r = 0; r < n; r++) out[r] = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    int devs[64];
    for (int r = 0; r < n; r++) {
        rc = comm_check_ctx(ctxs[r]);
        if (rc) return rc;
        if (ctxs[r]->n_shards != n || ctxs[r]->shard_rank != r) FR_FAIL(FR_ERR_INVALID, "ctxs[%d] is shard %d of %d, expected %d of %d", r, ctxs[r]->shard_rank, ctxs[r]->n_shards, r, n);
        devs[r] = ctxs[r]->device;
        for (int q = 0; q < r; q++)
            if (devs[q] == devs[r]) FR_FAIL(FR_ERR_INVALID, "shards %d and %d share device %d: RCCL needs one device per rank", q, r, devs[r]);
    }
    ncclComm_t comms[64];
    FR_NCCL(g_rccl.CommInitAll(comms, n, devs));
    for (int r = 0; r < n; r++) {
        fr_comm *c = new (std::nothrow) fr_comm();
        if (!c) {
            for (int q = 0; q < r; q++) {
                fr_comm_destroy(out[q]);
                out[q] = nullptr;
            }
            for (int q = r; q < n; q++) (void)g_rccl.CommDestroy(comms[q]);
            FR_FAIL(FR_ERR_OOM, "out of host memory");
        }
        c->comm = comms[r];
        c->rank = r;
        c->n_ranks = n;
        c->ctx = ctxs[r];
        out[r] = c;
    }
    return FR_OK;
}

extern "C" int fr_comm_set_wait_ms(fr_comm *c, int wait_ms) {
    if (!c || wait_ms < 1) FR_FAIL(FR_ERR_INVALID, "bad argument");
    c->wait_ms = wait_ms;
    return FR_OK;
}

static void comm_release(fr_comm *c) {
    if (c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    if (c->comm && g_rccl_ok) {
        if (c->ctx) (void)hipSetDevice(c->ctx->device);
        (void)g_rccl.CommDestroy(c->comm);
    }
    delete c;
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

// exchange buffers of a worker: slice [max_batch][F] and gathered [G][max_batch][F] sized for fp32 elements, score chunks
static int shard_buffers(fr_worker *w, int G) {
    if (w->sh_ranks == G && w->d_slice) return FR_OK;
    fr_ctx *c = w->ctx;
    const size_t slice = (size_t)w->max_batch * (size_t)c->slice_padded * sizeof(float);
    const size_t chunk = ((size_t)w->max_batch + G - 1) / G + 1;  // + the rank's status word (failure protocol, kind (2))
    if (G > FR_COMM_MAX_RANKS) FR_FAIL(FR_ERR_INVALID, "%d ranks: the status words of a sharded step hold %d", G, FR_COMM_MAX_RANKS);
    if (!w->h_sh_status) FR_HIP(hipHostMalloc((void **)&w->h_sh_status, sizeof(float) * (1 + FR_COMM_MAX_RANKS), hipHostMallocDefault));  // [0] sent, [1 .. G] received
    void **bufs[] = {&w->d_slice, &w->d_gathered, (void **)&w->d_score_part, (void **)&w->d_score_all};
    for (void **b : bufs)
        if (*b) {
            (void)hipFree(*b);
            *b = nullptr;
        }
    FR_HIP(hipMalloc(&w->d_slice, slice));
    FR_HIP(hipMalloc(&w->d_gathered, slice * G));
    FR_HIP(hipMalloc((void **)&w->d_score_part, chunk * sizeof(float)));
    FR_HIP(hipMalloc((void **)&w->d_score_all, chunk * G * sizeof(float)));
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

// buffers + H2D copies: a failure here is a device failure (kind (3))
static int sharded_prologue(fr_worker *w, fr_comm *comm, int batch) {
    fr_ctx *c = w->ctx;
    FR_HIP(hipSetDevice(c->device));
    int rc = shard_buffers(w, comm->n_ranks);
    if (rc) return rc;
    const size_t icols = c->model.index_mode == FR_INDEX_PER_TABLE ? (size_t)c->model.n_tables : (c->model.index_mode == FR_INDEX_PER_BANK ? (size_t)c->n_banks : 1);
    FR_HIP(hipMemcpyAsync(w->d_idx, w->h_idx, (size_t)batch * icols * sizeof(int32_t), hipMemcpyHostToDevice, w->stream));
    if (c->model.dense_len)
        FR_HIP(hipMemcpyAsync(w->d_dense, w->h_dense, (size_t)batch * c->model.dense_len * sizeof(float), hipMemcpyHostToDevice, w->stream));
    return FR_OK;
}

extern "C" int fr_worker_submit_sharded(fr_worker *w, fr_comm *comm, int batch) {
    if (comm && comm->broken) FR_FAIL(FR_ERR_COMM, "the communicator was aborted by an earlier failure on this rank");
    int rc = sharded_check_args(w, comm, batch);  // kind (1): returned as it is
    if (rc) return rc;
    rc = sharded_prologue(w, comm, batch);        // from here on: whatever fails aborts the communicator (kind (3))
    if (rc) return comm_fail(comm, rc);
    fr_ctx *c = w->ctx;
    const int G = comm->n_ranks, r = comm->rank;
    const int transport = c->fc_precision;  // slices travel in the chain's own operand type: fp32, bf16 (half) or e4m3 (a quarter of the bytes)
    const size_t esz = transport == FR_FC_FP32 ? 4 : (transport == FR_FC_BF16 ? 2 : 1);
    rc = fr_worker_gather_slices(w, batch, w->d_idx, w->d_dense, w->d_slice, transport);
    if (rc) return comm_fail(comm, rc);
    FR_NCCL_OR_ABORT(comm, g_rccl.AllGather(w->d_slice, w->d_gathered, (size_t)batch * c->slice_padded * esz, ncclChar, comm->comm, w->stream));
    const int base = batch / G, rem = batch % G, chunk = base + (rem ? 1 : 0);
    const int lo = r * base + (r < rem ? r : rem), n_mine = base + (r < rem ? 1 : 0);
    int fc_rc = FR_OK;
    if (n_mine > 0) {
        w->in_flight = false;  // fr_worker_fc_from_slices_lp is a public entry point with its own state checks
        fc_rc = fr_worker_fc_from_slices_lp(w, batch, lo, n_mine, w->d_gathered, transport, w->d_score_part);
        if (!fc_rc && FR_KNOB("SHARDED_INJECT_FC_FAIL", 0)) {  // experiments build only: the test of kind (2)
            fr_set_error("injected FC failure (FR_SHARDED_INJECT_FC_FAIL)");
            fc_rc = FR_ERR_STATE;
        }
    }
    // kind (2): the second collective is issued whatever the local FC chain returned -- the peers are already on their way into it -- but a
    // failed chain's chunk travels as NaN and its status word says so on every rank
    char fc_text[200] = "";
    if (fc_rc) {
        snprintf(fc_text, sizeof(fc_text), "%s", fr_last_error());
        if (hipMemsetAsync(w->d_score_part, 0xFF, (size_t)chunk * sizeof(float), w->stream) != hipSuccess) return comm_fail(comm, FR_ERR_HIP);
    }
    w->h_sh_status[0] = (float)fc_rc;
    if (hipMemcpyAsync(w->d_score_part + chunk, w->h_sh_status, sizeof(float), hipMemcpyHostToDevice, w->stream) != hipSuccess) {
        fr_set_error("status word H2D failed");
        return comm_fail(comm, FR_ERR_HIP);
    }
    FR_NCCL_OR_ABORT(comm, g_rccl.AllGather(w->d_score_part, w->d_score_all, (size_t)chunk + 1, ncclFloat, comm->comm, w->stream));
    hipError_t he = hipMemcpy2DAsync(w->h_sh_status + 1, sizeof(float), w->d_score_all + chunk, ((size_t)chunk + 1) * sizeof(float), sizeof(float), (size_t)G,
                                     hipMemcpyDeviceToHost, w->stream);
    for (int q = 0; q < G && he == hipSuccess; q++) {  // every rank ends up with all B scores in its pinned score buffer
        const int qlo = q * base + (q < rem ? q : rem), qn = base + (q < rem ? 1 : 0);
        if (qn > 0) he = hipMemcpyAsync(w->h_score + qlo, w->d_score_all + (size_t)q * (chunk + 1), (size_t)qn * sizeof(float), hipMemcpyDeviceToHost, w->stream);
    }
    if (he != hipSuccess) {
        fr_set_error("score D2H failed: %s", hipGetErrorString(he));
        return comm_fail(comm, FR_ERR_HIP);
    }
    w->sh_comm = comm;  // fr_worker_sync waits through fr_comm_wait and reads the G status words
    comm->refs.fetch_add(1, std::memory_order_relaxed);
    w->in_flight = true;
    if (fc_rc) FR_FAIL(fc_rc, "FC chain failed on this rank (its peers learn it from the status word): %s", fc_text);
    return FR_OK;
}

// fr_worker_sync of a worker with a sharded step in flight: a BOUNDED wait (failure protocol, kind (3)), then the status words (kind (2)).
static int comm_wait_step(fr_worker *w, fr_comm *comm);
int fr_comm_wait(fr_worker *w) {
    fr_comm *comm = w->sh_comm;
    w->sh_comm = nullptr;
    if (!comm) return FR_OK;
    const int rc = comm_wait_step(w, comm);
    comm_release(comm);   // the step's reference; the last one destroys a communicator that fr_comm_destroy has already let go
    return rc;
}

static int comm_wait_step(fr_worker *w, fr_comm *comm) {
    const int G = comm->n_ranks;
    if (comm->broken) FR_FAIL(FR_ERR_COMM, "the communicator was aborted by an earlier failure on this rank");
    // poll instead of hipStreamSynchronize: a peer that never arrives must not hold this rank for ever
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spin = 0;; spin++) {
        hipError_t q = hipStreamQuery(w->stream);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) {
            fr_set_error("sharded step: stream failed: %s", hipGetErrorString(q));
            return comm_fail(comm, FR_ERR_HIP);
        }
        if ((spin & 63) == 63) {
            ncclResult_t ar = ncclSuccess;
            if (g_rccl.CommGetAsyncError && comm->comm && g_rccl.CommGetAsyncError(comm->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                fr_set_error("sharded step: RCCL reported an asynchronous error: %s", g_rccl.GetErrorString(ar));
                return comm_fail(comm, FR_ERR_COMM);
            }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
            if (ms > comm->wait_ms) {
                fr_set_error("sharded step: the collectives did not complete within %d ms (a peer rank is missing): communicator aborted", comm->wait_ms);
                return comm_fail(comm, FR_ERR_COMM);
            }
            if (ms > 2.0) {  // long waits sleep between polls
                struct timespec nap = {0, 50000};
                nanosleep(&nap, nullptr);
            }
        }
    }
    for (int q = 0; q < G; q++)
        if (w->h_sh_status[1 + q] != 0.0f)
            FR_FAIL(FR_ERR_COMM, "shard rank %d reported a failed FC chain (status %d): the scores of its items are NaN", q, (int)w->h_sh_status[1 + q]);
    return FR_OK;
}

// fp8 chain on sharded contexts: every rank calibrates on the SAME all-gathered fp32 slices of the whole batch with the same
// (replicated) weights, so all ranks arrive at identical activation exponents without a further reduction.
extern "C" int fr_worker_calibrate_fp8_sharded(fr_worker *w, fr_comm *comm, int batch) {
    if (comm && comm->broken) FR_FAIL(FR_ERR_COMM, "the communicator was aborted by an earlier failure on this rank");
    int rc = sharded_check_args(w, comm, batch);
    if (rc) return rc;
    rc = sharded_prologue(w, comm, batch);
    if (rc) return comm_fail(comm, rc);
    fr_ctx *c = w->ctx;
    rc = fr_worker_gather_only(w, batch, w->d_idx, w->d_dense, reinterpret_cast<float *>(w->d_slice));
    if (rc) return comm_fail(comm, rc);
    FR_NCCL_OR_ABORT(comm, g_rccl.AllGather(w->d_slice, w->d_gathered, (size_t)batch * c->slice_padded * sizeof(float), ncclChar, comm->comm, w->stream));
    w->in_flight = false;
    return fr_worker_calibrate_fp8_slices(w, batch, 0, batch, reinterpret_cast<const float *>(w->d_gathered));
}
