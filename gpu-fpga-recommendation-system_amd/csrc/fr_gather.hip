// gather_pack kernels: the record-producing gather (fr_worker_gather_only, BLOCKED layout, sharded slices).
#include "fr_device.h"

// ---------------------------------------------------------------------------------------------------
// gather_pack: HBM-bound, no MFMA.
//
// One thread owns one 16-byte word position of the record (its FrWordDesc stays in registers) and
// walks ITEMS items: index load -> 16-byte row-word load -> 16-byte record store.  Consecutive lanes
// own consecutive record words, so record stores are fully coalesced (1 KiB per wave-instruction)
// and a dim-d row is read by d/4 adjacent lanes as one contiguous d*4-byte segment.
// All ITEMS index loads are issued before the row loads, and all row loads before the stores, so a
// wave keeps ITEMS x 1 KiB of gathers in flight.
// ---------------------------------------------------------------------------------------------------
// TP = what a record word is stored as: 0 fp32 (16 bytes, a bit copy), 1 bf16 (8 bytes, RNE), 2 e4m3 (4 bytes, x scale, saturated).
// The low-precision forms are the sharded mode's slice TRANSPORT formats: half / a quarter of the all-gather bytes, and exactly the
// values the bf16 / fp8 chain would have made of the fp32 slice on arrival.
// AUX = the cache-policy bits of a buffer store (gfx950: 1 sc0, 2 nt, 16 sc1; 16 = write-through): only the fp32 record stores of
// gather_pack_xcd_kernel use it, through a resource over `out` (the launcher keeps AUX = 0 when the records exceed 4 GiB)
template <int TP, int AUX = 0>
__device__ __forceinline__ void store_word(void *out, size_t word, const uint4 &v, float scale) {
    if constexpr (TP == 0 && AUX != 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0xffffffffu, 0x00020000);
        u32x4_t q;
        q.x = v.x, q.y = v.y, q.z = v.z, q.w = v.w;
        __builtin_amdgcn_raw_buffer_store_b128(q, rs, (unsigned)(word * 16), 0, AUX);
    } else if constexpr (TP == 0) {
        reinterpret_cast<uint4 *>(out)[word] = v;
    } else if constexpr (TP == 1) {
        reinterpret_cast<uint2 *>(out)[word] = make_uint2(pack_bf16x2(__uint_as_float(v.x), __uint_as_float(v.y)), pack_bf16x2(__uint_as_float(v.z), __uint_as_float(v.w)));
    } else {
        reinterpret_cast<uint32_t *>(out)[word] = pack_fp8_word(v, scale);
    }
}

template <int ITEMS, int TP>
__global__ void __launch_bounds__(256) gather_pack_kernel(const FrWordDesc *__restrict__ words, int n_words,
                                                          const int32_t *__restrict__ idx, int idx_stride,
                                                          const float *__restrict__ dense, void *__restrict__ out,
                                                          int batch, int *__restrict__ err_flag, float scale) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    const uint4 d0 = reinterpret_cast<const uint4 *>(words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(words)[2 * w + 1];
    const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
    const uint32_t stride = d0.z, idx_col = d0.w;
    const uint32_t rows = d1.x, dst_off = d1.y, dst_stride = d1.z, dst_blk = d1.w;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const char *base = is_dense ? reinterpret_cast<const char *>(dense) + src : reinterpret_cast<const char *>(src);
    const int b0 = blockIdx.y * ITEMS;

    uint32_t id[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int b = b0 + i;
        id[i] = 0;
        if (b < batch) id[i] = is_dense ? (uint32_t)b : (uint32_t)idx[(size_t)b * idx_stride + idx_col];
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        if (!is_dense && id[i] >= rows) {  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
            bad = true;
            id[i] = 0;
        }
    }
    uint4 v[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
    const size_t blk = (size_t)dst_blk * (size_t)batch + dst_off;
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const int b = b0 + i;
        if (b < batch) store_word<TP>(out, blk + (size_t)b * dst_stride, v[i], scale);
    }
    if (bad) atomicOr_system(err_flag, 1);  // pinned host word; error path only
}

// XCD-partitioned form for wide records and large batches: workgroup b serves XCD group b % 8 (workgroups are dealt
// round-robin over the 8 XCDs; a speed assumption only), and each group owns a fixed contiguous run of the record words for
// ALL items (FrGatherGroups: cut on source-row boundaries, so no table row / bank row is fetched by two XCDs).  Every table is then
// touched from ONE XCD only, so the 8 x 4 MiB L2s cache 8 different table sets instead of 8 copies of the same hottest 4 MiB --
// rows served by L2 cost ~5 cycles/CU instead of ~12 from the fabric (profiles/archive/r01_experiments.md, ta_cost2).
template <int ITEMS, int TP, int AUX = 0>
__global__ void __launch_bounds__(256) gather_pack_xcd_kernel(const FrWordDesc *__restrict__ words, const FrGatherGroups groups,
                                                              const int32_t *__restrict__ idx, int idx_stride,
                                                              const float *__restrict__ dense, void *__restrict__ out,
                                                              int batch, int *__restrict__ err_flag, float scale, int n_chunks) {
    const int group = blockIdx.x & 7;
    const int w0 = groups.start[group];
    if ((int)threadIdx.x >= groups.start[group + 1] - w0) return;
    const int w = w0 + threadIdx.x;
    const uint4 d0 = reinterpret_cast<const uint4 *>(words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(words)[2 * w + 1];
    const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
    const uint32_t stride = d0.z, idx_col = d0.w;
    const uint32_t rows = d1.x, dst_off = d1.y, dst_stride = d1.z, dst_blk = d1.w;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const char *base = is_dense ? reinterpret_cast<const char *>(dense) + src : reinterpret_cast<const char *>(src);
    const size_t blk = (size_t)dst_blk * (size_t)batch + dst_off;
    bool bad = false;
    // a thread keeps its descriptor and walks chunks of ITEMS items: chunk = blockIdx.x / 8, stepping by the grid's chunk count
    for (int chunk = blockIdx.x >> 3; chunk < n_chunks; chunk += gridDim.x >> 3) {
        const int b0 = chunk * ITEMS;
        uint32_t id[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int b = b0 + i;
            id[i] = 0;
            if (b < batch) id[i] = is_dense ? (uint32_t)b : (uint32_t)idx[(size_t)b * idx_stride + idx_col];
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            if (!is_dense && id[i] >= rows) {
                bad = true;
                id[i] = 0;
            }
        }
        uint4 v[ITEMS];
#pragma unroll
        for (int i = 0; i < ITEMS; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int b = b0 + i;
            if (b < batch) store_word<TP, AUX>(out, blk + (size_t)b * dst_stride, v[i], scale);
        }
    }
    if (bad) atomicOr_system(err_flag, 1);
}

// Software-pipelined form of gather_pack_xcd_kernel: a workgroup keeps its descriptors and walks NSTEP chunks of ITEMS items
// (chunks j, j + cs, j + 2 cs, ... of its XCD group) with the index loads TWO chunks ahead and the row loads ONE chunk ahead of the
// record stores.  A chunk's dependent chain (descriptor -> index -> row -> store) is then paid once per workgroup of NSTEP chunks, and
// every wave mixes loads and stores from its first stored chunk on (the one-chunk-per-workgroup form starts with a read-only phase
// of every resident workgroup at once).  Straight-line code (NSTEP is a template parameter), so every s_waitcnt is a counted one;
// items past the batch cost no branch: their index loads and record stores go through buffer resources whose bounds drop them.
// Needs batch * idx_stride * 4 and the record bytes below 4000 MiB (the launcher falls back to gather_pack_xcd_kernel otherwise).
// LEAD: one index load per RUN of adjacent lanes that share an index column (a bank row of a bank-interleaved context is 7-16 record
// words, a table row 1-8): only the run's first lane loads (the others are masked off the instruction), everybody takes the value
// from that lane through ds_bpermute -- the texture path sees an eighth to a half of the index lanes.
template <int ITEMS, int NSTEP, int TP, int AUX, bool LEAD>
__global__ void __launch_bounds__(256) gather_pack_stream_kernel(const FrWordDesc *__restrict__ words, const FrGatherGroups groups,
                                                                 const int32_t *__restrict__ idx, int idx_stride,
                                                                 const float *__restrict__ dense, void *__restrict__ out,
                                                                 int batch, int *__restrict__ err_flag, float scale, unsigned out_bytes) {
    const int group = blockIdx.x & 7;
    const int w0 = groups.start[group];
    if ((int)threadIdx.x >= groups.start[group + 1] - w0) return;
    const int w = w0 + threadIdx.x;
    const uint4 d0 = reinterpret_cast<const uint4 *>(words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(words)[2 * w + 1];
    const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
    const uint32_t stride = d0.z, idx_col = d0.w;
    const uint32_t rows = d1.x, dst_off = d1.y, dst_stride = d1.z, dst_blk = d1.w;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const uint64_t base = (is_dense ? (uint64_t)reinterpret_cast<uintptr_t>(dense) : 0ull) + src;
    constexpr unsigned ESZ = TP == 0 ? 16u : TP == 1 ? 8u : 4u;   // bytes a record word is stored as
    const unsigned blk = (dst_blk * (unsigned)batch + dst_off) * ESZ;
    const unsigned ostride = dst_stride * ESZ;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(idx), 0, (unsigned)batch * (unsigned)idx_stride * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, out_bytes, 0x00020000);
    const unsigned icol = is_dense ? 0u : idx_col * 4u;
    bool bad = false;
    const int cs = gridDim.x >> 3, c0 = blockIdx.x >> 3;
    uint32_t id[NSTEP][ITEMS];
    uint4 v[NSTEP][ITEMS];
    // run leaders (LEAD): lane l leads when it is the wave's first lane or its index column differs from lane l - 1's
    bool leader = true;
    int lead_byte = 0;   // 4 x the lane that holds this lane's index value
    if constexpr (LEAD) {
        const int lane = threadIdx.x & 63;
        const unsigned prev = (unsigned)__shfl_up((int)idx_col, 1);
        leader = lane == 0 || prev != idx_col;
        const unsigned long long lm = __ballot(leader);
        lead_byte = 4 * (63 - __builtin_clzll(lm & ((2ull << lane) - 1ull)));
    }
    auto load_idx = [&](int st) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const unsigned b = (unsigned)(c0 + st * cs) * ITEMS + i;
            // an item past the batch reads past the resource: 0, no branch (dense words do not use the value)
            if constexpr (LEAD) {
                id[st][i] = 0;
                if (leader) id[st][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, b * (unsigned)idx_stride * 4u + icol, 0, 0);
            } else {
                id[st][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, b * (unsigned)idx_stride * 4u + icol, 0, 0);
            }
        }
    };
    auto load_rows = [&](int st) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const unsigned b = (unsigned)(c0 + st * cs) * ITEMS + i;
            uint32_t r = id[st][i];
            if constexpr (LEAD) r = (uint32_t)__builtin_amdgcn_ds_bpermute(lead_byte, (int)r);   // the run leader's value
            const bool oob = !is_dense & (r >= rows);  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
            bad |= oob;
            r = oob ? 0u : r;
            r = is_dense ? (b < (unsigned)batch ? b : 0u) : r;
            typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;   // a global_load (not a flat one: no lgkmcnt, no aperture check)
            const u32x4_t q = *(gptr_t)(base + (uint64_t)r * stride);
            v[st][i] = make_uint4(q.x, q.y, q.z, q.w);
        }
    };
    auto store_rows = [&](int st) {
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const unsigned b = (unsigned)(c0 + st * cs) * ITEMS + i;
            const unsigned off = b < (unsigned)batch ? blk + b * ostride : out_bytes;   // past the batch: offset == num_records, dropped by the resource's bounds (no 32-bit wrap)
            const uint4 &q = v[st][i];
            if constexpr (TP == 0) {
                u32x4_t x;
                x.x = q.x, x.y = q.y, x.z = q.z, x.w = q.w;
                __builtin_amdgcn_raw_buffer_store_b128(x, rs_out, off, 0, AUX);
            } else if constexpr (TP == 1) {
                typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                u32x2_t x;
                x.x = pack_bf16x2(__uint_as_float(q.x), __uint_as_float(q.y)), x.y = pack_bf16x2(__uint_as_float(q.z), __uint_as_float(q.w));
                __builtin_amdgcn_raw_buffer_store_b64(x, rs_out, off, 0, AUX);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(pack_fp8_word(q, scale), rs_out, off, 0, AUX);
            }
        }
    };
    load_idx(0);
    if constexpr (NSTEP > 1) load_idx(1);
    load_rows(0);
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
        if (st + 2 < NSTEP) load_idx(st + 2);
        if (st + 1 < NSTEP) load_rows(st + 1);
        store_rows(st);
    }
    if (bad) atomicOr_system(err_flag, 1);
}

// stream_form: 0 = the one-chunk-per-workgroup kernel (gather_pack_xcd_kernel; also what records or index buffers of >= 4000 MiB take),
// n > 0 = gather_pack_stream_kernel with n chunks per workgroup (the product launches 2)
template <int ITEMS, int TP>
static int gather_launch_xcd(const FrWordDesc *words, const FrGatherGroups &groups, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch,
                             int *err_flag, float scale, hipStream_t s, int n_words, int out_words, int stream_form) {
    const size_t n_rec_bytes = (size_t)out_words * 16;
    const int bx = ((groups.max_words + 63) / 64) * 64;
    const int n_chunks = (batch + ITEMS - 1) / ITEMS;
    const int per_wg_knob = FR_KNOB("GATHER_LOOP", 1);  // experiment knob: chunks a workgroup of the one-chunk form walks
    const int per_wg = per_wg_knob > 0 ? per_wg_knob : 1;
    dim3 grid(8 * ((n_chunks + per_wg - 1) / per_wg));
    (void)n_rec_bytes;
    if constexpr (ITEMS == 4 || ITEMS == 2) {
        // default: the software-pipelined form, 2 chunks per workgroup, write-through record stores while the records fit the Infinity
        // Cache with room to spare (profiles/archive/r02_gather_stream_sweep.txt: Model-C batch 4096 per-bank 25.2 -> 22.0 us; beyond ~200 MB of
        // records write-back stores are as fast or 1-2 % faster).  Experiments build only: FR_GATHER_STREAM = chunks per workgroup,
        // FR_GATHER_STORE = 0 write-back / 16 write-through, both read per launch.
        const int nstep = FR_KNOB("GATHER_STREAM", stream_form);
        const size_t esz = TP == 0 ? 16 : TP == 1 ? 8 : 4;
        const size_t out_bytes = (size_t)batch * (size_t)out_words * esz, idx_bytes = (size_t)batch * (size_t)idx_stride * 4;
        if (nstep > 0 && out_bytes < ((size_t)4000 << 20) && idx_bytes < ((size_t)4000 << 20)) {  // 32-bit resource offsets, with room for the chunk past the batch
            const int st_knob = FR_KNOB("GATHER_STORE", -1);
            const bool wt = st_knob >= 0 ? st_knob == 16 : out_bytes <= ((size_t)200 << 20);
            const bool lead = FR_KNOB("GATHER_LEAD", 0) != 0;   // experiments build: one index load per run of lanes sharing an index column (slower: 22.3 -> 22.9 us)
#ifdef FR_EXPERIMENTS   /* the LEAD instantiations exist in the experiments build only (measured slower: profiles/archive/r03_experiments.md) */
#define FR_G_LEAD(NS)                                                                                                                       \
            if (wt) gather_pack_stream_kernel<ITEMS, NS, TP, 16, true><<<g2, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, (unsigned)out_bytes); \
            else gather_pack_stream_kernel<ITEMS, NS, TP, 0, true><<<g2, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, (unsigned)out_bytes);
#else
#define FR_G_LEAD(NS)
#endif
#define FR_G_STREAM(NS)                                                                                                                     \
    case NS: {                                                                                                                              \
        dim3 g2(8 * ((n_chunks + NS - 1) / NS));                                                                                            \
        if (lead) {                                                                                                                         \
            FR_G_LEAD(NS)                                                                                                                   \
        } else {                                                                                                                            \
            if (wt) gather_pack_stream_kernel<ITEMS, NS, TP, 16, false><<<g2, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, (unsigned)out_bytes); \
            else gather_pack_stream_kernel<ITEMS, NS, TP, 0, false><<<g2, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, (unsigned)out_bytes);    \
        }                                                                                                                                   \
        KCHECK();                                                                                                                           \
        fr_note_kernel("gather_pack_stream_kernel<%d, %d, %d, %d, %s>", ITEMS, NS, TP, wt ? 16 : 0, lead ? "true" : "false");               \
        return FR_OK;                                                                                                                       \
    }
            switch (nstep) {
                FR_G_STREAM(2)
#ifdef FR_EXPERIMENTS
                FR_G_STREAM(1)
                FR_G_STREAM(4)
                FR_G_STREAM(8)
#endif
                default: break;
            }
#undef FR_G_STREAM
#undef FR_G_LEAD
        }
    }
#ifdef FR_EXPERIMENTS
    if constexpr (ITEMS == 4 && TP == 0) {
        const int st = FR_KNOB("GATHER_STORE", 0);  // experiment knob: cache policy of the record stores (0 plain, 16 sc1, 2 nt) of the one-chunk form
        if (st && (size_t)batch * (size_t)n_rec_bytes < ((size_t)1 << 32)) {
#define FR_G_ST(A) case A: gather_pack_xcd_kernel<4, 0, A><<<grid, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, n_chunks); KCHECK(); return FR_OK;
            switch (st) {
                FR_G_ST(16)
                FR_G_ST(2)
                default: break;
            }
#undef FR_G_ST
        }
    }
#endif
    gather_pack_xcd_kernel<ITEMS, TP><<<grid, dim3(bx), 0, s>>>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, n_chunks);
    KCHECK();
    fr_note_kernel("gather_pack_xcd_kernel<%d, %d>", ITEMS, TP);
    return FR_OK;
}

template <int TP>
static int gather_launch(const FrWordDesc *words, int n_words, const FrGatherGroups &planned, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch, int *err_flag,
                         float scale, hipStream_t s, int out_words, int stream_form) {
    // experiments build only (tools/experiments/gather_sweep.py), read per launch: FR_GATHER_XCD = 0 / 1 forces the kernel form,
    // FR_GATHER_ITEMS = items per thread of the XCD-partitioned form, FR_GATHER_UNIFORM_GROUPS = n_words / 8 words per XCD group
    const int force = FR_KNOB("GATHER_XCD", -1);
    const bool xcd = force >= 0 ? force != 0 : (n_words >= 512 && batch >= 1024);
    if (xcd) {
        FrGatherGroups groups = planned;
        if (groups.max_words <= 0 || FR_KNOB("GATHER_UNIFORM_GROUPS", 0)) {  // no plan (or the experiment knob): n_words / 8 each
            const int wpg = (n_words + 7) / 8;
            for (int g = 0; g <= 8; g++) groups.start[g] = g * wpg < n_words ? g * wpg : n_words;
            groups.max_words = wpg;
        }
        if (groups.max_words <= 256) {
            switch (FR_KNOB("GATHER_ITEMS", 4)) {  // 4 items per thread: fastest in the r02 sweep (profiles/archive/r02_gather_sweep.txt)
#ifdef FR_EXPERIMENTS
                case 1: return gather_launch_xcd<1, TP>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, s, n_words, out_words, stream_form);
                case 2: return gather_launch_xcd<2, TP>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, s, n_words, out_words, stream_form);
                case 8: return gather_launch_xcd<8, TP>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, s, n_words, out_words, stream_form);
                case 16: return gather_launch_xcd<16, TP>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, s, n_words, out_words, stream_form);
#endif
                default: return gather_launch_xcd<4, TP>(words, groups, idx, idx_stride, dense, out, batch, err_flag, scale, s, n_words, out_words, stream_form);
            }
        }
    }
    // block width: whole waves, at most 256 lanes
    int bx = n_words >= 256 ? 256 : ((n_words + 63) / 64) * 64;
    dim3 block(bx);
    if (batch >= 2048) {
        constexpr int ITEMS = 8;
        dim3 grid((n_words + bx - 1) / bx, (batch + ITEMS - 1) / ITEMS);
        gather_pack_kernel<ITEMS, TP><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, scale);
        fr_note_kernel("gather_pack_kernel<%d, %d>", ITEMS, TP);
    } else {
        constexpr int ITEMS = 4;
        dim3 grid((n_words + bx - 1) / bx, (batch + ITEMS - 1) / ITEMS);
        gather_pack_kernel<ITEMS, TP><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, scale);
        fr_note_kernel("gather_pack_kernel<%d, %d>", ITEMS, TP);
    }
    KCHECK();
    return FR_OK;
}

// transport: FR_FC_FP32 (fp32 records, the reference's wire format), FR_FC_BF16 or FR_FC_FP8 (slice transport of the sharded mode)
int frk_gather(const FrWordDesc *words, int n_words, const FrGatherGroups &groups, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch, int *err_flag,
               int transport, int e_x, hipStream_t s, int out_words, bool one_chunk) {
    // out_words = 16-byte words per item of the DESTINATION (the record, or a shard's padded slice: >= n_words): the bounds of the
    // software-pipelined kernel's store resource.  one_chunk: the FR_GATHER_WORD_MAJOR_ONE_CHUNK variant (gather_pack_xcd_kernel).
    if (n_words <= 0 || batch <= 0) return FR_OK;
    if (out_words < n_words) out_words = n_words;
    const int stream_form = one_chunk ? 0 : 2;
    if (transport == FR_FC_BF16) return gather_launch<1>(words, n_words, groups, idx, idx_stride, dense, out, batch, err_flag, 1.0f, s, out_words, stream_form);
    if (transport == FR_FC_FP8) return gather_launch<2>(words, n_words, groups, idx, idx_stride, dense, out, batch, err_flag, ldexpf(1.0f, e_x), s, out_words, stream_form);
    return gather_launch<0>(words, n_words, groups, idx, idx_stride, dense, out, batch, err_flag, 1.0f, s, out_words, stream_form);
}

// ---------------------------------------------------------------------------------------------------
// gather_tile_kernel<DEDUP>: the item-tile form of the record-producing gather -- LDS-staged row packing, with (DEDUP) or without a
// wave-level merge of duplicate lookups (north_star: "LDS-staged row packing and wavefront ballot/shuffle for index dedup").
//
// A 256-thread workgroup owns 64 items x one chunk of <= 64 consecutive record words.  A wave-instruction ("pass") covers 64 / w items
// x the w 16-byte words of one source row each, so every row is one contiguous request of w adjacent lanes and a wave sees up to 64
// lookups of the SAME table at once -- which is what makes duplicates visible to it (in gather_pack_kernel a wave holds one item's
// words of many tables: no two lanes ever want the same row).  DEDUP: the lanes of a pass agree on one leader per distinct index
// through a 128-slot LDS hash of the index (last writer wins; a slot collision between different indices only costs a missed
// merge, never a wrong row: the leader's index is compared through a shuffle), only leaders load, the others take the leader's
// registers by ds_bpermute (__shfl); __ballot counts the merged lookups when a counter is supplied.
// All index loads of the wave's passes are issued first, then all row loads, then the rows go to the LDS tile; after one barrier the
// tile leaves as whole 1 KiB record pieces per item (fully coalesced, like gather_pack_kernel's stores).
// ---------------------------------------------------------------------------------------------------
constexpr int FR_TILE_LD = FR_TILE_WORDS + 1;     // LDS tile row stride in 16-byte words (+1: conflict-free column writes)
constexpr int FR_TILE_MAXP = FR_TILE_WORDS / 4;   // passes per wave: a chunk has <= 64 passes, dealt round-robin to 4 waves

template <bool DEDUP>
__global__ void __launch_bounds__(256) gather_tile_kernel(const FrPassDesc *__restrict__ passes, const FrChunkDesc *__restrict__ chunks, int n_chunks,
                                                          int chunks_per_group, const int32_t *__restrict__ idx, int idx_stride,
                                                          const float *__restrict__ dense, uint4 *__restrict__ out, int out_stride_words, int batch,
                                                          int *__restrict__ err_flag, unsigned long long *__restrict__ dup_counter) {
    __shared__ uint4 tile[FR_TILE_ITEMS * FR_TILE_LD];
    __shared__ uint32_t hash_slots[4][128];
    // workgroup b serves XCD group b % 8 (round-robin dealing; a speed assumption only): a chunk, hence a table, is always fetched by
    // the same XCD, so the 8 L2s cache 8 different table sets (as gather_pack_xcd_kernel does)
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int chunk = (j % chunks_per_group) * 8 + x;
    const int item_tile = j / chunks_per_group;
    if (chunk >= n_chunks) return;
    const FrChunkDesc ch = chunks[chunk];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform: descriptors load through SMEM
    const int m0 = item_tile * FR_TILE_ITEMS;

    uint32_t id[FR_TILE_MAXP];
    uint4 v[FR_TILE_MAXP];
    bool bad = false;
    // phase 1: index loads of every pass of this wave
#pragma unroll
    for (int q = 0; q < FR_TILE_MAXP; q++) {
        const int p = ch.pass_begin + 4 * q + wave;
        id[q] = 0;
        if (p < ch.pass_end) {
            const FrPassDesc d = passes[p];
            const int m = m0 + d.item0 + (lane >> d.log2_words);
            const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
            if (m < batch) id[q] = is_dense ? (uint32_t)m : (uint32_t)idx[(size_t)m * idx_stride + d.idx_col];
        }
    }
    // phase 2: (dedup,) row loads
    uint32_t src_lane[FR_TILE_MAXP];
    unsigned merged = 0;
#pragma unroll
    for (int q = 0; q < FR_TILE_MAXP; q++) {
        const int p = ch.pass_begin + 4 * q + wave;
        src_lane[q] = (uint32_t)lane;
        if (p < ch.pass_end) {
            const FrPassDesc d = passes[p];
            const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
            const uint32_t word = (uint32_t)lane & ((1u << d.log2_words) - 1u);
            if (!is_dense && id[q] >= d.rows) {  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
                bad = true;
                id[q] = 0;
            }
            bool dup = false;
            if (DEDUP && !is_dense) {
                const uint32_t slot = (uint32_t)lane >> d.log2_words;
                const uint32_t h = (id[q] * 0x9E3779B1u) >> 25;
                // same-wave LDS accesses execute in order: the read sees this pass's last writer.  volatile: without it hipcc forwards
                // the lane's own store to its load and no lane ever finds a leader
                volatile uint32_t *slots = hash_slots[wave];
                slots[h] = slot;
                __builtin_amdgcn_wave_barrier();
                const uint32_t winner = slots[h];
                const uint32_t wid = (uint32_t)__shfl((int)id[q], (int)(winner << d.log2_words));
                dup = wid == id[q] && winner != slot;
                if (dup) src_lane[q] = (winner << d.log2_words) + word;
                if (dup_counter) merged += (unsigned)__popcll(__ballot(dup && word == 0));
            }
            const char *base = is_dense ? reinterpret_cast<const char *>(dense) + d.src : reinterpret_cast<const char *>(d.src);
            if (!dup) v[q] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[q] * d.stride + 16u * word);
        }
    }
    // phase 3: duplicates take their leader's registers; rows -> LDS tile
#pragma unroll
    for (int q = 0; q < FR_TILE_MAXP; q++) {
        const int p = ch.pass_begin + 4 * q + wave;
        if (p < ch.pass_end) {
            const FrPassDesc d = passes[p];
            uint4 r = v[q];
            if (DEDUP) {
                r.x = (uint32_t)__shfl((int)v[q].x, (int)src_lane[q]);
                r.y = (uint32_t)__shfl((int)v[q].y, (int)src_lane[q]);
                r.z = (uint32_t)__shfl((int)v[q].z, (int)src_lane[q]);
                r.w = (uint32_t)__shfl((int)v[q].w, (int)src_lane[q]);
            }
            const uint32_t word = (uint32_t)lane & ((1u << d.log2_words) - 1u);
            tile[(d.item0 + (lane >> d.log2_words)) * FR_TILE_LD + d.tile_word + word] = r;
        }
    }
    __syncthreads();
    // phase 4: whole record pieces out, 16 items per wave
#pragma unroll 4
    for (int i = 0; i < FR_TILE_ITEMS / 4; i++) {
        const int it = wave * (FR_TILE_ITEMS / 4) + i, m = m0 + it;
        if (m < batch && lane < ch.n_words) out[(size_t)m * out_stride_words + ch.word0 + lane] = tile[it * FR_TILE_LD + lane];
    }
    if (bad) atomicOr_system(err_flag, 1);
    if (DEDUP && dup_counter && lane == 0 && merged) atomicAdd(dup_counter, (unsigned long long)merged);
}

int frk_gather_tile(const FrPassDesc *passes, const FrChunkDesc *chunks, int n_chunks, const int32_t *idx, int idx_stride, const float *dense, void *out,
                    int out_stride_words, int batch, int *err_flag, bool dedup, unsigned long long *dup_counter, hipStream_t s) {
    if (n_chunks <= 0 || batch <= 0) return FR_OK;
    const int cpg = (n_chunks + 7) / 8;
    const int tiles = (batch + FR_TILE_ITEMS - 1) / FR_TILE_ITEMS;
    dim3 grid(8 * cpg * tiles);
    if (dedup)
        gather_tile_kernel<true><<<grid, dim3(256), 0, s>>>(passes, chunks, n_chunks, cpg, idx, idx_stride, dense, reinterpret_cast<uint4 *>(out), out_stride_words, batch,
                                                           err_flag, dup_counter);
    else
        gather_tile_kernel<false><<<grid, dim3(256), 0, s>>>(passes, chunks, n_chunks, cpg, idx, idx_stride, dense, reinterpret_cast<uint4 *>(out), out_stride_words, batch,
                                                            err_flag, nullptr);
    KCHECK();
    fr_note_kernel("gather_tile_kernel<%s>", dedup ? "true" : "false");
    return FR_OK;
}
