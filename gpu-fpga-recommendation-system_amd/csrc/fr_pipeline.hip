// fr_pipeline_kernel: the stage pipeline of the FleetRec hot path (gather | FC1 | FC2 | FC3 | out of five consecutive batches in
// one launch) in its fp32, bf16 and fp8 flavours -- every model / precision / entry point that the fused item-tile kernels
// (fr_fused.hip) and the LDS-tiled GEMM (fr_gemm.hip) do not cover.  Wavefront = 64 lanes everywhere; gfx950 only.
#include "fr_device.h"


// ===================================================================================================
// FC chain on k-quad-packed feature-major activations, split-K inside the workgroup, stages
// software-pipelined across consecutive batches of a worker.
//
// Layout ("q4"): every activation matrix and every weight matrix is stored with 4 consecutive k per 16 bytes,
//     Xq[k/4][m][k%4]   (m = item, leading dimension ldm)         Wq[k/4][n][k%4]   (n = output feature)
// so that both MFMA operands are ONE coalesced 16-byte load per lane (512 contiguous bytes per half-wave) that goes
// straight from L2 to registers and feeds FOUR v_mfma_f32_32x32x2_f32:
//     a4 = Wq[2g + (lane>>5)][n0 + (lane&31)]   b4 = Xq[2g + (lane>>5)][m0 + (lane&31)]      (g = group of 8 k)
//     MFMA t (0..3) multiplies a4[t] x b4[t]: lane half h carries k = 8g + 4h + t on both operands.
// (4-byte operand loads cost ~37 cycles of the CU's vector-memory pipeline per wave-instruction and capped the whole
// chain at 4 TB/s; the 16-byte form moves 4x the bytes in ~25 cycles -- tools/experiments/ta_cost.hip.)
// The MFMA result D[n][m] has 4 consecutive n per lane in registers 4i..4i+3 (row = 8i + 4(lane>>5) + (r&3)), i.e. it is
// already a q4 element of the NEXT layer's B operand: the tile is stored as 16-byte elements, no shuffle.
// The reference's column-major W (cuda_server.c:215) is re-packed once when weights are set.
//
// At batch 256 one FC layer is only 64-256 output tiles of 32x32; a classic LDS-tiled GEMM leaves most of the 1024
// SIMDs idle and walks K serially (measured: 20 us per layer).  Here the 8 waves of a workgroup each own one slice of K
// for the SAME 32x32 output tile and are summed through LDS in a fixed order (deterministic, no atomics).  A layer may
// additionally be cut into `nsplit` workgroups along K that write partial tiles; the NEXT layer adds the partials
// while loading its B operand (launch-boundary reduce).  Item columns m >= batch are padding: every output column
// depends only on the same input column, so they never mix with real items.
//
// Why stage-pipelined launches: MI355X runs at most ~4 kernels of different streams concurrently and a dependent
// launch costs 2-5 us, so five narrow launches per batch cap throughput.  One launch of fr_pipeline_kernel carries
// ALL stages at once, each working on a different batch of the same worker:
//     launch L:  gather(batch L) | FC1(batch L-1) | FC2(batch L-2) | FC3(batch L-3) | out(batch L-4)
// Stage s reads what stage s-1 wrote in the PREVIOUS launch (activation buffers alternate by launch parity), so
// in-order execution of a stream's launches is the only synchronisation.  This mirrors the reference's hot loop,
// which enqueues batch after batch and never synchronises inside the loop (cuda_server.c:406-497).
// ===================================================================================================

constexpr int FR_PIPE_THREADS = 512;  // 8 waves per workgroup in every stage
constexpr int FR_PIPE_WAVES = 8;
// FrStageArgs / FrPipeArgs: fr_internal.h

// ---- stage 0: gather_q.  Lanes = 64 consecutive items; each wave walks WPW record words (wave-uniform descriptor ->
// scalar loads); each lane reads its item's 16-byte row word and stores it as ONE q4 element Xq[word][m]
// (1 KiB coalesced per wave-store).
__device__ __forceinline__ void gather_q_body(const FrPipeArgs &a, const FrStageArgs &st, int local) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m_blocks = (st.ldm + 63) / 64;
    const int mb = local % m_blocks, wb = local / m_blocks;  // padding workgroups land on w >= n_words below
    const int m = mb * 64 + lane;
    constexpr int WPW = 2;  // words per wave
    const int w_begin = (wb * FR_PIPE_WAVES + wave) * WPW;
    const bool live = m < st.batch;
    uint4 *Xq = reinterpret_cast<uint4 *>(st.out);
    bool bad = false;
#pragma unroll
    for (int i = 0; i < WPW; i++) {
        const int w = w_begin + i;
        if (w >= a.n_words) break;
        const FrWordDesc d = a.words[w];  // wave-uniform
        const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) {
            uint32_t id = is_dense ? (uint32_t)m : (uint32_t)a.idx[(size_t)m * a.idx_stride + d.idx_col];
            if (!is_dense && id >= d.rows) {  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
                bad = true;
                id = 0;
            }
            const char *base = is_dense ? reinterpret_cast<const char *>(a.dense) + d.src : reinterpret_cast<const char *>(d.src);
            v = *reinterpret_cast<const uint4 *>(base + (uint64_t)id * d.stride);
        }
        if (m < st.ldm) Xq[(size_t)d.dst_off * st.ldm + m] = v;  // SEMANTIC layout: dst_off = record word index = k/4
    }
    if (bad) atomicOr_system(a.err_flag, 1);  // pinned host word; error path only
}

// ---- stage 0, large batches: gather with an LDS transpose.  The lanes-along-items form above issues one 16-byte request
// per (item, record word): a dim-16 row is fetched by four separate wave-instructions, and all 64 lanes of an instruction
// hit 64 different lines.  Here a workgroup owns a tile of 32 items x 64 record words: phase 1 loads with lanes along
// WORDS (a row is read by dim/4 adjacent lanes of one instruction, like gather_pack_kernel) into an LDS tile, phase 2 reads
// the tile transposed (XOR-swizzled columns: conflict-free ds_read_b128) and stores with lanes along ITEMS (512 contiguous
// bytes per half-wave) in the chain's q4 (PREC 0) or bf16 q8 (PREC 1) layout.
constexpr int FR_GT_ITEMS = 32, FR_GT_WORDS = 64;
// tile element (item, word) lives at item * 64 + (word ^ (item & 15)): exactly 32 KiB (a padded stride of 65 would be 33,280 B and
// one such workgroup would no longer fit beside two 64 KiB GEMM workgroups on a CU), conflict-free both ways -- a b128 access is
// served 16 lanes at a time, and 16 consecutive words of one item (phase 1) or one word of 16 consecutive items (phase 2) land in
// 16 different 16-byte bank groups.
__device__ __forceinline__ int gt_at(int item, int word) { return item * FR_GT_WORDS + (word ^ (item & 15)); }

__device__ __forceinline__ uint32_t pack_fp8_word(const uint4 &v, float scale);

template <int PREC>
__device__ __forceinline__ void gather_tr_body(const FrPipeArgs &a, const FrStageArgs &st, int local, uint4 *tile /* [32][64], swizzled */) {
    const int m_blocks = st.ldm / FR_GT_ITEMS;
    const int mb = local % m_blocks, wb = local / m_blocks;
    const int m0 = mb * FR_GT_ITEMS, w0 = wb * FR_GT_WORDS;
    if (w0 >= a.n_words) return;  // padding workgroup
    {   // phase 1: lanes along words
        const int wl = threadIdx.x & 63, ig = threadIdx.x >> 6;
        const int w = w0 + wl;
        bool bad = false;
        if (w < a.n_words) {
            const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
            const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
            const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
            const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
            const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
            const char *base = is_dense ? reinterpret_cast<const char *>(a.dense) + src : reinterpret_cast<const char *>(src);
            uint32_t id[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = m0 + 4 * ig + i;
                id[i] = 0;
                if (m < st.batch) id[i] = is_dense ? (uint32_t)m : (uint32_t)a.idx[(size_t)m * a.idx_stride + idx_col];
                if (!is_dense && id[i] >= rows) {
                    bad = true;
                    id[i] = 0;
                }
            }
            uint4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int m = m0 + 4 * ig + i;
                tile[gt_at(4 * ig + i, wl)] = (m < st.batch) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();
    {   // phase 2: lanes along items
        const int il = threadIdx.x & 31, ws = threadIdx.x >> 5;  // 16 word slots
        const int m = m0 + il;
        if constexpr (PREC == 0) {
            uint4 *Xq = reinterpret_cast<uint4 *>(st.out);
#pragma unroll
            for (int j = 0; j < FR_GT_WORDS / 16; j++) {
                const int wl = ws + 16 * j, w = w0 + wl;
                if (w < a.n_words) Xq[(size_t)w * st.ldm + m] = tile[gt_at(il, wl)];  // SEMANTIC layout: dst word == w
            }
        } else if constexpr (PREC == 2) {
            uint4 *Xf = reinterpret_cast<uint4 *>(st.out);  // q16 element = record words 4e .. 4e+3 as e4m3 bytes (x 2^e_out, saturated)
            const float scale = __builtin_ldexpf(1.0f, st.e_out);
            const int KE = (st.K + 63) / 64 * 4;       // q16 rows including the zero pad up to a multiple of 64 k
            const int el = ws, e = (w0 >> 2) + el;     // 16 element slots x 32 items = one element per thread
            if (e < KE) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int wl = 4 * el + j;
                    o[j] = (w0 + wl < a.n_words) ? pack_fp8_word(tile[gt_at(il, wl)], scale) : 0u;
                }
                Xf[(size_t)e * st.ldm + m] = make_uint4(o[0], o[1], o[2], o[3]);
            }
        } else {
            uint4 *Xh = reinterpret_cast<uint4 *>(st.out);  // q8 element = record words 2p, 2p+1
#pragma unroll
            for (int j = 0; j < FR_GT_WORDS / 32; j++) {
                const int pl = ws + 16 * j, w = w0 + 2 * pl;
                if (w < a.n_words) {
                    const uint4 lo = tile[gt_at(il, 2 * pl)], hi = tile[gt_at(il, 2 * pl + 1)];
                    uint4 h;
                    h.x = pack_bf16x2(__uint_as_float(lo.x), __uint_as_float(lo.y));
                    h.y = pack_bf16x2(__uint_as_float(lo.z), __uint_as_float(lo.w));
                    h.z = pack_bf16x2(__uint_as_float(hi.x), __uint_as_float(hi.y));
                    h.w = pack_bf16x2(__uint_as_float(hi.z), __uint_as_float(hi.w));
                    Xh[(size_t)(w >> 1) * st.ldm + m] = h;
                }
            }
        }
    }
}

// Software-pipelined form of gather_tr_body (the same lesson as gather_pack_stream_kernel): a workgroup owns NT consecutive item
// tiles of one 64-word block; it keeps its descriptors, issues the index loads of ALL its tiles first and the row loads of all of
// them next, and transposes / stores tile t while the rows of tile t + 1 are still in flight.  Straight-line and branch-free in the
// load phase: items past the batch read index 0 through the bounds of a buffer resource; the image stores go through a resource
// over the workgroup's word block with the cache policy AUX (16 = write-through: the image is not left dirty in L2 for the
// end-of-kernel write-back).  Needs batch * idx_stride * 4 < 4000 MiB (variant chosen by the host).
template <int PREC, int NT, int AUX>
__device__ __forceinline__ void gather_tr_stream_body(const FrPipeArgs &a, const FrStageArgs &st, int local, uint4 *tile /* [32][64], swizzled */) {
    const int m_tiles = st.ldm / FR_GT_ITEMS, m_groups = (m_tiles + NT - 1) / NT;
    const int mg = local % m_groups, wb = local / m_groups;
    const int w0 = wb * FR_GT_WORDS;
    if (w0 >= a.n_words) return;  // padding workgroup
    const int wl = threadIdx.x & 63, ig = threadIdx.x >> 6;
    const int w = w0 + wl < a.n_words ? w0 + wl : a.n_words - 1;   // a lane past the record repeats the last word (its tile column is never read)
    const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
    const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const uint64_t base = (is_dense ? (uint64_t)reinterpret_cast<uintptr_t>(a.dense) : 0ull) + (((uint64_t)d0.y << 32) | d0.x);
    const unsigned icol = is_dense ? 0u : idx_col * 4u;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.idx), 0, (unsigned)st.batch * (unsigned)a.idx_stride * 4u, 0x00020000);
    uint32_t id[NT][4];
    uint4 v[NT][4];
    bool bad = false;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned m = (unsigned)((mg * NT + t) * FR_GT_ITEMS + 4 * ig + i);
            id[t][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, m * (unsigned)a.idx_stride * 4u + icol, 0, 0);
        }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned m = (unsigned)((mg * NT + t) * FR_GT_ITEMS + 4 * ig + i);
            uint32_t r = id[t][i];
            const bool oob = !is_dense & (r >= rows);  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
            bad |= oob;
            r = oob ? 0u : r;
            r = is_dense ? (m < (unsigned)st.batch ? m : 0u) : r;
            typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
            const u32x4_t q = *(gptr_t)(base + (uint64_t)r * stride);
            v[t][i] = make_uint4(q.x, q.y, q.z, q.w);
        }
    constexpr unsigned OSZ = 16;  // every image element (q4 word, q8 pair, q16 quad) is 16 bytes per item
    const int il = threadIdx.x & 31, ws = threadIdx.x >> 5;  // phase 2: lanes along items, 16 word slots
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int m0 = (mg * NT + t) * FR_GT_ITEMS;
        if (m0 >= st.ldm) break;  // workgroup-uniform: the last group of an odd tile count
        if (t > 0) __syncthreads();   // the previous tile has been read out
#pragma unroll
        for (int i = 0; i < 4; i++) tile[gt_at(4 * ig + i, wl)] = (m0 + 4 * ig + i < st.batch) ? v[t][i] : make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        const int m = m0 + il;
        auto put = [&](size_t elem_row, const uint4 &x) {   // image element row (word / pair / quad index) of item m
            if constexpr (AUX == 0) {
                reinterpret_cast<uint4 *>(st.out)[elem_row * st.ldm + m] = x;
            } else {
                const size_t row0 = PREC == 0 ? (size_t)w0 : PREC == 1 ? (size_t)(w0 >> 1) : (size_t)(w0 >> 2);
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(st.out) + row0 * st.ldm * OSZ, 0, 0xffffffffu, 0x00020000);
                u32x4_t q;
                q.x = x.x, q.y = x.y, q.z = x.z, q.w = x.w;
                __builtin_amdgcn_raw_buffer_store_b128(q, rs, (unsigned)(((elem_row - row0) * st.ldm + m) * OSZ), 0, AUX);
            }
        };
        if constexpr (PREC == 0) {
#pragma unroll
            for (int j = 0; j < FR_GT_WORDS / 16; j++) {
                const int wl2 = ws + 16 * j, w2 = w0 + wl2;
                if (w2 < a.n_words) put((size_t)w2, tile[gt_at(il, wl2)]);
            }
        } else if constexpr (PREC == 2) {
            const float scale = __builtin_ldexpf(1.0f, st.e_out);
            const int KE = (st.K + 63) / 64 * 4;
            const int el = ws, e = (w0 >> 2) + el;
            if (e < KE) {
                uint32_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int wl2 = 4 * el + j;
                    o[j] = (w0 + wl2 < a.n_words) ? pack_fp8_word(tile[gt_at(il, wl2)], scale) : 0u;
                }
                put((size_t)e, make_uint4(o[0], o[1], o[2], o[3]));
            }
        } else {
#pragma unroll
            for (int j = 0; j < FR_GT_WORDS / 32; j++) {
                const int pl = ws + 16 * j, w2 = w0 + 2 * pl;
                if (w2 < a.n_words) {
                    const uint4 lo = tile[gt_at(il, 2 * pl)], hi = tile[gt_at(il, 2 * pl + 1)];
                    uint4 h;
                    h.x = pack_bf16x2(__uint_as_float(lo.x), __uint_as_float(lo.y));
                    h.y = pack_bf16x2(__uint_as_float(lo.z), __uint_as_float(lo.w));
                    h.z = pack_bf16x2(__uint_as_float(hi.x), __uint_as_float(hi.y));
                    h.w = pack_bf16x2(__uint_as_float(hi.z), __uint_as_float(hi.w));
                    put((size_t)(w2 >> 1), h);
                }
            }
        }
    }
    if (bad) atomicOr_system(a.err_flag, 1);
}

// The same gather over the OPERAND-TYPE BANK IMAGE (a.src_lp; round 6): the table words a.words address are already in the chain's operand
// type -- 8 bytes (4 bf16) or 4 bytes (4 e4m3) per record word -- so a Model-C item costs 82 lines of the fabric instead of 142, the LDS
// tile is a half / a quarter of the size and phase 2 converts nothing: one ds_read_b128 IS one image element.  Dense words (the request's
// fp32 features) are converted by the lane that loads them, with the rounding the image was made with (pack_bf16x2 / pack_fp8_word), so the
// image written here is bit-identical to gather_tr_stream_body's.  Tile element = one record word (uint2 / uint32); element (item, word)
// sits at item * 64 + (((word / EW) ^ (item & SW)) * EW + word % EW), EW = words per image element (2 / 4): the words of an image element
// stay adjacent (phase 2 reads them in one piece), 16 consecutive words of an item (phase 1) and one element of 32 consecutive items
// (phase 2) spread over all banks.
template <int PREC, int NT, int AUX>
__device__ __forceinline__ void gather_tr_stream_lp_body(const FrPipeArgs &a, const FrStageArgs &st, int local, uint4 *tile_raw) {
    static_assert(PREC == 1 || PREC == 2, "operand-type rows exist for the bf16 and fp8 chains");
    constexpr int EW = PREC == 1 ? 2 : 4;          // record words per 16-byte image element
    constexpr int SW = 64 / EW - 1;                // swizzle mask over the element index
    using word_t = typename std::conditional<PREC == 1, uint2, uint32_t>::type;
    word_t *tile = reinterpret_cast<word_t *>(tile_raw);
    auto at = [](int item, int word) { return item * FR_GT_WORDS + ((((word / EW) ^ (item & SW)) * EW) | (word % EW)); };
    const int m_tiles = st.ldm / FR_GT_ITEMS, m_groups = (m_tiles + NT - 1) / NT;
    const int mg = local % m_groups, wb = local / m_groups;
    const int w0 = wb * FR_GT_WORDS;
    if (w0 >= a.n_words) return;  // padding workgroup
    const int wl = threadIdx.x & 63, ig = threadIdx.x >> 6;
    const bool w_live = w0 + wl < a.n_words;     // a lane past the record writes zeros (fp8: the image's zero pad up to 64 k)
    const int w = w_live ? w0 + wl : a.n_words - 1;
    const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
    const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const uint64_t base = (is_dense ? (uint64_t)reinterpret_cast<uintptr_t>(a.dense) : 0ull) + (((uint64_t)d0.y << 32) | d0.x);
    const unsigned icol = is_dense ? 0u : idx_col * 4u;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.idx), 0, (unsigned)st.batch * (unsigned)a.idx_stride * 4u, 0x00020000);
    const float scale = PREC == 2 ? __builtin_ldexpf(1.0f, st.e_out) : 1.0f;
    uint32_t id[NT][4];
    word_t v[NT][4];
    bool bad = false;
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned m = (unsigned)((mg * NT + t) * FR_GT_ITEMS + 4 * ig + i);
            id[t][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, m * (unsigned)a.idx_stride * 4u + icol, 0, 0);
        }
    if (is_dense) {   // wave-divergent only in the word block that holds the request's dense features
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned m = (unsigned)((mg * NT + t) * FR_GT_ITEMS + 4 * ig + i);
                const unsigned r = m < (unsigned)st.batch ? m : 0u;
                typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
                const u32x4_t q = *(gptr_t)(base + (uint64_t)r * stride);
                if constexpr (PREC == 1) v[t][i] = make_uint2(pack_bf16x2(__uint_as_float(q.x), __uint_as_float(q.y)), pack_bf16x2(__uint_as_float(q.z), __uint_as_float(q.w)));
                else v[t][i] = pack_fp8_word(make_uint4(q.x, q.y, q.z, q.w), scale);
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t r = id[t][i];
                const bool oob = r >= rows;  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
                bad |= oob;
                r = oob ? 0u : r;
                if constexpr (PREC == 1) {
                    typedef const unsigned __attribute__((ext_vector_type(2))) __attribute__((address_space(1))) * gptr2_t;
                    const auto q = *(gptr2_t)(base + (uint64_t)r * stride);
                    v[t][i] = make_uint2(q.x, q.y);
                } else {
                    typedef const unsigned __attribute__((address_space(1))) * gptr1_t;
                    v[t][i] = *(gptr1_t)(base + (uint64_t)r * stride);
                }
            }
    }
    const int il = threadIdx.x & 31, ws = threadIdx.x >> 5;  // phase 2: lanes along items, 16 element slots
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int m0 = (mg * NT + t) * FR_GT_ITEMS;
        if (m0 >= st.ldm) break;  // workgroup-uniform: the last group of an odd tile count
        if (t > 0) __syncthreads();   // the previous tile has been read out
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const bool keep = w_live && (m0 + 4 * ig + i < st.batch);
            if constexpr (PREC == 1) tile[at(4 * ig + i, wl)] = keep ? v[t][i] : make_uint2(0u, 0u);
            else tile[at(4 * ig + i, wl)] = keep ? v[t][i] : 0u;
        }
        __syncthreads();
        const int m = m0 + il;
        auto put = [&](size_t elem_row, const uint4 &x) {   // image element row (pair / quad index) of item m
            const size_t row0 = (size_t)(w0 / EW);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char *>(st.out) + row0 * st.ldm * 16, 0, 0xffffffffu, 0x00020000);
            u32x4_t q;
            q.x = x.x, q.y = x.y, q.z = x.z, q.w = x.w;
            __builtin_amdgcn_raw_buffer_store_b128(q, rs, (unsigned)(((elem_row - row0) * st.ldm + m) * 16), 0, AUX);
        };
        if constexpr (PREC == 2) {
            const int KE = (st.K + 63) / 64 * 4;
            const int e = (w0 >> 2) + ws;              // 16 quads of the block x 32 items = one element per thread
            if (e < KE) put((size_t)e, *reinterpret_cast<const uint4 *>(&tile[at(il, 4 * ws)]));
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {              // 32 pairs of the block x 32 items = two elements per thread
                const int pl = ws + 16 * j, w2 = w0 + 2 * pl;
                if (w2 < a.n_words) put((size_t)(w2 >> 1), *reinterpret_cast<const uint4 *>(&tile[at(il, 2 * pl)]));
            }
        }
    }
    if (bad) atomicOr_system(a.err_flag, 1);
}

// 2 = gather_tr_stream_body (two tiles per workgroup, write-through image stores: Model-C batch 4096 chain 55 -> 58 M inf/s in fp8,
// 35.7 -> 37 M in bf16, profiles/archive/r02_gather_tr_variants.txt); 1 = gather_tr_body (FR_GATHER_TR_VARIANT=1, and index buffers >= 4000 MiB)
int frk_gather_tr_variant(int batch, int idx_stride) {
    const int v = FR_KNOB_ONCE("GATHER_TR_VARIANT", 2);  // experiment knob
    return (v == 1 || (size_t)batch * (size_t)idx_stride * 4 >= ((size_t)4000 << 20)) ? 1 : 2;
}

int frk_gather_tr_blocks(int n_words, int ldm, int variant) {
    if (ldm % FR_GT_ITEMS || (n_words & 1)) return 0;
    const int nt = variant == 2 ? 2 : 1;
    const int blocks = ((ldm / FR_GT_ITEMS + nt - 1) / nt) * ((n_words + FR_GT_WORDS - 1) / FR_GT_WORDS);
    const int forced = FR_KNOB_ONCE("GATHER_TR", -1);  // experiment knob
    if (forced == 0) return 0;
    if (forced == 1) return blocks;
    return blocks >= 128 ? blocks : 0;  // needs enough workgroups to cover the chip; small batches keep the simple form
}

static int gather_q_blocks(int n_words, int ldm) { return ((ldm + 63) / 64) * ((n_words + FR_PIPE_WAVES * 2 - 1) / (FR_PIPE_WAVES * 2)); }

// ---- stages 1..3: one 32(n) x 32(m) output tile per workgroup, 8 waves split the workgroup's K range in groups of
// 8 k (one 16-byte load per operand per lane -> 4 MFMAs).

__device__ __forceinline__ void mfma4(f32x16 &acc, const float4 &a, const float4 &b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}

template <bool TWO_IN>
__device__ __forceinline__ void fc_q_body(const FrStageArgs &st, int local, float *red /* [8][16][64] */) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = st.N, ldm = st.ldm;
    const int tiles_n = N / 32, tiles_m = ldm / 32, tiles = tiles_n * tiles_m;
    // Workgroup -> (K part, n tile, m tile).  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the
    // XCD group; a speed assumption only, never correctness) and every stage starts at a multiple of 8, so `local % 8`
    // is the group.  Each group gets a fixed 1/NG of the weight columns and 1/MG of the items, so its slice of the
    // weights can stay in that XCD's 4 MiB L2 across launches.
    int part, n_tile, m_tile;
    const int MG = (tiles_m % 2 == 0) ? 2 : 1, NG = 8 / MG;  // (MG = 4 or 8 measured the same: the stage is not L2-miss bound)
    if (tiles_n % NG == 0) {
        const int x = local & 7, j = local >> 3;
        const int mg = x % MG, ng = x / MG;
        const int tn_x = tiles_n / NG, tm_x = tiles_m / MG;
        part = j / (tn_x * tm_x);
        const int r = j - part * (tn_x * tm_x);
        n_tile = (r % tn_x) * NG + ng;
        m_tile = (r / tn_x) * MG + mg;
    } else {
        part = local / tiles;
        const int tile = local - part * tiles;
        n_tile = tile % tiles_n;
        m_tile = tile / tiles_n;
    }
    if (part >= st.nsplit) return;  // padding workgroup (stage sizes are rounded up to a multiple of 8)
    const int n0 = n_tile * 32, m0 = m_tile * 32;
    const int hk = lane >> 5, lm = lane & 31;
    // this workgroup's K range in groups of 8 k, then this wave's slice of it
    const int groups = st.K / 8;
    const int wg_groups = (groups + st.nsplit - 1) / st.nsplit;
    const int wg_begin = part * wg_groups;
    int wg_ng = groups - wg_begin;
    wg_ng = wg_ng < 0 ? 0 : (wg_ng > wg_groups ? wg_groups : wg_ng);
    const int per = (wg_ng + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int g_begin = wg_begin + wave * per;
    int ng = wg_begin + wg_ng - g_begin;
    ng = ng < 0 ? 0 : (ng > per ? per : ng);

    // q4 element (k-quad 2g + hk, column) of an operand = byte (quad * ld + column) * 16.  Both operands come through buffer
    // loads: a constant per-lane VGPR offset plus a wave-uniform SGPR offset that advances by one group per step -- no 64-bit
    // VALU address arithmetic in the loop (it costs MFMA issue slots: tools/experiments/mfma_loop, 75 -> 68 cycles per MFMA).
    const unsigned KQ = (unsigned)(st.K / 4);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.w), 0, KQ * (unsigned)N * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.in), 0, KQ * (unsigned)ldm * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.in + (TWO_IN ? st.in_part_stride : 0)), 0, KQ * (unsigned)ldm * 16u, 0x00020000);
    const unsigned voA = ((unsigned)hk * N + n0 + lm) * 16u, voB = ((unsigned)hk * ldm + m0 + lm) * 16u;
    const unsigned stepA = 2u * (unsigned)N * 16u, stepB = 2u * (unsigned)ldm * 16u;
    auto ld_a = [&](int g) { return bload4(rsA, voA, (unsigned)g * stepA); };
    auto ld_b = [&](int g) {
        float4 b = bload4(rsB, voB, (unsigned)g * stepB);
        if constexpr (TWO_IN) {  // launch-boundary reduce of the previous layer's two K halves
            const float4 c = bload4(rsC, voB, (unsigned)g * stepB);
            b.x += c.x;
            b.y += c.y;
            b.z += c.z;
            b.w += c.w;
        }
        return b;
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;

    // Double-buffered loop over blocks of D groups: the next block's 2*D 16-byte loads are in flight while the current
    // block's 4*D MFMAs issue.  D = 2 keeps the kernel small (4 workgroups per CU): these workgroups are bound by
    // the CU's L2 ingest, not by issue slots, and measured faster than straight-line bodies that hold a whole K slice
    // in registers (110-165 VGPRs, 1-2 workgroups per CU).
    constexpr int D = 2;
    float4 ra[D], rb[D], na[D], nb[D];
    const int nb_full = ng / D;
    if (nb_full > 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = ld_a(g_begin + i);
            rb[i] = ld_b(g_begin + i);
        }
    }
    for (int blk = 0; blk < nb_full; blk++) {
        const int nx = (blk + 1 < nb_full) ? (blk + 1) : blk;  // the last block re-loads itself (harmless) -> branch-free body
#pragma unroll
        for (int i = 0; i < D; i++) {
            na[i] = ld_a(g_begin + nx * D + i);
            nb[i] = ld_b(g_begin + nx * D + i);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the scheduler from sinking the loads next to their consumers
#pragma unroll
        for (int i = 0; i < D; i++) mfma4(acc, ra[i], rb[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = na[i];
            rb[i] = nb[i];
        }
    }
    for (int g = g_begin + nb_full * D; g < g_begin + ng; g++)  // remainder (< D groups)
        mfma4(acc, ld_a(g), ld_b(g));
    // cross-wave reduction in fixed order, then the tile goes out as q4 elements (registers 4i..4i+3 = 4 consecutive n)
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    float4 *Yq = reinterpret_cast<float4 *>(st.out + (size_t)part * st.part_stride);
    if (threadIdx.x < 4 * 64) {
        const int i = threadIdx.x >> 6, l = threadIdx.x & 63;  // accumulator register quad i, lane l
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float s = red[(4 * i + c) * 64 + l];
#pragma unroll
            for (int w = 1; w < FR_PIPE_WAVES; w++) s += red[(w * 16 + 4 * i + c) * 64 + l];
            v[c] = s;
        }
        const int nq = (n0 >> 2) + 2 * i + (l >> 5);  // n = n0 + 8i + 4(l>>5) + c
        Yq[(size_t)nq * ldm + m0 + (l & 31)] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

static int fc_q_blocks(int N, int ldm, int nsplit) { return (N / 32) * (ldm / 32) * nsplit; }
static int pad8(int v) { return (v + 7) / 8 * 8; }

// ---- stage 4: score[m] = sum_n w[n] * (R3q[n/4][m][n%4] (+ second partial)); 64 items x 8 slices of n, one 16-byte
// load per 4 n, LDS reduce in fixed order.
__device__ __forceinline__ void fc_out_q_body(const FrStageArgs &st, int local, float *red) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = local * 64 + lane;  // padding workgroups have m >= ldm
    const int HQ = st.K / 4, ldm = st.ldm;
    const int per = (HQ + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int h0 = q * per;
    const int h1 = (h0 + per) < HQ ? (h0 + per) : HQ;
    const float4 *Rq = reinterpret_cast<const float4 *>(st.in);
    const float4 *Cq = reinterpret_cast<const float4 *>(st.in + st.in_part_stride);
    const float4 *wq = reinterpret_cast<const float4 *>(st.w);
    float s = 0.0f;
    if (m < ldm) {
        for (int h = h0; h < h1; h += 4) {
            float4 v[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int hh = (h + i < h1) ? (h + i) : (h1 - 1);
                v[i] = Rq[(size_t)hh * ldm + m];
                if (st.nparts_in == 2) {
                    const float4 c = Cq[(size_t)hh * ldm + m];
                    v[i].x += c.x;
                    v[i].y += c.y;
                    v[i].z += c.z;
                    v[i].w += c.w;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (h + i < h1) {
                    const float4 w4 = wq[h + i];
                    s = fmaf(w4.x, v[i].x, s);
                    s = fmaf(w4.y, v[i].y, s);
                    s = fmaf(w4.z, v[i].z, s);
                    s = fmaf(w4.w, v[i].w, s);
                }
            }
        }
    }
    red[q * 64 + lane] = s;
    __syncthreads();
    if (q == 0 && m < st.batch) {
        float t = red[lane];
#pragma unroll
        for (int i = 1; i < FR_PIPE_WAVES; i++) t += red[i * 64 + lane];
        st.out[m] = t;
    }
}

// ===================================================================================================
// bf16 variant of the chain (BASELINE configs 3/4: "bf16 MFMA FC, fused concat + first FC").
// Same structure, "q8" layout: 8 consecutive k per 16 bytes, Xh[k/8][m][k%8] / Wh[k/8][n][k%8] (bf16), so one 16-byte load
// per lane is exactly one v_mfma_f32_32x32x16_bf16 operand (lane half h carries k = 16g + 8h + j, j = 0..7).  fp32
// accumulation; activations are rounded to bf16 (RNE) once per layer when the tile is stored.  The gather stage converts
// the fp32 table rows to bf16 while concatenating -- the record never exists in fp32 (fused concat + FC1 operand).
// No K-split partials here (nsplit == 1): a bf16 layer is 16x cheaper than its fp32 form and partial sums would have to be
// rounded twice.
// ===================================================================================================

__device__ __forceinline__ void gather_h_body(const FrPipeArgs &a, const FrStageArgs &st, int local) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m_blocks = (st.ldm + 63) / 64;
    const int mb = local % m_blocks, wb = local / m_blocks;
    const int m = mb * 64 + lane;
    constexpr int WPW = 2;
    const int w_begin = (wb * FR_PIPE_WAVES + wave) * WPW;
    const bool live = m < st.batch;
    uint2 *Xh = reinterpret_cast<uint2 *>(st.out);  // 8-byte halves of the q8 elements
    bool bad = false;
#pragma unroll
    for (int i = 0; i < WPW; i++) {
        const int w = w_begin + i;
        if (w >= a.n_words) break;
        const FrWordDesc d = a.words[w];
        const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) {
            uint32_t id = is_dense ? (uint32_t)m : (uint32_t)a.idx[(size_t)m * a.idx_stride + d.idx_col];
            if (!is_dense && id >= d.rows) {
                bad = true;
                id = 0;
            }
            const char *base = is_dense ? reinterpret_cast<const char *>(a.dense) + d.src : reinterpret_cast<const char *>(d.src);
            v = *reinterpret_cast<const uint4 *>(base + (uint64_t)id * d.stride);
        }
        if (m < st.ldm) {
            uint2 h;
            h.x = pack_bf16x2(__uint_as_float(v.x), __uint_as_float(v.y));
            h.y = pack_bf16x2(__uint_as_float(v.z), __uint_as_float(v.w));
            // record word w = floats 4w..4w+3 = half (w & 1) of q8 element w / 2
            Xh[((size_t)(d.dst_off >> 1) * st.ldm + m) * 2 + (d.dst_off & 1)] = h;
        }
    }
    if (bad) atomicOr_system(a.err_flag, 1);
}

__device__ __forceinline__ void fc_h_body(const FrStageArgs &st, int local, float *red) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = st.N, ldm = st.ldm;
    const int tiles_n = N / 32, tiles_m = ldm / 32, tiles = tiles_n * tiles_m;
    int n_tile, m_tile;
    const int MG = (tiles_m % 2 == 0) ? 2 : 1, NG = 8 / MG;
    if (tiles_n % NG == 0) {  // same XCD-aware map as the fp32 body
        const int x = local & 7, j = local >> 3;
        const int mg = x % MG, ng = x / MG;
        const int tn_x = tiles_n / NG;
        if (j >= tn_x * (tiles_m / MG)) return;
        n_tile = (j % tn_x) * NG + ng;
        m_tile = (j / tn_x) * MG + mg;
    } else {
        if (local >= tiles) return;
        n_tile = local % tiles_n;
        m_tile = local / tiles_n;
    }
    const int n0 = n_tile * 32, m0 = m_tile * 32;
    const int hk = lane >> 5, lm = lane & 31;
    const int groups = st.K / 16;  // one MFMA (16 k) per group
    const int per = (groups + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int g_begin = wave * per;
    int ng_ = groups - g_begin;
    ng_ = ng_ < 0 ? 0 : (ng_ > per ? per : ng_);
    // buffer loads: constant per-lane VGPR offset + wave-uniform SGPR offset per group (see fc_q_body)
    const unsigned KO = (unsigned)(st.K / 8);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.w), 0, KO * (unsigned)N * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.in), 0, KO * (unsigned)ldm * 16u, 0x00020000);
    const unsigned voA = ((unsigned)hk * N + n0 + lm) * 16u, voB = ((unsigned)hk * ldm + m0 + lm) * 16u;
    const unsigned stepA = 2u * (unsigned)N * 16u, stepB = 2u * (unsigned)ldm * 16u;
    auto ld_a = [&](int g) { return bload4u(rsA, voA, (unsigned)g * stepA); };
    auto ld_b = [&](int g) { return bload4u(rsB, voB, (unsigned)g * stepB); };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    constexpr int D = 4;
    uint4 ra[D], rb[D], na[D], nb[D];
    const int nb_full = ng_ / D;
    if (nb_full > 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = ld_a(g_begin + i);
            rb[i] = ld_b(g_begin + i);
        }
    }
    for (int blk = 0; blk < nb_full; blk++) {
        const int nx = (blk + 1 < nb_full) ? (blk + 1) : blk;
#pragma unroll
        for (int i = 0; i < D; i++) {
            na[i] = ld_a(g_begin + nx * D + i);
            nb[i] = ld_b(g_begin + nx * D + i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i]), __builtin_bit_cast(bf16x8, rb[i]), acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = na[i];
            rb[i] = nb[i];
        }
    }
    for (int g = g_begin + nb_full * D; g < g_begin + ng_; g++) {
        const uint4 a8 = ld_a(g), b8 = ld_b(g);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a8), __builtin_bit_cast(bf16x8, b8), acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // fixed-order fp32 sum, ONE rounding to bf16, stored as the 8-byte half (4 consecutive n) of a q8 element
    uint2 *Yh = reinterpret_cast<uint2 *>(st.out);
    if (threadIdx.x < 4 * 64) {
        const int i = threadIdx.x >> 6, l = threadIdx.x & 63;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float s = red[(4 * i + c) * 64 + l];
#pragma unroll
            for (int w = 1; w < FR_PIPE_WAVES; w++) s += red[(w * 16 + 4 * i + c) * 64 + l];
            v[c] = s;
        }
        uint2 h;
        h.x = pack_bf16x2(v[0], v[1]);
        h.y = pack_bf16x2(v[2], v[3]);
        // n = n0 + 8i + 4(l>>5) + c  ->  q8 element (n0/8 + i), half (l>>5)
        Yh[((size_t)((n0 >> 3) + i) * ldm + m0 + (l & 31)) * 2 + (l >> 5)] = h;
    }
}

// score[m] = sum_n w[n] * R3h[n/8][m][n%8]; weights for this layer stay fp32 values rounded to bf16 (st.w = bf16 array)
__device__ __forceinline__ void fc_out_h_body(const FrStageArgs &st, int local, float *red) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = local * 64 + lane;
    const int HO = st.K / 8, ldm = st.ldm;
    const int per = (HO + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int h0 = q * per;
    const int h1 = (h0 + per) < HO ? (h0 + per) : HO;
    const uint4 *Rh = reinterpret_cast<const uint4 *>(st.in);
    const uint4 *wh = reinterpret_cast<const uint4 *>(st.w);
    float s = 0.0f;
    if (m < ldm) {
        for (int h = h0; h < h1; h++) {
            const uint4 r = Rh[(size_t)h * ldm + m];
            const uint4 w = wh[h];
            const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                s = fmaf(__uint_as_float(ww[c] << 16), __uint_as_float(rr[c] << 16), s);
                s = fmaf(__uint_as_float(ww[c] & 0xFFFF0000u), __uint_as_float(rr[c] & 0xFFFF0000u), s);
            }
        }
    }
    red[q * 64 + lane] = s;
    __syncthreads();
    if (q == 0 && m < st.batch) {
        float t = red[lane];
#pragma unroll
        for (int i = 1; i < FR_PIPE_WAVES; i++) t += red[i * 64 + lane];
        st.out[m] = t;
    }
}

// ===================================================================================================
// fp8 variant of the chain (BASELINE configs[4]: "fp8 MFMA FC on CDNA4").
// "q16" layout: 16 consecutive k per 16 bytes, Xf[k/16][m][k%16] / Wf[k/16][n][k%16] of OCP e4m3 bytes, K zero-padded to a
// multiple of 64.  One v_mfma_scale_f32_32x32x64_f8f6f4 takes 32 bytes per lane per operand = two q16 elements (lane half h
// carries k = 64g + 32h + j, j = 0..31 -- any assignment works as long as both operands use the same one).
// Quantisation is per tensor with power-of-two scales: weights are stored as e4m3(W * 2^e_w), activations as
// e4m3(sat(X * 2^e_x)); the MFMA's E8M0 block scales (127 - e_w, 127 - e_x) undo both inside the instruction, so the fp32
// accumulator is in real units.  v_cvt_pk_fp8_f32 rounds to nearest even but yields NaN above 448 (probed on gfx950:
// tools/experiments/fp8_probe.hip), hence the explicit clamp.  The activation exponents come from a calibration batch
// (fr_worker_calibrate_fp8) or from an rms estimate made when the weights are packed.  The output layer (N = 1) multiplies
// the decoded fp8 R3 by the fp32 master weights.  No K-split partials, no fused / tiled variants yet: stage pipeline only.
// ===================================================================================================

// stage 0, small batches: lanes = 64 consecutive items, each wave builds ONE q16 element (4 record words) per item
__device__ __forceinline__ void gather_f_body(const FrPipeArgs &a, const FrStageArgs &st, int local) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m_blocks = (st.ldm + 63) / 64;
    const int mb = local % m_blocks, eb = local / m_blocks;
    const int m = mb * 64 + lane;
    const int KE = (st.K + 63) / 64 * 4;  // q16 rows including the zero pad
    const int e = eb * FR_PIPE_WAVES + wave;
    if (e >= KE) return;
    const bool live = m < st.batch;
    const float scale = __builtin_ldexpf(1.0f, st.e_out);
    uint32_t out[4];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int w = 4 * e + j;
        out[j] = 0u;
        if (w < a.n_words && live) {
            const FrWordDesc d = a.words[w];  // wave-uniform; SEMANTIC layout: dst_off == w
            const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
            uint32_t id = is_dense ? (uint32_t)m : (uint32_t)a.idx[(size_t)m * a.idx_stride + d.idx_col];
            if (!is_dense && id >= d.rows) {
                bad = true;
                id = 0;
            }
            const char *base = is_dense ? reinterpret_cast<const char *>(a.dense) + d.src : reinterpret_cast<const char *>(d.src);
            out[j] = pack_fp8_word(*reinterpret_cast<const uint4 *>(base + (uint64_t)id * d.stride), scale);
        }
    }
    if (m < st.ldm) reinterpret_cast<uint4 *>(st.out)[(size_t)e * st.ldm + m] = make_uint4(out[0], out[1], out[2], out[3]);
    if (bad) atomicOr_system(a.err_flag, 1);
}
static int gather_f_blocks(int K, int ldm) { return ((ldm + 63) / 64) * (((K + 63) / 64 * 4 + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES); }

__device__ __forceinline__ void fc_f_body(const FrStageArgs &st, int local, float *red) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int N = st.N, ldm = st.ldm;
    const int tiles_n = N / 32, tiles_m = ldm / 32, tiles = tiles_n * tiles_m;
    int n_tile, m_tile;
    const int MG = (tiles_m % 2 == 0) ? 2 : 1, NG = 8 / MG;
    if (tiles_n % NG == 0) {  // same XCD-aware map as the fp32 body
        const int x = local & 7, j = local >> 3;
        const int mg = x % MG, ng = x / MG;
        const int tn_x = tiles_n / NG;
        if (j >= tn_x * (tiles_m / MG)) return;
        n_tile = (j % tn_x) * NG + ng;
        m_tile = (j / tn_x) * MG + mg;
    } else {
        if (local >= tiles) return;
        n_tile = local % tiles_n;
        m_tile = local / tiles_n;
    }
    const int n0 = n_tile * 32, m0 = m_tile * 32;
    const int hk = lane >> 5, lm = lane & 31;
    const int groups = (st.K + 63) / 64;  // one MFMA (64 k) per group
    const int per = (groups + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int g_begin = wave * per;
    int ng_ = groups - g_begin;
    ng_ = ng_ < 0 ? 0 : (ng_ > per ? per : ng_);
    const unsigned KE = (unsigned)groups * 4u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.w), 0, KE * (unsigned)N * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(st.in), 0, KE * (unsigned)ldm * 16u, 0x00020000);
    // lane half h owns q16 rows 4g + 2h and 4g + 2h + 1
    const unsigned voA = ((unsigned)(2 * hk) * N + n0 + lm) * 16u, voB = ((unsigned)(2 * hk) * ldm + m0 + lm) * 16u;
    const unsigned voA2 = voA + (unsigned)N * 16u, voB2 = voB + (unsigned)ldm * 16u;
    const unsigned stepA = 4u * (unsigned)N * 16u, stepB = 4u * (unsigned)ldm * 16u;
    const int sc_a = 127 - st.e_w, sc_b = 127 - st.e_in;  // E8M0: 2^(code - 127)
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    auto ld8 = [&](__amdgpu_buffer_rsrc_t rs, unsigned v0, unsigned v1, unsigned so) {
        const uint4 lo = bload4u(rs, v0, so), hi = bload4u(rs, v1, so);
        i32x8 r;
        r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
        r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
        return r;
    };
    constexpr int D = 2;
    i32x8 ra[D], rb[D], na[D], nb[D];
    const int nb_full = ng_ / D;
    if (nb_full > 0) {
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = ld8(rsA, voA, voA2, (unsigned)(g_begin + i) * stepA);
            rb[i] = ld8(rsB, voB, voB2, (unsigned)(g_begin + i) * stepB);
        }
    }
    for (int blk = 0; blk < nb_full; blk++) {
        const int nx = (blk + 1 < nb_full) ? (blk + 1) : blk;
#pragma unroll
        for (int i = 0; i < D; i++) {
            na[i] = ld8(rsA, voA, voA2, (unsigned)(g_begin + nx * D + i) * stepA);
            nb[i] = ld8(rsB, voB, voB2, (unsigned)(g_begin + nx * D + i) * stepB);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ra[i], rb[i], acc, 0, 0, 0, sc_a, 0, sc_b);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < D; i++) {
            ra[i] = na[i];
            rb[i] = nb[i];
        }
    }
    for (int g = g_begin + nb_full * D; g < g_begin + ng_; g++) {
        const i32x8 a8 = ld8(rsA, voA, voA2, (unsigned)g * stepA), b8 = ld8(rsB, voB, voB2, (unsigned)g * stepB);
        acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 0, 0, 0, sc_a, 0, sc_b);
    }
#pragma unroll
    for (int r = 0; r < 16; r++) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // fixed-order fp32 sum, ONE quantisation to e4m3 (x 2^e_out, saturated); 4 consecutive n = 4 bytes of a q16 element
    uint32_t *Yf = reinterpret_cast<uint32_t *>(st.out);
    const float oscale = __builtin_ldexpf(1.0f, st.e_out);
    if (threadIdx.x < 4 * 64) {
        const int i = threadIdx.x >> 6, l = threadIdx.x & 63;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float s = red[(4 * i + c) * 64 + l];
#pragma unroll
            for (int w = 1; w < FR_PIPE_WAVES; w++) s += red[(w * 16 + 4 * i + c) * 64 + l];
            v[c] = s;
        }
        const int n = n0 + 8 * i + 4 * (l >> 5);  // + c
        Yf[((size_t)(n >> 4) * ldm + m0 + (l & 31)) * 4 + ((n & 15) >> 2)] = pack_fp8x4(v[0], v[1], v[2], v[3], oscale);
    }
}

// stage 4: score[m] = 2^-e_in * sum_k w[k] * e4m3(R3)[k][m]; fp32 master weights, 64 items x 8 slices of q16 rows
__device__ __forceinline__ void fc_out_f_body(const FrStageArgs &st, int local, float *red) {
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = local * 64 + lane;
    const int KE = st.K / 16, ldm = st.ldm;  // hidden widths are multiples of 32: no pad rows here
    const int per = (KE + FR_PIPE_WAVES - 1) / FR_PIPE_WAVES;
    const int h0 = q * per;
    const int h1 = (h0 + per) < KE ? (h0 + per) : KE;
    const uint4 *Rf = reinterpret_cast<const uint4 *>(st.in);
    float s = 0.0f;
    if (m < ldm) {
        for (int h = h0; h < h1; h++) {
            const uint4 r = Rf[(size_t)h * ldm + m];
            const int rr[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
            const float *w = st.w + 16 * h;  // wave-uniform
#pragma unroll
            for (int c = 0; c < 4; c++) {
                s = fmaf(w[4 * c + 0], __builtin_amdgcn_cvt_f32_fp8(rr[c], 0), s);
                s = fmaf(w[4 * c + 1], __builtin_amdgcn_cvt_f32_fp8(rr[c], 1), s);
                s = fmaf(w[4 * c + 2], __builtin_amdgcn_cvt_f32_fp8(rr[c], 2), s);
                s = fmaf(w[4 * c + 3], __builtin_amdgcn_cvt_f32_fp8(rr[c], 3), s);
            }
        }
    }
    red[q * 64 + lane] = s;
    __syncthreads();
    if (q == 0 && m < st.batch) {
        float t = red[lane];
#pragma unroll
        for (int i = 1; i < FR_PIPE_WAVES; i++) t += red[i * 64 + lane];
        st.out[m] = t * __builtin_ldexpf(1.0f, -st.e_in);
    }
}

// STAGE = -1: all stages of one pipelined launch; STAGE = 0..4: that stage alone (separately named kernels so
// that rocprof attributes time per stage when a batch is run unpipelined).
// PREC: 0 = fp32 chain (q4 operands, exact-f32 MFMA), 1 = bf16 chain (q8 operands, bf16 MFMA, fp32 accumulate),
// 2 = fp8 chain (q16 e4m3 operands, scaled f8f6f4 MFMA, fp32 accumulate).
template <int STAGE, int PREC>
__global__ void __launch_bounds__(FR_PIPE_THREADS) fr_pipeline_kernel(const FrPipeArgs a) {
    __shared__ uint4 smem[FR_GT_ITEMS * FR_GT_WORDS];  // 32 KiB: the gather tile, or float red[8][16][64] of the FC stages
    float *red = reinterpret_cast<float *>(smem);
    const int b = blockIdx.x;
    int s = 0;
    if constexpr (STAGE >= 0) {
        s = STAGE;
    } else {
#pragma unroll
        for (int i = 1; i < FR_N_STAGES; i++) s += (b >= a.st[i].block_begin) ? 1 : 0;
    }
    const FrStageArgs &st = a.st[s];
    const int local = b - st.block_begin;
    unsigned long long t_in = 0;
    if (a.stamps) t_in = __builtin_amdgcn_s_memrealtime();  // diagnostics only; the values never feed an output
    if (s == 0 && st.variant == 1) {
        gather_tr_body<PREC>(a, st, local, smem);
    } else if (s == 0 && st.variant == 2) {
        if constexpr (PREC != 0) {
            if (a.src_lp) gather_tr_stream_lp_body<PREC, 2, 16>(a, st, local, smem);
            else gather_tr_stream_body<PREC, 2, 16>(a, st, local, smem);
        } else {
            gather_tr_stream_body<PREC, 2, 16>(a, st, local, smem);
        }
    } else if constexpr (PREC == 1) {
        if (s == 0) gather_h_body(a, st, local);
        else if (s == 4) fc_out_h_body(st, local, red);
        else fc_h_body(st, local, red);
    } else if constexpr (PREC == 2) {
        if (s == 0) gather_f_body(a, st, local);
        else if (s == 4) fc_out_f_body(st, local, red);
        else fc_f_body(st, local, red);
    } else {
        if (s == 0) {
            gather_q_body(a, st, local);
        } else if (s == 4) {
            fc_out_q_body(st, local, red);
        } else if (st.nparts_in == 2) {
            fc_q_body<true>(st, local, red);
        } else {
            fc_q_body<false>(st, local, red);
        }
    }
    if (a.stamps) {
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(hwid));
            unsigned long long *o = a.stamps + 4ull * b;
            o[0] = t_in;
            o[1] = __builtin_amdgcn_s_memrealtime();
            o[2] = (unsigned long long)s;
            o[3] = hwid;
        }
    }
}

// The gather | out launch of a chain whose FC layers run as GEMM kernels (Model-C at batch 4096), as a kernel of its own with a small
// register budget: fr_pipeline_kernel<-1, P> carries the FC stage bodies (86-90 VGPRs), and two waves of that size per SIMD do not fit
// beside a resident workgroup of the bf16 FC1 GEMM (2 x 200 of a SIMD's 512 registers: 112 left), so the gather could only start on a CU
// once its GEMM workgroup had gone.  FR_GO_VGPRS registers per wave (2 x 56 = 112) lets the two kernels share the CU.
#ifndef FR_GO_VGPRS
#define FR_GO_VGPRS 56
#endif
template <int PREC>
__global__ void __launch_bounds__(FR_PIPE_THREADS) __attribute__((amdgpu_num_vgpr(FR_GO_VGPRS))) fr_gather_out_kernel(const FrPipeArgs a) {
    __shared__ uint4 smem[FR_GT_ITEMS * FR_GT_WORDS];
    float *red = reinterpret_cast<float *>(smem);
    const int b = blockIdx.x;
    if (b < a.st[1].block_begin) {   // stage 0 (stages 1-3 are empty in this launch: their begins equal stage 4's)
        if constexpr (PREC != 0) {
            if (a.src_lp) gather_tr_stream_lp_body<PREC, 2, 16>(a, a.st[0], b - a.st[0].block_begin, smem);   // rows already in the operand type
            else gather_tr_stream_body<PREC, 2, 16>(a, a.st[0], b - a.st[0].block_begin, smem);
        } else {
            gather_tr_stream_body<PREC, 2, 16>(a, a.st[0], b - a.st[0].block_begin, smem);
        }
    } else {
        const FrStageArgs &st = a.st[4];
        const int local = b - st.block_begin;
        if constexpr (PREC == 1) fc_out_h_body(st, local, red);
        else if constexpr (PREC == 2) fc_out_f_body(st, local, red);
        else fc_out_q_body(st, local, red);
    }
}

// Launch one pipeline step.  `single_stage` >= 0 launches only that stage (its block_begin must be 0).
template <int PREC>
static int pipeline_launch_prec(const FrPipeArgs &a, int single_stage, hipStream_t s) {
    dim3 grid(a.n_blocks), block(FR_PIPE_THREADS);
    const int light = FR_KNOB_ONCE("GATHER_OUT_KERNEL", 1);  // experiment knob: 0 = always fr_pipeline_kernel<-1>
    if (single_stage == -1 && light && a.st[0].variant == 2 && a.st[1].block_begin > a.st[0].block_begin && a.st[1].block_begin == a.st[2].block_begin &&
        a.st[2].block_begin == a.st[3].block_begin && a.st[3].block_begin == a.st[4].block_begin) {
        fr_gather_out_kernel<PREC><<<grid, block, 0, s>>>(a);
        KCHECK();
        fr_note_kernel("fr_gather_out_kernel<%d>", PREC);
        return FR_OK;
    }
    switch (single_stage) {
        case -1: fr_pipeline_kernel<-1, PREC><<<grid, block, 0, s>>>(a); break;
        case 0: fr_pipeline_kernel<0, PREC><<<grid, block, 0, s>>>(a); break;
        case 1: fr_pipeline_kernel<1, PREC><<<grid, block, 0, s>>>(a); break;
        case 2: fr_pipeline_kernel<2, PREC><<<grid, block, 0, s>>>(a); break;
        case 3: fr_pipeline_kernel<3, PREC><<<grid, block, 0, s>>>(a); break;
        case 4: fr_pipeline_kernel<4, PREC><<<grid, block, 0, s>>>(a); break;
        default: FR_FAIL(FR_ERR_INVALID, "bad stage %d", single_stage);
    }
    KCHECK();
    fr_note_kernel("fr_pipeline_kernel<%d, %d>", single_stage, PREC);
    return FR_OK;
}

int frk_pipeline_launch(const FrPipeArgs &a, int single_stage, int precision, hipStream_t s) {
    if (a.n_blocks <= 0) return FR_OK;
    if (precision == FR_FC_FP8) return pipeline_launch_prec<2>(a, single_stage, s);
    return precision == FR_FC_BF16 ? pipeline_launch_prec<1>(a, single_stage, s) : pipeline_launch_prec<0>(a, single_stage, s);
}

int frk_stage_blocks_f8_gather(int K, int ldm) { return pad8(gather_f_blocks(K, ldm)); }

int frk_stage_blocks(int stage, int n_words, int K, int N, int ldm, int nsplit) {
    // every stage is padded to a multiple of 8 workgroups so that the next one starts on XCD group 0
    if (stage == 0) return pad8(gather_q_blocks(n_words, ldm));
    if (stage == 4) return pad8((ldm + 63) / 64);
    return pad8(fc_q_blocks(N, ldm, nsplit));
}



// ---- operand-type bank image: rows of an fp32 region -> the same rows as bf16 / e4m3 (fr_api.cpp lp_ensure_image) ------------------------
// One thread per (row, 16-byte fp32 word): the rounding functions are the gather's own (pack_bf16x2: RNE; pack_fp8_word: x 2^e_x,
// saturated, e4m3), so a row converted here and gathered as it is equals the fp32 row gathered and converted then, bit for bit.
template <int PREC>
__global__ void __launch_bounds__(256) convert_rows_lp_kernel(const char *src, size_t src_stride, char *dst, size_t dst_stride, long long rows, int words, float scale) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = i / words;
    const int w = (int)(i - r * words);
    if (r >= rows) return;
    const uint4 v = *reinterpret_cast<const uint4 *>(src + (size_t)r * src_stride + (size_t)w * 16);
    if constexpr (PREC == 1) {
        *reinterpret_cast<uint2 *>(dst + (size_t)r * dst_stride + (size_t)w * 8) =
            make_uint2(pack_bf16x2(__uint_as_float(v.x), __uint_as_float(v.y)), pack_bf16x2(__uint_as_float(v.z), __uint_as_float(v.w)));
    } else {
        *reinterpret_cast<uint32_t *>(dst + (size_t)r * dst_stride + (size_t)w * 4) = pack_fp8_word(v, scale);
    }
}

int frk_convert_rows_lp(int precision, const void *src, size_t src_stride, void *dst, size_t dst_stride, int64_t rows, int floats, int e_x, hipStream_t s) {
    if (rows <= 0 || floats <= 0) return FR_OK;
    if (floats % 4) FR_FAIL(FR_ERR_INVALID, "internal: a table row of %d floats is not whole 16-byte words", floats);
    const int words = floats / 4;
    const long long total = (long long)rows * words;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) FR_FAIL(FR_ERR_INVALID, "internal: %lld rows x %d words exceed one conversion launch", (long long)rows, words);
    if (precision == FR_FC_BF16)
        convert_rows_lp_kernel<1><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(static_cast<const char *>(src), src_stride, static_cast<char *>(dst), dst_stride, rows, words, 1.0f);
    else if (precision == FR_FC_FP8)
        convert_rows_lp_kernel<2><<<dim3((unsigned)blocks), dim3(256), 0, s>>>(static_cast<const char *>(src), src_stride, static_cast<char *>(dst), dst_stride, rows, words,
                                                                                  ldexpf(1.0f, e_x));
    else
        FR_FAIL(FR_ERR_INVALID, "internal: no operand-type rows for precision %d", precision);
    KCHECK();
    return FR_OK;
}
