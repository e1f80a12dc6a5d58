// gfx950 (MI355X / CDNA4) kernels of the FleetRec hot path + their launchers.
//
//   fill_*        device-side table / weight synthesis (the FPGA host's init_vectors,
//                 FPGA/host/embedding_47_krnl/host.cpp:66-88, without a 63 GB host staging copy)
//   gather_pack   per-table embedding row gather + bit-copy concat into the per-item record
//                 (load_single_embedding_*_tables + group_* + gather_embeddings of
//                 FPGA/kernel/user_krnl/embedding_{47,98,377}_krnl/src/hls/embedding_*_krnl.cpp)
//   gather_t      the same gather, written feature-major (Xt[k][item]) for the FC chain
//   fc_t          one column-major GEMM of the chain R = W * X on the exact-f32 MFMA, split-K inside the
//                 workgroup (cublasLtMatmul, GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c:468-485)
//   fc_out_t      the OUT == 1 layer (cuda_server.c:486-491), a per-item dot product
//
// Wavefront = 64 lanes everywhere; no CUDA-compat shims.
#include <cstdlib>

#include <hip/hip_bf16.h>
#include <hip/hip_runtime.h>

#include "fr_internal.h"

#define KCHECK()                                                                          \
    do {                                                                                  \
        hipError_t e_ = hipGetLastError();                                                \
        if (e_ != hipSuccess) {                                                           \
            fr_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return FR_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

// ---------------------------------------------------------------------------------------------------
// Procedural contents.  Bit-for-bit the same functions as oracle/fleetrec_oracle.c content_bits().
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__device__ __forceinline__ uint32_t content_bits(int mode, uint32_t h_seed_uid, uint32_t uid, uint64_t row, uint32_t col) {
    if (mode == FR_FILL_EVEN_ODD) return (row & 1) ? 0u : 0x3F800000u;
    if (mode == FR_FILL_TAGGED) {
        uint32_t source = uid >> 10, cls = (uid >> 8) & 3, tid = uid & 255;
        return (source << 31) | (cls << 29) | (tid << 21) | ((uint32_t)(row & 0xFFFF) << 5) | (col & 31);
    }
    uint32_t h = fmix32(h_seed_uid ^ (uint32_t)row);
    h = fmix32(h ^ (uint32_t)(row >> 32) ^ (col * 0x27D4EB2Fu));
    float v = (float)(int32_t)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    return __float_as_uint(v);
}

// one thread per 16-byte word, grid-stride; stores are 16 B/lane fully coalesced
__global__ void __launch_bounds__(256) fill_table_kernel(uint4 *base, uint64_t n_words, uint32_t words_per_row, int mode,
                                                          uint32_t seed, uint32_t uid) {
    const uint32_t h0 = fmix32(seed ^ (uid * 0x9E3779B1u));
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint64_t row = w / words_per_row;
        const uint32_t c0 = (uint32_t)(w - row * words_per_row) * 4;
        uint4 v;
        v.x = content_bits(mode, h0, uid, row, c0 + 0);
        v.y = content_bits(mode, h0, uid, row, c0 + 1);
        v.z = content_bits(mode, h0, uid, row, c0 + 2);
        v.w = content_bits(mode, h0, uid, row, c0 + 3);
        base[w] = v;
    }
}

int frk_fill_table(float *base, int64_t rows, int dim, int mode, uint32_t seed, uint32_t uid, hipStream_t s) {
    const uint64_t n_words = (uint64_t)rows * (uint64_t)(dim / 4);
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) return FR_OK;
    fill_table_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((uint4 *)base, n_words, (uint32_t)(dim / 4), mode, seed, uid);
    KCHECK();
    return FR_OK;
}

__global__ void __launch_bounds__(256) fill_weights_kernel(float *w, uint64_t n, int mode, uint32_t seed, uint32_t layer, float scale) {
    const uint32_t h0 = fmix32(seed ^ ((layer + 1u) * 0x9E3779B1u));
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = 1.0f;
        if (mode == FR_WEIGHTS_UNIFORM) {
            uint32_t h = fmix32(h0 ^ (uint32_t)i);
            h = fmix32(h ^ (uint32_t)(i >> 32));
            v = ((float)(int32_t)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
        }
        w[i] = v;
    }
}

int frk_fill_weights(float *w, size_t count, int mode, uint32_t seed, uint32_t layer, float scale, hipStream_t s) {
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return FR_OK;
    fill_weights_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(w, count, mode, seed, layer, scale);
    KCHECK();
    return FR_OK;
}


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
template <int ITEMS>
__global__ void __launch_bounds__(256) gather_pack_kernel(const FrWordDesc *__restrict__ words, int n_words,
                                                          const int32_t *__restrict__ idx, int idx_stride,
                                                          const float *__restrict__ dense, uint4 *__restrict__ out,
                                                          int batch, int *__restrict__ err_flag) {
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
        if (b < batch) out[blk + (size_t)b * dst_stride] = v[i];
    }
    if (bad) atomicOr_system(err_flag, 1);  // pinned host word; error path only
}

// XCD-partitioned form for wide records and large batches: workgroup b serves XCD group b % 8 (workgroups are dealt
// round-robin over the 8 XCDs; a speed assumption only), and each group owns a fixed contiguous 1/8 of the record words for
// ALL items.  Every table is then touched from ONE XCD only, so the 8 x 4 MiB L2s cache 8 different table sets instead of
// 8 copies of the same hottest 4 MiB -- rows served by L2 cost ~5 cycles/CU instead of ~12 from the fabric
// (profiles/r01_experiments.md, ta_cost2).
template <int ITEMS>
__global__ void __launch_bounds__(256) gather_pack_xcd_kernel(const FrWordDesc *__restrict__ words, int n_words, int words_per_group,
                                                              const int32_t *__restrict__ idx, int idx_stride,
                                                              const float *__restrict__ dense, uint4 *__restrict__ out,
                                                              int batch, int *__restrict__ err_flag) {
    const int group = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    const int w = group * words_per_group + threadIdx.x;
    if ((int)threadIdx.x >= words_per_group || w >= n_words) return;
    const uint4 d0 = reinterpret_cast<const uint4 *>(words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(words)[2 * w + 1];
    const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
    const uint32_t stride = d0.z, idx_col = d0.w;
    const uint32_t rows = d1.x, dst_off = d1.y, dst_stride = d1.z, dst_blk = d1.w;
    const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
    const char *base = is_dense ? reinterpret_cast<const char *>(dense) + src : reinterpret_cast<const char *>(src);
    const int b0 = chunk * ITEMS;
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
        if (!is_dense && id[i] >= rows) {
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
        if (b < batch) out[blk + (size_t)b * dst_stride] = v[i];
    }
    if (bad) atomicOr_system(err_flag, 1);
}

int frk_gather(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, float *out,
               int batch, int *err_flag, hipStream_t s) {
    if (n_words <= 0 || batch <= 0) return FR_OK;
    static const int force = getenv("FR_GATHER_XCD") ? atoi(getenv("FR_GATHER_XCD")) : -1;  // experiment knob
    const bool xcd = force >= 0 ? force != 0 : (n_words >= 512 && batch >= 1024);
    if (xcd) {
        constexpr int ITEMS = 8;
        const int wpg = (n_words + 7) / 8;
        if (wpg <= 256) {
            const int bx = ((wpg + 63) / 64) * 64;
            dim3 grid(8 * ((batch + ITEMS - 1) / ITEMS));
            gather_pack_xcd_kernel<ITEMS><<<grid, dim3(bx), 0, s>>>(words, n_words, wpg, idx, idx_stride, dense, (uint4 *)out, batch, err_flag);
            KCHECK();
            return FR_OK;
        }
    }
    // block width: whole waves, at most 256 lanes
    int bx = n_words >= 256 ? 256 : ((n_words + 63) / 64) * 64;
    dim3 block(bx);
    if (batch >= 2048) {
        constexpr int ITEMS = 8;
        dim3 grid((n_words + bx - 1) / bx, (batch + ITEMS - 1) / ITEMS);
        gather_pack_kernel<ITEMS><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, (uint4 *)out, batch, err_flag);
    } else {
        constexpr int ITEMS = 4;
        dim3 grid((n_words + bx - 1) / bx, (batch + ITEMS - 1) / ITEMS);
        gather_pack_kernel<ITEMS><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, (uint4 *)out, batch, err_flag);
    }
    KCHECK();
    return FR_OK;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi);
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

int frk_gather_tr_blocks(int n_words, int ldm) {
    if (ldm % FR_GT_ITEMS || (n_words & 1)) return 0;
    const int blocks = (ldm / FR_GT_ITEMS) * ((n_words + FR_GT_WORDS - 1) / FR_GT_WORDS);
    static const int forced = getenv("FR_GATHER_TR") ? atoi(getenv("FR_GATHER_TR")) : -1;  // experiment knob
    if (forced == 0) return 0;
    if (forced == 1) return blocks;
    return blocks >= 128 ? blocks : 0;  // needs enough workgroups to cover the chip; small batches keep the simple form
}

static int gather_q_blocks(int n_words, int ldm) { return ((ldm + 63) / 64) * ((n_words + FR_PIPE_WAVES * 2 - 1) / (FR_PIPE_WAVES * 2)); }

// ---- stages 1..3: one 32(n) x 32(m) output tile per workgroup, 8 waves split the workgroup's K range in groups of
// 8 k (one 16-byte load per operand per lane -> 4 MFMAs).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bload4u(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
// 16 bytes per lane through a buffer resource: per-lane byte offset in a VGPR, wave-uniform byte offset in an SGPR
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

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
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d, float scale) {
    const float lim = 448.0f;  // largest finite e4m3fn
    a = fminf(fmaxf(a * scale, -lim), lim);
    b = fminf(fmaxf(b * scale, -lim), lim);
    c = fminf(fmaxf(c * scale, -lim), lim);
    d = fminf(fmaxf(d * scale, -lim), lim);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t pack_fp8_word(const uint4 &v, float scale) {
    return pack_fp8x4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), scale);
}

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

// Launch one pipeline step.  `single_stage` >= 0 launches only that stage (its block_begin must be 0).
template <int PREC>
static int pipeline_launch_prec(const FrPipeArgs &a, int single_stage, hipStream_t s) {
    dim3 grid(a.n_blocks), block(FR_PIPE_THREADS);
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

// Reference weight layout (column-major H x K, W[h + k*H], cuda_server.c:215) -> Wq[k/4][h][k%4]
__global__ void __launch_bounds__(256) pack_weights_q4_kernel(const float *__restrict__ W, float4 *__restrict__ Wq, int K, int H) {
    const size_t n = (size_t)(K / 4) * H;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t kq = e / H, h = e - kq * H;
        Wq[e] = make_float4(W[h + (4 * kq + 0) * H], W[h + (4 * kq + 1) * H], W[h + (4 * kq + 2) * H], W[h + (4 * kq + 3) * H]);
    }
}

int frk_pack_weights_q4(const float *W, float *Wq, int K, int H, hipStream_t s) {
    if (K % 4) FR_FAIL(FR_ERR_INVALID, "pack_weights_q4 needs K %% 4 == 0 (K=%d)", K);
    size_t n = (size_t)(K / 4) * H;
    unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    pack_weights_q4_kernel<<<dim3(blocks), dim3(256), 0, s>>>(W, reinterpret_cast<float4 *>(Wq), K, H);
    KCHECK();
    return FR_OK;
}

// item-major records [B][K] -> Xq[K/4][ldm][4]: a transpose of 16-byte elements (only used by the fc_only diagnostic
// entry point and the BLOCKED layout)
__global__ void __launch_bounds__(256) transpose_records_kernel(const float4 *__restrict__ X, float4 *__restrict__ Xq, int batch, int KQ, int ldm) {
    __shared__ float4 tile[16][17];
    const int q0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16
    {
        const int m = m0 + ty, q = q0 + tx;
        tile[ty][tx] = (m < batch && q < KQ) ? X[(size_t)m * KQ + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    {
        const int q = q0 + ty, m = m0 + tx;
        if (q < KQ && m < ldm) Xq[(size_t)q * ldm + m] = tile[tx][ty];
    }
}

int frk_transpose_records(const float *X, float *Xq, int batch, int K, int ldm, hipStream_t s) {
    dim3 grid((K / 4 + 15) / 16, (ldm + 15) / 16);
    transpose_records_kernel<<<grid, dim3(256), 0, s>>>(reinterpret_cast<const float4 *>(X), reinterpret_cast<float4 *>(Xq), batch, K / 4, ldm);
    KCHECK();
    return FR_OK;
}

// Sharded mode: all-gathered padded slices [G][B][F] (item-major per shard) -> Xq[K/4][ldm][4] for items [item0, item0+n).
// One launch per shard (slice offsets/lengths are host data); a transpose of 16-byte elements like the one above.
__global__ void __launch_bounds__(256) transpose_slice_kernel(const float4 *__restrict__ S /* [B][F/4] of this shard */, int FQ, int item0,
                                                              int n_items, int q_off, int q_len, float4 *__restrict__ Xq, int ldm) {
    __shared__ float4 tile[16][17];
    const int q0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    {
        const int m = m0 + ty, q = q0 + tx;
        tile[ty][tx] = (m < n_items && q < q_len) ? S[(size_t)(item0 + m) * FQ + q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    {
        const int q = q0 + ty, m = m0 + tx;
        if (q < q_len && m < ldm) Xq[(size_t)(q_off + q) * ldm + m] = tile[tx][ty];
    }
}

int frk_transpose_slices(const float *gathered, int n_shards, int batch_total, int slice_padded, const int *h_offsets, const int *h_lens,
                         int item0, int n_items, float *Xq, int ldm, hipStream_t s) {
    if (slice_padded % 4) FR_FAIL(FR_ERR_INVALID, "slice_padded %d must be a multiple of 4", slice_padded);
    const int FQ = slice_padded / 4;
    for (int g = 0; g < n_shards; g++) {
        const int q_len = h_lens[g] / 4;
        if (q_len == 0) continue;
        dim3 grid((q_len + 15) / 16, (ldm + 15) / 16);
        transpose_slice_kernel<<<grid, dim3(256), 0, s>>>(reinterpret_cast<const float4 *>(gathered) + (size_t)g * batch_total * FQ, FQ, item0,
                                                         n_items, h_offsets[g] / 4, q_len, reinterpret_cast<float4 *>(Xq), ldm);
    }
    KCHECK();
    return FR_OK;
}

// fp32 master weights (column-major H x K) -> Wh[k/8][h][k%8] bf16 (RNE)
__global__ void __launch_bounds__(256) pack_weights_q8_bf16_kernel(const float *__restrict__ W, uint4 *__restrict__ Wh, int K, int H) {
    const size_t n = (size_t)(K / 8) * H;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t ko = e / H, h = e - ko * H;
        uint4 o;
        o.x = pack_bf16x2(W[h + (8 * ko + 0) * H], W[h + (8 * ko + 1) * H]);
        o.y = pack_bf16x2(W[h + (8 * ko + 2) * H], W[h + (8 * ko + 3) * H]);
        o.z = pack_bf16x2(W[h + (8 * ko + 4) * H], W[h + (8 * ko + 5) * H]);
        o.w = pack_bf16x2(W[h + (8 * ko + 6) * H], W[h + (8 * ko + 7) * H]);
        Wh[e] = o;
    }
}

int frk_pack_weights_q8_bf16(const float *W, uint16_t *Wh, int K, int H, hipStream_t s) {
    if (K % 8) FR_FAIL(FR_ERR_INVALID, "pack_weights_q8 needs K %% 8 == 0 (K=%d)", K);
    size_t n = (size_t)(K / 8) * H;
    unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    pack_weights_q8_bf16_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(W, reinterpret_cast<uint4 *>(Wh), K, H);
    KCHECK();
    return FR_OK;
}

// fp32 master weights (column-major H x K) -> Wf[k/16][h][k%16] e4m3(W * scale), K zero-padded to KP (multiple of 64)
__global__ void __launch_bounds__(256) pack_weights_q16_fp8_kernel(const float *__restrict__ W, uint4 *__restrict__ Wf, int K, int KP, int H, float scale) {
    const size_t n = (size_t)(KP / 16) * H;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t ke = e / H, h = e - ke * H;
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const size_t k = 16 * ke + 4 * j + c;
                v[c] = (k < (size_t)K) ? W[h + k * H] : 0.0f;
            }
            o[j] = pack_fp8x4(v[0], v[1], v[2], v[3], scale);
        }
        Wf[e] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int frk_pack_weights_q16_fp8(const float *W, void *Wf, int K, int H, int e_w, hipStream_t s) {
    const int KP = (K + 63) / 64 * 64;
    size_t n = (size_t)(KP / 16) * H;
    unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    pack_weights_q16_fp8_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(W, reinterpret_cast<uint4 *>(Wf), K, KP, H, ldexpf(1.0f, e_w));
    KCHECK();
    return FR_OK;
}

// item-major fp32 records [B][K] -> Xf[KP/16][ldm][16] e4m3(x * scale) (fc_only diagnostic / BLOCKED layout in fp8 mode)
__global__ void __launch_bounds__(256) records_to_q16_fp8_kernel(const float *__restrict__ X, uint4 *__restrict__ Xf, int batch, int K, int KE, int ldm, float scale) {
    const size_t n = (size_t)KE * ldm;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int ke = (int)(e / ldm), m = (int)(e - (size_t)ke * ldm);
        uint32_t o[4] = {0u, 0u, 0u, 0u};
        if (m < batch) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int k = 16 * ke + 4 * j;
                if (k < K) {  // K % 4 == 0
                    const float4 v = *reinterpret_cast<const float4 *>(X + (size_t)m * K + k);
                    o[j] = pack_fp8x4(v.x, v.y, v.z, v.w, scale);
                }
            }
        }
        Xf[e] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int frk_records_to_q16_fp8(const float *X, void *Xf, int batch, int K, int ldm, int e_x, hipStream_t s) {
    const int KE = (K + 63) / 64 * 4;
    size_t n = (size_t)KE * ldm;
    unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    records_to_q16_fp8_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(X, reinterpret_cast<uint4 *>(Xf), batch, K, KE, ldm, ldexpf(1.0f, e_x));
    KCHECK();
    return FR_OK;
}

// max |x| and sum x^2 of a float array (fp8 scale selection): out[0] = bits of max |x| (atomicMax on the uint pattern), out[1] = sum
__global__ void __launch_bounds__(256) stats_kernel(const float *__restrict__ p, size_t n, unsigned *out_max, float *out_sumsq) {
    float mx = 0.0f, ss = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = p[i];
        if (v == v) {  // NaN never drives a scale
            mx = fmaxf(mx, fabsf(v));
            ss = fmaf(v, v, ss);
        }
    }
    __shared__ float smx[256], sss[256];
    smx[threadIdx.x] = mx;
    sss[threadIdx.x] = ss;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            smx[threadIdx.x] = fmaxf(smx[threadIdx.x], smx[threadIdx.x + o]);
            sss[threadIdx.x] += sss[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(out_max, __float_as_uint(smx[0]));
        atomicAdd(out_sumsq, sss[0]);
    }
}

// d_out: 2 words (zeroed here); returns after the launch is enqueued
int frk_stats(const float *p, size_t n, void *d_out, hipStream_t s) {
    if (hipMemsetAsync(d_out, 0, 8, s) != hipSuccess) FR_FAIL(FR_ERR_HIP, "hipMemsetAsync failed");
    unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    stats_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(p, n, reinterpret_cast<unsigned *>(d_out), reinterpret_cast<float *>(d_out) + 1);
    KCHECK();
    return FR_OK;
}

// item-major fp32 records [B][K] -> Xh[K/8][ldm][8] bf16 (fc_only diagnostic / BLOCKED layout in bf16 mode)
__global__ void __launch_bounds__(256) records_to_q8_bf16_kernel(const float *__restrict__ X, uint4 *__restrict__ Xh, int batch, int KO, int ldm) {
    const size_t n = (size_t)KO * ldm;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t ko = e / ldm, m = e - ko * ldm;
        uint4 o = make_uint4(0u, 0u, 0u, 0u);
        if ((int)m < batch) {
            const float *x = X + m * (size_t)KO * 8 + ko * 8;
            o.x = pack_bf16x2(x[0], x[1]);
            o.y = pack_bf16x2(x[2], x[3]);
            o.z = pack_bf16x2(x[4], x[5]);
            o.w = pack_bf16x2(x[6], x[7]);
        }
        Xh[e] = o;
    }
}

int frk_records_to_q8_bf16(const float *X, void *Xh, int batch, int K, int ldm, hipStream_t s) {
    size_t n = (size_t)(K / 8) * ldm;
    unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    records_to_q8_bf16_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(X, reinterpret_cast<uint4 *>(Xh), batch, K / 8, ldm);
    KCHECK();
    return FR_OK;
}

// ===================================================================================================
// fc_lp_gemm_kernel<PREC>: LDS-tiled GEMM for low-precision layers that fill the chip on their own (Model-C FC1 at batch 4096).
// PREC 0 = fp32 (q4 elements, v_mfma_f32_32x32x2_f32: the same k-ordered exact-f32 sums as the other fp32 kernels, full K per
// output), PREC 1 = bf16 (q8 elements, v_mfma_f32_32x32x16_bf16), PREC 2 = fp8 (q16 elements, v_mfma_scale_f32_32x32x64_f8f6f4).
// Block tile 128 (n) x 256 (m), 8 waves as 2 x 4 with 64 x 64 wave tiles (2 x 2 MFMA tiles: 4 fragment reads per 4 MFMAs).
// A K step is 8 rows of 16-byte elements (64 k in bf16, 128 k in fp8): 16 KiB of W + 32 KiB of X, three steps in LDS (144 KiB).
// Operands go global -> LDS directly (buffer_load ... lds, 64 consecutive elements per wave-instruction, no VGPR staging) two
// steps ahead of the MFMAs; one barrier per step.  Fragment reads are conflict-free ds_read_b128 (a b128 access is served 16
// lanes at a time, and 16 consecutive elements of a row are 256 contiguous bytes).  L2 -> CU traffic per output is 25 % below
// that of 128 x 128 tiles.
// ===================================================================================================
constexpr int FR_GN = 128, FR_GM = 256, FR_GR = 8;
// STAGES K steps in LDS: 3 = 144 KiB, loads two steps ahead (the workgroup owns its CU); 2 = 96 KiB, one step ahead, leaves room
// for a 32 KiB stage-pipeline workgroup of another stream on the same CU (FR_LP_GEMM_STAGES=2).

template <int PREC, int STAGES>
__global__ void __launch_bounds__(512) fc_lp_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                         int N, int ldm, int sc_a, int sc_b, float oscale) {
    extern __shared__ uint4 glds[];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tn = N / FR_GN, tm = ldm / FR_GM;
    // XCD-aware tile map: workgroup b runs on XCD b % 8.  The XCDs form a 2 (n) x 4 (m) grid and each owns a tn/2 x tm/4 block
    // of tiles, so its L2 sees tn/2 weight panels + tm/4 activation panels instead of (with a linear map) two weight panels and
    // EVERY activation panel: half the traffic from beyond L2 for Model-C FC1.  The workgroups of an XCD walk K in step, so a
    // panel row is fetched once and hit by the others whatever the L2 capacity.
    int n_tile, m_tile;
    if (tn % 2 == 0 && tm % 4 == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int tnx = tn / 2;
        n_tile = (x & 1) * tnx + j % tnx;
        m_tile = (x >> 1) * (tm / 4) + j / tnx;
    } else {
        n_tile = blockIdx.x % tn;
        m_tile = blockIdx.x / tn;
    }
    const int n0 = n_tile * FR_GN, m0 = m_tile * FR_GM;
    const int r = lane & 31, h = lane >> 5;
    // Buffer resources built by hand (SGPR quads for the inline asm below): base, stride 0, bytes, gfx9 raw-buffer flags.
    auto make_rs = [](const void *p, unsigned bytes) {
        const unsigned long long a = (unsigned long long)p;
        i32x4_t rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        rs[3] = 0x00020000;
        return rs;
    };
    const i32x4_t rsW = make_rs(W, (unsigned)KE * (unsigned)N * 16u), rsX = make_rs(X, (unsigned)KE * (unsigned)ldm * 16u);
    auto As = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * (FR_GN + FR_GM); };          // 128 elements
    auto Bs = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * (FR_GN + FR_GM) + FR_GN; };  // 256 elements
    // global -> LDS without a VGPR round trip: lane i's 16 bytes land at M0 + 16 i.  Inline asm on purpose: through the builtin the
    // compiler treats every LDS read as a possible alias of the DMA write and waits for vmcnt(0) before each fragment read, which
    // removes the two-step prefetch distance; the counted s_waitcnt below are the only synchronisation these loads need.
    auto dma = [&](const i32x4_t &rs, const uint4 *dst, unsigned voff, unsigned soff) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr)dst);
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    // staging: 48 wave-instructions of 64 elements per step; every wave issues 2 of W's 16 and 4 of X's 32
    const unsigned vW = (unsigned)(n0 + 64 * (wave & 1) + lane) * 16u, vX = (unsigned)(m0 + lane) * 16u;
    auto issue = [&](int step, int st) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (wave >> 1) + 4 * i;  // 8 waves x 2 = rows 0..7 x 2 halves
            dma(rsW, As(st, row) + 64 * (wave & 1), vW, (unsigned)(step * FR_GR + row) * (unsigned)N * 16u);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)  // row = wave, four quarters
            dma(rsX, Bs(st, wave) + 64 * i, vX + 64u * 16u * i, (unsigned)(step * FR_GR + wave) * (unsigned)ldm * 16u);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
    const int nsteps = KE / FR_GR;
    issue(0, 0);
    if (STAGES == 3 && nsteps > 1) issue(1, 1);
    for (int s = 0; s < nsteps; s++) {
        // this wave's loads of step s have landed once at most the 6 of step s+1 are outstanding
        if (STAGES == 3 && s + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        __syncthreads();  // everyone's step-s data is in LDS, and everyone is done reading the buffer the next issue overwrites
        if (s + STAGES - 1 < nsteps) issue(s + STAGES - 1, (s + STAGES - 1) % STAGES);
        const int st = s % STAGES;
        if constexpr (PREC == 0) {  // q4 fp32 elements: one element per lane feeds four v_mfma_f32_32x32x2_f32 (k = 8 kk + 4 h + c)
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 64 + r;
                const uint4 a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
                const uint32_t a0c[4] = {a0.x, a0.y, a0.z, a0.w}, a1c[4] = {a1.x, a1.y, a1.z, a1.w};
                const uint32_t b0c[4] = {b0.x, b0.y, b0.z, b0.w}, b1c[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a0c[c]), __uint_as_float(b0c[c]), acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a1c[c]), __uint_as_float(b0c[c]), acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a0c[c]), __uint_as_float(b1c[c]), acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a1c[c]), __uint_as_float(b1c[c]), acc[1][1], 0, 0, 0);
                }
            }
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 64 + r;
                const uint4 a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 4; kk++) {
                i32x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint4 alo = As(st, 4 * kk + 2 * h)[wn * 64 + 32 * t + r], ahi = As(st, 4 * kk + 2 * h + 1)[wn * 64 + 32 * t + r];
                    const uint4 blo = Bs(st, 4 * kk + 2 * h)[wm * 64 + 32 * t + r], bhi = Bs(st, 4 * kk + 2 * h + 1)[wm * 64 + 32 * t + r];
                    a[t][0] = (int)alo.x; a[t][1] = (int)alo.y; a[t][2] = (int)alo.z; a[t][3] = (int)alo.w;
                    a[t][4] = (int)ahi.x; a[t][5] = (int)ahi.y; a[t][6] = (int)ahi.z; a[t][7] = (int)ahi.w;
                    b[t][0] = (int)blo.x; b[t][1] = (int)blo.y; b[t][2] = (int)blo.z; b[t][3] = (int)blo.w;
                    b[t][4] = (int)bhi.x; b[t][5] = (int)bhi.y; b[t][6] = (int)bhi.z; b[t][7] = (int)bhi.w;
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int t = 0; t < 2; t++) acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], b[u], acc[t][u], 0, 0, 0, sc_a, 0, sc_b);
            }
        }
    }
    // epilogue: ONE rounding per output; registers 4i..4i+3 of a tile are 4 consecutive n
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const f32x16 &c = acc[t][u];
            const int m = m0 + wm * 64 + 32 * u + r;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int n = n0 + wn * 64 + 32 * t + 8 * i + 4 * h;  // + c
                if constexpr (PREC == 0) {  // q4 element (n / 4) of the next layer's operand
                    reinterpret_cast<float4 *>(Y)[(size_t)(n >> 2) * ldm + m] = make_float4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
                } else if constexpr (PREC == 1) {
                    uint2 hv;
                    hv.x = pack_bf16x2(c[4 * i + 0], c[4 * i + 1]);
                    hv.y = pack_bf16x2(c[4 * i + 2], c[4 * i + 3]);
                    reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
                } else {
                    reinterpret_cast<uint32_t *>(Y)[((size_t)(n >> 4) * ldm + m) * 4 + ((n & 15) >> 2)] =
                        pack_fp8x4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], oscale);
                }
            }
        }
}

// precision: FR_FC_BF16 (K % 64 == 0) or FR_FC_FP8 (K padded to 128 by the caller's layout: KE % 8 == 0)
bool frk_fc_lp_gemm_ok(int precision, int K, int N, int ldm) {
    static const int forced = getenv("FR_LP_GEMM") ? atoi(getenv("FR_LP_GEMM")) : -1;  // experiment knob: 0 = never, 1 = whenever legal
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if ((precision == FR_FC_BF16 && K % 8) || (precision == FR_FC_FP32 && K % 4) || KE % FR_GR || KE / FR_GR < 2 || N % FR_GN || ldm % FR_GM) return false;
    if (forced == 0) return false;
    if (forced == 1) return true;
    static const int min_tiles = getenv("FR_LP_GEMM_MIN_TILES") ? atoi(getenv("FR_LP_GEMM_MIN_TILES")) : 64;
    return (long)(N / FR_GN) * (ldm / FR_GM) >= min_tiles;  // 64 tiles = a quarter of the CUs (Model-C FC2 at batch 4096: runs beside another stream's gather)
}

template <int PREC, int STAGES>
static int lp_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)STAGES * FR_GR * (FR_GN + FR_GM) * 16;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fc_lp_gemm_kernel<PREC, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");
        attr_set = true;
    }
    dim3 grid((N / FR_GN) * (ldm / FR_GM));
    fc_lp_gemm_kernel<PREC, STAGES><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale);
    KCHECK();
    return FR_OK;
}

int frk_fc_lp_gemm(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, hipStream_t s) {
    static const int stages = getenv("FR_LP_GEMM_STAGES") ? atoi(getenv("FR_LP_GEMM_STAGES")) : 2;  // 2: co-resident with other streams' stage kernels
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if (precision == FR_FC_FP32)
        return stages == 2 ? lp_gemm_launch<0, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<0, 3>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    if (precision == FR_FC_FP8) {
        const float os = ldexpf(1.0f, e_out);
        return stages == 2 ? lp_gemm_launch<2, 2>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s)
                           : lp_gemm_launch<2, 3>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s);
    }
    return stages == 2 ? lp_gemm_launch<1, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<1, 3>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
}

// ===================================================================================================
// fr_fused_tile_kernel: the whole hot path of 32 items in ONE workgroup, activations never leave LDS.
//
// The stage pipeline above re-reads every activation and weight panel once per 32x32 output tile: at batch 256 the CUs'
// L2 ingest (not the MFMA pipe) bounds it at ~43 % of the f32 MFMA peak.  Here a workgroup owns 32 items for ALL layers:
//   gather  -> Xq[K/4][32] in LDS (the record of its 32 items; 16-byte q4 elements, row stride 33)
//   FC1     -> in chunks of 256 outputs (one 32-wide n tile per wave, full K, no split-K) -> R1 chunk in LDS
//   FC2     -> every wave keeps its H2/256 n tiles in accumulators and adds each R1 chunk as it appears
//   FC3     -> one n tile per wave from R2 in LDS, out layer from R3 in LDS, 32 scores stored
// Only the weights stream from L2 (each element once per workgroup: 4 MB for Model-A = 38 GB/s per CU at the MFMA rate);
// the B operand of every MFMA is a conflict-free ds_read_b128.  One launch carries the tiles of up to 64 queued batches
// of a worker (8 workgroups per batch of 256): 32 batches put one workgroup on every CU.
// Needs (K/4 + 64) * 33 * 16 B of LDS <= 160 KiB (one R1 buffer; two when they fit), H1 % 256 == 0, H2 in {256, 512}, H3 == 256,
// and K % 32 == 0 unless FC1 has a straight-line instantiation (K = 352, 880).
// ===================================================================================================
constexpr int FR_FT_LD = 33;  // LDS row stride in 16-byte elements (32 items + 1 pad: conflict-free writes and reads)

// acc[t] += Wq[k][n0 + 32 t ..] x B, for groups [g0, g0 + cnt) of 8 k; B group g' = g - g0 + gb0 lives in LDS at
// Bq[(2 g' + hk) * 33 + lm].  Double-buffered blocks of D groups: the next block's weight loads fly during the MFMAs.
template <int NT, int D>
__device__ __forceinline__ void ft_load_a(float4 (&r)[D][NT], const float4 *__restrict__ aq, int N, int g) {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) r[i][t] = aq[(size_t)(2 * (g + i)) * N + 32 * t];
}

// MFMAs of one block.  Consecutive MFMAs of a wave always target different accumulators (the NT tiles when NT >= 2, else two
// partial accumulators for even / odd k that the caller adds once at the end), so that no MFMA waits for its predecessor.
template <int NT, int D>
__device__ __forceinline__ void ft_compute(f32x16 (&acc)[NT], f32x16 &alt, const float4 (&r)[D][NT], const uint4 *bl, int gb) {
    uint4 rb[D];
#pragma unroll
    for (int i = 0; i < D; i++) rb[i] = bl[(size_t)(2 * (gb + i)) * FR_FT_LD];
    __builtin_amdgcn_sched_barrier(0);  // weight loads of the other register set stay ahead of these MFMAs
#pragma unroll
    for (int i = 0; i < D; i++) {
        const float bx = __uint_as_float(rb[i].x), by = __uint_as_float(rb[i].y), bz = __uint_as_float(rb[i].z), bw = __uint_as_float(rb[i].w);
        if constexpr (NT == 1) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].x, bx, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].y, by, alt, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].z, bz, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].w, bw, alt, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].x, bx, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].y, by, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].z, bz, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].w, bw, acc[t], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Compile-time trip count, fully unrolled, weight operands in a RING of R register slots refilled one group at a time:
// group g's MFMAs are followed by the load of group g + R into the slot they just freed.  The loads are spread evenly over
// the MFMA stream (no bursts into the CU's ~70 GB/s L2 ingest) and the prefetch distance is R - 1 groups, i.e. almost the
// whole in-flight window -- a lone wave (its SIMD partner parked at a barrier) no longer exposes L2/Infinity-Cache latency
// once per block.  Straight-line code: hipcc keeps counted vmcnt waits (a branch in here would degrade them to vmcnt(0)).
// Weight operands come through BUFFER loads: one resource descriptor per weight matrix (SGPRs), one constant per-lane byte offset
// (a single VGPR, computed once per kernel), and the group / n-tile position as a uniform SGPR offset.  A global_load with a
// 64-bit per-lane address needs a v_lshl_add_u64 per load and reads two address VGPRs; measured on this loop shape
// (tools/experiments/mfma_loop) that alone stretches the MFMA issue interval from 65 to 75-77 cycles, buffer loads: 68.
struct FtW {
    __amdgpu_buffer_rsrc_t rs;  // base = weight matrix, num_records = its bytes (out-of-range lanes read 0, never fault)
    unsigned voff;              // (hk * N + lm) * 16
    unsigned row2;              // 2 * N * 16: byte step of one group (two q4 rows)
};
__device__ __forceinline__ FtW ft_w(const float4 *wq, int K, int N, int hk, int lm) {
    FtW w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(wq), 0, (unsigned)(K / 4) * (unsigned)N * 16u, 0x00020000);
    w.voff = (unsigned)(hk * N + lm) * 16u;
    w.row2 = 2u * (unsigned)N * 16u;
    return w;
}
__device__ __forceinline__ float4 ft_wload(const FtW &w, unsigned soff, int imm) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + imm, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Prologue of ft_gemm_ct, separated so that the caller can issue it BEFORE the barrier / LDS stores that precede the GEMM: the
// weights do not depend on them, and a GEMM that starts with a cold ring exposes a full L2 round trip on all 8 waves at once.
// n0 and g0 must be wave-uniform (they become the SGPR offset).
template <int NT, int R, int CNT>
__device__ __forceinline__ void ft_ring_fill(float4 (&ring)[R][NT], const FtW &w, int n0, int g0) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
#pragma unroll
    for (int g = 0; g < R && g < CNT; g++)
#pragma unroll
        for (int t = 0; t < NT; t++) ring[g][t] = ft_wload(w, s0 + (unsigned)g * w.row2, 512 * t);
    __builtin_amdgcn_sched_barrier(0);  // the loads stay where the caller put them
}

template <int NT, int R, int CNT>
__device__ __forceinline__ void ft_gemm_ct(f32x16 (&acc)[NT], float4 (&ring)[R][NT], const FtW &w, int n0, const uint4 *Bq, int gb0, int g0, int hk,
                                           int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
    const uint4 *bl = Bq + (size_t)(2 * gb0 + hk) * FR_FT_LD + lm;
    f32x16 alt;
#pragma unroll
    for (int i = 0; i < 16; i++) alt[i] = 0.0f;
    uint4 bcur = bl[0];
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        const uint4 bnext = bl[(size_t)(2 * ((g + 1 < CNT) ? g + 1 : g)) * FR_FT_LD];  // next group's B fragment from LDS
        const float bx = __uint_as_float(bcur.x), by = __uint_as_float(bcur.y), bz = __uint_as_float(bcur.z), bw = __uint_as_float(bcur.w);
        const float4(&a4)[NT] = ring[g % R];
        if constexpr (NT == 1) {  // two partial accumulators (even / odd k): consecutive MFMAs are independent
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].x, bx, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].y, by, alt, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].z, bz, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].w, bw, alt, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].x, bx, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].y, by, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].z, bz, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].w, bw, acc[t], 0, 0, 0);
        }
        if (g + R < CNT) {  // compile-time after unrolling: refill the slot just consumed
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = ft_wload(w, s0 + (unsigned)(g + R) * w.row2, 512 * t);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep this interleave: the scheduler would otherwise sink the refills
        bcur = bnext;
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[0][i] += alt[i];
    }
}

// Run-time trip count (models without a straight-line instantiation): one register set + copies, branch-free loop body.
template <int NT, int D>
__device__ __forceinline__ void ft_gemm(f32x16 (&acc)[NT], const float4 *__restrict__ wq, int N, int n0, const uint4 *Bq, int gb0, int g0,
                                        int cnt, int hk, int lm) {
    const float4 *aq = wq + (size_t)hk * N + n0 + lm;
    const uint4 *bl = Bq + hk * FR_FT_LD + lm;
    float4 ra[D][NT], na[D][NT];
    f32x16 alt;
#pragma unroll
    for (int i = 0; i < 16; i++) alt[i] = 0.0f;
    const int nblk = cnt / D;  // launcher guarantees cnt % D == 0
    ft_load_a<NT, D>(ra, aq, N, g0);
    for (int blk = 0; blk < nblk; blk++) {
        const int nx = (blk + 1 < nblk) ? (blk + 1) : blk;  // last block re-loads itself: branch-free body
        ft_load_a<NT, D>(na, aq, N, g0 + nx * D);
        ft_compute<NT, D>(acc, alt, ra, bl, gb0 + blk * D);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int t = 0; t < NT; t++) ra[i][t] = na[i][t];
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[0][i] += alt[i];
    }
}

// store a 32(n) x 32(m) accumulator tile as q4 elements into an LDS operand image: n_local = tile's first n inside the image
__device__ __forceinline__ void ft_store_tile(uint4 *img, const f32x16 &acc, int n_local, int hk, int lm) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v;
        v.x = __float_as_uint(acc[4 * i + 0]);
        v.y = __float_as_uint(acc[4 * i + 1]);
        v.z = __float_as_uint(acc[4 * i + 2]);
        v.w = __float_as_uint(acc[4 * i + 3]);
        img[(size_t)((n_local >> 2) + 2 * i + hk) * FR_FT_LD + lm] = v;  // n = n_local + 8i + 4hk + c
    }
}

// T2W = H2 / 256 (n tiles of FC2 per wave); KG = K / 8 when FC1 has a straight-line instantiation, else 0 (run-time loop).
// WPE = waves per SIMD the kernel is built for.  2: one workgroup per CU, deep weight rings.  4: TWO workgroups per CU
// (<= 128 VGPRs, <= 80 KiB LDS, half-depth rings -- the same bytes in flight per CU) so that one workgroup's gather / barrier /
// epilogue phases run under the other one's MFMAs.  DB: double-buffered R1 (one barrier per chunk); !DB: single R1 buffer and
// R3 overlaying R2 -- the small-LDS layout that WPE 4 needs and that lets a K = 880 record (Model-B) fit at WPE 2.
template <int T2W, int KG, int WPE, bool DB>
__global__ void __launch_bounds__(512, WPE) fr_fused_tile_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int RD = (WPE == 2) ? 1 : 2;  // ring depth divisor: half-depth rings when two workgroups share a CU's registers
    const int KQ = a.K / 4;
    // LDS: [ Xq: KQ rows | R1 chunk buffer 0: 64 rows | R1 chunk buffer 1: 64 rows ]; R2 (H2/4 rows) overlays the start once
    // Xq and the R1 chunks are dead, R3 (64 rows) follows R2.
    uint4 *Xq = lds;                                  // [KQ][33]
    uint4 *R1b[2] = {Xq + (size_t)KQ * FR_FT_LD, Xq + (size_t)(KQ + (DB ? 64 : 0)) * FR_FT_LD};  // [64][33] each: 256 outputs of FC1
    uint4 *R2 = lds;                                  // [H2/4][33]
    uint4 *R3 = DB ? lds + (size_t)(a.H2 / 4) * FR_FT_LD : lds;  // [64][33]; WPE 4: overlays R2 once FC3 has read it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * 32;
    if (m0 >= bt.batch) return;
    int n_st = 0;
    auto stamp = [&]() {  // diagnostic build aid; wave 0 lane 0 only, values never feed an output
        if (a.stamps && tid == 0 && n_st < 14) {
            a.stamps[16ull * blockIdx.x + n_st] = __builtin_amdgcn_s_memrealtime();
            if (n_st == 1) a.stamps[16ull * blockIdx.x + 14] = __builtin_amdgcn_s_memtime();   // shader-clock cycles at "gather done"
            if (n_st == 11) a.stamps[16ull * blockIdx.x + 15] = __builtin_amdgcn_s_memtime();  // ... and at "FC3 + R3"
        }
        n_st++;
    };
    auto wstamp = [&](int k) {  // per-wave stamps around chunk 1: FC1 done, barrier released, FC2 done, next FC1 done
        if (a.stamps && lane == 0) a.stamps[16ull * gridDim.x + (8ull * blockIdx.x + wave) * 4 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp();
    const FtW W1 = ft_w(a.w1q, a.K, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2, a.H3, hk, lm);
    constexpr int RA = 16 / RD, RB = 24 / T2W / RD;     // ring slots: FC1 / FC3 (one n tile), FC2 (T2W n tiles)
    constexpr int KG1 = KG > 0 ? KG : 1;
    float4 ring1[RA][1];                                // FC1's weight ring; chunk 0's first groups are requested before the gather
    if constexpr (KG > 0) ft_ring_fill<1, RA, KG1>(ring1, W1, 32 * wave, 0);

    // ---- gather: lanes along record words (a row is read by dim/4 adjacent lanes), 4 items per thread ----
    {
        const int wl = tid & 63, ig = tid >> 6;
        bool bad = false;
        for (int w0 = 0; w0 < a.n_words; w0 += 64) {
            const int w = w0 + wl;
            if (w < a.n_words) {
                const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
                const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
                const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
                const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
                const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
                const char *base = is_dense ? reinterpret_cast<const char *>(bt.dense) + src : reinterpret_cast<const char *>(src);
                uint32_t id[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int m = m0 + 4 * ig + i;
                    id[i] = 0;
                    if (m < bt.batch) id[i] = is_dense ? (uint32_t)m : (uint32_t)bt.idx[(size_t)m * a.idx_stride + idx_col];
                    if (!is_dense && id[i] >= rows) {
                        bad = true;
                        id[i] = 0;
                    }
                }
                uint4 v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    Xq[(size_t)w * FR_FT_LD + 4 * ig + i] = (m0 + 4 * ig + i < bt.batch) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();
    stamp();

    // ---- FC1 in chunks of 256 outputs, FC2 accumulating each chunk ----
    f32x16 acc2[T2W];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[t][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1];
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[0][i] = 0.0f;
        if constexpr (KG > 0) ft_gemm_ct<1, RA, KG1>(acc1, ring1, W1, c * 256 + 32 * wave, Xq, 0, 0, hk, lm);
        else ft_gemm<1, 4>(acc1, a.w1q, a.H1, c * 256 + 32 * wave, Xq, 0, 0, a.K / 8, hk, lm);
        stamp();
        if (c == 1) wstamp(0);
        if (c == 2) wstamp(3);
        float4 ring2[RB][T2W];  // FC2's first weight groups are requested before the R1 store and the barrier
        ft_ring_fill<T2W, RB, 32>(ring2, W2, 32 * T2W * wave, 32 * c);
        uint4 *R1 = R1b[c & 1];  // double-buffered: ONE barrier per chunk (the buffer written now was last read two chunks ago)
        if (!DB && c > 0) __syncthreads();  // single buffer: every wave must be done with the previous chunk's FC2
        ft_store_tile(R1, acc1[0], 32 * wave, hk, lm);
        __syncthreads();
        stamp();
        // FC2: K range [256 c, 256 c + 256) = groups [32 c, 32 c + 32); this wave's n tiles start at 32 * T2W * wave
        if (c == 1) wstamp(1);
        ft_gemm_ct<T2W, RB, 32>(acc2, ring2, W2, 32 * T2W * wave, R1, 0, 32 * c, hk, lm);
        if (c == 1) wstamp(2);
        if constexpr (KG > 0) {  // next chunk's FC1 ring (the last chunk re-requests its own: branch-free, 16 harmless loads)
            const int cn = (c + 1 < n_chunks) ? c + 1 : c;
            ft_ring_fill<1, RA, KG1>(ring1, W1, cn * 256 + 32 * wave, 0);
        }
    }
    float4 ring3[RA][1];  // FC3's first weight groups, requested before the two barriers around the R2 store
    ft_ring_fill<1, RA, 32 * T2W>(ring3, W3, 32 * wave, 0);
    __syncthreads();  // every wave is done with Xq and both R1 buffers: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++) ft_store_tile(R2, acc2[t], 32 * (T2W * wave + t), hk, lm);
    __syncthreads();  // R2 complete
    stamp();

    // ---- FC3: one n tile per wave (H3 == 256), then the output layer ----
    f32x16 acc3[1];
#pragma unroll
    for (int i = 0; i < 16; i++) acc3[0][i] = 0.0f;
    ft_gemm_ct<1, RA, 32 * T2W>(acc3, ring3, W3, 32 * wave, R2, 0, 0, hk, lm);  // H2 / 8 = 32 * T2W groups
    if (!DB) __syncthreads();                       // R3 overlays R2: every wave must have finished reading R2
    ft_store_tile(R3, acc3[0], 32 * wave, hk, lm);  // R3 image [H3/4][33]
    __syncthreads();
    stamp();
    {   // score[m] = sum_n wout[n] * R3[n][m]: 32 items x 16 slices of 4 q4 rows, fixed-order reduction through LDS (reuses Xq)
        const int il = tid & 31, sl = tid >> 5;
        const int rows_per = (a.H3 / 4) / 16;
        float s = 0.0f;
        for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
            const uint4 r = R3[(size_t)q * FR_FT_LD + il];
            const float4 w4 = reinterpret_cast<const float4 *>(a.wout)[q];
            s = fmaf(w4.x, __uint_as_float(r.x), s);
            s = fmaf(w4.y, __uint_as_float(r.y), s);
            s = fmaf(w4.z, __uint_as_float(r.z), s);
            s = fmaf(w4.w, __uint_as_float(r.w), s);
        }
        float *part = reinterpret_cast<float *>(R3 + (size_t)64 * FR_FT_LD);  // 2 KiB behind R3
        part[sl * 32 + il] = s;
        __syncthreads();
        if (tid < 32 && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < 16; i++) t += part[i * 32 + tid];
            bt.scores[m0 + tid] = t;
        }
    }
    stamp();
}

size_t frk_fused_lds_bytes(int K, int H2, int wpe) {  // wpe 4 stands for the single-buffer layout here
    if (wpe == 4) {  // Xq + one R1 buffer | R2, then R3 + scratch over R2
        const size_t phase1 = (size_t)(K / 4) + 64, phase2 = (size_t)(H2 / 4) > 68 ? (size_t)(H2 / 4) : 68;
        return (phase1 > phase2 ? phase1 : phase2) * FR_FT_LD * 16;
    }
    const size_t phase1 = (size_t)(K / 4) + 128;      // Xq + two R1 chunk buffers
    const size_t phase2 = (size_t)(H2 / 4) + 64 + 4;  // R2 + R3 + the 2 KiB reduction scratch
    return (phase1 > phase2 ? phase1 : phase2) * FR_FT_LD * 16;
}

bool frk_fused_ok(int K, int H1, int H2, int H3) {
    const bool straight = (K == 352 || K == 880) && H2 == 512;  // straight-line FC1 instantiations (Model-A, Model-B)
    if ((!straight && K % 32) || H1 % 256 || (H2 != 256 && H2 != 512) || H3 != 256) return false;  // run-time loop: K/8 groups in blocks of 4
    return frk_fused_lds_bytes(K, H2, 4) <= 160 * 1024;  // the single-buffer layout is the smallest
}

// two workgroups per CU need <= 80 KiB each (FR_FUSED_WPE=2 forces the one-workgroup build, for A/B measurements)
static int fused_wpe(int K, int H2) {
    static const int forced = [] {
        const char *e = getenv("FR_FUSED_WPE");
        return e ? atoi(e) : 0;
    }();
    if (forced == 2 || forced == 4) return forced == 4 && frk_fused_lds_bytes(K, H2, 4) > 80 * 1024 ? 2 : forced;
    return frk_fused_lds_bytes(K, H2, 4) <= 80 * 1024 ? 4 : 2;
}

template <int T2W, int KG, int WPE, bool DB>
static int fused_launch_inst(const FrFusedArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;  // per instantiation
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fr_fused_tile_kernel<T2W, KG, WPE, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");
        attr_set = true;
    }
    fr_fused_tile_kernel<T2W, KG, WPE, DB><<<grid, dim3(512), lds, s>>>(a);
    KCHECK();
    return FR_OK;
}

int frk_fused_launch(const FrFusedArgs &a, hipStream_t s) {
    int wpe = fused_wpe(a.K, a.H2);
    const bool db = wpe == 2 && frk_fused_lds_bytes(a.K, a.H2, 2) <= 160 * 1024;
    const size_t lds = frk_fused_lds_bytes(a.K, a.H2, db ? 2 : 4);
    dim3 grid(a.n_batches * a.tiles_per_batch);
    if (a.H2 == 512) {
        if (a.K == 352) {  // Model-A: straight-line FC1
            if (wpe == 4) return fused_launch_inst<2, 44, 4, false>(a, grid, lds, s);
            return fused_launch_inst<2, 44, 2, true>(a, grid, lds, s);
        }
        if (a.K == 880) return fused_launch_inst<2, 110, 2, false>(a, grid, lds, s);  // Model-B: 150 KiB with one R1 buffer
        if (db) return fused_launch_inst<2, 0, 2, true>(a, grid, lds, s);
        return fused_launch_inst<2, 0, 2, false>(a, grid, lds, s);
    }
    if (db) return fused_launch_inst<1, 0, 2, true>(a, grid, lds, s);
    return fused_launch_inst<1, 0, 2, false>(a, grid, lds, s);
}

// ===================================================================================================
// fr_fused_tile_m2_kernel<KG>: the fp32 fused item-tile kernel with 64 items (two m tiles) per workgroup.
// Every weight fragment now feeds two m tiles: half as many weight loads per MFMA (they cost MFMA issue slots, see
// tools/experiments/mfma_loop) and half as many barriers, ring cold starts and gather phases per item.  Needs the whole CU
// (158 KiB of LDS: Xq[K/4][65] + ONE R1 buffer; R3 overlays R2) and 64 queued batches of 256 to put one workgroup on every CU,
// so it is used only when the launch group is 64 (fr_ctx_set_stream_group) -- twice the queueing latency of the 32-item kernel.
// Same arithmetic as fr_fused_tile_kernel: full-K k-ordered f32 sums, bit-identical scores.
// ===================================================================================================
constexpr int FR_M2_LD = 65;

template <int NT, int MT, int R, int CNT>
__device__ __forceinline__ void ftm_gemm_ct(f32x16 (&acc)[NT][MT], float4 (&ring)[R][NT], const FtW &w, int n0, const uint4 *Bq, int gb0, int g0, int hk,
                                            int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
    const uint4 *bl = Bq + (size_t)(2 * gb0 + hk) * FR_M2_LD + lm;
    uint4 bcur[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) bcur[mt] = bl[32 * mt];
    // NT == 1: even / odd k partial accumulators exactly as in fr_fused_tile_kernel, so that both kernels return the same bits
    f32x16 alt[MT];
    if constexpr (NT == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) alt[mt][i] = 0.0f;
    }
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        uint4 bnext[MT];
        const int gn = (g + 1 < CNT) ? g + 1 : g;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bnext[mt] = bl[(size_t)(2 * gn) * FR_M2_LD + 32 * mt];  // next group's B fragments from LDS
        const float4(&a4)[NT] = ring[g % R];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const float av = c == 0 ? a4[t].x : c == 1 ? a4[t].y : c == 2 ? a4[t].z : a4[t].w;
                    const uint32_t bv = c == 0 ? bcur[mt].x : c == 1 ? bcur[mt].y : c == 2 ? bcur[mt].z : bcur[mt].w;
                    if (NT == 1 && (c & 1)) alt[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, __uint_as_float(bv), alt[mt], 0, 0, 0);
                    else acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, __uint_as_float(bv), acc[t][mt], 0, 0, 0);
                }
        if (g + R < CNT) {
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = ft_wload(w, s0 + (unsigned)(g + R) * w.row2, 512 * t);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bcur[mt] = bnext[mt];
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[0][mt][i] += alt[mt][i];
    }
}

__device__ __forceinline__ void ftm_store_tile(uint4 *img, const f32x16 &acc, int n_local, int m_local, int hk, int lm) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v;
        v.x = __float_as_uint(acc[4 * i + 0]);
        v.y = __float_as_uint(acc[4 * i + 1]);
        v.z = __float_as_uint(acc[4 * i + 2]);
        v.w = __float_as_uint(acc[4 * i + 3]);
        img[(size_t)((n_local >> 2) + 2 * i + hk) * FR_M2_LD + m_local + lm] = v;  // n = n_local + 8i + 4hk + c
    }
}

template <int KG>
__global__ void __launch_bounds__(512) fr_fused_tile_m2_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int LD = FR_M2_LD, MT = 2, T2W = 2, TI = 64;
    constexpr int RA = 12, RB = 8;
    const int KQ = a.K / 4;
    uint4 *Xq = lds;                     // [KQ][65]
    uint4 *R1 = Xq + (size_t)KQ * LD;    // [64][65]: 256 outputs of FC1
    uint4 *R2 = lds;                     // [H2/4][65], overlays Xq / R1 once they are dead
    uint4 *R3 = lds;                     // [64][65], overlays R2 once FC3 has read it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * TI;
    if (m0 >= bt.batch) return;
    const FtW W1 = ft_w(a.w1q, a.K, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2, a.H3, hk, lm);
    float4 ring1[RA][1];
    ft_ring_fill<1, RA, KG>(ring1, W1, 32 * wave, 0);  // FC1 chunk 0's first weight groups: requested before the gather

    {   // ---- gather: lanes along record words, 8 items per thread ----
        const int wl = tid & 63, ig = tid >> 6;
        bool bad = false;
        for (int w0 = 0; w0 < a.n_words; w0 += 64) {
            const int w = w0 + wl;
            if (w < a.n_words) {
                const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
                const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
                const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
                const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
                const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
                const char *base = is_dense ? reinterpret_cast<const char *>(bt.dense) + src : reinterpret_cast<const char *>(src);
                uint32_t id[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int m = m0 + 8 * ig + i;
                    id[i] = 0;
                    if (m < bt.batch) id[i] = is_dense ? (uint32_t)m : (uint32_t)bt.idx[(size_t)m * a.idx_stride + idx_col];
                    if (!is_dense && id[i] >= rows) {
                        bad = true;
                        id[i] = 0;
                    }
                }
                uint4 v[8];
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
                for (int i = 0; i < 8; i++) Xq[(size_t)w * LD + 8 * ig + i] = (m0 + 8 * ig + i < bt.batch) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();

    f32x16 acc2[T2W][MT];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc1[0][mt][i] = 0.0f;
        ftm_gemm_ct<1, MT, RA, KG>(acc1, ring1, W1, c * 256 + 32 * wave, Xq, 0, 0, hk, lm);
        float4 ring2[RB][T2W];  // FC2's first weight groups are requested before the R1 store and the barriers
        ft_ring_fill<T2W, RB, 32>(ring2, W2, 32 * T2W * wave, 32 * c);
        if (c > 0) __syncthreads();  // single R1 buffer: every wave must be done with the previous chunk's FC2
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ftm_store_tile(R1, acc1[0][mt], 32 * wave, 32 * mt, hk, lm);
        __syncthreads();
        ftm_gemm_ct<T2W, MT, RB, 32>(acc2, ring2, W2, 32 * T2W * wave, R1, 0, 32 * c, hk, lm);
        const int cn = (c + 1 < n_chunks) ? c + 1 : c;  // the last chunk re-requests its own: branch-free, harmless
        ft_ring_fill<1, RA, KG>(ring1, W1, cn * 256 + 32 * wave, 0);
    }
    float4 ring3[RA][1];
    ft_ring_fill<1, RA, 32 * T2W>(ring3, W3, 32 * wave, 0);
    __syncthreads();  // Xq and R1 dead: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ftm_store_tile(R2, acc2[t][mt], 32 * (T2W * wave + t), 32 * mt, hk, lm);
    __syncthreads();

    f32x16 acc3[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc3[0][mt][i] = 0.0f;
    ftm_gemm_ct<1, MT, RA, 32 * T2W>(acc3, ring3, W3, 32 * wave, R2, 0, 0, hk, lm);
    __syncthreads();  // R3 overlays R2: every wave must have finished reading R2
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ftm_store_tile(R3, acc3[0][mt], 32 * wave, 32 * mt, hk, lm);
    __syncthreads();
    {   // score[m] = sum_n wout[n] * R3[n][m]: 64 items x 16 slices of 4 q4 rows (two slices per thread), fixed-order reduction
        // through LDS -- the same partial sums in the same order as fr_fused_tile_kernel
        const int il = tid & 63, sg = tid >> 6;
        const int rows_per = (a.H3 / 4) / 16;
        float *part = reinterpret_cast<float *>(R3 + (size_t)(a.H3 / 4) * LD);
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const int sl = 2 * sg + h2;
            float s = 0.0f;
            for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
                const uint4 r = R3[(size_t)q * LD + il];
                const float4 w4 = reinterpret_cast<const float4 *>(a.wout)[q];
                s = fmaf(w4.x, __uint_as_float(r.x), s);
                s = fmaf(w4.y, __uint_as_float(r.y), s);
                s = fmaf(w4.z, __uint_as_float(r.z), s);
                s = fmaf(w4.w, __uint_as_float(r.w), s);
            }
            part[sl * 64 + il] = s;
        }
        __syncthreads();
        if (tid < 64 && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < 16; i++) t += part[i * 64 + tid];
            bt.scores[m0 + tid] = t;
        }
    }
}

bool frk_fused_m2_ok(int K, int H1, int H2, int H3) { return K == 352 && H1 % 256 == 0 && H2 == 512 && H3 == 256; }

// a.tiles_per_batch counts 64-item tiles
int frk_fused_m2_launch(const FrFusedArgs &a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fr_fused_tile_m2_kernel<44>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");
        attr_set = true;
    }
    const size_t rows1 = (size_t)(a.K / 4) + 64, rows2 = (size_t)(a.H2 / 4), rows3 = 64 + 4;  // R3 + 4 KiB of reduction scratch
    const size_t rows = rows1 > rows2 ? (rows1 > rows3 ? rows1 : rows3) : (rows2 > rows3 ? rows2 : rows3);
    fr_fused_tile_m2_kernel<44><<<dim3(a.n_batches * a.tiles_per_batch), dim3(512), rows * FR_M2_LD * 16, s>>>(a);
    KCHECK();
    return FR_OK;
}

// ===================================================================================================
// fr_fused_tile_h_kernel: the fused item-tile kernel in bf16 (BASELINE configs 2/3: "bf16 MFMA FC, fused concat + first FC").
// Same phases as fr_fused_tile_kernel, q8 operands (8 bf16 per 16 bytes = one v_mfma_f32_32x32x16_bf16 operand per lane),
// fp32 accumulation, ONE bf16 rounding per hidden activation.  A bf16 MFMA is 16x cheaper than the f32 one while the weights
// are only 2x smaller, so the kernel is bound by streaming the weights into the CU (~70 GB/s): a workgroup therefore owns
// 32 * MT items (MT = 2: every weight fragment feeds two m tiles) and the weight ring is as deep as the registers allow.
// ===================================================================================================
template <int NT, int MT, int R, int CNT>
__device__ __forceinline__ void fth_gemm_ct(f32x16 (&acc)[NT][MT], const FtW &w, int n0, const uint4 *Bh, int ld, int gb0, int g0, int hk, int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;  // n0, g0 wave-uniform
    const uint4 *bl = Bh + (size_t)(2 * gb0 + hk) * ld + lm;
    uint4 ring[R][NT];
#pragma unroll
    for (int g = 0; g < R && g < CNT; g++)
#pragma unroll
        for (int t = 0; t < NT; t++) ring[g][t] = __builtin_bit_cast(uint4, ft_wload(w, s0 + (unsigned)g * w.row2, 512 * t));
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        uint4 b8[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) b8[mt] = bl[(size_t)(2 * g) * ld + 32 * mt];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % R][t]), __builtin_bit_cast(bf16x8, b8[mt]),
                                                                     acc[t][mt], 0, 0, 0);
        if (g + R < CNT) {
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = __builtin_bit_cast(uint4, ft_wload(w, s0 + (unsigned)(g + R) * w.row2, 512 * t));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// 32(n) x 32(m) fp32 accumulator tile -> bf16, stored as 8-byte halves of q8 elements of an LDS operand image (row stride ld)
__device__ __forceinline__ void fth_store_tile(uint4 *img, int ld, const f32x16 &acc, int n_local, int m_local, int hk, int lm) {
    uint2 *h = reinterpret_cast<uint2 *>(img);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint2 v;
        v.x = pack_bf16x2(acc[4 * i + 0], acc[4 * i + 1]);
        v.y = pack_bf16x2(acc[4 * i + 2], acc[4 * i + 3]);
        h[((size_t)((n_local >> 3) + i) * ld + m_local + lm) * 2 + hk] = v;  // n = n_local + 8i + 4hk + c
    }
}

template <int MT, int T2W, int KG, bool DB>
__global__ void __launch_bounds__(512) fr_fused_tile_h_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int LD = 32 * MT + 1, TI = 32 * MT;
    const int KO = a.K / 8;
    uint4 *Xh = lds;                                                   // [K/8][LD]
    uint4 *R1b[2] = {Xh + (size_t)KO * LD, Xh + (size_t)(KO + (DB ? 32 : 0)) * LD};  // [32][LD]: 256 outputs of FC1 (bf16)
    uint4 *R2 = lds;                                                   // [H2/8][LD], overlays Xh / R1 once they are dead
    uint4 *R3 = lds + (size_t)(a.H2 / 8) * LD;                         // [H3/8][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * TI;
    if (m0 >= bt.batch) return;

    {   // ---- gather + bf16 conversion: lanes along record words, TI / 8 items per thread ----
        const int wl = tid & 63, ig = tid >> 6;
        constexpr int IPT = TI / 8;
        uint2 *Xh2 = reinterpret_cast<uint2 *>(Xh);
        bool bad = false;
        for (int w0 = 0; w0 < a.n_words; w0 += 64) {
            const int w = w0 + wl;
            if (w < a.n_words) {
                const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
                const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
                const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
                const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
                const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
                const char *base = is_dense ? reinterpret_cast<const char *>(bt.dense) + src : reinterpret_cast<const char *>(src);
                uint32_t id[IPT];
#pragma unroll
                for (int i = 0; i < IPT; i++) {
                    const int m = m0 + IPT * ig + i;
                    id[i] = 0;
                    if (m < bt.batch) id[i] = is_dense ? (uint32_t)m : (uint32_t)bt.idx[(size_t)m * a.idx_stride + idx_col];
                    if (!is_dense && id[i] >= rows) {
                        bad = true;
                        id[i] = 0;
                    }
                }
                uint4 v[IPT];
#pragma unroll
                for (int i = 0; i < IPT; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
                for (int i = 0; i < IPT; i++) {
                    uint2 hv = make_uint2(0u, 0u);
                    if (m0 + IPT * ig + i < bt.batch) {
                        hv.x = pack_bf16x2(__uint_as_float(v[i].x), __uint_as_float(v[i].y));
                        hv.y = pack_bf16x2(__uint_as_float(v[i].z), __uint_as_float(v[i].w));
                    }
                    Xh2[((size_t)(w >> 1) * LD + IPT * ig + i) * 2 + (w & 1)] = hv;  // record word w = half (w & 1) of q8 element w / 2
                }
            }
        }
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();

    f32x16 acc2[T2W][MT];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    // q8 weights: (K / 8) rows of N 16-byte elements -> ft_w's K / 4 rows of the q4 layout is K / 2 here
    const FtW W1 = ft_w(a.w1q, a.K / 2, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1 / 2, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2 / 2, a.H3, hk, lm);
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc1[0][mt][i] = 0.0f;
        fth_gemm_ct<1, MT, 16, KG>(acc1, W1, c * 256 + 32 * wave, Xh, LD, 0, 0, hk, lm);
        uint4 *R1 = R1b[DB ? (c & 1) : 0];
        if (!DB && c > 0) __syncthreads();  // single R1 buffer: every wave must be done reading the previous chunk
#pragma unroll
        for (int mt = 0; mt < MT; mt++) fth_store_tile(R1, LD, acc1[0][mt], 32 * wave, 32 * mt, hk, lm);
        __syncthreads();
        // FC2: K range [256 c, 256 c + 256) = 16 groups of 16 k
        fth_gemm_ct<T2W, MT, 16 / T2W, 16>(acc2, W2, 32 * T2W * wave, R1, LD, 0, 16 * c, hk, lm);
    }
    __syncthreads();  // Xh and R1 dead: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) fth_store_tile(R2, LD, acc2[t][mt], 32 * (T2W * wave + t), 32 * mt, hk, lm);
    __syncthreads();

    f32x16 acc3[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc3[0][mt][i] = 0.0f;
    fth_gemm_ct<1, MT, 16, 16 * T2W>(acc3, W3, 32 * wave, R2, LD, 0, 0, hk, lm);  // H2 / 16 groups
#pragma unroll
    for (int mt = 0; mt < MT; mt++) fth_store_tile(R3, LD, acc3[0][mt], 32 * wave, 32 * mt, hk, lm);
    __syncthreads();
    {   // score[m] = sum_n wout[n] * R3[n][m] (bf16 x bf16, fp32 sum): TI items x (512 / TI) slices of q8 rows, fixed-order reduction
        const int il = tid % TI, sl = tid / TI;
        constexpr int NSL = 512 / TI;
        const int rows_per = (a.H3 / 8) / NSL;
        const uint4 *wh = reinterpret_cast<const uint4 *>(a.wout);  // bf16 vector w[k], 8 per element
        float s = 0.0f;
        for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
            const uint4 r = R3[(size_t)q * LD + il];
            const uint4 w = wh[q];
            const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s = fmaf(__uint_as_float(ww[e] << 16), __uint_as_float(rr[e] << 16), s);
                s = fmaf(__uint_as_float(ww[e] & 0xFFFF0000u), __uint_as_float(rr[e] & 0xFFFF0000u), s);
            }
        }
        float *part = reinterpret_cast<float *>(R3 + (size_t)(a.H3 / 8) * LD);
        part[sl * TI + il] = s;
        __syncthreads();
        if (tid < TI && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < NSL; i++) t += part[i * TI + tid];
            bt.scores[m0 + tid] = t;
        }
    }
}

static size_t fused_h_lds_bytes(int K, int H2, int H3, int MT, bool db) {
    const size_t LD = 32 * MT + 1;
    const size_t phase1 = (size_t)(K / 8) + (db ? 64 : 32);
    const size_t phase2 = (size_t)(H2 / 8) + (size_t)(H3 / 8) + 2;  // + 2 rows: 512 floats of reduction scratch
    return (phase1 > phase2 ? phase1 : phase2) * LD * 16;
}

// bf16 fused kernel: straight-line instantiations exist for K = 352 (Model-A) and K = 880 (Model-B), 64 items per workgroup
bool frk_fused_h_ok(int K, int H1, int H2, int H3) {
    if (H1 % 256 || H2 != 512 || H3 != 256) return false;
    return K == 352 || K == 880;
}

int frk_fused_h_items_per_wg() { return 64; }

template <int MT, int T2W, int KG, bool DB>
static int fused_h_launch_inst(const FrFusedArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fr_fused_tile_h_kernel<MT, T2W, KG, DB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");
        attr_set = true;
    }
    fr_fused_tile_h_kernel<MT, T2W, KG, DB><<<grid, dim3(512), lds, s>>>(a);
    KCHECK();
    return FR_OK;
}

// a.w1q/w2q/w3q/wout must point at the bf16 q8 weights; a.tiles_per_batch counts 64-item tiles
int frk_fused_h_launch(const FrFusedArgs &a, hipStream_t s) {
    dim3 grid(a.n_batches * a.tiles_per_batch);
    if (a.K == 352) return fused_h_launch_inst<2, 2, 22, true>(a, grid, fused_h_lds_bytes(a.K, a.H2, a.H3, 2, true), s);
    if (a.K == 880) return fused_h_launch_inst<2, 2, 55, false>(a, grid, fused_h_lds_bytes(a.K, a.H2, a.H3, 2, false), s);
    FR_FAIL(FR_ERR_INVALID, "no bf16 fused instantiation for K=%d", a.K);
}
