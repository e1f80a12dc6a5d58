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

__global__ void __launch_bounds__(256) f32_to_bf16_kernel(const float *src, uint16_t *dst, uint64_t n) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        __hip_bfloat16 b = __float2bfloat16(src[i]);  // round-to-nearest-even, NaN stays NaN
        dst[i] = *reinterpret_cast<uint16_t *>(&b);
    }
}

int frk_f32_to_bf16(const float *src, uint16_t *dst, size_t count, hipStream_t s) {
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return FR_OK;
    f32_to_bf16_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(src, dst, count);
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

int frk_gather(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, float *out,
               int batch, int *err_flag, hipStream_t s) {
    if (n_words <= 0 || batch <= 0) return FR_OK;
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

// ===================================================================================================
// Small-batch pipeline: feature-major activations + split-K inside the workgroup.
//
// At batch 256 one FC layer is only 64-128 output tiles of 32x32: a classic LDS-tiled GEMM leaves
// most of the 1024 SIMDs idle and walks K serially (measured: 20 us per layer).  Here every
// activation matrix is kept FEATURE-major, Xt[k][m] (m = item, leading dimension ldm), so that both
// MFMA operands are plain coalesced 128-byte row segments that go straight from L2 to registers:
//     A fragment: Wt[k + (lane>>5)][n0 + (lane&31)]     (the reference's column-major W, cuda_server.c:215)
//     B fragment: Xt[k + (lane>>5)][m0 + (lane&31)]
//     D[n][m] accumulates in the 32x32 MFMA; its store Yt[n][m0 + (lane&31)] is coalesced again.
// The SPLITK waves of a workgroup each own one slice of K for the SAME 32x32 output tile and are
// summed through LDS in a fixed order (deterministic, no atomics).  Item columns m >= batch are
// padding: every output column depends only on the same input column, so they never mix with real items.
// ===================================================================================================

// gather_t: per-table rows -> Xt[k][m].  Lanes = 64 consecutive items, one record word per wave step
// (its descriptor is wave-uniform -> scalar loads); each lane reads its item's 16-byte row word and
// writes 4 floats to 4 feature rows (256-byte coalesced stores).
__global__ void __launch_bounds__(256) gather_t_kernel(const FrWordDesc *__restrict__ words, int n_words,
                                                       const int32_t *__restrict__ idx, int idx_stride,
                                                       const float *__restrict__ dense, float *__restrict__ Xt,
                                                       int batch, int ldm, int words_per_wave, int *__restrict__ err_flag) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int w_begin = (blockIdx.y * 4 + wave) * words_per_wave;
    const bool live = m < batch;
    bool bad = false;
    for (int i = 0; i < words_per_wave; i++) {
        const int w = w_begin + i;
        if (w >= n_words) break;
        const FrWordDesc d = words[w];  // wave-uniform
        const bool is_dense = (d.idx_col & FR_DESC_DENSE) != 0;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) {
            uint32_t id = is_dense ? (uint32_t)m : (uint32_t)idx[(size_t)m * idx_stride + d.idx_col];
            if (!is_dense && id >= d.rows) {
                bad = true;
                id = 0;
            }
            const char *base = is_dense ? reinterpret_cast<const char *>(dense) + d.src : reinterpret_cast<const char *>(d.src);
            v = *reinterpret_cast<const uint4 *>(base + (uint64_t)id * d.stride);
        }
        // feature index of this word inside the record (SEMANTIC layout only on this path)
        float *o = Xt + (size_t)(4 * d.dst_off) * ldm + m;
        if (m < ldm) {
            o[0] = __uint_as_float(v.x);
            o[(size_t)ldm] = __uint_as_float(v.y);
            o[(size_t)2 * ldm] = __uint_as_float(v.z);
            o[(size_t)3 * ldm] = __uint_as_float(v.w);
        }
    }
    if (bad) atomicOr_system(err_flag, 1);  // pinned host word; error path only
}

int frk_gather_t(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, float *Xt,
                 int batch, int ldm, int *err_flag, hipStream_t s) {
    if (n_words <= 0 || batch <= 0) return FR_OK;
    const int words_per_wave = 2;
    dim3 grid((ldm + 63) / 64, (n_words + 4 * words_per_wave - 1) / (4 * words_per_wave));
    gather_t_kernel<<<grid, dim3(256), 0, s>>>(words, n_words, idx, idx_stride, dense, Xt, batch, ldm, words_per_wave, err_flag);
    KCHECK();
    return FR_OK;
}

// item-major records [B][K] -> Xt[K][ldm] (only used by the fc_only diagnostic entry point)
__global__ void __launch_bounds__(256) transpose_records_kernel(const float *__restrict__ X, float *__restrict__ Xt, int batch, int K, int ldm) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r, k = k0 + tx;
        tile[r][tx] = (m < batch && k < K) ? X[(size_t)m * K + k] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int k = k0 + r, m = m0 + tx;
        if (k < K && m < ldm) Xt[(size_t)k * ldm + m] = tile[tx][r];
    }
}

int frk_transpose_records(const float *X, float *Xt, int batch, int K, int ldm, hipStream_t s) {
    dim3 grid((K + 31) / 32, (ldm + 31) / 32);
    transpose_records_kernel<<<grid, dim3(256), 0, s>>>(X, Xt, batch, K, ldm);
    KCHECK();
    return FR_OK;
}

// fc_t: Yt[N][ldm] = W * X with Wt[K][N], Xt[K][ldm]; one 32(n) x 32(m) tile per workgroup, SPLITK waves.
//
// NP > 0 : the wave's K slice is exactly NP k-pairs, known at compile time.  The body is straight-line: all
//          2*NP operand loads are issued up front (row base in SGPRs + one per-lane VGPR offset), then NP MFMAs
//          consume them behind counted vmcnt waits -- one L2 round trip per wave instead of one per k-group.
// NP == 0: generic shapes; double-buffered groups of 8 k-pairs, branch-free inside the loop.
template <int SPLITK, int NP>
__global__ void __launch_bounds__(64 * SPLITK) fc_t_kernel(const float *__restrict__ Wt, const float *__restrict__ Xt,
                                                           float *__restrict__ Yt, int K, int N, int ldm) {
    __shared__ float red[SPLITK > 1 ? SPLITK : 1][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int hk = lane >> 5, lm = lane & 31;
    const int a_off = hk * N + n0 + lm;    // per-lane, loop-invariant
    const int b_off = hk * ldm + m0 + lm;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;

    if constexpr (NP > 0) {
        const float *a_row = Wt + (size_t)(2 * NP * wave) * N;   // wave-uniform
        const float *b_row = Xt + (size_t)(2 * NP * wave) * ldm;
        float ra[NP], rb[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            ra[i] = a_row[(size_t)(2 * i) * N + a_off];
            rb[i] = b_row[(size_t)(2 * i) * ldm + b_off];
        }
        // keep the scheduler from sinking the loads back next to their MFMAs (it would, to save VGPRs)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NP; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i], rb[i], acc, 0, 0, 0);
    } else {
        const int pairs = K / 2;
        const int per = (pairs + SPLITK - 1) / SPLITK;
        const int p_begin = wave * per;
        int np = pairs - p_begin;
        np = np < 0 ? 0 : (np > per ? per : np);
        const float *a_row = Wt + (size_t)(2 * p_begin) * N;
        const float *b_row = Xt + (size_t)(2 * p_begin) * ldm;
        constexpr int D = 8;
        const int ng = np / D;
        float ra[D], rb[D], na[D], nb[D];
        if (ng > 0) {
#pragma unroll
            for (int i = 0; i < D; i++) {
                ra[i] = a_row[(size_t)(2 * i) * N + a_off];
                rb[i] = b_row[(size_t)(2 * i) * ldm + b_off];
            }
        }
        for (int g = 0; g < ng; g++) {
            const int nx = (g + 1 < ng) ? (g + 1) : g;  // the last group re-loads itself (harmless) -> no branch in the body
#pragma unroll
            for (int i = 0; i < D; i++) {
                na[i] = a_row[(size_t)(2 * (nx * D + i)) * N + a_off];
                nb[i] = b_row[(size_t)(2 * (nx * D + i)) * ldm + b_off];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < D; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i], rb[i], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < D; i++) {
                ra[i] = na[i];
                rb[i] = nb[i];
            }
        }
        for (int p = ng * D; p < np; p++) {  // remainder (< D pairs)
            const float a = a_row[(size_t)(2 * p) * N + a_off];
            const float b = b_row[(size_t)(2 * p) * ldm + b_off];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    if constexpr (SPLITK == 1) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * hk;
            Yt[(size_t)n * ldm + m0 + lm] = acc[r];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; r++) red[wave][r][lane] = acc[r];
        __syncthreads();
        // fixed-order sum over the K slices, then the coalesced store of the tile
        for (int e = threadIdx.x; e < 16 * 64; e += 64 * SPLITK) {
            const int r = e >> 6, l = e & 63;
            float s = red[0][r][l];
#pragma unroll
            for (int w = 1; w < SPLITK; w++) s += red[w][r][l];
            const int n = n0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            Yt[(size_t)n * ldm + m0 + (l & 31)] = s;
        }
    }
}

template <int SPLITK, int NP>
static void fc_t_launch(dim3 grid, const float *Wt, const float *Xt, float *Yt, int K, int N, int ldm, hipStream_t s) {
    fc_t_kernel<SPLITK, NP><<<grid, dim3(64 * SPLITK), 0, s>>>(Wt, Xt, Yt, K, N, ldm);
}

int frk_fc_t(const float *Wt, const float *Xt, float *Yt, int K, int N, int ldm, int splitk, hipStream_t s) {
    if (K % 4 || N % 32 || ldm % 32) FR_FAIL(FR_ERR_INVALID, "fc_t needs K%%4==0, N%%32==0, ldm%%32==0 (K=%d N=%d ldm=%d)", K, N, ldm);
    dim3 grid(N / 32, ldm / 32);
    const int pairs = K / 2;
    const int np = (pairs % splitk == 0) ? pairs / splitk : -1;
    // straight-line instantiations for the reference models' small-batch shapes
    if (splitk == 4 && np == 44) fc_t_launch<4, 44>(grid, Wt, Xt, Yt, K, N, ldm, s);        // A FC1: K=352
    else if (splitk == 8 && np == 55) fc_t_launch<8, 55>(grid, Wt, Xt, Yt, K, N, ldm, s);   // B FC1: K=880
    else if (splitk == 8 && np == 64) fc_t_launch<8, 64>(grid, Wt, Xt, Yt, K, N, ldm, s);   // FC2: K=1024
    else if (splitk == 16 && np == 64) fc_t_launch<16, 64>(grid, Wt, Xt, Yt, K, N, ldm, s); // C FC2: K=2048
    else if (splitk == 8 && np == 32) fc_t_launch<8, 32>(grid, Wt, Xt, Yt, K, N, ldm, s);   // FC3: K=512
    else if (splitk == 16 && np == 16) fc_t_launch<16, 16>(grid, Wt, Xt, Yt, K, N, ldm, s); // FC3: K=512
    else if (splitk == 4 && np == 64) fc_t_launch<4, 64>(grid, Wt, Xt, Yt, K, N, ldm, s);
    else if (splitk == 2 && np == 64) fc_t_launch<2, 64>(grid, Wt, Xt, Yt, K, N, ldm, s);
    else {
        switch (splitk) {
            case 1: fc_t_launch<1, 0>(grid, Wt, Xt, Yt, K, N, ldm, s); break;
            case 2: fc_t_launch<2, 0>(grid, Wt, Xt, Yt, K, N, ldm, s); break;
            case 4: fc_t_launch<4, 0>(grid, Wt, Xt, Yt, K, N, ldm, s); break;
            case 8: fc_t_launch<8, 0>(grid, Wt, Xt, Yt, K, N, ldm, s); break;
            case 16: fc_t_launch<16, 0>(grid, Wt, Xt, Yt, K, N, ldm, s); break;
            default: FR_FAIL(FR_ERR_INVALID, "fc_t: unsupported splitk %d", splitk);
        }
    }
    KCHECK();
    return FR_OK;
}

// fc_out_t: score[m] = sum_n w[n] * Rt[n][m]; block = 64 items x 16 slices of n (all loads of a slice in flight),
// LDS reduce in fixed order.
__global__ void __launch_bounds__(1024) fc_out_t_kernel(const float *__restrict__ Rt, const float *__restrict__ w,
                                                        float *__restrict__ score, int batch, int H, int ldm) {
    __shared__ float part[16][64];
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = blockIdx.x * 64 + lane;
    const int per = (H + 15) / 16;
    const int h0 = q * per;
    float s = 0.0f;
    if (m < ldm) {
        int h = h0;
        for (; h + 8 <= h0 + per && h + 8 <= H; h += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; i++) v[i] = Rt[(size_t)(h + i) * ldm + m];
#pragma unroll
            for (int i = 0; i < 8; i++) s = fmaf(w[h + i], v[i], s);
        }
        for (; h < h0 + per && h < H; h++) s = fmaf(w[h], Rt[(size_t)h * ldm + m], s);
    }
    part[q][lane] = s;
    __syncthreads();
    if (q == 0 && m < batch) {
        float t = part[0][lane];
#pragma unroll
        for (int i = 1; i < 16; i++) t += part[i][lane];
        score[m] = t;
    }
}

int frk_fc_out_t(const float *Rt, const float *w, float *score, int batch, int H, int ldm, hipStream_t s) {
    fc_out_t_kernel<<<dim3((ldm + 63) / 64), dim3(1024), 0, s>>>(Rt, w, score, batch, H, ldm);
    KCHECK();
    return FR_OK;
}
