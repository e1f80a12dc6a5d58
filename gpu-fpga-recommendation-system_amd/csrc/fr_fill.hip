// Table / weight fills, operand re-packs and reductions: one-off or diagnostic kernels around the hot path.
#include <type_traits>

#include "fr_content.h"
#include "fr_device.h"

// Procedural contents: fr_content.h (shared with the CPU back-end; oracle/fleetrec_oracle.c content_bits() restates the same functions).
// one thread per 16-byte word, grid-stride; stores are 16 B/lane fully coalesced.  Row r of the table (r = row0 + local row) lives at
// base + local_row * row_stride_words: row_stride_words == words_per_row for a table stored on its own, larger for a table that is one
// column block of a bank-interleaved region (FR_INDEX_PER_BANK, fr_api.cpp build_words).
__global__ void __launch_bounds__(256) fill_table_kernel(uint4 *base, uint64_t n_words, uint32_t words_per_row, uint64_t row_stride_words, uint64_t row0,
                                                          int mode, uint32_t seed, uint32_t uid) {
    const uint32_t h0 = fr_table_hash_seed(seed, uid);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += stride) {
        const uint64_t lrow = w / words_per_row, row = row0 + lrow;
        const uint32_t wc = (uint32_t)(w - lrow * words_per_row), c0 = wc * 4;
        uint4 v;
        v.x = fr_content_bits(mode, h0, uid, row, c0 + 0);
        v.y = fr_content_bits(mode, h0, uid, row, c0 + 1);
        v.z = fr_content_bits(mode, h0, uid, row, c0 + 2);
        v.w = fr_content_bits(mode, h0, uid, row, c0 + 3);
        base[lrow * row_stride_words + wc] = v;
    }
}

int frk_fill_table(float *base, int64_t row0, int64_t rows, int dim, int64_t row_stride_bytes, int mode, uint32_t seed, uint32_t uid, hipStream_t s) {
    const uint64_t n_words = (uint64_t)rows * (uint64_t)(dim / 4);
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks == 0) return FR_OK;
    fill_table_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>((uint4 *)base, n_words, (uint32_t)(dim / 4), (uint64_t)row_stride_bytes / 16, (uint64_t)row0, mode, seed,
                                                                 uid);
    KCHECK();
    return FR_OK;
}

__global__ void __launch_bounds__(256) fill_weights_kernel(float *w, uint64_t n, int mode, uint32_t seed, uint32_t layer, float scale) {
    const uint32_t h0 = fr_weight_hash_seed(seed, layer);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) w[i] = fr_weight_value(mode, h0, i, scale);
}

int frk_fill_weights(float *w, size_t count, int mode, uint32_t seed, uint32_t layer, float scale, hipStream_t s) {
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) return FR_OK;
    fill_weights_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(w, count, mode, seed, layer, scale);
    KCHECK();
    return FR_OK;
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
// ONE launch for all shards (blockIdx.z = shard; the slice offsets / lengths travel as a kernel argument): a rank of an 8-way job used
// to issue eight small launches one behind the other in front of every FC chain.  A transpose of 16-byte elements like the one above.
constexpr int FR_SLICE_TABLE = 64;   // shards per launch (more: one launch per 64)
struct FrSliceTable {
    int q_off[FR_SLICE_TABLE], q_len[FR_SLICE_TABLE];   // record words (16 bytes of fp32) per shard
};

// shards [g0, g0 + n) -> the table; -> the longest slice among them
static int slice_table(FrSliceTable &t, int g0, int n, const int *h_offsets, const int *h_lens) {
    int max_q = 0;
    for (int g = 0; g < n; g++) {
        t.q_off[g] = h_offsets[g0 + g] / 4, t.q_len[g] = h_lens[g0 + g] / 4;
        if (t.q_len[g] > max_q) max_q = t.q_len[g];
    }
    return max_q;
}

__global__ void __launch_bounds__(256) transpose_slices_kernel(const float4 *__restrict__ S /* [G][B][F/4] */, size_t shard_stride, int FQ, int item0, int n_items,
                                                               const FrSliceTable t, float4 *__restrict__ Xq, int ldm) {
    __shared__ float4 tile[16][17];
    const int g = blockIdx.z, q_len = t.q_len[g], q_off = t.q_off[g];
    const int q0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    if (q0 >= q_len) return;
    S += (size_t)g * shard_stride;
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
    for (int g0 = 0; g0 < n_shards; g0 += FR_SLICE_TABLE) {
        const int n = n_shards - g0 < FR_SLICE_TABLE ? n_shards - g0 : FR_SLICE_TABLE;
        FrSliceTable t;
        const int max_q = slice_table(t, g0, n, h_offsets, h_lens);
        if (max_q == 0) continue;
        dim3 grid((max_q + 15) / 16, (ldm + 15) / 16, n);
        transpose_slices_kernel<<<grid, dim3(256), 0, s>>>(reinterpret_cast<const float4 *>(gathered) + (size_t)g0 * batch_total * FQ, (size_t)batch_total * FQ, FQ, item0,
                                                          n_items, t, reinterpret_cast<float4 *>(Xq), ldm);
    }
    KCHECK();
    return FR_OK;
}

// Sharded mode with low-precision transport: all-gathered slices [G][B][F] of bf16 (PREC 1: 8 bytes per record word) or e4m3
// (PREC 2: 4 bytes per word) -> the q8 / q16 operand image for items [item0, item0+n).  One launch for all shards, as above.  The slices cover
// every record word and the kernel writes zeros for the items past n_items, so nothing is cleared beforehand except the fp8 image's rows past
// K / 16 (its k is padded to a multiple of 64) -- the whole image used to be zeroed in front of every FC chain (32 MB at batch 4096).
template <int PREC>
__global__ void __launch_bounds__(256) transpose_slices_lp_kernel(const char *__restrict__ S /* [G][B][F/4] words */, size_t shard_stride_bytes, int FQ, int item0,
                                                                   int n_items, const FrSliceTable t, void *__restrict__ X, int ldm) {
    typedef typename std::conditional<PREC == 1, uint2, uint32_t>::type word_t;
    __shared__ word_t tile[16][17];
    const int g = blockIdx.z, q_len = t.q_len[g], q_off = t.q_off[g];
    const int q0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    if (q0 >= q_len) return;
    const word_t *Sg = reinterpret_cast<const word_t *>(S + (size_t)g * shard_stride_bytes);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    {
        const int m = m0 + ty, q = q0 + tx;
        word_t z{};
        tile[ty][tx] = (m < n_items && q < q_len) ? Sg[(size_t)(item0 + m) * FQ + q] : z;
    }
    __syncthreads();
    {
        const int q = q0 + ty, m = m0 + tx;
        if (q < q_len && m < ldm) {
            const int w = q_off + q;  // record word
            if constexpr (PREC == 1) reinterpret_cast<uint2 *>(X)[((size_t)(w >> 1) * ldm + m) * 2 + (w & 1)] = tile[tx][ty];
            else reinterpret_cast<uint32_t *>(X)[((size_t)(w >> 2) * ldm + m) * 4 + (w & 3)] = tile[tx][ty];
        }
    }
}

int frk_transpose_slices_lp(int precision, const void *gathered, int n_shards, int batch_total, int slice_padded, const int *h_offsets, const int *h_lens,
                            int item0, int n_items, void *X, int K, int ldm, hipStream_t s) {
    if (slice_padded % 4) FR_FAIL(FR_ERR_INVALID, "slice_padded %d must be a multiple of 4", slice_padded);
    const int FQ = slice_padded / 4;
    const size_t esz = precision == FR_FC_BF16 ? 8 : 4;  // bytes per record word on the wire
    const size_t rows = precision == FR_FC_FP8 ? (size_t)(K + 63) / 64 * 4 : (size_t)K / 8;
    const size_t covered = precision == FR_FC_FP8 ? (size_t)K / 16 : (size_t)K / 8;   // element rows whose every word a slice writes
    if (covered < rows && hipMemsetAsync(reinterpret_cast<char *>(X) + covered * ldm * 16, 0, (rows - covered) * ldm * 16, s) != hipSuccess)
        FR_FAIL(FR_ERR_HIP, "hipMemsetAsync failed");
    const size_t stride = (size_t)batch_total * FQ * esz;
    for (int g0 = 0; g0 < n_shards; g0 += FR_SLICE_TABLE) {
        const int n = n_shards - g0 < FR_SLICE_TABLE ? n_shards - g0 : FR_SLICE_TABLE;
        FrSliceTable t;
        const int max_q = slice_table(t, g0, n, h_offsets, h_lens);
        if (max_q == 0) continue;
        dim3 grid((max_q + 15) / 16, (ldm + 15) / 16, n);
        const char *src = reinterpret_cast<const char *>(gathered) + (size_t)g0 * stride;
        if (precision == FR_FC_BF16) transpose_slices_lp_kernel<1><<<grid, dim3(256), 0, s>>>(src, stride, FQ, item0, n_items, t, X, ldm);
        else transpose_slices_lp_kernel<2><<<grid, dim3(256), 0, s>>>(src, stride, FQ, item0, n_items, t, X, ldm);
    }
    KCHECK();
    return FR_OK;
}

// q4 fp32 activations Xq[K/4][ldm] -> the low-precision operand image of the same tensor: PREC 1 = q8 bf16 Xh[K/8][ldm],
// PREC 2 = q16 e4m3 Xf[KP/16][ldm] (x 2^e, saturated, zero rows up to a multiple of 64 k).  Used by the sharded mode's FC entry
// point, whose slices arrive as fp32.
template <int PREC>
__global__ void __launch_bounds__(256) q4_to_lp_kernel(const uint4 *__restrict__ Xq, uint4 *__restrict__ Xo, int KQ, int rows_out, int ldm, float scale) {
    const size_t n = (size_t)rows_out * ldm;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const int ro = (int)(e / ldm), m = (int)(e - (size_t)ro * ldm);
        constexpr int PER = PREC == 1 ? 2 : 4;  // q4 rows per output element
        uint4 in[PER];
#pragma unroll
        for (int j = 0; j < PER; j++) in[j] = (PER * ro + j < KQ) ? Xq[(size_t)(PER * ro + j) * ldm + m] : make_uint4(0u, 0u, 0u, 0u);
        uint4 o;
        if constexpr (PREC == 1) {
            o.x = pack_bf16x2(__uint_as_float(in[0].x), __uint_as_float(in[0].y));
            o.y = pack_bf16x2(__uint_as_float(in[0].z), __uint_as_float(in[0].w));
            o.z = pack_bf16x2(__uint_as_float(in[1].x), __uint_as_float(in[1].y));
            o.w = pack_bf16x2(__uint_as_float(in[1].z), __uint_as_float(in[1].w));
        } else {
            o.x = pack_fp8_word(in[0], scale);
            o.y = pack_fp8_word(in[1], scale);
            o.z = pack_fp8_word(in[2], scale);
            o.w = pack_fp8_word(in[3], scale);
        }
        Xo[e] = o;
    }
}

int frk_q4_to_lp(int precision, const float *Xq, void *Xo, int K, int ldm, int e_x, hipStream_t s) {
    const int KQ = K / 4;
    const int rows_out = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : K / 8;
    if (precision == FR_FC_BF16 && K % 8) FR_FAIL(FR_ERR_INVALID, "bf16 operands need K %% 8 == 0 (K=%d)", K);
    const size_t n = (size_t)rows_out * ldm;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
    if (precision == FR_FC_FP8)
        q4_to_lp_kernel<2><<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(reinterpret_cast<const uint4 *>(Xq), reinterpret_cast<uint4 *>(Xo), KQ, rows_out, ldm, ldexpf(1.0f, e_x));
    else
        q4_to_lp_kernel<1><<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(reinterpret_cast<const uint4 *>(Xq), reinterpret_cast<uint4 *>(Xo), KQ, rows_out, ldm, 1.0f);
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

// fp32 master weights (column-major H x K) -> "q16h" Wg[2 (k / 32) + kh][h][16 B], K zero-padded to a multiple of 32: the operand layout
// of the NON-scaled fp8 MFMA (v_mfma_f32_32x32x16_fp8_fp8: lane half kh supplies k 8 kh .. 8 kh + 7 of a 16-k step).  The 16 bytes of
// (row 2 j + kh, h) are the lane's 8 bytes of step 2 j followed by its 8 bytes of step 2 j + 1: byte b holds k = 32 j + 16 (b / 8) + 8 kh + b % 8.
// One 16-byte load per lane then feeds two MFMAs, with the addressing of the bf16 q8 layout (fr_fused_tile_hs_kernel, fp8 form).
__global__ void __launch_bounds__(256) pack_weights_q16h_fp8_kernel(const float *__restrict__ W, uint4 *__restrict__ Wg, int K, int KP, int H, float scale) {
    const size_t n = (size_t)(KP / 16) * H;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t row = e / H, h = e - row * H;
        const size_t j = row >> 1, kh = row & 1;
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {   // dword q: bytes 4 q .. 4 q + 3 -> step q / 2, position 4 (q % 2) ..
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const size_t k = 32 * j + 16 * (size_t)(q >> 1) + 8 * kh + 4 * (size_t)(q & 1) + c;
                v[c] = (k < (size_t)K) ? W[h + k * H] : 0.0f;
            }
            o[q] = pack_fp8x4(v[0], v[1], v[2], v[3], scale);
        }
        Wg[e] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

int frk_pack_weights_q16h_fp8(const float *W, void *Wg, int K, int H, int e_w, hipStream_t s) {
    const int KP = (K + 31) / 32 * 32;
    size_t n = (size_t)(KP / 16) * H;
    unsigned blocks = (unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    pack_weights_q16h_fp8_kernel<<<dim3(blocks ? blocks : 1), dim3(256), 0, s>>>(W, reinterpret_cast<uint4 *>(Wg), K, KP, H, ldexpf(1.0f, e_w));
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




#ifdef FR_EXPERIMENTS
// experiments build: a spin kernel shaped like a part-chip GEMM launch of a chain model (workgroups x 512 threads, dynamic LDS) for the queue-pairing
// probe of tools/experiments/queue_aging.py (fr_exp_burst_probe).  Every wave exits: bounded by the 100 MHz clock and an iteration count.
__global__ void __launch_bounds__(512) spin_probe_kernel(unsigned ticks, unsigned *sink) {
    extern __shared__ unsigned spin_lds[];
    if (threadIdx.x == 0) spin_lds[0] = ticks;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    for (int it = 0; it < 4000000; it++) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)spin_lds[0]) break;
        __builtin_amdgcn_s_sleep(8);
        n++;
    }
    if (n == 0xffffffffu && sink) sink[0] = n;
}
int frk_spin_probe(hipStream_t s, int wgs, int lds_bytes, unsigned ticks) {
    static FrLdsAttrOnce once;
    if (int rc_ = fr_allow_full_lds(&spin_probe_kernel, once)) return rc_;
    spin_probe_kernel<<<dim3(wgs), dim3(512), lds_bytes, s>>>(ticks, nullptr);
    KCHECK();
    return FR_OK;
}
#endif
