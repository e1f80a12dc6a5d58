// gfx950 (MI355X / CDNA4) kernels of the FleetRec hot path + their launchers.
//
//   fill_*        device-side table / weight synthesis (the FPGA host's init_vectors,
//                 FPGA/host/embedding_47_krnl/host.cpp:66-88, without a 63 GB host staging copy)
//   gather_pack   per-table embedding row gather + bit-copy concat into the per-item record
//                 (load_single_embedding_*_tables + group_* + gather_embeddings of
//                 FPGA/kernel/user_krnl/embedding_{47,98,377}_krnl/src/hls/embedding_*_krnl.cpp)
//   fc_f32        one column-major GEMM of the chain R = W * X on the exact-f32 MFMA
//                 (cublasLtMatmul, GPU/final_network_cublasLt_1_node_no_FIFO_scatter/cuda_server.c:468-485)
//   fc_out        the OUT == 1 layer (cuda_server.c:486-491), a per-item dot product
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
    if (bad) atomicOr(err_flag, 1);
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

// ---------------------------------------------------------------------------------------------------
// fc_f32: Y[B][N] = X[B][K] * Wt[K][N], all fp32.
//
// The reference's column-major operands map onto row-major ones without any transpose:
//   X  (K x B, ld=K)  == item-major records X[b][k]                      (cuda_server.c:216)
//   W  (H x K, ld=H)  == K-major weights   Wt[k][h] = W[h + k*H]         (cuda_server.c:215)
//   R  (H x B, ld=H)  == item-major        Y[b][h]                       (cuda_server.c:217)
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain
// (no TF32-like shortcut exists on gfx950), i.e. CUBLAS_COMPUTE_32F semantics (cuda_server.c:211).
//
// Block = 256 threads = 4 waves (2 x 2), block tile 64 items x 64 outputs, wave tile 32 x 32
// (one 16-register accumulator), K step 16, two LDS buffers, one barrier per K step.
// LDS images are k-major ([k][m] and [k][n]) so that the A and B fragment reads
// (lane l -> element [k + (l>>5)][base + (l&31)]) are conflict-free ds_read_b32.
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FC_BM = 64, FC_BN = 64, FC_BK = 16;

__global__ void __launch_bounds__(256) fc_f32_kernel(const float *__restrict__ X, const float *__restrict__ Wt,
                                                     float *__restrict__ Y, int B, int K, int N) {
    __shared__ float lds[2][2][FC_BK][FC_BM];  // [buf][A|B][k][m or n]  = 16 KiB
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FC_BM, n0 = blockIdx.x * FC_BN;

    // staging roles
    const int a_m = tid & 63, a_kq = tid >> 6;   // A: row m0+a_m, floats k0+4*a_kq .. +3
    const int b_k = tid >> 4, b_nq = tid & 15;   // B: row k0+b_k, floats n0+4*b_nq .. +3
    const bool a_row_ok = (m0 + a_m) < B;
    const float *a_ptr = X + (size_t)(a_row_ok ? (m0 + a_m) : 0) * K + 4 * a_kq;
    const bool b_col_ok = (n0 + 4 * b_nq) < N;  // N % 4 == 0 is required by the launcher
    const float *b_ptr = Wt + (size_t)b_k * N + (b_col_ok ? (n0 + 4 * b_nq) : 0);

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;

    const int n_kt = (K + FC_BK - 1) / FC_BK;
    float4 ra, rb;
    auto load_tile = [&](int kt) {
        const int k0 = kt * FC_BK;
        ra = make_float4(0.f, 0.f, 0.f, 0.f);
        rb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a_row_ok && (k0 + 4 * a_kq) < K) ra = *reinterpret_cast<const float4 *>(a_ptr + k0);  // K % 4 == 0
        if (b_col_ok && (k0 + b_k) < K) rb = *reinterpret_cast<const float4 *>(b_ptr + (size_t)k0 * N);
    };
    load_tile(0);
    for (int kt = 0; kt < n_kt; kt++) {
        const int buf = kt & 1;
        lds[buf][0][4 * a_kq + 0][a_m] = ra.x;
        lds[buf][0][4 * a_kq + 1][a_m] = ra.y;
        lds[buf][0][4 * a_kq + 2][a_m] = ra.z;
        lds[buf][0][4 * a_kq + 3][a_m] = ra.w;
        *reinterpret_cast<float4 *>(&lds[buf][1][b_k][4 * b_nq]) = rb;
        __syncthreads();
        if (kt + 1 < n_kt) load_tile(kt + 1);
        const int hk = lane >> 5, lm = lane & 31;
#pragma unroll
        for (int kk = 0; kk < FC_BK; kk += 2) {
            const float a = lds[buf][0][kk + hk][wm * 32 + lm];
            const float b = lds[buf][1][kk + hk][wn * 32 + lm];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < N) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < B) Y[(size_t)row * N + col] = acc[r];
        }
    }
}

int frk_fc_f32(const float *X, const float *Wt, float *Y, int B, int K, int N, hipStream_t s) {
    if (B <= 0) return FR_OK;
    if (K % 4 || N % 4) FR_FAIL(FR_ERR_INVALID, "fc_f32 needs K and N multiples of 4 (got K=%d N=%d)", K, N);
    dim3 grid((N + FC_BN - 1) / FC_BN, (B + FC_BM - 1) / FC_BM);
    fc_f32_kernel<<<grid, dim3(256), 0, s>>>(X, Wt, Y, B, K, N);
    KCHECK();
    return FR_OK;
}

// ---------------------------------------------------------------------------------------------------
// fc_out: score[b] = sum_h R[b][h] * w[h]  (OUTPUT_FEATURE_LEN == 1; Wout is 1 x H column-major = w[h]).
// One wave per item, 16-byte loads, wave-level shuffle reduction.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fc_out_kernel(const float *__restrict__ R, const float *__restrict__ w,
                                                     float *__restrict__ score, int B, int H) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= B) return;
    const float *r = R + (size_t)item * H;
    float s = 0.0f;
    for (int h = 4 * lane; h < H; h += 256) {
        const float4 a = *reinterpret_cast<const float4 *>(r + h);
        const float4 b = *reinterpret_cast<const float4 *>(w + h);
        s = fmaf(a.x, b.x, s);
        s = fmaf(a.y, b.y, s);
        s = fmaf(a.z, b.z, s);
        s = fmaf(a.w, b.w, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) score[item] = s;
}

int frk_fc_out(const float *R, const float *w, float *score, int B, int H, hipStream_t s) {
    if (B <= 0) return FR_OK;
    if (H % 4) FR_FAIL(FR_ERR_INVALID, "fc_out needs H multiple of 4 (got %d)", H);
    fc_out_kernel<<<dim3((B + 3) / 4), dim3(256), 0, s>>>(R, w, score, B, H);
    KCHECK();
    return FR_OK;
}
