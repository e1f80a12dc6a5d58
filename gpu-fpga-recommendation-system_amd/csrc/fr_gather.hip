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
template <int TP>
__device__ __forceinline__ void store_word(void *out, size_t word, const uint4 &v, float scale) {
    if constexpr (TP == 0) {
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
// round-robin over the 8 XCDs; a speed assumption only), and each group owns a fixed contiguous 1/8 of the record words for
// ALL items.  Every table is then touched from ONE XCD only, so the 8 x 4 MiB L2s cache 8 different table sets instead of
// 8 copies of the same hottest 4 MiB -- rows served by L2 cost ~5 cycles/CU instead of ~12 from the fabric
// (profiles/r01_experiments.md, ta_cost2).
template <int ITEMS, int TP>
__global__ void __launch_bounds__(256) gather_pack_xcd_kernel(const FrWordDesc *__restrict__ words, int n_words, int words_per_group,
                                                              const int32_t *__restrict__ idx, int idx_stride,
                                                              const float *__restrict__ dense, void *__restrict__ out,
                                                              int batch, int *__restrict__ err_flag, float scale) {
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
        if (b < batch) store_word<TP>(out, blk + (size_t)b * dst_stride, v[i], scale);
    }
    if (bad) atomicOr_system(err_flag, 1);
}

template <int TP>
static int gather_launch(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch, int *err_flag,
                         float scale, hipStream_t s) {
    static const int force = getenv("FR_GATHER_XCD") ? atoi(getenv("FR_GATHER_XCD")) : -1;  // experiment knob
    const bool xcd = force >= 0 ? force != 0 : (n_words >= 512 && batch >= 1024);
    if (xcd) {
        constexpr int ITEMS = 8;
        const int wpg = (n_words + 7) / 8;
        if (wpg <= 256) {
            const int bx = ((wpg + 63) / 64) * 64;
            dim3 grid(8 * ((batch + ITEMS - 1) / ITEMS));
            gather_pack_xcd_kernel<ITEMS, TP><<<grid, dim3(bx), 0, s>>>(words, n_words, wpg, idx, idx_stride, dense, out, batch, err_flag, scale);
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
        gather_pack_kernel<ITEMS, TP><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, scale);
    } else {
        constexpr int ITEMS = 4;
        dim3 grid((n_words + bx - 1) / bx, (batch + ITEMS - 1) / ITEMS);
        gather_pack_kernel<ITEMS, TP><<<grid, block, 0, s>>>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, scale);
    }
    KCHECK();
    return FR_OK;
}

// transport: FR_FC_FP32 (fp32 records, the reference's wire format), FR_FC_BF16 or FR_FC_FP8 (slice transport of the sharded mode)
int frk_gather(const FrWordDesc *words, int n_words, const int32_t *idx, int idx_stride, const float *dense, void *out, int batch, int *err_flag,
               int transport, int e_x, hipStream_t s) {
    if (n_words <= 0 || batch <= 0) return FR_OK;
    if (transport == FR_FC_BF16) return gather_launch<1>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, 1.0f, s);
    if (transport == FR_FC_FP8) return gather_launch<2>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, ldexpf(1.0f, e_x), s);
    return gather_launch<0>(words, n_words, idx, idx_stride, dense, out, batch, err_flag, 1.0f, s);
}
