// Procedural table / weight contents (FR_FILL_*, FR_WEIGHTS_*): ONE definition for the gfx950 fill kernels (fr_fill.hip) and the CPU
// back-end (fr_cpu.cpp), so that a table filled on either side holds the same bits.  Every step is exact in fp32 (a 24-bit integer
// times 2^-23, minus 1), so host and device cannot round differently.
#pragma once
#include <cstdint>

#include "fleetrec.h"

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define FR_HD static __host__ __device__ __forceinline__
#else
#define FR_HD static inline
#endif

FR_HD uint32_t fr_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// seed and table -> the per-table hash seed every element of the table starts from (uid = fr_table_uid)
FR_HD uint32_t fr_table_hash_seed(uint32_t seed, uint32_t uid) { return fr_fmix32(seed ^ (uid * 0x9E3779B1u)); }

// Bits of element (row, col) of the table `uid` under fill mode `mode`.
//  EVEN_ODD: host.cpp:66-88 init_vectors / embedding_47_krnl.cpp:869-897 (even rows 1.0f, odd rows 0.0f);
//  TAGGED  : source<<31 | class<<29 | table_id<<21 | (row & 0xffff)<<5 | (col & 31);
//  HASH    : hash32(seed, table, row, col) mapped to [-1, 1).
FR_HD uint32_t fr_content_bits(int mode, uint32_t h_seed_uid, uint32_t uid, uint64_t row, uint32_t col) {
    if (mode == FR_FILL_EVEN_ODD) return (row & 1) ? 0u : 0x3F800000u;
    if (mode == FR_FILL_TAGGED) {
        uint32_t source = uid >> 10, cls = (uid >> 8) & 3, tid = uid & 255;
        return (source << 31) | (cls << 29) | (tid << 21) | ((uint32_t)(row & 0xFFFF) << 5) | (col & 31);
    }
    uint32_t h = fr_fmix32(h_seed_uid ^ (uint32_t)row);
    h = fr_fmix32(h ^ (uint32_t)(row >> 32) ^ (col * 0x27D4EB2Fu));
    const float v = (float)(int32_t)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    return __builtin_bit_cast(uint32_t, v);
}

FR_HD uint32_t fr_weight_hash_seed(uint32_t seed, uint32_t layer) { return fr_fmix32(seed ^ ((layer + 1u) * 0x9E3779B1u)); }

// Element i of layer `layer`'s weights: 1.0f (init_array(w, n, 1.0f), cuda_server.c:152-160) or U(-1, 1) * scale.
FR_HD float fr_weight_value(int mode, uint32_t h_seed_layer, uint64_t i, float scale) {
    if (mode != FR_WEIGHTS_UNIFORM) return 1.0f;
    uint32_t h = fr_fmix32(h_seed_layer ^ (uint32_t)i);
    h = fr_fmix32(h ^ (uint32_t)(i >> 32));
    return ((float)(int32_t)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}
