// Device-side helpers shared by the kernel translation units of libfleetrec (gfx950 only): vector typedefs of the MFMA
// builtins, the launch-error macro, bf16 / e4m3 packing, 16-byte buffer loads.
#pragma once
#include <atomic>
#include <cstdlib>
#include <mutex>

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

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: raise it once per (instantiation, device),
// safely from concurrent driver threads.  One FrLdsAttrOnce lives next to each launcher (function-local static).
struct FrLdsAttrOnce {
    std::mutex m;
    std::atomic<unsigned long long> done[4] = {};  // one bit per device ordinal (256 devices)
};
template <class Kernel>
static inline int fr_allow_full_lds(Kernel kernel, FrLdsAttrOnce &once) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 256) FR_FAIL(FR_ERR_HIP, "hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.done[dev >> 6].load(std::memory_order_acquire) & bit) return FR_OK;
    std::lock_guard<std::mutex> g(once.m);
    if (once.done[dev >> 6].load(std::memory_order_relaxed) & bit) return FR_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed on device %d", dev);
    once.done[dev >> 6].fetch_or(bit, std::memory_order_release);
    return FR_OK;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const __bf16 a = (__bf16)lo, b = (__bf16)hi;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, a) | ((uint32_t)__builtin_bit_cast(unsigned short, b) << 16);
}

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d, float scale) {
    const float lim = 448.0f;  // largest finite e4m3fn: values beyond it saturate (v_cvt_pk_fp8_f32 alone would return NaN above 448)
    // compares, not fminf / fmaxf: a NaN fails both and stays a NaN (the conversion keeps it), as in the fp32 and bf16 chains --
    // fminf(fmaxf(NaN, -448), 448) would quietly turn a corrupt table value into -448
    auto sat = [lim](float x) { x = x > lim ? lim : x; return x < -lim ? -lim : x; };
    a = sat(a * scale);
    b = sat(b * scale);
    c = sat(c * scale);
    d = sat(d * scale);
    int v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (uint32_t)v;
}
__device__ __forceinline__ uint32_t pack_fp8_word(const uint4 &v, float scale) {
    return pack_fp8x4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w), scale);
}

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
