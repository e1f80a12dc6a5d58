// Peak L2 -> CU bandwidth for coalesced 16-byte-per-lane loads of an L2-resident buffer (what an LDS-tiled GEMM's staging loads see).
// Every workgroup streams the same `span` bytes (so they stay in the XCD's L2) `reps` times; waves = 4, 8 or 16 per CU;
// variant 0: buffer_load -> VGPR (summed), 1: buffer_load ... lds (direct to LDS, no VGPR round trip).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int VARIANT>
__global__ void __launch_bounds__(1024) k(const uint4 *buf, size_t span_el, int reps, unsigned *out) {
    __shared__ uint4 lds[1024 * 4];
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(buf), 0, (unsigned)(span_el * 16), 0x00020000);
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned acc = 0;
    const unsigned per_iter = nthr * 4;  // elements per workgroup iteration (4 loads per thread in flight)
    for (int r = 0; r < reps; r++) {
        for (unsigned base = 0; base + per_iter <= span_el; base += per_iter) {
            if (VARIANT == 0) {
                u32x4 v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (tid + j * nthr) * 16, base * 16, 0);
#pragma unroll
                for (int j = 0; j < 4; j++) acc += v[j].x ^ v[j].w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(lds + j * nthr + (tid & ~63)), 16, (tid + j * nthr) * 16,
                                                         base * 16, 0, 0);
            }
        }
        if (VARIANT == 1) {
            __builtin_amdgcn_s_waitcnt(0);
            acc += lds[tid].x;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int VARIANT> void run(const uint4 *buf, unsigned *out, int waves, size_t span_bytes) {
    const int reps = 40;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<VARIANT><<<256, 64 * waves>>>(buf, span_bytes / 16, 2, out);
    (void)hipEventRecord(e0, 0);
    k<VARIANT><<<256, 64 * waves>>>(buf, span_bytes / 16, reps, out);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)span_bytes * reps * 256;
    printf("variant %d waves/CU %2d span %5zu KiB: %.1f GB/s per CU, %.2f TB/s chip\n", VARIANT, waves, span_bytes / 1024, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
}
int main() {
    uint4 *buf; unsigned *out;
    (void)hipMalloc(&buf, 64 << 20); (void)hipMemset(buf, 1, 64 << 20); (void)hipMalloc(&out, 4);
    for (size_t span : {(size_t)256 << 10, (size_t)2 << 20, (size_t)16 << 20})
        for (int waves : {4, 8, 16}) { run<0>(buf, out, waves, span); run<1>(buf, out, waves, span); }
    return 0;
}
