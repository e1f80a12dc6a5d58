// v_mfma_f32_32x32x2_f32 issue rate per SIMD: operands in registers, W waves per workgroup (one workgroup per CU),
// CHAIN = 1 (every MFMA depends on the previous one) or 2/4 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAIN>
__global__ void __launch_bounds__(1024) k(float *out, int iters) {
    f32x16 acc[CHAIN];
    for (int c = 0; c < CHAIN; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-4f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++)
#pragma unroll
            for (int c = 0; c < CHAIN; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAIN; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
    if (s == 12345.f) out[0] = s;
}
template <int CHAIN> void run(int waves) {
    float *o; (void)hipMalloc(&o, 4);
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<CHAIN><<<256, 64 * waves>>>(o, 10);
    (void)hipEventRecord(e0, 0);
    k<CHAIN><<<256, 64 * waves>>>(o, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double mfma_per_simd = (double)iters * 16 * CHAIN * waves / 4.0;
    printf("waves/WG=%2d chain=%d: %.1f us, %.1f ns per MFMA per SIMD (= %.0f cycles at 2.4 GHz), %.1f TFLOP/s\n", waves, CHAIN, ms * 1e3,
           ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, 256.0 * waves * iters * 16 * CHAIN * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    for (int w : {4, 8, 16}) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
