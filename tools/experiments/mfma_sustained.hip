// Sustained v_mfma_f32_32x32x2_f32 rate: back-to-back launches for ~3 s, 8 waves per CU (2 per SIMD), 4 independent accumulators per
// wave.  Reports the TFLOP/s of consecutive windows and the shader clock (s_memtime cycles per s_memrealtime 100 MHz tick):
// the "peak" a kernel can be priced against under a sustained MFMA load, as opposed to the 2.4 GHz boost figure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(512) k(float *out, int iters, uint64_t *clk, int random) {
    f32x16 acc[4];
    for (int c = 0; c < 4; c++) for (int i = 0; i < 16; i++) acc[c][i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-4f;
    float ra[8], rb[8];  // RANDOM != 0: eight distinct random operand values per lane, cycled (realistic operand toggling)
    for (int u = 0; u < 8; u++) {
        unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u) ^ (u * 0x9E3779B9u);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
        ra[u] = random ? ((int)(h & 0xFFFFF) - 524288) * 1e-6f : a;
        rb[u] = random ? ((int)((h >> 12) & 0xFFFFF) - 524288) * 1e-6f : b;
    }
    uint64_t t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[u], rb[(u + c) & 7], acc[c], 0, 0, 0);
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int c = 0; c < 4; c++) for (int i = 0; i < 16; i++) s += acc[c][i];
    if (s == 12345.f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main(int argc, char **argv) {
    const int random = argc > 1 ? atoi(argv[1]) : 0;
    float *o; (void)hipMalloc(&o, 4);
    uint64_t *clk; (void)hipHostMalloc(&clk, 16, hipHostMallocMapped);
    const int iters = 20000, waves = 8, per_window = 20;   // one launch = 20000*32 MFMAs per wave = 2 per SIMD -> ~34 ms
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<<<256, 64 * waves>>>(o, 10, clk, random); (void)hipDeviceSynchronize();
    for (int w = 0; w < 6; w++) {
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < per_window; i++) k<<<256, 64 * waves>>>(o, iters, clk, random);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = 256.0 * waves * iters * 32.0 * 4096 * per_window;
        printf("window %d: %.1f ms, %.1f TFLOP/s, memtime/realtime = %.3f (x100 MHz)\n", w, ms, flops / (ms * 1e-3) / 1e12, (double)clk[0] / (double)clk[1]);
    }
    return 0;
}
