// Scattered 16-byte loads, NO L1 reuse: offsets are hashed per (wave, iteration, lane group).  Cost per distinct line when the
// line comes from L2 (8 MB buffer), Infinity Cache (128 MB) or HBM (4 GB).  lanes_per_row = 1,2,4,8 (row = 16*lanes_per_row bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned mix(unsigned h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
template <int LPR>
__global__ void __launch_bounds__(256) k(const char *__restrict__ base, unsigned long long n_rows, int n_iter, unsigned *sink) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned acc = 0;
    for (int it = 0; it < n_iter; it += 8) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            unsigned long long row = (unsigned long long)mix(wave * 7919u + (it + j) * 104729u + (lane / LPR) * 31u) % n_rows;
            v[j] = *reinterpret_cast<const uint4 *>(base + row * (16ull * LPR) + 16 * (lane % LPR));
        }
#pragma unroll
        for (int j = 0; j < 8; j++) acc ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    if (acc == 0x12345u) *sink = acc;
}
int main(int argc, char **argv) {
    const int uncached = argc > 1 ? atoi(argv[1]) : 0;
    printf("allocation: %s\n", uncached ? "hipDeviceMallocUncached" : "hipMalloc");
    unsigned *sink; (void)hipMalloc(&sink, 4);
    for (size_t mb : {4096ul}) {
        const size_t BUF = mb << 20;
        char *buf; if (uncached) { if (hipExtMallocWithFlags((void **)&buf, BUF, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc failed\n"); return 1; } } else (void)hipMalloc(&buf, BUF); (void)hipMemset(buf, 0, BUF);
        for (int lpr : {1, 2, 4, 8}) {
            const int n_iter = 64, blocks = 256 * 8;
            const unsigned long long n_rows = BUF / (16ull * lpr);
            auto go = [&]() {
                if (lpr == 1) k<1><<<blocks, 256>>>(buf, n_rows, n_iter, sink);
                if (lpr == 2) k<2><<<blocks, 256>>>(buf, n_rows, n_iter, sink);
                if (lpr == 4) k<4><<<blocks, 256>>>(buf, n_rows, n_iter, sink);
                if (lpr == 8) k<8><<<blocks, 256>>>(buf, n_rows, n_iter, sink);
            };
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            go(); go();
            (void)hipEventRecord(a, 0); for (int r = 0; r < 5; r++) go(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
            double instr_per_cu = (double)blocks * 4 * n_iter / 256;
            double rows = (double)blocks * 256 * n_iter / lpr;
            printf("buffer %5zu MB, row %3d B: %8.1f us, %6.1f cycles/wave-instr/CU, %5.2f cycles/row/CU, %6.1f G rows/s, %6.0f GB/s useful\n", mb, 16 * lpr,
                   ms * 1e3, ms * 1e-3 * 2.4e9 / instr_per_cu, ms * 1e-3 * 2.4e9 / (rows / 256), rows / ms / 1e6, rows * 16 * lpr / ms / 1e6);
        }
        (void)hipFree(buf);
    }
    return 0;
}
