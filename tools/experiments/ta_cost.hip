// Cost model of scattered vector loads on gfx950: cycles per wave-instruction as a function of how the 64 lanes'
// 16-byte (or 4/8-byte) pieces fall on cache lines.  Buffer is L2-resident (8 MB).  All CUs busy.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int W>  // bytes per lane: 4, 8, 16
__global__ void __launch_bounds__(256) k(const char *__restrict__ base, const unsigned *__restrict__ offs, int n_iter, unsigned *sink) {
    // offs[iter*64 + lane] = byte offset; same pattern for every wave (different iter rows per wave via blockIdx)
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    const unsigned *o = offs + ((blockIdx.x * 4 + (threadIdx.x >> 6)) % 64) * 64 * 16;
    for (int it = 0; it < n_iter; it += 8) {
        unsigned off[8];
#pragma unroll
        for (int j = 0; j < 8; j++) off[j] = o[((it + j) % 16) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (W == 16) { uint4 v = *reinterpret_cast<const uint4 *>(base + off[j]); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
            if (W == 8) { uint2 v = *reinterpret_cast<const uint2 *>(base + off[j]); acc ^= v.x ^ v.y; }
            if (W == 4) { unsigned v = *reinterpret_cast<const unsigned *>(base + off[j]); acc ^= v; }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}
int main() {
    const size_t BUF = 8u << 20;
    char *buf; (void)hipMalloc(&buf, BUF); (void)hipMemset(buf, 0, BUF);
    unsigned *doffs, *sink; (void)hipMalloc(&doffs, 64 * 16 * 64 * 4); (void)hipMalloc(&sink, 4);
    unsigned h[64 * 16 * 64];
    const char *names[] = {"broadcast (all lanes same 16B)", "contiguous 1 KiB", "each lane its own 128B line (random)", "lane pairs share a 32B row, rows random",
                           "lane quads share a 64B row, rows random", "8 lanes share a 128B row, rows random", "each lane own 64B-aligned piece, random", "16 lanes x 64B contiguous, 4 random rows of 256B"};
    for (int pat = 0; pat < 8; pat++) {
        for (int w = 0; w < 64; w++) for (int it = 0; it < 16; it++) for (int l = 0; l < 64; l++) {
            unsigned r = (unsigned)rand();
            unsigned off = 0;
            switch (pat) {
                case 0: off = (r % (BUF / 128)) * 128; off = h[(w * 16 + it) * 64] = (l == 0 ? off : h[(w * 16 + it) * 64]); break;
                case 1: off = (l == 0 ? (r % (BUF / 1024)) * 1024 : h[(w * 16 + it) * 64] + 16 * l); break;
                case 2: off = (r % (BUF / 128)) * 128; break;
                case 3: off = (l % 2 == 0) ? (r % (BUF / 32)) * 32 : h[(w * 16 + it) * 64 + l - 1] + 16; break;
                case 4: off = (l % 4 == 0) ? (r % (BUF / 64)) * 64 : h[(w * 16 + it) * 64 + l - 1] + 16; break;
                case 5: off = (l % 8 == 0) ? (r % (BUF / 128)) * 128 : h[(w * 16 + it) * 64 + l - 1] + 16; break;
                case 6: off = (r % (BUF / 64)) * 64; break;
                case 7: off = (l % 16 == 0) ? (r % (BUF / 256)) * 256 : h[(w * 16 + it) * 64 + l - 1] + 16; break;
            }
            h[(w * 16 + it) * 64 + l] = off;
        }
        (void)hipMemcpy(doffs, h, sizeof(h), hipMemcpyHostToDevice);
        for (int W : {16, 4}) {
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            const int n_iter = 256, blocks = 256 * 8;
            auto go = [&]() { if (W == 16) k<16><<<blocks, 256>>>(buf, doffs, n_iter, sink); else k<4><<<blocks, 256>>>(buf, doffs, n_iter, sink); };
            go(); go();
            (void)hipEventRecord(a, 0); for (int r = 0; r < 5; r++) go(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= 5;
            double instr_per_cu = (double)blocks * 4 * n_iter / 256;
            printf("%-52s W=%2d: %8.1f us, %6.1f cycles/wave-instr/CU (2.4 GHz), %6.0f GB/s useful\n", names[pat], W, ms * 1e3,
                   ms * 1e-3 * 2.4e9 / instr_per_cu, (double)blocks * 256 * n_iter * W / ms / 1e6);
        }
    }
    return 0;
}
