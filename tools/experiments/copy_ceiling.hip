// What one launch that reads R bytes cold and writes W bytes can reach on this chip at the SIZE of a Model-C batch-4096 gather
// (65.7 MB in, 65 MB out), back to back on one stream: the practical ceiling of the record-producing gather, ramp and
// end-of-kernel write-back included.  Sources rotate over NSRC buffers (so reads miss the Infinity Cache as table rows do), the
// destination is one buffer (as the record buffer is).  Store policy: plain / sc1 (write-through) / nt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int AUX, int PER>
__global__ void __launch_bounds__(256) copy_kernel(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, unsigned n) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0xffffffffu, 0x00020000);
    unsigned i = (blockIdx.x * PER) * 256 + threadIdx.x;
    u32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = i + k * 256 < n ? src[i + k * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (i + k * 256 < n) {
            if constexpr (AUX == 0) dst[i + k * 256] = v[k];
            else __builtin_amdgcn_raw_buffer_store_b128(v[k], rs, (i + k * 256) * 16u, 0, AUX);
        }
}

template <int AUX, int PER>
static float run(const std::vector<u32x4 *> &src, u32x4 *dst, unsigned n, int reps, hipStream_t s) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const unsigned grid = (n + 256 * PER - 1) / (256 * PER);
    for (int r = 0; r < 20; r++) copy_kernel<AUX, PER><<<grid, 256, 0, s>>>(src[r % src.size()], dst, n);
    CK(hipEventRecord(a, s));
    for (int r = 0; r < reps; r++) copy_kernel<AUX, PER><<<grid, 256, 0, s>>>(src[r % src.size()], dst, n);
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return 1e3f * ms / reps;
}

int main() {
    const int NSRC = 24, reps = 200;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (size_t mb : {32, 65, 130, 260, 1040}) {
        const size_t bytes = mb * 1000 * 1000 / 4096 * 4096;
        const unsigned n = (unsigned)(bytes / 16);
        const int nsrc = mb > 300 ? 4 : NSRC;
        std::vector<u32x4 *> src(nsrc);
        for (auto &p : src) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
        u32x4 *dst;
        CK(hipMalloc(&dst, bytes));
        CK(hipDeviceSynchronize());
        for (int rnd = 0; rnd < 2; rnd++) {
            const float t0 = run<0, 4>(src, dst, n, reps, s), t1 = run<16, 4>(src, dst, n, reps, s), t2 = run<2, 4>(src, dst, n, reps, s), t3 = run<0, 1>(src, dst, n, reps, s),
                        t4 = run<16, 1>(src, dst, n, reps, s);
            printf("copy %4zu MB in + %4zu MB out per launch: plain %.2f us = %.2f TB/s | sc1 %.2f us = %.2f TB/s | nt %.2f us = %.2f TB/s | 1 x 16 B per thread: plain %.2f us = %.2f, sc1 %.2f us = %.2f TB/s\n",
                   mb, mb, t0, 2e-6 * bytes / t0, t1, 2e-6 * bytes / t1, t2, 2e-6 * bytes / t2, t3, 2e-6 * bytes / t3, t4, 2e-6 * bytes / t4);
            fflush(stdout);
        }
        for (auto p : src) CK(hipFree(p));
        CK(hipFree(dst));
    }
    return 0;
}
