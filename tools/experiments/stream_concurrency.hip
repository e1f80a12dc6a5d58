// How many small-grid kernels from different HIP streams run concurrently on MI355X?
// Decides whether a "one workgroup per item tile, whole chain fused" design (8-16 workgroups per
// batch-256 launch) can fill 256 CUs from N worker streams.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(long long cycles, int *sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
int main(int argc, char **argv) {
    int wgs = argc > 1 ? atoi(argv[1]) : 8;
    long long cyc = argc > 2 ? atoll(argv[2]) : 10000;  // s_memtime ticks at 100 MHz -> 10000 = 100 us
    const int max_s = 64;
    std::vector<hipStream_t> st(max_s);
    for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int S : {1, 2, 4, 8, 16, 32, 64}) {
        const int per = 20;
        for (int i = 0; i < S; i++) spin<<<wgs, 256, 0, st[i]>>>(cyc, nullptr);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < per; r++)
            for (int i = 0; i < S; i++) spin<<<wgs, 256, 0, st[i]>>>(cyc, nullptr);
        hipDeviceSynchronize();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("streams=%2d wgs=%d: %.1f us total for %d launches -> %.2f us per launch, concurrency ~%.1f\n", S, wgs, us,
               per * S, us / (per * S), (per * S) * (cyc / 100.0) / us);
    }
    return 0;
}
