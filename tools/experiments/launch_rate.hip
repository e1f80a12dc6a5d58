// Host launch throughput: trivial kernels from T threads on T streams.  Is hipLaunchKernel serialised across threads?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void noop(int *p) { if (p && threadIdx.x == 9999) *p = 1; }
int main() {
    for (int T : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(T);
        for (auto &s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        const int N = 20000;
        (void)hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([&, t]() {
            for (int i = 0; i < N; i++) { noop<<<64, 256, 0, st[t]>>>(nullptr); if ((i & 63) == 63) (void)hipStreamSynchronize(st[t]); }
            (void)hipStreamSynchronize(st[t]);
        });
        for (auto &x : th) x.join();
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("threads=%d: %.2f us per launch per thread, %.2f us per launch overall (%.0f k launches/s)\n", T, us / N, us / (N * T), N * T / us * 1e3);
        for (auto &s : st) (void)hipStreamDestroy(s);
    }
    // graph of 5 kernels: replay cost
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 5; i++) noop<<<64, 256, 0, s>>>(nullptr);
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 100; i++) (void)hipGraphLaunch(ge, s);
    (void)hipStreamSynchronize(s);
    const int N = 5000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { (void)hipGraphLaunch(ge, s); if ((i & 15) == 15) (void)hipStreamSynchronize(s); }
    (void)hipStreamSynchronize(s);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("graph of 5 noop kernels: %.2f us per replay\n", us / N);
    return 0;
}
