// Does marking streaming traffic non-temporal keep the small embedding tables L2-resident?
// Model-C-like set: per half 8 x dim4 + 128 x dim8 small tables (SMALL_ROWS rows) and 48 x dim16 + 4 x dim32 big tables (2 M rows);
// workgroup b serves XCD group b % 8 = a fixed 1/8 of the record words.  Variants: plain | NT stores | NT stores + NT big-table loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Desc { unsigned long long src; unsigned stride, idx_col, rows, dst_off, dst_stride, big; };
template <int ITEMS, int MODE>
__global__ void __launch_bounds__(256) gk(const Desc *__restrict__ words, int n_words, int wpg, const int *__restrict__ idx, int idx_stride,
                                          uint4 *__restrict__ out, int batch) {
    const int group = blockIdx.x & 7, chunk = blockIdx.x >> 3;
    const int w = group * wpg + threadIdx.x;
    if ((int)threadIdx.x >= wpg || w >= n_words) return;
    const uint4 d0 = reinterpret_cast<const uint4 *>(words)[2 * w];
    const uint4 d1 = reinterpret_cast<const uint4 *>(words)[2 * w + 1];
    const unsigned long long src = ((unsigned long long)d0.y << 32) | d0.x;
    const unsigned stride = d0.z, idx_col = d0.w, dst_off = d1.y, dst_stride = d1.z, big = d1.w;
    const int b0 = chunk * ITEMS;
    unsigned id[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) id[i] = (unsigned)idx[(size_t)(b0 + i) * idx_stride + idx_col];
    uint4 v[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        const u32x4 *p = reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(src) + (unsigned long long)id[i] * stride);
        u32x4 t;
        if ((MODE & 2) && big) t = __builtin_nontemporal_load(p);
        else t = *p;
        v[i] = make_uint4(t.x, t.y, t.z, t.w);
    }
#pragma unroll
    for (int i = 0; i < ITEMS; i++) {
        u32x4 *q = reinterpret_cast<u32x4 *>(out + (size_t)(b0 + i) * dst_stride + dst_off);
        u32x4 t = {v[i].x, v[i].y, v[i].z, v[i].w};
        if (MODE & 1) __builtin_nontemporal_store(t, q);
        else *q = t;
    }
}
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++) f(i);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f(i);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / reps * 1000.f;
}
int main() {
    const int B = 4096, NW = 992;
    const unsigned small_rows = getenv("SMALL_ROWS") ? (unsigned)atoi(getenv("SMALL_ROWS")) : 20000;
    std::vector<int> dims; std::vector<unsigned> rows;
    for (int h = 0; h < 2; h++) { for (int i = 0; i < 8; i++) { dims.push_back(4); rows.push_back(small_rows); } for (int i = 0; i < 128; i++) { dims.push_back(8); rows.push_back(small_rows); }
        for (int i = 0; i < 48; i++) { dims.push_back(16); rows.push_back(2000000); } for (int i = 0; i < 4; i++) { dims.push_back(32); rows.push_back(2000000); } }
    const int T = (int)dims.size();
    size_t tot = 0, small = 0; std::vector<size_t> off; for (int t = 0; t < T; t++) { off.push_back(tot); size_t b = (size_t)rows[t] * dims[t] * 4; tot += b; if (rows[t] == small_rows) small += b; }
    char *tab; (void)hipMalloc(&tab, tot + 4096); (void)hipMemset(tab, 1, tot);
    std::vector<Desc> h; for (int t = 0; t < T; t++) for (int j = 0; j < dims[t] / 4; j++) {
        Desc d{}; d.src = (unsigned long long)(tab + off[t] + 16 * j); d.stride = dims[t] * 4; d.idx_col = t; d.rows = rows[t]; d.dst_off = (unsigned)h.size(); d.dst_stride = NW; d.big = rows[t] != small_rows; h.push_back(d); }
    while ((int)h.size() < NW) { Desc d = h[h.size() - 976]; d.dst_off = (unsigned)h.size(); h.push_back(d); }
    printf("tables %.1f MB total, small tables %.1f MB (%.1f MB per XCD group), %d words\n", tot / 1e6, small / 1e6, small / 8e6, (int)h.size());
    Desc *dw; (void)hipMalloc(&dw, sizeof(Desc) * NW); (void)hipMemcpy(dw, h.data(), sizeof(Desc) * NW, hipMemcpyHostToDevice);
    const int NB = 8; int *idx[NB]; std::vector<int> hi((size_t)B * T);
    for (int n = 0; n < NB; n++) { for (size_t i = 0; i < hi.size(); i++) hi[i] = rand() % rows[i % T]; (void)hipMalloc(&idx[n], hi.size() * 4); (void)hipMemcpy(idx[n], hi.data(), hi.size() * 4, hipMemcpyHostToDevice); }
    uint4 *out; (void)hipMalloc(&out, (size_t)B * NW * 16);
    const double bytes = (double)B * (NW * 16 * 2 + T * 4);
    const int wpg = (NW + 7) / 8;
#define RUN(MODE, name) { float us = timeit([&](int i) { gk<8, MODE><<<dim3(8 * (B / 8)), 128>>>(dw, NW, wpg, idx[i % NB], T, out, B); }, 50); \
    printf("%-44s %7.2f us  %6.0f GB/s(alg)\n", name, us, bytes / us / 1e3); }
    RUN(0, "xcd-partitioned, plain");
    RUN(1, "NT stores");
    RUN(2, "NT big-table loads");
    RUN(3, "NT stores + NT big-table loads");
    return 0;
}
