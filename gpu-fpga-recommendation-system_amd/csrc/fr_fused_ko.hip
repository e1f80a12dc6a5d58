#include "fr_device.h"

// ===================================================================================================
// fr_fused_tile_hk_kernel: the bf16 fused item-tile kernel, K-OUTER and persistent (BASELINE configs[2]: Model-B 1024, "bf16 MFMA FC,
// fused concat + first FC").  It replaces fr_fused_tile_h_kernel (fr_fused.hip), whose workgroup first gathered its 64 records with
// the matrix pipes idle (21-23 of 63-68 us per launch) and then walked FC1 in four chunks of 256 outputs, re-reading the whole record
// image from LDS per chunk.
//
// Here FC1 runs over K ONCE with all 1024 outputs of the 64 items in accumulators (a wave owns 128 outputs x 64 items = 128
// registers), so the record is consumed in K order -- which is the order the gather produces it in.  The record image therefore never
// exists as a whole: a ring of TWO slices of KGS k-groups (14 q8 rows = 14.2 KiB for Model-B) sits in LDS, and while the MFMAs of slice
// s run, the same waves write slice s + 1 into the other buffer and have the row loads of slices s + 2, s + 3 and the index loads of
// slice s + 4 in flight in registers.  The gather is spread over the whole tile instead of preceding it, and it keeps running through FC2 /
// FC3 for the NEXT tile of the workgroup (persistent: a workgroup walks tiles b, b + grid, ...), so a tile starts with its first two
// slices in LDS and two more on their way.  What the freed LDS pays for: the complete bf16 R1 image (1024 x 64 = 128 KiB) as FC2's B
// operand, so FC2 is one K-outer pass as well (64 outputs x 64 items per wave) and the chunk loop with its eight barriers is gone.
//   LDS  [ R1 128 KiB | X ring 2 x 2 KGS rows x 65 x 16 B | packed word descriptors 16 B each ]   (R2 / R3 / scratch overlay R1)
//   per k-group and wave: FC1 4 weight fragments + 2 B fragments -> 8 MFMAs (the chunked kernel: 1 + 2 -> 2: a quarter of its LDS reads)
// Weights stream from L2 through ONE register ring across FC1 -> FC2 -> FC3 -> the next tile's FC1 (all slot indices compile-time).
// Arithmetic per output: fp32 accumulation over k in ascending order, one bf16 rounding per activation -- the same values as the
// chunked kernel, bit for bit (the order of the sums inside an output is unchanged; only the order of the outputs moved).
// ===================================================================================================
namespace {

constexpr int HK_LD = 64;   // R1 / R2 / R3 images: 64 items per q8 row, no pad (fragment reads and tile stores are whole 512-byte runs)
constexpr int HK_LDX = 65;  // X ring: + 1 pad, the gather writes a column of rows per item
constexpr int HK_H1 = 1024, HK_H2 = 512, HK_H3 = 256;

// q8 weights Wh[k/8][n][8 bf16]: the fragment of k-group g (16 k) and n tile n0 is rows 2g + hk, columns n0 + lm -- one 16-byte buffer
// load per lane: constant per-lane byte offset in a VGPR, the (k-group, n tile) position as a wave-uniform SGPR offset (+ 512 t).
struct FtWk {
    __amdgpu_buffer_rsrc_t rs;  // base = weight matrix, num_records = its bytes (out-of-range lanes read 0, never fault)
    unsigned voff;              // (hk * N + lm) * 16
    unsigned row2;              // 2 * N * 16: byte step of one k-group (two q8 rows)
};
__device__ __forceinline__ FtWk ftk_w(const float4 *wq, int rows, int N, int hk, int lm) {
    FtWk w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(wq), 0, (unsigned)rows * (unsigned)N * 16u, 0x00020000);
    w.voff = (unsigned)(hk * N + lm) * 16u;
    w.row2 = 2u * (unsigned)N * 16u;
    return w;
}
__device__ __forceinline__ uint4 ftk_load(const FtWk &w, unsigned soff, int imm) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + imm, soff, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct HkTile {  // wave-uniform
    const int32_t *idx;
    const float *dense;
    float *scores;
    int batch, m0;
};

__device__ __forceinline__ void hk_store_tile(uint4 *img, const f32x16 &acc, int n_local, int m_local, int hk, int lm) {
    uint2 *h = reinterpret_cast<uint2 *>(img);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint2 v;
        v.x = pack_bf16x2(acc[4 * i + 0], acc[4 * i + 1]);
        v.y = pack_bf16x2(acc[4 * i + 2], acc[4 * i + 3]);
        h[((size_t)((n_local >> 3) + i) * HK_LD + m_local + lm) * 2 + hk] = v;  // n = n_local + 8 i + 4 hk + c
    }
}

// KG = K / 16 k-groups; KGS = k-groups per slice (4 KGS record words <= LW lanes); LW = lanes along the words of a slice (16 or 32);
// RD = weight ring slots.
template <int KG, int KGS, int LW, int RD>
__global__ void __launch_bounds__(512) fr_fused_tile_hk_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int NSL = (KG + KGS - 1) / KGS;          // slices per tile
    constexpr int IPT = LW / 8;                        // items per thread in the gather (512 threads = LW x 64 / IPT)
    constexpr int XROWS = 2 * KGS;                     // q8 rows of one X ring buffer
    constexpr int Q1 = 4 * KG, Q2 = Q1 + 128, Q = Q2 + 32;  // weight fragments of a tile: FC1 | FC2 | FC3
    constexpr int QP = (Q + RD - 1) / RD * RD;         // padded to a multiple of the ring: every tile starts at slot 0
    static_assert(NSL >= 6 && NSL % 2 == 0, "the gather pipeline runs 4 slices ahead and alternates two buffers: even NSL >= 6");
    static_assert(4 * KGS <= LW && (LW == 16 || LW == 32), "a slice's record words ride the lanes of one half / quarter wave");
    static_assert(RD % 4 == 0 && RD >= 8, "FC1 consumes 4 fragments per k-group");
    uint4 *R1 = lds;                                   // [128][64]
    uint4 *R2 = lds;                                   // [64][64], overlays R1 once FC2 has read it
    uint4 *R3 = lds + 64 * HK_LD;                      // [32][64]
    float *part = reinterpret_cast<float *>(lds + 96 * HK_LD);  // 8 x 64 partial scores
    uint4 *Xr = lds + 128 * HK_LD;                     // [2][XROWS][65]
    uint4 *Dsc = Xr + 2 * XROWS * HK_LDX;              // [n_words] packed descriptors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int n_tiles = a.n_batches * a.tiles_per_batch;

    auto tile_at = [&](int t) {
        const int bi = t / a.tiles_per_batch;
        HkTile r;
        r.idx = a.b[bi].idx, r.dense = a.b[bi].dense, r.scores = a.b[bi].scores, r.batch = a.b[bi].batch;
        r.m0 = (t - bi * a.tiles_per_batch) * 64;
        return r;
    };
    auto next_tile = [&](int t) {  // the workgroup's next non-empty tile after t (n_tiles: none); wave-uniform
        for (t += gridDim.x; t < n_tiles; t += gridDim.x) {
            const int bi = t / a.tiles_per_batch;
            if ((t - bi * a.tiles_per_batch) * 64 < a.b[bi].batch) break;
        }
        return t;
    };
    int t_cur = next_tile((int)blockIdx.x - (int)gridDim.x);
    if (t_cur >= n_tiles) return;

    unsigned long long *st = a.stamps ? a.stamps + 16ull * (8ull * blockIdx.x + wave) : nullptr;
    auto stamp = [&](int k) {  // diagnostic build aid (tools/experiments/fused_hk_stamps.py); values never feed an output
        if (st && lane == 0) st[k] = __builtin_amdgcn_s_memrealtime();
    };
    auto cstamp = [&](int k) {
        if (st && lane == 0) st[k] = __builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- weights: one ring of RD fragments across FC1 -> FC2 -> FC3 -> next tile ----
    const FtWk W1 = ftk_w(a.w1q, KG * 2, HK_H1, hk, lm), W2 = ftk_w(a.w2q, HK_H1 / 8, HK_H2, hk, lm), W3 = ftk_w(a.w3q, HK_H2 / 8, HK_H3, hk, lm);
    // n1 / n2 / n3 and the lane offsets below are re-declared opaque at the top of every tile: everything derived from them is tile-loop
    // invariant, and hoisted out of the loop the 380 SGPR offsets of the weight stream alone spilled 200 scalars into vector registers
    unsigned n1 = (unsigned)(128 * wave) * 16u, n2 = (unsigned)(64 * wave) * 16u, n3 = (unsigned)(32 * wave) * 16u;
    auto wfrag = [&](int q) -> uint4 {  // fragment q of the tile's weight stream (q compile-time after unrolling)
        if (q < Q1) return ftk_load(W1, (unsigned)(q >> 2) * W1.row2 + n1, 512 * (q & 3));                  // FC1: k-group q / 4, n tile q % 4
        if (q < Q2) return ftk_load(W2, (unsigned)((q - Q1) >> 1) * W2.row2 + n2, 512 * ((q - Q1) & 1));    // FC2: k-group, n tile
        if (q < Q) return ftk_load(W3, (unsigned)(q - Q2) * W3.row2 + n3, 0);                               // FC3: k-group
        return ftk_load(W3, n3, 0);                                                                         // pad: never consumed
    };
    uint4 ring[RD];
#pragma unroll
    for (int i = 0; i < RD; i++) ring[i] = wfrag(i);

    // ---- packed word descriptors -> LDS (read just in time by the gather: no registers, no vector-memory queue slots) ----
    for (int w = tid; w < a.n_words; w += 512) {
        const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
        const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
        // {src[31:0], src[47:32] | stride << 16, rows, idx byte offset | DENSE}
        Dsc[w] = make_uint4(d0.x, (d0.y & 0xFFFFu) | (d0.z << 16), d1.x, ((d0.w & ~FR_DESC_DENSE) * 4u) | (d0.w & FR_DESC_DENSE));
    }
    __syncthreads();

    // ---- gather pipeline state ----
    int wl = tid & (LW - 1);                    // word of the slice this thread moves            } re-derived per tile from the opaque
    int it0 = (tid / LW) * IPT;                 // first of its IPT items inside the tile         } thread id (see the tile loop)
    uint32_t idxr[IPT];                         // index values of the slice whose rows are loaded next
    uint4 rows[2][IPT];                         // row words in flight: two slices
    unsigned bad = 0u;                         // out-of-range index seen (a lane flag, OR-ed: no compare mask is kept)
    uint2 *Xh2 = reinterpret_cast<uint2 *>(Xr);
    auto slice_word = [&](int s) {  // this thread's record word in slice s; lanes past the slice repeat its last word (same row as their
        const int nw = 4 * (KG - KGS * s < KGS ? KG - KGS * s : KGS);  // neighbour: no extra line is fetched) and never store it
        return 4 * KGS * s + (wl < nw ? wl : nw - 1);
    };
    auto I_op = [&](const HkTile &t, int s) {  // index loads of slice s
        const uint4 d = Dsc[slice_word(s)];
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(t.idx), 0, (unsigned)t.batch * (unsigned)a.idx_stride * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < IPT; i++)   // items past the batch: out of the resource's bounds, 0 comes back (no branch)
            idxr[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(t.m0 + it0 + i) * (unsigned)a.idx_stride * 4u + (d.w & 0x7FFFFFFFu), 0, 0);
    };
    auto R_op = [&](const HkTile &t, int s, uint4 (&r)[IPT]) {  // row loads of slice s (its indices are in idxr)
        const uint4 d = Dsc[slice_word(s)];
        const bool dense = (d.w >> 31) != 0;
        const uint64_t base = (((uint64_t)(d.y & 0xFFFFu) << 32) | d.x) + (dense ? (uint64_t)reinterpret_cast<uintptr_t>(t.dense) : 0ull);
        const uint32_t stride = d.y >> 16, nrows = d.z;
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const unsigned m = (unsigned)(t.m0 + it0 + i);
            uint32_t x = idxr[i];
            const bool oob = !dense & (x >= nrows);  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933); here reported
            bad |= oob ? 1u : 0u;
            x = oob ? 0u : x;
            x = dense ? (m < (unsigned)t.batch ? m : 0u) : x;
            typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
            const u32x4_t q = *(gptr_t)(base + (uint64_t)x * stride);
            r[i] = make_uint4(q.x, q.y, q.z, q.w);
        }
    };
    auto W_op = [&](const HkTile &t, int s, const uint4 (&r)[IPT]) {  // slice s: fp32 rows -> bf16 -> X ring buffer s % 2
        const int nw = 4 * (KG - KGS * s < KGS ? KG - KGS * s : KGS);  // words of this slice
        if (wl < nw) {
            uint2 *xb = Xh2 + (size_t)(s & 1) * (XROWS * HK_LDX * 2);
#pragma unroll
            for (int i = 0; i < IPT; i++) {
                const uint32_t in = 0u - (uint32_t)(t.m0 + it0 + i < t.batch);  // all ones / zero: items past the batch are zero rows, branch-free
                uint2 hv;
                hv.x = pack_bf16x2(__uint_as_float(r[i].x), __uint_as_float(r[i].y)) & in;
                hv.y = pack_bf16x2(__uint_as_float(r[i].z), __uint_as_float(r[i].w)) & in;
                xb[((size_t)(wl >> 1) * HK_LDX + it0 + i) * 2 + (wl & 1)] = hv;  // slice word wl = half (wl & 1) of q8 row wl / 2
            }
        }
    };

    HkTile cur = tile_at(t_cur);
    // prologue: the first tile's slices 0, 1 into LDS, 2 and 3 requested, the indices of 4 requested (the state every tile starts in)
    I_op(cur, 0);
    R_op(cur, 0, rows[0]);
    I_op(cur, 1);
    R_op(cur, 1, rows[1]);
    W_op(cur, 0, rows[0]);
    I_op(cur, 2);
    R_op(cur, 2, rows[0]);
    W_op(cur, 1, rows[1]);
    I_op(cur, 3);
    R_op(cur, 3, rows[1]);
    I_op(cur, 4);
    stamp(1);
    cstamp(14);

    bool first = true;
    while (true) {
        const int t_nxt = next_tile(t_cur);
        const bool has_next = t_nxt < n_tiles;
        HkTile nxt = cur;
        if (has_next) nxt = tile_at(t_nxt);
        else nxt.batch = 0;  // no next tile: the run-ahead gather reads row 0 of every table (index loads out of bounds return 0) into buffers nobody consumes
        asm volatile("" : "+s"(n1), "+s"(n2), "+s"(n3));
        // lane geometry, re-derived per tile from an opaque copy of the thread id: hoisted out of the tile loop (they are all loop
        // invariant) the lane-constant LDS addresses of every phase were live through FC1 and spilled
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int hk = (tid_o >> 5) & 1, lm = tid_o & 31;
        wl = tid_o & (LW - 1), it0 = ((tid_o & 511) / LW) * IPT;
        const unsigned xlane = (unsigned)(128 * HK_LD + hk * HK_LDX + lm), rlane = (unsigned)(hk * HK_LD + lm);  // B-fragment lane bases (16-byte units): X ring, R1 / R2
        // gather slot of step s (s >= 1): write slice s + 1, request the rows of s + 3 and the indices of s + 4; slices >= NSL are the next tile's
        auto tref = [&](int s) -> const HkTile & { return s >= NSL ? nxt : cur; };

        // ---- FC1, K-outer: 128 outputs x 64 items per wave ----
        f32x16 acc1[4][2];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc1[t][mt][i] = 0.0f;
#pragma unroll
        for (int s = 0; s < NSL; s++) {
            const int kgs = KG - KGS * s < KGS ? KG - KGS * s : KGS;
            __syncthreads();  // slice s is complete in X[s % 2]; everybody is done reading slice s - 1
            const uint4 *xb = lds + xlane + (s & 1) * (XROWS * HK_LDX);
            uint4 b0 = xb[0], b1 = xb[32];
#pragma unroll
            for (int gl = 0; gl < kgs; gl++) {
                const int g = KGS * s + gl;
                const int gn = gl + 1 < kgs ? gl + 1 : gl;
                const uint4 bn0 = xb[(size_t)(2 * gn) * HK_LDX], bn1 = xb[(size_t)(2 * gn) * HK_LDX + 32];
                if (s >= 1 && gl == 0) W_op(tref(s + 1), (s + 1) % NSL, rows[(s + 1) & 1]);  // requested two steps ago; frees the register set for ...
                if (s >= 1 && gl == 1) {  // ... the loads of the run-ahead gather, issued together (one exposure of their latency to the ring's waits)
                    R_op(tref(s + 3), (s + 3) % NSL, rows[(s + 3) & 1]);
                    I_op(tref(s + 4), (s + 4) % NSL);
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int p = 4 * g + t;
                    acc1[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b0), acc1[t][0], 0, 0, 0);
                    acc1[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b1), acc1[t][1], 0, 0, 0);
                    ring[p % RD] = wfrag((p + RD) % QP);
                }
                __builtin_amdgcn_sched_barrier(0);
                b0 = bn0, b1 = bn1;
            }
        }
        if (first) stamp(2), cstamp(15);

        // ---- R1 -> LDS (bf16), FC2 K-outer over it: 64 outputs x 64 items per wave ----
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) hk_store_tile(R1, acc1[t][mt], 128 * wave + 32 * t, 32 * mt, hk, lm);
        __syncthreads();  // R1 complete; the X ring is free (every wave is past the last slice)
        if (first) stamp(3);
        f32x16 acc2[2][2];
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
        {
            const uint4 *bl = lds + rlane;
            uint4 b0 = bl[0], b1 = bl[32];
#pragma unroll
            for (int j = 0; j < 64; j++) {
                const int jn = j + 1 < 64 ? j + 1 : j;
                const uint4 bn0 = bl[(size_t)(2 * jn) * HK_LD], bn1 = bl[(size_t)(2 * jn) * HK_LD + 32];
                if (j == 2) {  // the next tile's gather: slice 1 to LDS (buffer 1 is free now), rows of 3, indices of 4
                    R_op(nxt, 3, rows[1]);
                    I_op(nxt, 4);
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int p = Q1 + 2 * j + t;
                    acc2[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b0), acc2[t][0], 0, 0, 0);
                    acc2[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b1), acc2[t][1], 0, 0, 0);
                    ring[p % RD] = wfrag((p + RD) % QP);
                }
                if (j == 1) W_op(nxt, 1, rows[1]);
                __builtin_amdgcn_sched_barrier(0);
                b0 = bn0, b1 = bn1;
            }
        }
        __syncthreads();  // every wave is done reading R1: R2 may overlay it
        if (first) stamp(4);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) hk_store_tile(R2, acc2[t][mt], 64 * wave + 32 * t, 32 * mt, hk, lm);
        __syncthreads();

        // ---- FC3: 32 outputs x 64 items per wave ----
        f32x16 acc3[2];
#pragma unroll
        for (int mt = 0; mt < 2; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc3[mt][i] = 0.0f;
        {
            const uint4 *bl = lds + rlane;  // R2 overlays R1
            uint4 b0 = bl[0], b1 = bl[32];
#pragma unroll
            for (int j = 0; j < 32; j++) {
                const int jn = j + 1 < 32 ? j + 1 : j;
                const uint4 bn0 = bl[(size_t)(2 * jn) * HK_LD], bn1 = bl[(size_t)(2 * jn) * HK_LD + 32];
                const int p = Q2 + j;
                acc3[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b0), acc3[0], 0, 0, 0);
                acc3[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[p % RD]), __builtin_bit_cast(bf16x8, b1), acc3[1], 0, 0, 0);
                ring[p % RD] = wfrag((p + RD) % QP);
                __builtin_amdgcn_sched_barrier(0);
                b0 = bn0, b1 = bn1;
            }
#pragma unroll
            for (int p = Q; p < QP; p++) ring[p % RD] = wfrag((p + RD) % QP);  // the pad positions pass their slots on to the next tile's FC1
        }
#pragma unroll
        for (int mt = 0; mt < 2; mt++) hk_store_tile(R3, acc3[mt], 32 * wave, 32 * mt, hk, lm);
        __syncthreads();
        if (first) stamp(5);
        {   // score[m] = sum_n wout[n] * R3[n][m] (bf16 x bf16, fp32 sum): 64 items x 8 slices of 4 q8 rows, fixed-order reduction
            const int il = tid & 63, sl = tid >> 6;
            const uint4 *wh = reinterpret_cast<const uint4 *>(a.wout);  // bf16 vector w[k], 8 per element
            float s = 0.0f;
            for (int q = 4 * sl; q < 4 * sl + 4; q++) {
                const uint4 r = R3[(size_t)q * HK_LD + il];
                const uint4 w = wh[q];
                const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    s = fmaf(__uint_as_float(ww[e] << 16), __uint_as_float(rr[e] << 16), s);
                    s = fmaf(__uint_as_float(ww[e] & 0xFFFF0000u), __uint_as_float(rr[e] & 0xFFFF0000u), s);
                }
            }
            part[sl * 64 + il] = s;
            __syncthreads();
            if (tid < 64 && cur.m0 + tid < cur.batch) {
                float t = part[tid];
#pragma unroll
                for (int i = 1; i < 8; i++) t += part[i * 64 + tid];
                cur.scores[cur.m0 + tid] = t;
            }
        }
        if (first) stamp(6);
        first = false;
        if (!has_next) break;
        cur = nxt;
        t_cur = t_nxt;
    }
    stamp(7);
    if (bad) atomicOr_system(a.err_flag, 1);
}

}  // namespace

// Every word descriptor must fit the packed 16-byte form: 48-bit source address, 16-bit row stride.
bool frk_fused_hk_ok(int K, int H1, int H2, int H3, const FrWordDesc *h_words, int n_words) {
    if ((K != 880 && K != 352) || H1 != HK_H1 || H2 != HK_H2 || H3 != HK_H3 || n_words != K / 4) return false;
    for (int w = 0; w < n_words; w++)
        if (h_words[w].stride >= 65536u || (h_words[w].src >> 48) != 0 || (h_words[w].idx_col & ~FR_DESC_DENSE) >= (1u << 28)) return false;
    return true;
}

template <int KG, int KGS, int LW, int RD>
static int fused_hk_launch_inst(const FrFusedArgs &a, int n_cu, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_hk_kernel<KG, KGS, LW, RD>, lds_once)) return rc_;
    const size_t lds = ((size_t)128 * HK_LD + (size_t)2 * 2 * KGS * HK_LDX + (size_t)a.n_words) * 16;
    const int tiles = a.n_batches * a.tiles_per_batch;
    fr_fused_tile_hk_kernel<KG, KGS, LW, RD><<<dim3(tiles < n_cu ? tiles : n_cu), dim3(512), lds, s>>>(a);
    KCHECK();
    return FR_OK;
}

// a.w1q/w2q/w3q/wout point at the bf16 q8 weights; a.tiles_per_batch counts 64-item tiles; one persistent workgroup per CU
int frk_fused_hk_launch(const FrFusedArgs &a, int n_cu, hipStream_t s) {
    if (a.K == 880) return fused_hk_launch_inst<55, 4, 16, 8>(a, n_cu, s);   // Model-B: 14 slices of 4 (3) k-groups
    if (a.K == 352) return fused_hk_launch_inst<22, 4, 16, 8>(a, n_cu, s);   // Model-A: 6 slices of 4 (2) k-groups
    FR_FAIL(FR_ERR_INVALID, "no K-outer bf16 fused instantiation for K=%d", a.K);
}
