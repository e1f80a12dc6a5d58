#include "fr_device.h"

// ===================================================================================================
// fr_fused_tile_kernel: the whole hot path of 32 items in ONE workgroup, activations never leave LDS.
//
// The stage pipeline above re-reads every activation and weight panel once per 32x32 output tile: at batch 256 the CUs'
// L2 ingest (not the MFMA pipe) bounds it at ~43 % of the f32 MFMA peak.  Here a workgroup owns 32 items for ALL layers:
//   gather  -> Xq[K/4][32] in LDS (the record of its 32 items; 16-byte q4 elements, row stride 33)
//   FC1     -> in chunks of 256 outputs (one 32-wide n tile per wave, full K, no split-K) -> R1 chunk in LDS
//   FC2     -> every wave keeps its H2/256 n tiles in accumulators and adds each R1 chunk as it appears
//   FC3     -> one n tile per wave from R2 in LDS, out layer from R3 in LDS, 32 scores stored
// Only the weights stream from L2 (each element once per workgroup: 4 MB for Model-A = 38 GB/s per CU at the MFMA rate);
// the B operand of every MFMA is a conflict-free ds_read_b128.  One launch carries the tiles of up to 64 queued batches
// of a worker (8 workgroups per batch of 256): 32 batches put one workgroup on every CU.
// Needs (K/4 + 64) * 33 * 16 B of LDS <= 160 KiB (one R1 buffer; two when they fit), H1 % 256 == 0, H2 in {256, 512}, H3 == 256,
// and K % 32 == 0 unless FC1 has a straight-line instantiation (K = 352, 880).
// ===================================================================================================
constexpr int FR_FT_LD = 33;  // LDS row stride in 16-byte elements (32 items + 1 pad: conflict-free writes and reads)

// acc[t] += Wq[k][n0 + 32 t ..] x B, for groups [g0, g0 + cnt) of 8 k; B group g' = g - g0 + gb0 lives in LDS at
// Bq[(2 g' + hk) * 33 + lm].  Double-buffered blocks of D groups: the next block's weight loads fly during the MFMAs.
template <int NT, int D>
__device__ __forceinline__ void ft_load_a(float4 (&r)[D][NT], const float4 *__restrict__ aq, int N, int g) {
#pragma unroll
    for (int i = 0; i < D; i++)
#pragma unroll
        for (int t = 0; t < NT; t++) r[i][t] = aq[(size_t)(2 * (g + i)) * N + 32 * t];
}

// MFMAs of one block.  Consecutive MFMAs of a wave always target different accumulators (the NT tiles when NT >= 2, else two
// partial accumulators for even / odd k that the caller adds once at the end), so that no MFMA waits for its predecessor.
template <int NT, int D>
__device__ __forceinline__ void ft_compute(f32x16 (&acc)[NT], f32x16 &alt, const float4 (&r)[D][NT], const uint4 *bl, int gb) {
    uint4 rb[D];
#pragma unroll
    for (int i = 0; i < D; i++) rb[i] = bl[(size_t)(2 * (gb + i)) * FR_FT_LD];
    __builtin_amdgcn_sched_barrier(0);  // weight loads of the other register set stay ahead of these MFMAs
#pragma unroll
    for (int i = 0; i < D; i++) {
        const float bx = __uint_as_float(rb[i].x), by = __uint_as_float(rb[i].y), bz = __uint_as_float(rb[i].z), bw = __uint_as_float(rb[i].w);
        if constexpr (NT == 1) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].x, bx, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].y, by, alt, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].z, bz, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][0].w, bw, alt, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].x, bx, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].y, by, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].z, bz, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(r[i][t].w, bw, acc[t], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Compile-time trip count, fully unrolled, weight operands in a RING of R register slots refilled one group at a time:
// group g's MFMAs are followed by the load of group g + R into the slot they just freed.  The loads are spread evenly over
// the MFMA stream (no bursts into the CU's ~70 GB/s L2 ingest) and the prefetch distance is R - 1 groups, i.e. almost the
// whole in-flight window -- a lone wave (its SIMD partner parked at a barrier) no longer exposes L2/Infinity-Cache latency
// once per block.  Straight-line code: hipcc keeps counted vmcnt waits (a branch in here would degrade them to vmcnt(0)).
// Weight operands come through BUFFER loads: one resource descriptor per weight matrix (SGPRs), one constant per-lane byte offset
// (a single VGPR, computed once per kernel), and the group / n-tile position as a uniform SGPR offset.  A global_load with a
// 64-bit per-lane address needs a v_lshl_add_u64 per load and reads two address VGPRs; measured on this loop shape
// (tools/experiments/mfma_loop) that alone stretches the MFMA issue interval from 65 to 75-77 cycles, buffer loads: 68.
struct FtW {
    __amdgpu_buffer_rsrc_t rs;  // base = weight matrix, num_records = its bytes (out-of-range lanes read 0, never fault)
    unsigned voff;              // (hk * N + lm) * 16
    unsigned row2;              // 2 * N * 16: byte step of one group (two q4 rows)
};
__device__ __forceinline__ FtW ft_w(const float4 *wq, int K, int N, int hk, int lm) {
    FtW w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(wq), 0, (unsigned)(K / 4) * (unsigned)N * 16u, 0x00020000);
    w.voff = (unsigned)(hk * N + lm) * 16u;
    w.row2 = 2u * (unsigned)N * 16u;
    return w;
}
__device__ __forceinline__ float4 ft_wload(const FtW &w, unsigned soff, int imm) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + imm, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Prologue of ft_gemm_ct, separated so that the caller can issue it BEFORE the barrier / LDS stores that precede the GEMM: the
// weights do not depend on them, and a GEMM that starts with a cold ring exposes a full L2 round trip on all 8 waves at once.
// n0 and g0 must be wave-uniform (they become the SGPR offset).
template <int NT, int R, int CNT>
__device__ __forceinline__ void ft_ring_fill(float4 (&ring)[R][NT], const FtW &w, int n0, int g0) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
#pragma unroll
    for (int g = 0; g < R && g < CNT; g++)
#pragma unroll
        for (int t = 0; t < NT; t++) ring[g][t] = ft_wload(w, s0 + (unsigned)g * w.row2, 512 * t);
    __builtin_amdgcn_sched_barrier(0);  // the loads stay where the caller put them
}

template <int NT, int R, int CNT>
__device__ __forceinline__ void ft_gemm_ct(f32x16 (&acc)[NT], float4 (&ring)[R][NT], const FtW &w, int n0, const uint4 *Bq, int gb0, int g0, int hk,
                                           int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
    // The refill's SGPR offset is a RUNNING value (one s_add per group), re-declared opaque every group: written as s0 + (g + R) * row2 in
    // fully unrolled code, hipcc computes dozens of them ahead of their loads and runs out of scalar registers -- Model-B's 110-group FC1
    // spilled 92 SGPRs into vector-register lanes and read them back inside the MFMA stream.
    unsigned so = s0 + (unsigned)R * w.row2;
    const uint4 *bl = Bq + (size_t)(2 * gb0 + hk) * FR_FT_LD + lm;
    f32x16 alt;
#pragma unroll
    for (int i = 0; i < 16; i++) alt[i] = 0.0f;
    uint4 bcur = bl[0];
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        const uint4 bnext = bl[(size_t)(2 * ((g + 1 < CNT) ? g + 1 : g)) * FR_FT_LD];  // next group's B fragment from LDS
        const float bx = __uint_as_float(bcur.x), by = __uint_as_float(bcur.y), bz = __uint_as_float(bcur.z), bw = __uint_as_float(bcur.w);
        const float4(&a4)[NT] = ring[g % R];
        if constexpr (NT == 1) {  // two partial accumulators (even / odd k): consecutive MFMAs are independent
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].x, bx, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].y, by, alt, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].z, bz, acc[0], 0, 0, 0);
            alt = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[0].w, bw, alt, 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].x, bx, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].y, by, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].z, bz, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[t].w, bw, acc[t], 0, 0, 0);
        }
        if (g + R < CNT) {  // compile-time after unrolling: refill the slot just consumed
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = ft_wload(w, so, 512 * t);
            so += w.row2;
            asm volatile("" : "+s"(so));
        }
        __builtin_amdgcn_sched_barrier(0);  // keep this interleave: the scheduler would otherwise sink the refills
        bcur = bnext;
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[0][i] += alt[i];
    }
}

// Run-time trip count (models without a straight-line instantiation): one register set + copies, branch-free loop body.
template <int NT, int D>
__device__ __forceinline__ void ft_gemm(f32x16 (&acc)[NT], const float4 *__restrict__ wq, int N, int n0, const uint4 *Bq, int gb0, int g0,
                                        int cnt, int hk, int lm) {
    const float4 *aq = wq + (size_t)hk * N + n0 + lm;
    const uint4 *bl = Bq + hk * FR_FT_LD + lm;
    float4 ra[D][NT], na[D][NT];
    f32x16 alt;
#pragma unroll
    for (int i = 0; i < 16; i++) alt[i] = 0.0f;
    const int nblk = cnt / D;  // launcher guarantees cnt % D == 0
    ft_load_a<NT, D>(ra, aq, N, g0);
    for (int blk = 0; blk < nblk; blk++) {
        const int nx = (blk + 1 < nblk) ? (blk + 1) : blk;  // last block re-loads itself: branch-free body
        ft_load_a<NT, D>(na, aq, N, g0 + nx * D);
        ft_compute<NT, D>(acc, alt, ra, bl, gb0 + blk * D);
#pragma unroll
        for (int i = 0; i < D; i++)
#pragma unroll
            for (int t = 0; t < NT; t++) ra[i][t] = na[i][t];
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; i++) acc[0][i] += alt[i];
    }
}

// store a 32(n) x 32(m) accumulator tile as q4 elements into an LDS operand image: n_local = tile's first n inside the image
__device__ __forceinline__ void ft_store_tile(uint4 *img, const f32x16 &acc, int n_local, int hk, int lm) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v;
        v.x = __float_as_uint(acc[4 * i + 0]);
        v.y = __float_as_uint(acc[4 * i + 1]);
        v.z = __float_as_uint(acc[4 * i + 2]);
        v.w = __float_as_uint(acc[4 * i + 3]);
        img[(size_t)((n_local >> 2) + 2 * i + hk) * FR_FT_LD + lm] = v;  // n = n_local + 8i + 4hk + c
    }
}

// T2W = H2 / 256 (n tiles of FC2 per wave); KG = K / 8 when FC1 has a straight-line instantiation, else 0 (run-time loop).
// WPE = waves per SIMD the kernel is built for.  2: one workgroup per CU, deep weight rings.  4: TWO workgroups per CU
// (<= 128 VGPRs, <= 80 KiB LDS, half-depth rings -- the same bytes in flight per CU) so that one workgroup's gather / barrier /
// epilogue phases run under the other one's MFMAs.  DB: double-buffered R1 (one barrier per chunk); !DB: single R1 buffer and
// R3 overlaying R2 -- the small-LDS layout that WPE 4 needs and that lets a K = 880 record (Model-B) fit at WPE 2.
// The gather phase of an item tile, software-pipelined (the code shape of gather_pack_stream_kernel): NB passes of 64 record words, IPT
// items per thread.  Every pass's descriptors first, then every pass's index loads (through a buffer resource whose bounds return 0 for
// the items past the batch: no branches), then the row loads with the next pass in flight while a pass is handed to `sink(w, i, v)`
// (w = record word, i = 0 .. IPT-1 the thread's item, v zeroed past the batch; lanes past the record are masked off).  Straight-line
// code: the chain descriptor -> index -> row is paid once per tile, not once per pass.  Returns the out-of-range flag.
template <int NB, int IPT, typename Sink>
__device__ __forceinline__ bool ft_gather_tile(const FrFusedArgs &a, const FrFusedBatch &bt, int m0, int wl, int ig, Sink &&sink) {
    uint64_t base[NB];
    uint32_t stride[NB], rows[NB], icol[NB];
    bool dense[NB];
#pragma unroll
    for (int p = 0; p < NB; p++) {
        const int w = 64 * p + wl < a.n_words ? 64 * p + wl : a.n_words - 1;   // a lane past the record repeats the last word (never stored)
        const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
        const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
        dense[p] = (d0.w & FR_DESC_DENSE) != 0;
        base[p] = (dense[p] ? (uint64_t)reinterpret_cast<uintptr_t>(bt.dense) : 0ull) + (((uint64_t)d0.y << 32) | d0.x);
        stride[p] = d0.z, rows[p] = d1.x, icol[p] = dense[p] ? 0u : d0.w * 4u;
    }
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(bt.idx), 0, (unsigned)bt.batch * (unsigned)a.idx_stride * 4u, 0x00020000);
    uint32_t id[NB][IPT];
#pragma unroll
    for (int p = 0; p < NB; p++)
#pragma unroll
        for (int i = 0; i < IPT; i++)
            id[p][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, (unsigned)(m0 + IPT * ig + i) * (unsigned)a.idx_stride * 4u + icol[p], 0, 0);
    bool bad = false;
    uint4 v[NB][IPT];
    auto load_rows = [&](int p) {
#pragma unroll
        for (int i = 0; i < IPT; i++) {
            const unsigned m = (unsigned)(m0 + IPT * ig + i);
            uint32_t r = id[p][i];
            const bool oob = !dense[p] & (r >= rows[p]);  // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933)
            bad |= oob;
            r = oob ? 0u : r;
            r = dense[p] ? (m < (unsigned)bt.batch ? m : 0u) : r;
            typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
            const u32x4_t q = *(gptr_t)(base[p] + (uint64_t)r * stride[p]);
            v[p][i] = make_uint4(q.x, q.y, q.z, q.w);
        }
    };
    load_rows(0);
#pragma unroll
    for (int p = 0; p < NB; p++) {
        if (p + 1 < NB) load_rows(p + 1);
        const int w = 64 * p + wl;
        if (w < a.n_words) {
#pragma unroll
            for (int i = 0; i < IPT; i++) sink(w, i, (m0 + IPT * ig + i < bt.batch) ? v[p][i] : make_uint4(0u, 0u, 0u, 0u));
        }
    }
    return bad;
}

template <int T2W, int KG, int WPE, bool DB>
__global__ void __launch_bounds__(512, WPE) fr_fused_tile_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int RD = (WPE == 2) ? 1 : 2;  // ring depth divisor: half-depth rings when two workgroups share a CU's registers
    const int KQ = a.K / 4;
    // LDS: [ Xq: KQ rows | R1 chunk buffer 0: 64 rows | R1 chunk buffer 1: 64 rows ]; R2 (H2/4 rows) overlays the start once
    // Xq and the R1 chunks are dead, R3 (64 rows) follows R2.
    uint4 *Xq = lds;                                  // [KQ][33]
    uint4 *R1b[2] = {Xq + (size_t)KQ * FR_FT_LD, Xq + (size_t)(KQ + (DB ? 64 : 0)) * FR_FT_LD};  // [64][33] each: 256 outputs of FC1
    uint4 *R2 = lds;                                  // [H2/4][33]
    uint4 *R3 = DB ? lds + (size_t)(a.H2 / 4) * FR_FT_LD : lds;  // [64][33]; WPE 4: overlays R2 once FC3 has read it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * 32;
    if (m0 >= bt.batch) return;
    int n_st = 0;
    auto stamp = [&]() {  // diagnostic build aid; wave 0 lane 0 only, values never feed an output
        if (a.stamps && tid == 0 && n_st < 14) {
            a.stamps[16ull * blockIdx.x + n_st] = __builtin_amdgcn_s_memrealtime();
            if (n_st == 1) a.stamps[16ull * blockIdx.x + 14] = __builtin_amdgcn_s_memtime();   // shader-clock cycles at "gather done"
            if (n_st == 11) a.stamps[16ull * blockIdx.x + 15] = __builtin_amdgcn_s_memtime();  // ... and at "FC3 + R3"
        }
        n_st++;
    };
    auto wstamp = [&](int k) {  // per-wave stamps around chunk 1: FC1 done, barrier released, FC2 done, next FC1 done
        if (a.stamps && lane == 0) a.stamps[16ull * gridDim.x + (8ull * blockIdx.x + wave) * 4 + k] = __builtin_amdgcn_s_memrealtime();
    };
    stamp();
    const FtW W1 = ft_w(a.w1q, a.K, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2, a.H3, hk, lm);
    constexpr int RA = 16 / RD, RB = 24 / T2W / RD;     // ring slots: FC1 / FC3 (one n tile), FC2 (T2W n tiles)
    constexpr int KG1 = KG > 0 ? KG : 1;
    float4 ring1[RA][1];                                // FC1's weight ring; chunk 0's first groups are requested before the gather
    if constexpr (KG > 0) ft_ring_fill<1, RA, KG1>(ring1, W1, 32 * wave, 0);

    // ---- gather: lanes along record words (a row is read by dim/4 adjacent lanes), 4 items per thread ----
    if constexpr (KG > 0) {
        const int wl = tid & 63, ig = tid >> 6;
        constexpr int NB = (2 * KG1 + 63) / 64;   // K = 8 KG floats = 2 KG record words
        const bool bad = ft_gather_tile<NB, 4>(a, bt, m0, wl, ig, [&](int w, int i, const uint4 &x) { Xq[(size_t)w * FR_FT_LD + 4 * ig + i] = x; });
        if (bad) atomicOr_system(a.err_flag, 1);
    } else {
        const int wl = tid & 63, ig = tid >> 6;
        bool bad = false;
        for (int w0 = 0; w0 < a.n_words; w0 += 64) {
            const int w = w0 + wl;
            if (w < a.n_words) {
                const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
                const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
                const uint64_t src = ((uint64_t)d0.y << 32) | d0.x;
                const uint32_t stride = d0.z, idx_col = d0.w, rows = d1.x;
                const bool is_dense = (idx_col & FR_DESC_DENSE) != 0;
                const char *base = is_dense ? reinterpret_cast<const char *>(bt.dense) + src : reinterpret_cast<const char *>(src);
                uint32_t id[4];
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int m = m0 + 4 * ig + i;
                    id[i] = 0;
                    if (m < bt.batch) id[i] = is_dense ? (uint32_t)m : (uint32_t)bt.idx[(size_t)m * a.idx_stride + idx_col];
                    if (!is_dense && id[i] >= rows) {
                        bad = true;
                        id[i] = 0;
                    }
                }
                uint4 v[4];
#pragma unroll
                for (int i = 0; i < 4; i++) v[i] = *reinterpret_cast<const uint4 *>(base + (uint64_t)id[i] * stride);
#pragma unroll
                for (int i = 0; i < 4; i++)
                    Xq[(size_t)w * FR_FT_LD + 4 * ig + i] = (m0 + 4 * ig + i < bt.batch) ? v[i] : make_uint4(0u, 0u, 0u, 0u);
            }
        }
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();
    stamp();

    // ---- FC1 in chunks of 256 outputs, FC2 accumulating each chunk ----
    f32x16 acc2[T2W];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc2[t][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1];
#pragma unroll
        for (int i = 0; i < 16; i++) acc1[0][i] = 0.0f;
        if constexpr (KG > 0) ft_gemm_ct<1, RA, KG1>(acc1, ring1, W1, c * 256 + 32 * wave, Xq, 0, 0, hk, lm);
        else ft_gemm<1, 4>(acc1, a.w1q, a.H1, c * 256 + 32 * wave, Xq, 0, 0, a.K / 8, hk, lm);
        stamp();
        if (c == 1) wstamp(0);
        if (c == 2) wstamp(3);
        float4 ring2[RB][T2W];  // FC2's first weight groups are requested before the R1 store and the barrier
        ft_ring_fill<T2W, RB, 32>(ring2, W2, 32 * T2W * wave, 32 * c);
        uint4 *R1 = R1b[c & 1];  // double-buffered: ONE barrier per chunk (the buffer written now was last read two chunks ago)
        if (!DB && c > 0) __syncthreads();  // single buffer: every wave must be done with the previous chunk's FC2
        ft_store_tile(R1, acc1[0], 32 * wave, hk, lm);
        __syncthreads();
        stamp();
        // FC2: K range [256 c, 256 c + 256) = groups [32 c, 32 c + 32); this wave's n tiles start at 32 * T2W * wave
        if (c == 1) wstamp(1);
        ft_gemm_ct<T2W, RB, 32>(acc2, ring2, W2, 32 * T2W * wave, R1, 0, 32 * c, hk, lm);
        if (c == 1) wstamp(2);
        if constexpr (KG > 0) {  // next chunk's FC1 ring (the last chunk re-requests its own: branch-free, 16 harmless loads)
            const int cn = (c + 1 < n_chunks) ? c + 1 : c;
            ft_ring_fill<1, RA, KG1>(ring1, W1, cn * 256 + 32 * wave, 0);
        }
    }
    float4 ring3[RA][1];  // FC3's first weight groups, requested before the two barriers around the R2 store
    ft_ring_fill<1, RA, 32 * T2W>(ring3, W3, 32 * wave, 0);
    __syncthreads();  // every wave is done with Xq and both R1 buffers: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++) ft_store_tile(R2, acc2[t], 32 * (T2W * wave + t), hk, lm);
    __syncthreads();  // R2 complete
    stamp();

    // ---- FC3: one n tile per wave (H3 == 256), then the output layer ----
    f32x16 acc3[1];
#pragma unroll
    for (int i = 0; i < 16; i++) acc3[0][i] = 0.0f;
    ft_gemm_ct<1, RA, 32 * T2W>(acc3, ring3, W3, 32 * wave, R2, 0, 0, hk, lm);  // H2 / 8 = 32 * T2W groups
    if (!DB) __syncthreads();                       // R3 overlays R2: every wave must have finished reading R2
    ft_store_tile(R3, acc3[0], 32 * wave, hk, lm);  // R3 image [H3/4][33]
    __syncthreads();
    stamp();
    {   // score[m] = sum_n wout[n] * R3[n][m]: 32 items x 16 slices of 4 q4 rows, fixed-order reduction through LDS (reuses Xq)
        const int il = tid & 31, sl = tid >> 5;
        const int rows_per = (a.H3 / 4) / 16;
        float s = 0.0f;
        for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
            const uint4 r = R3[(size_t)q * FR_FT_LD + il];
            const float4 w4 = reinterpret_cast<const float4 *>(a.wout)[q];
            s = fmaf(w4.x, __uint_as_float(r.x), s);
            s = fmaf(w4.y, __uint_as_float(r.y), s);
            s = fmaf(w4.z, __uint_as_float(r.z), s);
            s = fmaf(w4.w, __uint_as_float(r.w), s);
        }
        float *part = reinterpret_cast<float *>(R3 + (size_t)64 * FR_FT_LD);  // 2 KiB behind R3
        part[sl * 32 + il] = s;
        __syncthreads();
        if (tid < 32 && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < 16; i++) t += part[i * 32 + tid];
            bt.scores[m0 + tid] = t;
        }
    }
    stamp();
}

size_t frk_fused_lds_bytes(int K, int H2, int wpe) {  // wpe 4 stands for the single-buffer layout here
    if (wpe == 4) {  // Xq + one R1 buffer | R2, then R3 + scratch over R2
        const size_t phase1 = (size_t)(K / 4) + 64, phase2 = (size_t)(H2 / 4) > 68 ? (size_t)(H2 / 4) : 68;
        return (phase1 > phase2 ? phase1 : phase2) * FR_FT_LD * 16;
    }
    const size_t phase1 = (size_t)(K / 4) + 128;      // Xq + two R1 chunk buffers
    const size_t phase2 = (size_t)(H2 / 4) + 64 + 4;  // R2 + R3 + the 2 KiB reduction scratch
    return (phase1 > phase2 ? phase1 : phase2) * FR_FT_LD * 16;
}

bool frk_fused_ok(int K, int H1, int H2, int H3) {
    const bool straight = (K == 352 || K == 880) && H2 == 512;  // straight-line FC1 instantiations (Model-A, Model-B)
    if ((!straight && K % 32) || H1 % 256 || (H2 != 256 && H2 != 512) || H3 != 256) return false;  // run-time loop: K/8 groups in blocks of 4
    return frk_fused_lds_bytes(K, H2, 4) <= 160 * 1024;  // the single-buffer layout is the smallest
}

// two workgroups per CU need <= 80 KiB each (FR_FUSED_WPE=2 forces the one-workgroup build, for A/B measurements)
static int fused_wpe(int K, int H2) {
    const int forced = FR_KNOB_ONCE("FUSED_WPE", 0);
    if (forced == 2 || forced == 4) return forced == 4 && frk_fused_lds_bytes(K, H2, 4) > 80 * 1024 ? 2 : forced;
    return frk_fused_lds_bytes(K, H2, 4) <= 80 * 1024 ? 4 : 2;
}

template <int T2W, int KG, int WPE, bool DB>
static int fused_launch_inst(const FrFusedArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_kernel<T2W, KG, WPE, DB>, lds_once)) return rc_;
    fr_fused_tile_kernel<T2W, KG, WPE, DB><<<grid, dim3(512), lds, s>>>(a);
    fr_note_kernel("fr_fused_tile_kernel<%d, %d, %d, %s>", T2W, KG, WPE, DB ? "true" : "false");
    KCHECK();
    return FR_OK;
}

int frk_fused_launch(const FrFusedArgs &a, hipStream_t s) {
    int wpe = fused_wpe(a.K, a.H2);
    const bool db = wpe == 2 && frk_fused_lds_bytes(a.K, a.H2, 2) <= 160 * 1024;
    const size_t lds = frk_fused_lds_bytes(a.K, a.H2, db ? 2 : 4);
    dim3 grid(a.n_batches * a.tiles_per_batch);
    if (a.H2 == 512) {
        if (a.K == 352) {  // Model-A: straight-line FC1
            if (wpe == 4) return fused_launch_inst<2, 44, 4, false>(a, grid, lds, s);
            return fused_launch_inst<2, 44, 2, true>(a, grid, lds, s);
        }
        if (a.K == 880) return fused_launch_inst<2, 110, 2, false>(a, grid, lds, s);  // Model-B: 150 KiB with one R1 buffer
        if (db) return fused_launch_inst<2, 0, 2, true>(a, grid, lds, s);
        return fused_launch_inst<2, 0, 2, false>(a, grid, lds, s);
    }
    if (db) return fused_launch_inst<1, 0, 2, true>(a, grid, lds, s);
    return fused_launch_inst<1, 0, 2, false>(a, grid, lds, s);
}

// ===================================================================================================
// fr_fused_tile_m2_kernel<KG>: the fp32 fused item-tile kernel with 64 items (two m tiles) per workgroup.
// Every weight fragment now feeds two m tiles: half as many weight loads per MFMA (they cost MFMA issue slots, see
// tools/experiments/mfma_loop) and half as many barriers, ring cold starts and gather phases per item.  Needs the whole CU
// (158 KiB of LDS: Xq[K/4][65] + ONE R1 buffer; R3 overlays R2) and 64 queued batches of 256 to put one workgroup on every CU,
// so it is used only when the launch group is 64 (fr_ctx_set_stream_group) -- twice the queueing latency of the 32-item kernel.
// Same arithmetic as fr_fused_tile_kernel: full-K k-ordered f32 sums, bit-identical scores.
// ===================================================================================================
constexpr int FR_M2_LD = 65;

template <int NT, int MT, int R, int CNT>
__device__ __forceinline__ void ftm_gemm_ct(f32x16 (&acc)[NT][MT], float4 (&ring)[R][NT], const FtW &w, int n0, const uint4 *Bq, int gb0, int g0, int hk,
                                            int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;
    const uint4 *bl = Bq + (size_t)(2 * gb0 + hk) * FR_M2_LD + lm;
    uint4 bcur[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) bcur[mt] = bl[32 * mt];
    // NT == 1: even / odd k partial accumulators exactly as in fr_fused_tile_kernel, so that both kernels return the same bits
    f32x16 alt[MT];
    if constexpr (NT == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) alt[mt][i] = 0.0f;
    }
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        uint4 bnext[MT];
        const int gn = (g + 1 < CNT) ? g + 1 : g;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bnext[mt] = bl[(size_t)(2 * gn) * FR_M2_LD + 32 * mt];  // next group's B fragments from LDS
        const float4(&a4)[NT] = ring[g % R];
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int t = 0; t < NT; t++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const float av = c == 0 ? a4[t].x : c == 1 ? a4[t].y : c == 2 ? a4[t].z : a4[t].w;
                    const uint32_t bv = c == 0 ? bcur[mt].x : c == 1 ? bcur[mt].y : c == 2 ? bcur[mt].z : bcur[mt].w;
                    if (NT == 1 && (c & 1)) alt[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, __uint_as_float(bv), alt[mt], 0, 0, 0);
                    else acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, __uint_as_float(bv), acc[t][mt], 0, 0, 0);
                }
        if (g + R < CNT) {
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = ft_wload(w, s0 + (unsigned)(g + R) * w.row2, 512 * t);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bcur[mt] = bnext[mt];
    }
    if constexpr (NT == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[0][mt][i] += alt[mt][i];
    }
}

__device__ __forceinline__ void ftm_store_tile(uint4 *img, const f32x16 &acc, int n_local, int m_local, int hk, int lm) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint4 v;
        v.x = __float_as_uint(acc[4 * i + 0]);
        v.y = __float_as_uint(acc[4 * i + 1]);
        v.z = __float_as_uint(acc[4 * i + 2]);
        v.w = __float_as_uint(acc[4 * i + 3]);
        img[(size_t)((n_local >> 2) + 2 * i + hk) * FR_M2_LD + m_local + lm] = v;  // n = n_local + 8i + 4hk + c
    }
}

template <int KG>
__global__ void __launch_bounds__(512) fr_fused_tile_m2_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int LD = FR_M2_LD, MT = 2, T2W = 2, TI = 64;
    constexpr int RA = 12, RB = 8;
    const int KQ = a.K / 4;
    uint4 *Xq = lds;                     // [KQ][65]
    uint4 *R1 = Xq + (size_t)KQ * LD;    // [64][65]: 256 outputs of FC1
    uint4 *R2 = lds;                     // [H2/4][65], overlays Xq / R1 once they are dead
    uint4 *R3 = lds;                     // [64][65], overlays R2 once FC3 has read it
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * TI;
    if (m0 >= bt.batch) return;
    const FtW W1 = ft_w(a.w1q, a.K, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2, a.H3, hk, lm);
    float4 ring1[RA][1];
    ft_ring_fill<1, RA, KG>(ring1, W1, 32 * wave, 0);  // FC1 chunk 0's first weight groups: requested before the gather

    {   // ---- gather: lanes along record words, 8 items per thread ----
        const int wl = tid & 63, ig = tid >> 6;
        constexpr int NB = (2 * KG + 63) / 64;   // K = 8 KG floats = 2 KG record words
        const bool bad = ft_gather_tile<NB, 8>(a, bt, m0, wl, ig, [&](int w, int i, const uint4 &x) { Xq[(size_t)w * LD + 8 * ig + i] = x; });
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();

    f32x16 acc2[T2W][MT];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc1[0][mt][i] = 0.0f;
        ftm_gemm_ct<1, MT, RA, KG>(acc1, ring1, W1, c * 256 + 32 * wave, Xq, 0, 0, hk, lm);
        float4 ring2[RB][T2W];  // FC2's first weight groups are requested before the R1 store and the barriers
        ft_ring_fill<T2W, RB, 32>(ring2, W2, 32 * T2W * wave, 32 * c);
        if (c > 0) __syncthreads();  // single R1 buffer: every wave must be done with the previous chunk's FC2
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ftm_store_tile(R1, acc1[0][mt], 32 * wave, 32 * mt, hk, lm);
        __syncthreads();
        ftm_gemm_ct<T2W, MT, RB, 32>(acc2, ring2, W2, 32 * T2W * wave, R1, 0, 32 * c, hk, lm);
        const int cn = (c + 1 < n_chunks) ? c + 1 : c;  // the last chunk re-requests its own: branch-free, harmless
        ft_ring_fill<1, RA, KG>(ring1, W1, cn * 256 + 32 * wave, 0);
    }
    float4 ring3[RA][1];
    ft_ring_fill<1, RA, 32 * T2W>(ring3, W3, 32 * wave, 0);
    __syncthreads();  // Xq and R1 dead: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ftm_store_tile(R2, acc2[t][mt], 32 * (T2W * wave + t), 32 * mt, hk, lm);
    __syncthreads();

    f32x16 acc3[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc3[0][mt][i] = 0.0f;
    ftm_gemm_ct<1, MT, RA, 32 * T2W>(acc3, ring3, W3, 32 * wave, R2, 0, 0, hk, lm);
    __syncthreads();  // R3 overlays R2: every wave must have finished reading R2
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ftm_store_tile(R3, acc3[0][mt], 32 * wave, 32 * mt, hk, lm);
    __syncthreads();
    {   // score[m] = sum_n wout[n] * R3[n][m]: 64 items x 16 slices of 4 q4 rows (two slices per thread), fixed-order reduction
        // through LDS -- the same partial sums in the same order as fr_fused_tile_kernel
        const int il = tid & 63, sg = tid >> 6;
        const int rows_per = (a.H3 / 4) / 16;
        float *part = reinterpret_cast<float *>(R3 + (size_t)(a.H3 / 4) * LD);
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
            const int sl = 2 * sg + h2;
            float s = 0.0f;
            for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
                const uint4 r = R3[(size_t)q * LD + il];
                const float4 w4 = reinterpret_cast<const float4 *>(a.wout)[q];
                s = fmaf(w4.x, __uint_as_float(r.x), s);
                s = fmaf(w4.y, __uint_as_float(r.y), s);
                s = fmaf(w4.z, __uint_as_float(r.z), s);
                s = fmaf(w4.w, __uint_as_float(r.w), s);
            }
            part[sl * 64 + il] = s;
        }
        __syncthreads();
        if (tid < 64 && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < 16; i++) t += part[i * 64 + tid];
            bt.scores[m0 + tid] = t;
        }
    }
}

bool frk_fused_m2_ok(int K, int H1, int H2, int H3) { return K == 352 && H1 % 256 == 0 && H2 == 512 && H3 == 256; }

// a.tiles_per_batch counts 64-item tiles
int frk_fused_m2_launch(const FrFusedArgs &a, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_m2_kernel<44>, lds_once)) return rc_;
    const size_t rows1 = (size_t)(a.K / 4) + 64, rows2 = (size_t)(a.H2 / 4), rows3 = 64 + 4;  // R3 + 4 KiB of reduction scratch
    const size_t rows = rows1 > rows2 ? (rows1 > rows3 ? rows1 : rows3) : (rows2 > rows3 ? rows2 : rows3);
    fr_fused_tile_m2_kernel<44><<<dim3(a.n_batches * a.tiles_per_batch), dim3(512), rows * FR_M2_LD * 16, s>>>(a);
    fr_note_kernel("fr_fused_tile_m2_kernel<44>");
    KCHECK();
    return FR_OK;
}

// ===================================================================================================
// fr_fused_tile_h_kernel: the fused item-tile kernel in bf16 (BASELINE configs 2/3: "bf16 MFMA FC, fused concat + first FC").
// Same phases as fr_fused_tile_kernel, q8 operands (8 bf16 per 16 bytes = one v_mfma_f32_32x32x16_bf16 operand per lane),
// fp32 accumulation, ONE bf16 rounding per hidden activation.  A bf16 MFMA is 16x cheaper than the f32 one while the weights
// are only 2x smaller, so the kernel is bound by streaming the weights into the CU (~70 GB/s): a workgroup therefore owns
// 32 * MT items (MT = 2: every weight fragment feeds two m tiles) and the weight ring is as deep as the registers allow.
// ===================================================================================================
template <int NT, int MT, int R, int CNT>
__device__ __forceinline__ void fth_gemm_ct(f32x16 (&acc)[NT][MT], const FtW &w, int n0, const uint4 *Bh, int ld, int gb0, int g0, int hk, int lm) {
    const unsigned s0 = (unsigned)g0 * w.row2 + (unsigned)n0 * 16u;  // n0, g0 wave-uniform
    const uint4 *bl = Bh + (size_t)(2 * gb0 + hk) * ld + lm;
    uint4 ring[R][NT];
#pragma unroll
    for (int g = 0; g < R && g < CNT; g++)
#pragma unroll
        for (int t = 0; t < NT; t++) ring[g][t] = __builtin_bit_cast(uint4, ft_wload(w, s0 + (unsigned)g * w.row2, 512 * t));
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        uint4 b8[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) b8[mt] = bl[(size_t)(2 * g) * ld + 32 * mt];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                acc[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % R][t]), __builtin_bit_cast(bf16x8, b8[mt]),
                                                                     acc[t][mt], 0, 0, 0);
        if (g + R < CNT) {
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = __builtin_bit_cast(uint4, ft_wload(w, s0 + (unsigned)(g + R) * w.row2, 512 * t));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// 32(n) x 32(m) fp32 accumulator tile -> bf16, stored as 8-byte halves of q8 elements of an LDS operand image (row stride ld)
__device__ __forceinline__ void fth_store_tile(uint4 *img, int ld, const f32x16 &acc, int n_local, int m_local, int hk, int lm) {
    uint2 *h = reinterpret_cast<uint2 *>(img);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint2 v;
        v.x = pack_bf16x2(acc[4 * i + 0], acc[4 * i + 1]);
        v.y = pack_bf16x2(acc[4 * i + 2], acc[4 * i + 3]);
        h[((size_t)((n_local >> 3) + i) * ld + m_local + lm) * 2 + hk] = v;  // n = n_local + 8i + 4hk + c
    }
}

template <int MT, int T2W, int KG, bool DB, int RD, int PD>
__global__ void __launch_bounds__(512) fr_fused_tile_h_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int LD = 32 * MT + 1, TI = 32 * MT;
    const int KO = a.K / 8;
    uint4 *Xh = lds;                                                   // [K/8][LD]
    uint4 *R1b[2] = {Xh + (size_t)KO * LD, Xh + (size_t)(KO + (DB ? 32 : 0)) * LD};  // [32][LD]: 256 outputs of FC1 (bf16)
    uint4 *R2 = lds;                                                   // [H2/8][LD], overlays Xh / R1 once they are dead
    uint4 *R3 = lds + (size_t)(a.H2 / 8) * LD;                         // [H3/8][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * TI;
    if (m0 >= bt.batch) return;
    int n_st = 0;
    auto stamp = [&]() {  // diagnostic build aid (tools/experiments/fused_h_stamps.py); lane 0 of every wave, values never feed an output
        if (a.stamps && lane == 0 && n_st < 14) {
            a.stamps[16ull * (8ull * blockIdx.x + wave) + n_st] = __builtin_amdgcn_s_memrealtime();
            if (n_st == 1) a.stamps[16ull * (8ull * blockIdx.x + wave) + 14] = __builtin_amdgcn_s_memtime();   // shader-clock cycles at "gather done"
            if (n_st == 9) a.stamps[16ull * (8ull * blockIdx.x + wave) + 15] = __builtin_amdgcn_s_memtime();   // ... and at the end of the last chunk's FC2
        }
        n_st++;
    };
    stamp();

    {   // ---- gather + bf16 conversion: lanes along record words, TI / 8 items per thread ----
        const int wl = tid & 63, ig = tid >> 6;
        constexpr int IPT = TI / 8, NB = (4 * KG + 63) / 64;   // K = 16 KG floats = 4 KG record words
        uint2 *Xh2 = reinterpret_cast<uint2 *>(Xh);
        const bool bad = ft_gather_tile<NB, IPT>(a, bt, m0, wl, ig, [&](int w, int i, const uint4 &x) {
            uint2 hv;
            hv.x = pack_bf16x2(__uint_as_float(x.x), __uint_as_float(x.y));
            hv.y = pack_bf16x2(__uint_as_float(x.z), __uint_as_float(x.w));
            Xh2[((size_t)(w >> 1) * LD + IPT * ig + i) * 2 + (w & 1)] = hv;  // record word w = half (w & 1) of q8 element w / 2
        });
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();
    stamp();

    f32x16 acc2[T2W][MT];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    // q8 weights: (K / 8) rows of N 16-byte elements -> ft_w's K / 4 rows of the q4 layout is K / 2 here
    const FtW W1 = ft_w(a.w1q, a.K / 2, a.H1, hk, lm), W2 = ft_w(a.w2q, a.H1 / 2, a.H2, hk, lm), W3 = ft_w(a.w3q, a.H2 / 2, a.H3, hk, lm);
    // ONE weight ring of RD 16-byte register fragments feeds FC1, FC2 and FC3 as a continuous stream: a slot freed by the last RD
    // groups of FC1 is refilled with the first elements of this chunk's FC2 weights, a slot freed by the last RD elements of FC2 with the
    // first groups of the next chunk's FC1 weights (or of W3 after the last chunk).  With a ring of its own every GEMM phase used to start
    // cold -- all 8 waves waiting for an L2 round trip at once: FC2's 16 groups took 5.4 us for 1.95 us of MFMA work, FC3 6.0 us
    // (tools/experiments/fused_h_stamps.py).  FC1 group g lives in slot g % RD, W3 group g likewise, FC2 element e = 2 j + t (group j,
    // n tile t) in slot (e + OFF2) % RD with OFF2 = KG % RD -- the slot FC1's tail frees first.  Every index is a compile-time
    // constant (the bodies are unrolled), so the ring stays in registers and hipcc counts the vmcnt waits.  The order of the sums does
    // not depend on RD: scores are bit-identical for every depth.  Depths of 8, 12 and 16 take the same time (the weight stream is not
    // latency-bound, profiles/archive/r02_experiments.md section 5.4); 12 leaves the registers for the B-fragment ring below.
    static_assert(T2W == 2 && KG >= RD && RD >= 8 && RD <= 32 && RD % 2 == 0, "the stream ring assumes two n tiles per wave in FC2 (32 elements per chunk) and at least RD k groups in FC1");
    constexpr int OFF2 = KG % RD;
    uint4 ring[RD];
    auto w1load = [&](int c, int g) { return __builtin_bit_cast(uint4, ft_wload(W1, (unsigned)g * W1.row2 + (unsigned)(c * 256 + 32 * wave) * 16u, 0)); };
    auto w2load = [&](int c, int e) {
        return __builtin_bit_cast(uint4, ft_wload(W2, (unsigned)(16 * c + (e >> 1)) * W2.row2 + (unsigned)(32 * T2W * wave) * 16u, 512 * (e & 1)));
    };
#pragma unroll
    for (int g = 0; g < RD; g++) ring[g] = w1load(0, g);
    const uint4 *blx = Xh + (size_t)hk * LD + lm;
    // B fragments (LDS reads) run PD groups ahead of the MFMAs that use them, in a ring of PD register slots refilled like the weight
    // ring (group g lives in slot g % PD).  Read in place, a wave stalled ~90 cycles on the read before every pair of MFMAs (64
    // cycles): per-wave stamps showed 154 cycles per FC1 group and the matrix pipe half idle in the FC phases.  The record image is
    // static, so FC2's tail already fetches the first PD fragments of the next chunk's FC1 (16 % PD == 0 keeps the slots aligned).
    static_assert(PD == 1 || PD == 2 || PD == 4, "B ring: 16 % PD must be 0");
    uint4 bq[PD][MT];
#pragma unroll
    for (int i = 0; i < PD; i++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) bq[i][mt] = blx[(size_t)(2 * i) * LD + 32 * mt];
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc1[0][mt][i] = 0.0f;
#pragma unroll
        for (int g = 0; g < KG; g++) {  // FC1, chunk c: K / 16 groups
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                acc1[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % RD]), __builtin_bit_cast(bf16x8, bq[g % PD][mt]), acc1[0][mt], 0, 0, 0);
            if (g + PD < KG) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) bq[g % PD][mt] = blx[(size_t)(2 * (g + PD)) * LD + 32 * mt];
            }
            if (g + RD < KG) ring[g % RD] = w1load(c, g + RD);
            else ring[g % RD] = w2load(c, (g % RD - OFF2 + RD) % RD);  // FC1's tail: the first RD elements of this chunk's FC2 weights
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
        uint4 *R1 = R1b[DB ? (c & 1) : 0];
        if (!DB && c > 0) __syncthreads();  // single R1 buffer: every wave must be done reading the previous chunk
#pragma unroll
        for (int mt = 0; mt < MT; mt++) fth_store_tile(R1, LD, acc1[0][mt], 32 * wave, 32 * mt, hk, lm);
        __syncthreads();
        // FC2: K range [256 c, 256 c + 256) = 16 groups of 16 k; its tail requests the next stream: W1 chunk c + 1, or W3 after the last chunk
        const bool last = c + 1 == n_chunks;
        FtW WN = W1;
        if (last) WN = W3;
        const unsigned nsoff = last ? (unsigned)(32 * wave) * 16u : (unsigned)((c + 1) * 256 + 32 * wave) * 16u;
        const uint4 *blr = R1 + (size_t)hk * LD + lm;
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) bq[i][mt] = blr[(size_t)(2 * i) * LD + 32 * mt];
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
            for (int t = 0; t < T2W; t++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc2[t][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[(2 * j + t + OFF2) % RD]), __builtin_bit_cast(bf16x8, bq[j % PD][mt]),
                                                                         acc2[t][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)  // refill: FC2 group j + PD, or (tail) group j + PD - 16 of the next chunk's FC1 -- the record image is static
                bq[j % PD][mt] = j + PD < 16 ? blr[(size_t)(2 * (j + PD)) * LD + 32 * mt] : blx[(size_t)(2 * (j + PD - 16)) * LD + 32 * mt];
#pragma unroll
            for (int t = 0; t < T2W; t++) {
                const int r = (2 * j + t + OFF2) % RD;
                if (2 * j + t + RD < 32) ring[r] = w2load(c, 2 * j + t + RD);
                else ring[r] = __builtin_bit_cast(uint4, ft_wload(WN, (unsigned)r * WN.row2 + nsoff, 0));  // group r of the next stream lives in slot r
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
    }
    __syncthreads();  // Xh and R1 dead: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) fth_store_tile(R2, LD, acc2[t][mt], 32 * (T2W * wave + t), 32 * mt, hk, lm);
    __syncthreads();

    f32x16 acc3[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc3[0][mt][i] = 0.0f;
    {   // FC3: H2 / 16 = 32 groups; groups 0..15 are already in (or on their way into) the ring
        const uint4 *bl3 = R2 + (size_t)hk * LD + lm;
#pragma unroll
        for (int i = 0; i < PD; i++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) bq[i][mt] = bl3[(size_t)(2 * i) * LD + 32 * mt];
#pragma unroll
        for (int g = 0; g < 16 * T2W; g++) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                acc3[0][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[g % RD]), __builtin_bit_cast(bf16x8, bq[g % PD][mt]), acc3[0][mt], 0, 0, 0);
            if (g + PD < 16 * T2W) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) bq[g % PD][mt] = bl3[(size_t)(2 * (g + PD)) * LD + 32 * mt];
            }
            if (g + RD < 16 * T2W) ring[g % RD] = __builtin_bit_cast(uint4, ft_wload(W3, (unsigned)(g + RD) * W3.row2 + (unsigned)(32 * wave) * 16u, 0));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) fth_store_tile(R3, LD, acc3[0][mt], 32 * wave, 32 * mt, hk, lm);
    __syncthreads();
    stamp();
    {   // score[m] = sum_n wout[n] * R3[n][m] (bf16 x bf16, fp32 sum): TI items x (512 / TI) slices of q8 rows, fixed-order reduction
        const int il = tid % TI, sl = tid / TI;
        constexpr int NSL = 512 / TI;
        const int rows_per = (a.H3 / 8) / NSL;
        const uint4 *wh = reinterpret_cast<const uint4 *>(a.wout);  // bf16 vector w[k], 8 per element
        float s = 0.0f;
        for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
            const uint4 r = R3[(size_t)q * LD + il];
            const uint4 w = wh[q];
            const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                s = fmaf(__uint_as_float(ww[e] << 16), __uint_as_float(rr[e] << 16), s);
                s = fmaf(__uint_as_float(ww[e] & 0xFFFF0000u), __uint_as_float(rr[e] & 0xFFFF0000u), s);
            }
        }
        float *part = reinterpret_cast<float *>(R3 + (size_t)(a.H3 / 8) * LD);
        part[sl * TI + il] = s;
        __syncthreads();
        if (tid < TI && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < NSL; i++) t += part[i * TI + tid];
            bt.scores[m0 + tid] = t;
        }
    }
}

static size_t fused_h_lds_bytes(int K, int H2, int H3, int MT, bool db) {
    const size_t LD = 32 * MT + 1;
    const size_t phase1 = (size_t)(K / 8) + (db ? 64 : 32);
    const size_t phase2 = (size_t)(H2 / 8) + (size_t)(H3 / 8) + 2;  // + 2 rows: 512 floats of reduction scratch
    return (phase1 > phase2 ? phase1 : phase2) * LD * 16;
}

// bf16 fused kernel: straight-line instantiations exist for K = 352 (Model-A) and K = 880 (Model-B), 64 items per workgroup
bool frk_fused_h_ok(int K, int H1, int H2, int H3) {
    if (H1 % 256 || H2 != 512 || H3 != 256) return false;
    return K == 352 || K == 880;
}

int frk_fused_h_items_per_wg() { return 64; }

template <int MT, int T2W, int KG, bool DB, int RD, int PD>
static int fused_h_launch_inst(const FrFusedArgs &a, dim3 grid, size_t lds, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_h_kernel<MT, T2W, KG, DB, RD, PD>, lds_once)) return rc_;
    fr_fused_tile_h_kernel<MT, T2W, KG, DB, RD, PD><<<grid, dim3(512), lds, s>>>(a);
    fr_note_kernel("fr_fused_tile_h_kernel<%d, %d, %d, %s, %d, %d>", MT, T2W, KG, DB ? "true" : "false", RD, PD);
    KCHECK();
    return FR_OK;
}

// a.w1q/w2q/w3q/wout must point at the bf16 q8 weights; a.tiles_per_batch counts 64-item tiles
int frk_fused_h_launch(const FrFusedArgs &a, hipStream_t s) {
    dim3 grid(a.n_batches * a.tiles_per_batch);
    // weight ring 16 (Model-A: 22 k groups per chunk) / 12 (Model-B) fragments, B fragments two groups ahead.  Ring depths 8 / 12 / 16 and
    // B-ring depths 1 / 2 / 4 take the same time on the chip and return the same bits (profiles/archive/r02_experiments.md section 5.4).
    if (a.K == 352) return fused_h_launch_inst<2, 2, 22, true, 16, 2>(a, grid, fused_h_lds_bytes(a.K, a.H2, a.H3, 2, true), s);
    if (a.K == 880) return fused_h_launch_inst<2, 2, 55, false, 12, 2>(a, grid, fused_h_lds_bytes(a.K, a.H2, a.H3, 2, false), s);
    FR_FAIL(FR_ERR_INVALID, "no bf16 fused instantiation for K=%d", a.K);
}

// ===================================================================================================
// fr_fused_tile_f8_kernel: the fused item-tile kernel in fp8 (e4m3 q16 operands, v_mfma_scale_f32_32x32x64_f8f6f4).
// Same phases as the bf16 kernel, 64 items per workgroup, every activation image in LDS as e4m3 bytes (x 2^e_act[l], saturated),
// the MFMA's E8M0 block scales undo the weight and activation exponents, the output layer multiplies the decoded R3 by the fp32
// master weights.  A group is 64 k: lane half h supplies q16 rows 4g + 2h and 4g + 2h + 1 of both operands.  The record is
// zero-padded to a multiple of 64 k inside the LDS image (Model-A 352 -> 384, Model-B 880 -> 896).
// ===================================================================================================
struct FtW8 {
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff, voff2;  // rows 2h and 2h + 1 of a group
    unsigned grp;          // byte step of one group (4 rows)
};
__device__ __forceinline__ FtW8 ft_w8(const void *wf, int KE, int N, int hk, int lm) {
    FtW8 w;
    w.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wf), 0, (unsigned)KE * (unsigned)N * 16u, 0x00020000);
    w.voff = ((unsigned)(2 * hk) * N + lm) * 16u;
    w.voff2 = w.voff + (unsigned)N * 16u;
    w.grp = 4u * (unsigned)N * 16u;
    return w;
}
__device__ __forceinline__ i32x8 ft_w8load(const FtW8 &w, unsigned soff, int imm) {
    const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff + imm, soff, 0), hi = __builtin_amdgcn_raw_buffer_load_b128(w.rs, w.voff2 + imm, soff, 0);
    i32x8 r;
    r[0] = (int)lo.x; r[1] = (int)lo.y; r[2] = (int)lo.z; r[3] = (int)lo.w;
    r[4] = (int)hi.x; r[5] = (int)hi.y; r[6] = (int)hi.z; r[7] = (int)hi.w;
    return r;
}

template <int NT, int MT, int R, int CNT>
__device__ __forceinline__ void ft8_gemm_ct(f32x16 (&acc)[NT][MT], const FtW8 &w, int n0, const uint4 *Bf, int ld, int gb0, int g0, int hk, int lm, int sc_a,
                                            int sc_b) {
    const unsigned s0 = (unsigned)g0 * w.grp + (unsigned)n0 * 16u;  // n0, g0 wave-uniform
    const uint4 *bl = Bf + (size_t)(4 * gb0 + 2 * hk) * ld + lm;
    i32x8 ring[R][NT];
#pragma unroll
    for (int g = 0; g < R && g < CNT; g++)
#pragma unroll
        for (int t = 0; t < NT; t++) ring[g][t] = ft_w8load(w, s0 + (unsigned)g * w.grp, 512 * t);
#pragma unroll
    for (int g = 0; g < CNT; g++) {
        i32x8 b8[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const uint4 lo = bl[(size_t)(4 * g) * ld + 32 * mt], hi = bl[(size_t)(4 * g + 1) * ld + 32 * mt];
            b8[mt][0] = (int)lo.x; b8[mt][1] = (int)lo.y; b8[mt][2] = (int)lo.z; b8[mt][3] = (int)lo.w;
            b8[mt][4] = (int)hi.x; b8[mt][5] = (int)hi.y; b8[mt][6] = (int)hi.z; b8[mt][7] = (int)hi.w;
        }
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++) acc[t][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ring[g % R][t], b8[mt], acc[t][mt], 0, 0, 0, sc_a, 0, sc_b);
        if (g + R < CNT) {
#pragma unroll
            for (int t = 0; t < NT; t++) ring[g % R][t] = ft_w8load(w, s0 + (unsigned)(g + R) * w.grp, 512 * t);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// 32(n) x 32(m) fp32 accumulator tile -> e4m3 (x oscale, saturated), stored as 4-byte quarters of q16 elements of an LDS image
__device__ __forceinline__ void ft8_store_tile(uint4 *img, int ld, const f32x16 &acc, int n_local, int m_local, int hk, int lm, float oscale) {
    uint32_t *q = reinterpret_cast<uint32_t *>(img);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n = n_local + 8 * i + 4 * hk;  // + c
        q[((size_t)(n >> 4) * ld + m_local + lm) * 4 + ((n & 15) >> 2)] = pack_fp8x4(acc[4 * i + 0], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3], oscale);
    }
}

template <int G1 /* FC1 groups of 64 k */>
__global__ void __launch_bounds__(512) fr_fused_tile_f8_kernel(const FrFusedArgs a) {
    extern __shared__ uint4 lds[];
    constexpr int MT = 2, T2W = 2, LD = 32 * MT + 1, TI = 32 * MT;
    constexpr int KE1 = 4 * G1;                       // q16 rows of the (padded) record
    uint4 *Xf = lds;                                  // [KE1][LD]
    uint4 *R1b[2] = {Xf + (size_t)KE1 * LD, Xf + (size_t)(KE1 + 16) * LD};  // [16][LD] each: 256 outputs of FC1
    uint4 *R2 = lds;                                  // [H2/16][LD], overlays Xf / R1 once they are dead
    uint4 *R3 = lds + (size_t)(a.H2 / 16) * LD;       // [H3/16][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hk = lane >> 5, lm = lane & 31;
    const int bi = blockIdx.x / a.tiles_per_batch, tile = blockIdx.x - bi * a.tiles_per_batch;
    const FrFusedBatch &bt = a.b[bi];
    const int m0 = tile * TI;
    if (m0 >= bt.batch) return;

    {   // ---- gather + e4m3 conversion: lanes along record words (incl. the zero pad), 8 items per thread ----
        const int wl = tid & 63, ig = tid >> 6;
        uint32_t *Xw = reinterpret_cast<uint32_t *>(Xf);
        const float scale = __builtin_ldexpf(1.0f, a.e_act[0]);
        constexpr int NB = (4 * KE1 + 63) / 64;
        const bool bad = ft_gather_tile<NB, 8>(a, bt, m0, wl, ig, [&](int w, int i, const uint4 &x) {
            Xw[((size_t)(w >> 2) * LD + 8 * ig + i) * 4 + (w & 3)] = pack_fp8_word(x, scale);   // a zero row (item past the batch) packs to 0
        });
        for (int w = a.n_words + tid; w < 4 * KE1; w += 512)   // pad up to a multiple of 64 k
#pragma unroll 1
            for (int i = 0; i < TI; i++) Xw[((size_t)(w >> 2) * LD + i) * 4 + (w & 3)] = 0u;
        if (bad) atomicOr_system(a.err_flag, 1);
    }
    __syncthreads();

    f32x16 acc2[T2W][MT];
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc2[t][mt][i] = 0.0f;
    const int n_chunks = a.H1 / 256;
    const FtW8 W1 = ft_w8(a.w1q, KE1, a.H1, hk, lm), W2 = ft_w8(a.w2q, a.H1 / 16, a.H2, hk, lm), W3 = ft_w8(a.w3q, a.H2 / 16, a.H3, hk, lm);
    const float os1 = __builtin_ldexpf(1.0f, a.e_act[1]), os2 = __builtin_ldexpf(1.0f, a.e_act[2]), os3 = __builtin_ldexpf(1.0f, a.e_act[3]);
    for (int c = 0; c < n_chunks; c++) {
        f32x16 acc1[1][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc1[0][mt][i] = 0.0f;
        ft8_gemm_ct<1, MT, (G1 < 8 ? G1 : 8), G1>(acc1, W1, c * 256 + 32 * wave, Xf, LD, 0, 0, hk, lm, 127 - a.e_w[0], 127 - a.e_act[0]);
        uint4 *R1 = R1b[c & 1];  // double-buffered: one barrier per chunk
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ft8_store_tile(R1, LD, acc1[0][mt], 32 * wave, 32 * mt, hk, lm, os1);
        __syncthreads();
        // FC2: K range [256 c, 256 c + 256) = 4 groups of 64 k
        ft8_gemm_ct<T2W, MT, 4, 4>(acc2, W2, 32 * T2W * wave, R1, LD, 0, 4 * c, hk, lm, 127 - a.e_w[1], 127 - a.e_act[1]);
    }
    __syncthreads();  // Xf and R1 dead: R2 may overlay them
#pragma unroll
    for (int t = 0; t < T2W; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) ft8_store_tile(R2, LD, acc2[t][mt], 32 * (T2W * wave + t), 32 * mt, hk, lm, os2);
    __syncthreads();

    f32x16 acc3[1][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc3[0][mt][i] = 0.0f;
    ft8_gemm_ct<1, MT, 8, 8>(acc3, W3, 32 * wave, R2, LD, 0, 0, hk, lm, 127 - a.e_w[2], 127 - a.e_act[2]);  // H2 / 64 = 8 groups
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ft8_store_tile(R3, LD, acc3[0][mt], 32 * wave, 32 * mt, hk, lm, os3);
    __syncthreads();
    {   // score[m] = 2^-e3 * sum_n wout[n] * e4m3(R3)[n][m]: 64 items x 8 slices of 2 q16 rows, fp32 master weights, fixed order
        const int il = tid & 63, sl = tid >> 6;
        const int rows_per = (a.H3 / 16) / 8;
        float s = 0.0f;
        for (int q = sl * rows_per; q < (sl + 1) * rows_per; q++) {
            const uint4 r = R3[(size_t)q * LD + il];
            const int rr[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
            const float *w = a.wout + 16 * q;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                s = fmaf(w[4 * c + 0], __builtin_amdgcn_cvt_f32_fp8(rr[c], 0), s);
                s = fmaf(w[4 * c + 1], __builtin_amdgcn_cvt_f32_fp8(rr[c], 1), s);
                s = fmaf(w[4 * c + 2], __builtin_amdgcn_cvt_f32_fp8(rr[c], 2), s);
                s = fmaf(w[4 * c + 3], __builtin_amdgcn_cvt_f32_fp8(rr[c], 3), s);
            }
        }
        float *part = reinterpret_cast<float *>(R3 + (size_t)(a.H3 / 16) * LD);
        part[sl * 64 + il] = s;
        __syncthreads();
        if (tid < 64 && m0 + tid < bt.batch) {
            float t = part[tid];
#pragma unroll
            for (int i = 1; i < 8; i++) t += part[i * 64 + tid];
            bt.scores[m0 + tid] = t * __builtin_ldexpf(1.0f, -a.e_act[3]);
        }
    }
}

bool frk_fused_f8_ok(int K, int H1, int H2, int H3) { return (K == 352 || K == 880) && H1 % 256 == 0 && H2 == 512 && H3 == 256; }

template <int G1>
static int fused_f8_launch_inst(const FrFusedArgs &a, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_f8_kernel<G1>, lds_once)) return rc_;
    const size_t rows1 = 4 * G1 + 32, rows2 = (size_t)(a.H2 / 16) + (size_t)(a.H3 / 16) + 2;  // + 2 rows: 512 floats of reduction scratch
    fr_fused_tile_f8_kernel<G1><<<dim3(a.n_batches * a.tiles_per_batch), dim3(512), (rows1 > rows2 ? rows1 : rows2) * 65 * 16, s>>>(a);
    fr_note_kernel("fr_fused_tile_f8_kernel<%d>", G1);
    KCHECK();
    return FR_OK;
}

// a.w1q/w2q/w3q point at the e4m3 q16 weights, a.wout at the fp32 output weights; a.tiles_per_batch counts 64-item tiles
int frk_fused_f8_launch(const FrFusedArgs &a, hipStream_t s) {
    if (a.K == 352) return fused_f8_launch_inst<6>(a, s);
    if (a.K == 880) return fused_f8_launch_inst<14>(a, s);
    FR_FAIL(FR_ERR_INVALID, "no fp8 fused instantiation for K=%d", a.K);
}
