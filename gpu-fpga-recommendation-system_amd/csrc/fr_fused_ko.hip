#include "fr_device.h"

// ===================================================================================================
// fr_fused_tile_hs_kernel: the bf16 fused item-tile kernel -- K-OUTER, PERSISTENT and WAVE-SPECIALISED (BASELINE configs[2]: Model-B
// 1024, "bf16 MFMA FC, fused concat + first FC").  It replaces fr_fused_tile_h_kernel (fr_fused.hip), whose workgroup first gathered
// its 64 records with the matrix pipes idle (21-23 of 63-68 us per launch) and then walked FC1 in four chunks of 256 outputs,
// re-reading the whole record image from LDS per chunk.
//
// A workgroup is 12 waves: 8 CONSUMERS (MFMA) and 4 PRODUCERS (gather), three per SIMD (two consumers + one producer), 168 registers
// each.  Why separate waves: a wave's vector-memory operations return IN ORDER (one vmcnt), so a wave that streams weights from L2
// through a register ring AND has HBM-latency row loads outstanding waits for the rows every time it waits for a weight fragment --
// the 8-symmetric-wave form of this kernel (round 3, profiles/archive/r03_fused_hk_8wave_stamps.txt) spent 31-36 us in an FC1 whose MFMA
// work is 11.7 us.  The producers have nothing but the gather in their queue, and their registers ARE the in-flight window: 4 row
// sets x 8 items x 16 B per thread = 128 KiB of rows on their way per CU, all the time, through every phase of the consumers.
//
// Consumers: FC1 runs over K ONCE with all 1024 outputs of the 64 items in accumulators (128 outputs x 64 items = 128 registers per
// wave), so the record is consumed in K order -- the order the gather produces it in.  The record image never exists as a whole: a
// ring of TWO slices of KGS k-groups sits in LDS; while the consumers multiply slice s, the producers convert the rows of slice s + 1
// to bf16 into the other buffer (one s_barrier per slice for all 12 waves).  The freed LDS holds the complete bf16 R1 image
// (1024 x 64 = 128 KiB), so FC2 is one K-outer pass as well (64 outputs x 64 items per wave): no chunk loop.  The producers keep
// running through FC2 / FC3 / the output layer for the NEXT tile (persistent: a workgroup walks tiles b, b + grid, ...): a tile
// starts with its first two slices in LDS and D more in registers.
//   LDS  [ R1 128 KiB | X ring 2 x 2 KGS rows x 65 x 16 B | packed word descriptors 16 B each ]   (R2 / R3 / scratch overlay R1)
//   per k-group and consumer: FC1 4 weight fragments + 2 B fragments -> 8 MFMAs (the chunked kernel: 1 + 2 -> 2: a quarter of its LDS reads)
// Arithmetic per output: fp32 accumulation over k in ascending order, one bf16 rounding per activation -- the same values as the
// chunked kernel, bit for bit (the order of the sums inside an output is unchanged; only the order of the outputs moved).
// ===================================================================================================
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
// the per-lane part again, from a freshly made lane number (the K-outer kernel re-derives it per phase instead of holding three of them)
__device__ __forceinline__ void ftk_lane(FtWk &w, int hk, int lm) { w.voff = (unsigned)hk * (w.row2 >> 1) + (unsigned)lm * 16u; }
// FR_HS_W_AUX (hand-built experiment variants only, tools/jobs/r03_waux.sh): cache-policy bits of the consumers' weight loads
#ifndef FR_HS_W_AUX
#define FR_HS_W_AUX 0
#endif
__device__ __forceinline__ uint4 ftk_load(const FtWk &w, unsigned soff, int imm) {
    // The per-lane offset is made opaque at every load: seen through, hipcc knows its low bits (voff + 512 t == voff | 512 t), keeps one
    // VGPR per n tile for the whole kernel, runs out of registers in FC1, spills exactly those to scratch and reloads each of them -- behind
    // an s_waitcnt vmcnt(0) that drains the weight ring -- in front of the load that needs it.  Opaque, the tile offset rides the
    // instruction's immediate field (buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen offset:512 t).
    unsigned vo = w.voff;
    asm volatile("" : "+v"(vo));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rs, vo + imm, soff, FR_HS_W_AUX);
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

// fp8 form of the tile store: 32(n) x 32(m) fp32 accumulators -> e4m3 (x oscale, saturated) in the "q16h" operand layout of the non-scaled
// fp8 MFMA: element (row 2 (n / 32) + ((n % 16) / 8), m), byte 8 ((n % 32) / 16) + n % 8.  A lane's four consecutive n are one dword.
__device__ __forceinline__ void hk_store_tile_f8(uint4 *img, const f32x16 &acc, int n_local, int m_local, int hk, int lm, float oscale) {
    uint32_t *q = reinterpret_cast<uint32_t *>(img);
#pragma unroll
    for (int i = 0; i < 4; i++)   // n = n_local + 8 i + 4 hk + c, n_local a multiple of 32: step i / 2, lane half i % 2, position 4 hk + c
        q[((size_t)(2 * (n_local >> 5) + (i & 1)) * HK_LD + m_local + lm) * 4 + 2 * (i >> 1) + hk] = pack_fp8x4(acc[4 * i + 0], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3], oscale);
}

// one "k-group" = the k one 16-byte operand fragment per lane covers: 16 (bf16: one v_mfma_f32_32x32x16_bf16) or 32 (fp8: two
// v_mfma_f32_32x32x16_fp8_fp8, the fragment's low and high 8 bytes).  acc += A(fragment) x B(fragment), k ascending.
template <int PREC>
__device__ __forceinline__ void hk_mma(f32x16 &acc, const uint4 &a4, const uint4 &b4) {
    if constexpr (PREC == 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc, 0, 0, 0);
    } else {
        const long a_lo = (long)(((unsigned long long)a4.y << 32) | a4.x), a_hi = (long)(((unsigned long long)a4.w << 32) | a4.z);
        const long b_lo = (long)(((unsigned long long)b4.y << 32) | b4.x), b_hi = (long)(((unsigned long long)b4.w << 32) | b4.z);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_lo, b_lo, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a_hi, b_hi, acc, 0, 0, 0);
    }
}

// PREC = 1 bf16 (q8 operands), 2 fp8 (e4m3 "q16h" operands on the NON-scaled fp8 MFMA: operand registers as in bf16 -- the scaled
// 32x32x64 form needs 8 registers per fragment, which the consumers' 168 do not have beside 128 accumulators; the power-of-two
// exponents are folded into the scale of the next activation's quantisation, exactly).  KG = k-groups of the (zero-padded) record (16 k
// each in bf16, 32 k in fp8); KGS = k-groups per slice; LW = producer lanes along the record words of a slice (16 or 32); D = row sets
// a producer thread keeps in flight (NSL % D == 0: the set of a slice is a compile-time index); R1D = FC1 weight fragments in registers.
// fp8 (round 4): the e4m3 R1 image is 64 KiB, so the LDS the bf16 form spends on R1 holds the WHOLE record image of a tile here (NSL
// slice buffers instead of a ring of two), and the producers gather tile t + 1 while the consumers are anywhere in tile t: E2 of a tile's
// NSL + D + 1 gather events run under FC2, E3 under FC3, the rest under FC1's first slices (one per slice barrier) -- with the bf16
// schedule (every event under FC1) the fp8 consumers, whose operand stream is half as long, waited for the gather in every slice
// (340 M inf/s against the chunked kernel's 398 M, profiles/archive/r03_fused_hs_fp8_ab.txt).
// SRC = 1 (bf16, round 6): a.words address the OPERAND-TYPE row image (fr_ctx::lp_arena: every table / bank row once more as bf16, made with
// the rounding W_op applies) -- a row word is 8 bytes, a row set costs half the registers, so D = 4 sets ride where 2 did; W_op stores the
// words as they come.  Dense words (the request's fp32 features) are converted by the lane that loads them.  Same X image, bit for bit.
template <int PREC, int KG, int KGS, int LW, int D, int R1D, int E2 = 0, int E3 = 0, int SRC = 0>
__global__ void __launch_bounds__(768) fr_fused_tile_hs_kernel(const FrFusedArgs a) {
    static_assert(SRC == 0 || PREC == 1, "operand-type rows: the bf16 form only");
    using row_t = typename std::conditional<SRC == 1, uint2, uint4>::type;
    extern __shared__ uint4 lds[];
    constexpr int WPG = PREC == 2 ? 8 : 4;             // record words (4 floats each) per k-group
    constexpr int KPG = 4 * WPG;                       // k per k-group
    constexpr int KG2 = HK_H1 / KPG, KG3 = HK_H2 / KPG, KG4 = HK_H3 / KPG;   // k-groups of FC2 / FC3 / the output layer
    constexpr int NSL = (KG + KGS - 1) / KGS;          // slices per tile
    constexpr int IPT = LW / 4;                        // items per producer thread (256 threads = LW words x 64 / IPT item slots)
    constexpr int XROWS = 2 * KGS;                     // operand rows of one X ring buffer
    constexpr int NBAR = NSL + 5;                      // barriers per tile (stamp slots)
    constexpr bool FULLX = PREC == 2;                  // the X image of a whole tile in LDS (see above); bf16: a ring of two slices
    constexpr int NXB = FULLX ? NSL : 2;               // X slice buffers
    constexpr int NEV = NSL + D + 1;                   // FULLX: gather events per tile: event e = { W(e - 1 - D), R(e - 1), I(e) }
    constexpr int NE1 = NEV - E2 - E3;                 // ... of which under FC1's slice barriers 0 .. NE1 - 1
    static_assert(FULLX || (NSL % 2 == 0 && NSL % D == 0 && NSL >= D + 3), "two X buffers, D row sets, and the run-ahead stays inside the next tile");
    static_assert(!FULLX || (NE1 >= 1 && NE1 <= NSL && E2 >= 0 && E3 >= 0), "FC1 carries at most one gather event per slice barrier");
    static_assert(WPG * KGS <= LW && (LW == 16 || LW == 32), "a slice's record words ride the lanes of one half / quarter wave");
    uint4 *R1 = lds;                                   // [2 KG2][64]: bf16 128 KiB, fp8 64 KiB
    uint4 *R2 = lds;                                   // [2 KG3][64], overlays R1 once FC2 has read it
    uint4 *R3 = lds + 2 * KG3 * HK_LD;                 // [2 KG4][64]
    float *part = reinterpret_cast<float *>(lds + (2 * KG3 + 2 * KG4) * HK_LD);  // 8 x 64 partial scores
    uint4 *Xr = lds + 2 * KG2 * HK_LD;                 // [NXB][XROWS][65]
    uint4 *Dsc = Xr + NXB * XROWS * HK_LDX;            // [n_words] packed descriptors
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 8;                   // waves 8..11: one per SIMD beside two consumers (waves are dealt to SIMDs cyclically)
    const int n_tiles = a.n_batches * a.tiles_per_batch;

    const FrFusedBatch *bl = a.blist;   // the launch's batches: device memory (up to FR_FUSED_MAX_QUEUE of them), read with scalar loads
    auto tile_at = [&](int t) {
        const int bi = t / a.tiles_per_batch;
        HkTile r;
        r.idx = bl[bi].idx, r.dense = bl[bi].dense, r.scores = bl[bi].scores, r.batch = bl[bi].batch;
        r.m0 = (t - bi * a.tiles_per_batch) * 64;
        return r;
    };
    auto next_tile = [&](int t) {  // the workgroup's next non-empty tile after t (n_tiles: none); wave-uniform
        for (t += gridDim.x; t < n_tiles; t += gridDim.x) {
            const int bi = t / a.tiles_per_batch;
            if ((t - bi * a.tiles_per_batch) * 64 < bl[bi].batch) break;
        }
        return t;
    };
    int t_cur = next_tile((int)blockIdx.x - (int)gridDim.x);
    if (t_cur >= n_tiles) return;

    // Diagnostic build aid (tools/experiments/fused_hk_stamps.py; a.stamps is NULL in normal operation, values never feed an output): 128
    // s_memrealtime slots per wave.  0 start, 1 set-up done, 2 / 3 s_memtime around tile 0's FC1; the barriers of the workgroup's first two
    // tiles: 4 + 2 NBAR tile + 2 b = arrival at barrier b, + 1 = release (b = slice s for the step barriers, NSL + p for the five phase
    // barriers R1 stored / FC2 done / R2 stored / R3 stored / partial scores; NBAR = NSL + 5); 126 kernel end.
    // The stamps (and the producers' timing ablations below) are compiled into the DIAGNOSTIC build only (`make -C csrc diag` ->
    // libfleetrec_diag.so = the experiments build + -DFR_HS_DIAG on this file): even on the producer side alone they cost 30 more spilled
    // registers and 17 % of the kernel's speed (Model-B 1024: 272 vs 225 us per launch), so an A/B taken in a build that has them says
    // little about the product kernel.  Previously they were part of every EXPERIMENTS build: in the product kernel their pointer and tile counter cost registers the
    // consumers do not have (168, 128 of them accumulators) -- with them in, hipcc spilled 11-18 registers to scratch, and a kernel that
    // uses scratch pays for its set-up at every dispatch.
#ifdef FR_HS_DIAG
    constexpr bool kStamps = true;
#else
    constexpr bool kStamps = false;
#endif
    unsigned long long *st = (kStamps && a.stamps) ? a.stamps + 128ull * (12ull * blockIdx.x + wave) : nullptr;
    auto stamp = [&](int k) {
        if constexpr (kStamps)
            if (st && lane == 0) st[k] = __builtin_amdgcn_s_memrealtime();
    };
    auto cstamp = [&](int k) {
        if constexpr (kStamps)
            if (st && lane == 0) st[k] = __builtin_amdgcn_s_memtime();
    };
    int tile_no = 0;
    // The thread's lane number, re-made where it is needed (volatile: never hoisted, never kept): the thread id the kernel starts with is
    // one more register held across every phase, and FC1 has none to spare -- kept, it was spilled to scratch and reloaded once per tile.
    auto lane_now = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // The workgroup barrier; the PRODUCERS' copy is stamped on both sides for the first two tiles (a release is common to all waves, and a
    // step the producers did not arrive last at waited for the consumers).  The consumers stamp only under FR_STAMP_CONSUMERS: their code is
    // then not the product's -- the stamp pointer and tile counter cost registers, hipcc spills, and the spills show up as slow steps.
#if defined(FR_HS_DIAG) && defined(FR_STAMP_CONSUMERS)
    constexpr bool kCStamps = true;
#else
    constexpr bool kCStamps = false;
#endif
    auto pbar = [&](int b) {
        if constexpr (kStamps)
            if (st && tile_no < 2) stamp(4 + 2 * NBAR * tile_no + 2 * b);
        __syncthreads();
        if constexpr (kStamps)
            if (st && tile_no < 2) stamp(5 + 2 * NBAR * tile_no + 2 * b);
    };
    auto bar = [&](int b) {
        if constexpr (kCStamps) pbar(b);
        else __syncthreads();
    };
    stamp(0);

    // ---- packed word descriptors -> LDS (read just in time by the producers: no registers, no vector-memory queue slots) ----
    for (int w = tid; w < a.n_words; w += 768) {
        const uint4 d0 = reinterpret_cast<const uint4 *>(a.words)[2 * w];
        const uint4 d1 = reinterpret_cast<const uint4 *>(a.words)[2 * w + 1];
        // {src[31:0], src[47:32] | stride << 16, rows, idx byte offset | DENSE}
        Dsc[w] = make_uint4(d0.x, (d0.y & 0xFFFFu) | (d0.z << 16), d1.x, ((d0.w & ~FR_DESC_DENSE) * 4u) | (d0.w & FR_DESC_DENSE));
    }
    __syncthreads();

    if (producer) {
        // =========================================== PRODUCERS: the gather ===========================================
        int wl = 0, it0 = 0;                        // word of the slice this thread moves; first of its IPT items inside the tile
        uint32_t idxr[IPT];                         // index values of the slice whose rows are loaded next
        row_t rows[D][IPT];                         // row words in flight: D slices
        unsigned bad = 0u;                          // out-of-range index seen (a lane flag, OR-ed: no compare mask is kept)
        uint2 *Xh2 = reinterpret_cast<uint2 *>(Xr);
        uint32_t *Xw = reinterpret_cast<uint32_t *>(Xr);
        const float xscale = PREC == 2 ? __builtin_ldexpf(1.0f, a.e_act[0]) : 1.0f;   // fp8: features are stored as e4m3(sat(x 2^e_x))
        auto slice_word = [&](int s) {  // this thread's record word in slice s; lanes past the slice (or, fp8, past the record: the zero pad
            const int nw = WPG * (KG - KGS * s < KGS ? KG - KGS * s : KGS);  // up to a whole k-group) repeat the last valid word -- the same row
            const int w = WPG * KGS * s + (wl < nw ? wl : nw - 1);           // as their neighbour, no extra line is fetched -- and never store it
            return w < a.n_words ? w : a.n_words - 1;
        };
        auto I_op = [&](const HkTile &t, int s) {  // index loads of slice s
            const uint4 d = Dsc[slice_word(s)];
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(t.idx), 0, (unsigned)t.batch * (unsigned)a.idx_stride * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < IPT; i++)   // items past the batch: out of the resource's bounds, 0 comes back (no branch)
                idxr[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (unsigned)(t.m0 + it0 + i) * (unsigned)a.idx_stride * 4u + (d.w & 0x7FFFFFFFu), 0, 0);
        };
#ifdef FR_HS_DIAG
        // timing ablations (FR_FUSED_HS_ABLATE, wrong scores).  1: the producers load no rows -- the consumers' own pace.  2: every row load reads
        // row 0 of its table (the same instructions; all lanes of a table share one line).  4 / 5 / 6 / 7: row 0 for the tables of >= 10 M / 1 M /
        // 500 k / 100 k rows only (which table class the producers wait for)
        const bool no_rows = PREC == 1 && a.e_act[3] == -777;
        const uint32_t zero_above = PREC != 1 ? 0u : a.e_act[3] == -778 ? 1u : a.e_act[3] == -780 ? 10000000u : a.e_act[3] == -781 ? 1000000u : a.e_act[3] == -782 ? 500000u : a.e_act[3] == -783 ? 100000u : 0u;
#else
        constexpr bool no_rows = false;
        constexpr uint32_t zero_above = 0u;
#endif
        auto R_op = [&](const HkTile &t, int s, row_t (&r)[IPT]) {  // row loads of slice s (its indices are in idxr)
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
                if (zero_above && nrows >= zero_above) x = 0u;
                if constexpr (SRC == 1) {
                    if (dense) {   // the request's fp32 features: converted here, with W_op's rounding
                        typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
                        const u32x4_t q = *(gptr_t)(base + (uint64_t)x * stride);
                        r[i] = make_uint2(pack_bf16x2(__uint_as_float(q.x), __uint_as_float(q.y)), pack_bf16x2(__uint_as_float(q.z), __uint_as_float(q.w)));
                    } else {
                        typedef const unsigned __attribute__((ext_vector_type(2))) __attribute__((address_space(1))) * gptr2_t;
                        const auto q = *(gptr2_t)(base + (uint64_t)x * stride);
                        r[i] = make_uint2(q.x, q.y);
                    }
                } else {
                    if (no_rows) {
                        r[i] = make_uint4(x, x, x, x);
                        continue;
                    }
                    typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
                    const u32x4_t q = *(gptr_t)(base + (uint64_t)x * stride);
                    r[i] = make_uint4(q.x, q.y, q.z, q.w);
                }
            }
        };
        auto W_op = [&](const HkTile &t, int s, const row_t (&r)[IPT]) {  // slice s: fp32 rows -> bf16 / e4m3 (SRC = 1: as they are) -> X ring buffer s % 2
            const int nw = WPG * (KG - KGS * s < KGS ? KG - KGS * s : KGS);  // words of this slice
            if (wl < nw) {
                const uint32_t real = 0u - (uint32_t)(WPG * KGS * s + wl < a.n_words);  // fp8: words past the record are the zero pad of its last k-group
#pragma unroll
                for (int i = 0; i < IPT; i++) {
                    const uint32_t in = (0u - (uint32_t)(t.m0 + it0 + i < t.batch)) & real;  // all ones / zero: items past the batch are zero rows, branch-free
                    if constexpr (SRC == 1) {
                        uint2 *xb = Xh2 + (size_t)(FULLX ? s : (s & 1)) * (XROWS * HK_LDX * 2);
                        xb[((size_t)(wl >> 1) * HK_LDX + it0 + i) * 2 + (wl & 1)] = make_uint2(r[i].x & in, r[i].y & in);
                    } else if constexpr (PREC == 1) {
                        uint2 *xb = Xh2 + (size_t)(FULLX ? s : (s & 1)) * (XROWS * HK_LDX * 2);
                        uint2 hv;
                        hv.x = pack_bf16x2(__uint_as_float(r[i].x), __uint_as_float(r[i].y)) & in;
                        hv.y = pack_bf16x2(__uint_as_float(r[i].z), __uint_as_float(r[i].w)) & in;
                        xb[((size_t)(wl >> 1) * HK_LDX + it0 + i) * 2 + (wl & 1)] = hv;  // slice word wl = half (wl & 1) of q8 row wl / 2
                    } else {
                        // word wl = k 4 wl .. 4 wl + 3 of the slice: k-group wl / 8, step (wl % 8) / 4, lane half ((wl % 8) % 4) / 2, dword wl % 2
                        uint32_t *xb = Xw + (size_t)(FULLX ? s : (s & 1)) * (XROWS * HK_LDX * 4);
                        xb[((size_t)(2 * (wl >> 3) + ((wl >> 1) & 1)) * HK_LDX + it0 + i) * 4 + 2 * ((wl >> 2) & 1) + (wl & 1)] = pack_fp8_word(r[i], xscale) & in;
                    }
                }
            }
        };
        HkTile cur = tile_at(t_cur);
        if constexpr (FULLX) {
            // ---- whole-tile X image: the gather of tile t + 1 runs under every phase of tile t ----
            // event e of a tile (0 <= e < NEV): convert slice e - 1 - D into its buffer (its rows were requested D events ago), request the
            // rows of slice e - 1 (its indices were requested by the previous event), request the indices of slice e
            auto event = [&](const HkTile &t, int e) {
                if (e - 1 - D >= 0 && e - 1 - D < NSL) W_op(t, e - 1 - D, rows[(e - 1) % D]);
                if (e >= 1 && e - 1 < NSL) R_op(t, e - 1, rows[(e - 1) % D]);
                if (e < NSL) I_op(t, e);
            };
            {
                const int t_ = 64 * (wave - 8) + lane_now();
                wl = t_ & (LW - 1), it0 = (t_ / LW) * IPT;
            }
#pragma unroll
            for (int e = 0; e < NEV; e++) event(cur, e);   // the workgroup's first tile: everything before the consumers' first barrier
            while (true) {
                const int t_nxt = next_tile(t_cur);
                const bool has_next = t_nxt < n_tiles;   // wave-uniform
                HkTile nxt = cur;
                if (has_next) nxt = tile_at(t_nxt);
                {
                    const int t_ = 64 * (wave - 8) + lane_now();
                    wl = t_ & (LW - 1), it0 = (t_ / LW) * IPT;
                }
#pragma unroll
                for (int s = 0; s < NSL; s++) {
                    pbar(s);       // the consumers are done with slice s - 1 of this tile: buffer s - 1 may take the next tile's slice
                    if (has_next && s < NE1) event(nxt, s);   // event s converts slice s - 1 - D at the latest
                }
                pbar(NSL);         // R1 stored; FC2 runs
                if (has_next) {
#pragma unroll
                    for (int e = NE1; e < NE1 + E2; e++) event(nxt, e);
                }
                pbar(NSL + 1);     // FC2 done
                pbar(NSL + 2);     // R2 stored; FC3 runs
                if (has_next) {
#pragma unroll
                    for (int e = NE1 + E2; e < NEV; e++) event(nxt, e);
                }
                pbar(NSL + 3);     // R3 stored
                pbar(NSL + 4);     // partial scores
                tile_no++;
                if (!has_next) break;
                cur = nxt;
                t_cur = t_nxt;
            }
        } else {
            {   // prologue: the first tile's slices 0, 1 into LDS, 2 .. D + 1 requested, the indices of D + 2 requested (every tile starts so).
                // (Issuing every index load first -- two dependent latencies instead of D + 2 -- measured no different: at launch start, with
                // every workgroup in its prologue, the chain takes 13-14 us either way, and the extra index registers spill.)
                const int t_ = 64 * (wave - 8) + lane_now();   // (not the kernel's thread id: held for this, it was live -- and spilled -- across the consumers' code)
                wl = t_ & (LW - 1), it0 = (t_ / LW) * IPT;
    #pragma unroll
                for (int j = 0; j < D; j++) {
                    I_op(cur, j);
                    R_op(cur, j, rows[j]);
                }
                W_op(cur, 0, rows[0]);
                I_op(cur, D);
                R_op(cur, D, rows[0]);
                W_op(cur, 1, rows[1]);
                I_op(cur, D + 1);
                R_op(cur, D + 1, rows[1]);
                I_op(cur, D + 2);
            }
            while (true) {
                const int t_nxt = next_tile(t_cur);
                const bool has_next = t_nxt < n_tiles;
                HkTile nxt = cur;
                if (has_next) nxt = tile_at(t_nxt);
                else nxt.batch = 0;  // no next tile: the run-ahead gather reads row 0 of every table (index loads out of bounds return 0) into buffers nobody consumes
                const int tid_o = 64 * (wave - 8) + lane_now();   // lane geometry re-derived per tile: nothing of it is hoisted and kept live
                wl = tid_o & (LW - 1), it0 = ((tid_o & 255) / LW) * IPT;
                auto tref = [&](int s) -> const HkTile & { return s >= NSL ? nxt : cur; };
    #pragma unroll
                for (int s = 0; s < NSL; s++) {
                    pbar(s);           // consumers start slice s; X[(s + 1) % 2] is free
                    if (s >= 1) {     // write slice s + 1, request the rows of s + 1 + D and the indices of s + 2 + D (slices >= NSL: the next tile's)
                        W_op(tref(s + 1), (s + 1) % NSL, rows[(s + 1) % D]);
                        R_op(tref(s + 1 + D), (s + 1 + D) % NSL, rows[(s + 1) % D]);
                        I_op(tref(s + 2 + D), (s + 2 + D) % NSL);
                    }
                }
                pbar(NSL);             // R1 stored, every consumer is past the last slice: X[1] is free
                W_op(nxt, 1, rows[1 % D]);
                R_op(nxt, 1 + D, rows[1 % D]);
                I_op(nxt, 2 + D);
                pbar(NSL + 1);         // FC2 done
                pbar(NSL + 2);         // R2 stored
                pbar(NSL + 3);         // R3 stored
                pbar(NSL + 4);         // partial scores
                tile_no++;
                if (!has_next) break;
                cur = nxt;
                t_cur = t_nxt;
            }
        }
        stamp(126);
        if (bad) atomicOr_system(a.err_flag, 1);
        return;
    }

    // =============================================== CONSUMERS: the FC chain ===============================================
    // n1 / n2 / n3 and the lane offsets are re-declared opaque at the top of every tile: everything derived from them is tile-loop
    // invariant, and hoisted out of the loop the SGPR offsets of the weight stream alone spilled 200 scalars into vector registers
    unsigned n1 = (unsigned)(128 * wave) * 16u, n2 = (unsigned)(64 * wave) * 16u, n3 = (unsigned)(32 * wave) * 16u;
    FtWk W1 = ftk_w(a.w1q, a.K / 8 /* the model's own q8 rows: an instantiation may be wider than the record, rows past it read 0 */, HK_H1, lane >> 5, lane & 31), W2 = ftk_w(a.w2q, KG2 * 2, HK_H2, lane >> 5, lane & 31), W3 = ftk_w(a.w3q, KG3 * 2, HK_H3, lane >> 5, lane & 31);
    // fp8: the accumulators hold sums of (w 2^e_w)(x 2^e_x); the next activation is quantised as e4m3(sat(value 2^e_next)), so the store
    // scales by 2^(e_next - e_w - e_x) -- powers of two, exact
    const float os1 = PREC == 2 ? __builtin_ldexpf(1.0f, a.e_act[1] - a.e_w[0] - a.e_act[0]) : 1.0f, os2 = PREC == 2 ? __builtin_ldexpf(1.0f, a.e_act[2] - a.e_w[1] - a.e_act[1]) : 1.0f,
                os3 = PREC == 2 ? __builtin_ldexpf(1.0f, a.e_act[3] - a.e_w[2] - a.e_act[2]) : 1.0f;
    auto store_tile = [&](uint4 *img, const f32x16 &acc, int n_local, int m_local, int hk_, int lm_, float os) {
        if constexpr (PREC == 1) hk_store_tile(img, acc, n_local, m_local, hk_, lm_);
        else hk_store_tile_f8(img, acc, n_local, m_local, hk_, lm_, os);
    };
    // The SGPR offset of a fragment is kept as a RUNNING value (one s_add per k-group) that is re-declared opaque once per slice / per
    // 8 k-groups: written as g * row2 + n1, hipcc computes dozens of them ahead of their loads and spills scalars into vector registers.
    unsigned so1 = 0, so2 = 0;   // byte offset of FC1's NEXT k-group; of the k-group FC2's current block of refills counts from
    auto w2load = [&](int dq) { return ftk_load(W2, so2 + (unsigned)(dq >> 1) * W2.row2, 512 * (dq & 1)); };  // FC2: fragment dq past so2's k-group
    auto w3load = [&](int g) { return ftk_load(W3, (unsigned)g * W3.row2 + n3, 0); };                        // FC3: k-group g
    uint4 ring1[R1D]; // FC1: the weight fragments of R1D / 4 k-groups (fragment 4 g + t in slot (4 g + t) % R1D), each refilled right after its second MFMA
    static_assert(R1D == 4 || R1D == 6 || R1D == 8, "one, one and a half or two k-groups of FC1 weights in registers");
    constexpr int RB = 16;  // FC2 / FC3: 16 fragments (their accumulators are 64 / 32 registers: room for a deep ring)
    static_assert(2 * KG2 >= RB && KG3 >= 8 && KG2 % 8 == 0, "the FC2 / FC3 ring arithmetic below");
    uint4 ringb[RB];
    auto w1frag = [&](int q) {  // fragment q = 4 g + t of FC1, addressed from n1 (prologues only; the loop uses the running so1)
        return ftk_load(W1, n1 + (unsigned)(q >> 2) * W1.row2, 512 * (q & 3));
    };
    auto ring1_fill = [&]() {  // fragments 0 .. R1D - 1
#pragma unroll
        for (int i = 0; i < R1D; i++) ring1[i] = w1frag(i);
    };
    ring1_fill();
    if constexpr (kCStamps) stamp(1);
    while (true) {
        // (the tile's descriptor -- scores pointer, first item, batch size -- is read where the scores are stored, and the next tile is looked
        // up at the bottom of the loop: held across the tile they cost registers the FC1 phase does not have and went to scratch)
        asm volatile("" : "+s"(n1), "+s"(n2), "+s"(n3));
        int hk, lm;          // lane geometry, re-derived per phase (lane_now): nothing of it is held across FC1
        {
            const int l_ = lane_now();
            hk = (l_ >> 5) & 1, lm = l_ & 31;
        }
        const unsigned xlane = (unsigned)(2 * KG2 * HK_LD + hk * HK_LDX + lm);  // B-fragment lane base of FC1 (16-byte units): the X ring
        ftk_lane(W1, hk, lm);

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
            bar(s);           // slice s is complete in X[s % 2]; everybody is done reading slice s - 1
            if (kCStamps && s == 0 && tile_no == 0) cstamp(2);
            if (s == 0) so1 = n1;  // k-group 0; advanced once per k-group: inside k-group g it points at g + 1
            asm volatile("" : "+s"(so1));
            const uint4 *xb = lds + xlane + (FULLX ? s : (s & 1)) * (XROWS * HK_LDX);
            uint4 b0 = xb[0], b1 = xb[32];
#pragma unroll
            for (int gl = 0; gl < kgs; gl++) {
                const int g = KGS * s + gl;
                // Issue order inside a k-group (pinned by the sched_barriers): every fragment register is refilled for the next k-group right
                // after its second MFMA and every B register right after its fourth, so each load has 5-6 MFMAs (plus the partner wave's)
                // to land and no second register set is needed -- the consumers have 168 registers, 128 of them accumulators.
                auto mm = [&](int t, int mt, const uint4 &b) { hk_mma<PREC>(acc1[t][mt], ring1[(4 * g + t) % R1D], b); };
                // the slot of fragment q = 4 g + t takes fragment q + R1D: k-group g + (t + R1D) / 4 (so1 + ((t + R1D) / 4 - 1) row2), n tile (t + R1D) % 4
                auto refill = [&](int t) {
                    const int q2 = 4 * g + t + R1D;
#ifdef FR_HS_W1_SKIP   // TIMING ABLATION (wrong scores; `make w1skip` only): every FR_HS_W1_SKIP-th FC1 weight fragment is not loaded, its slot keeps the old one --
                       // an upper bound on what a design that streams fewer W1 bytes per item (a two-CU supertile) could gain (profiles/r05_experiments.md section 3)
                    if (q2 % FR_HS_W1_SKIP == 0) return;
#endif
                    if (q2 < 4 * KG) ring1[(4 * g + t) % R1D] = ftk_load(W1, so1 + (unsigned)((t + R1D) / 4 - 1) * W1.row2, 512 * (q2 & 3));
                };
                const bool more_b = gl + 1 < kgs;
                mm(0, 0, b0), mm(1, 0, b0), mm(0, 1, b1);
                so1 += W1.row2;
                refill(0);
                __builtin_amdgcn_sched_barrier(0);
                mm(1, 1, b1);
                refill(1);
                mm(2, 0, b0), mm(3, 0, b0);
                if (more_b) b0 = xb[(size_t)(2 * (gl + 1)) * HK_LDX];
                __builtin_amdgcn_sched_barrier(0);
                mm(2, 1, b1);
                refill(2);
                mm(3, 1, b1);
                refill(3);
                if (more_b) b1 = xb[(size_t)(2 * (gl + 1)) * HK_LDX + 32];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (kCStamps && tile_no == 0) cstamp(3);

        // ---- R1 -> LDS (bf16), FC2 K-outer over it: 64 outputs x 64 items per wave ----
        so2 = n2;
        {
            const int l_ = lane_now();
            hk = (l_ >> 5) & 1, lm = l_ & 31;
        }
        ftk_lane(W2, hk, lm);
        // FC2's first fragments are requested before the barrier that follows the R1 store -- but only as many as there are registers for:
        // FC1's 128 accumulators are live until their tiles are stored, so half the ring goes out first and the other half once four of the
        // eight tiles have been converted (all 16 at once needed 192 registers of the 168: that was the kernel's scratch)
#pragma unroll
        for (int i = 0; i < RB / 2; i++) ringb[i] = w2load(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) store_tile(R1, acc1[t][mt], 128 * wave + 32 * t, 32 * mt, hk, lm, os1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = RB / 2; i < RB; i++) ringb[i] = w2load(i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 2; t < 4; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) store_tile(R1, acc1[t][mt], 128 * wave + 32 * t, 32 * mt, hk, lm, os1);
        bar(NSL);         // R1 complete; the X ring is free (every consumer is past the last slice)
        const unsigned rlane = (unsigned)(hk * HK_LD + lm);  // B-fragment lane base of FC2 / FC3 (R1 / R2)
        ftk_lane(W3, hk, lm);   // (FC2's tail requests W3's first fragments)
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
            for (int j = 0; j < KG2; j++) {
                const int jn = j + 1 < KG2 ? j + 1 : j;
                const uint4 bn0 = bl[(size_t)(2 * jn) * HK_LD], bn1 = bl[(size_t)(2 * jn) * HK_LD + 32];
                if (j % 8 == 0) {  // so2 = k-group j + RB / 2: the refills of this block of 8 k-groups are fragments 0 .. 15 past it
                    so2 = n2 + (unsigned)(j + RB / 2) * W2.row2;
                    asm volatile("" : "+s"(so2));
                }
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int q = 2 * j + t;
                    hk_mma<PREC>(acc2[t][0], ringb[q % RB], b0);
                    hk_mma<PREC>(acc2[t][1], ringb[q % RB], b1);
                    // fragment q + RB = (q - 16 (j / 8)) past so2; FC2's tail requests W3's first k-groups (as many as it has, up to RB)
                    if (q + RB < 2 * KG2) ringb[q % RB] = w2load(q - 16 * (j / 8));
                    else if (q + RB - 2 * KG2 < KG3) ringb[q % RB] = w3load(q + RB - 2 * KG2);
                }
                __builtin_amdgcn_sched_barrier(0);
                b0 = bn0, b1 = bn1;
            }
        }
        bar(NSL + 1);     // every wave is done reading R1: R2 may overlay it
        {
            const int l_ = lane_now();
            hk = (l_ >> 5) & 1, lm = l_ & 31;
        }
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int mt = 0; mt < 2; mt++) store_tile(R2, acc2[t][mt], 64 * wave + 32 * t, 32 * mt, hk, lm, os2);
        bar(NSL + 2);

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
            for (int j = 0; j < KG3; j++) {
                const int jn = j + 1 < KG3 ? j + 1 : j;
                const uint4 bn0 = bl[(size_t)(2 * jn) * HK_LD], bn1 = bl[(size_t)(2 * jn) * HK_LD + 32];
                hk_mma<PREC>(acc3[0], ringb[j % RB], b0);
                hk_mma<PREC>(acc3[1], ringb[j % RB], b1);
                if (j + RB < KG3) ringb[j % RB] = w3load(j + RB);
                __builtin_amdgcn_sched_barrier(0);
                b0 = bn0, b1 = bn1;
            }
        }
        {
            const int l_ = lane_now();
            hk = (l_ >> 5) & 1, lm = l_ & 31;
        }
        ftk_lane(W1, hk, lm);
        ring1_fill();  // the next tile's first k-group(s) of FC1: requested before the R3 store and the barriers
#pragma unroll
        for (int mt = 0; mt < 2; mt++) store_tile(R3, acc3[mt], 32 * wave, 32 * mt, hk, lm, os3);
        bar(NSL + 3);
        {   // score[m] = sum_n wout[n] * R3[n][m], fp32 sum: 64 items x 8 slices of the image's rows, fixed-order reduction
            const int il = lane_now(), sl = wave;
            float sc = 0.0f;
            if constexpr (PREC == 1) {   // bf16 x bf16: 4 q8 rows per slice
                const uint4 *wh = reinterpret_cast<const uint4 *>(a.wout);  // bf16 vector w[k], 8 per element
                for (int q = 4 * sl; q < 4 * sl + 4; q++) {
                    const uint4 r = R3[(size_t)q * HK_LD + il];
                    const uint4 w = wh[q];
                    const uint32_t rr[4] = {r.x, r.y, r.z, r.w}, ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        sc = fmaf(__uint_as_float(ww[e] << 16), __uint_as_float(rr[e] << 16), sc);
                        sc = fmaf(__uint_as_float(ww[e] & 0xFFFF0000u), __uint_as_float(rr[e] & 0xFFFF0000u), sc);
                    }
                }
            } else {                     // fp32 master weights x decoded e4m3: 2 "q16h" rows per slice; byte b of (row 2 j + kh) is n = 32 j + 16 (b / 8) + 8 kh + b % 8
                for (int row = 2 * sl; row < 2 * sl + 2; row++) {
                    const uint4 r = R3[(size_t)row * HK_LD + il];
                    const int rr[4] = {(int)r.x, (int)r.y, (int)r.z, (int)r.w};
                    const float *w = a.wout + 32 * (row >> 1) + 8 * (row & 1);
#pragma unroll
                    for (int d = 0; d < 4; d++) {   // dword d: step d / 2, positions 4 (d % 2) .. + 3
                        const float *wd = w + 16 * (d >> 1) + 4 * (d & 1);
                        sc = fmaf(wd[0], __builtin_amdgcn_cvt_f32_fp8(rr[d], 0), sc);
                        sc = fmaf(wd[1], __builtin_amdgcn_cvt_f32_fp8(rr[d], 1), sc);
                        sc = fmaf(wd[2], __builtin_amdgcn_cvt_f32_fp8(rr[d], 2), sc);
                        sc = fmaf(wd[3], __builtin_amdgcn_cvt_f32_fp8(rr[d], 3), sc);
                    }
                }
            }
            part[sl * 64 + il] = sc;
            bar(NSL + 4);
            const int tl = 64 * wave + lane_now();  // (the score address is computed here, not at the top of the tile)
            const HkTile cur = tile_at(t_cur);
            if (tl < 64 && cur.m0 + tl < cur.batch) {
                float t = part[tl];
#pragma unroll
                for (int i = 1; i < 8; i++) t += part[i * 64 + tl];
                if constexpr (PREC == 2) t *= __builtin_ldexpf(1.0f, -a.e_act[3]);
                cur.scores[cur.m0 + tl] = t;
            }
        }
        tile_no++;
        t_cur = next_tile(t_cur);
        if (t_cur >= n_tiles) break;
    }
    if constexpr (kCStamps) stamp(126);
}

// Which models the K-outer kernel takes: the reference's FC widths (1024 / 512 / 256, constant.h:24-27 -- the LDS plan is built on them) and
// ANY record of 64 .. 880 floats in whole k-groups of 16 (the generated constants.hpp of a user kernel, embedding_47_krnl/src/hls/
// constants.hpp:28,505, may hold any table set): the record rides the narrowest of four instantiations (22 / 33 / 44 / 55 k-groups = 352 /
// 528 / 704 / 880 floats) that holds it, the k-groups past the record are zeros on both sides (weight rows past the matrix read 0 through
// the buffer resource's bounds, the producers store zero words).  Every word descriptor must fit the packed 16-byte form: 48-bit source
// address, 16-bit row stride.
bool frk_fused_hk_ok(int K, int H1, int H2, int H3, const FrWordDesc *h_words, int n_words) {
    if (K % 16 || K < 64 || K > 880 || H1 != HK_H1 || H2 != HK_H2 || H3 != HK_H3 || n_words != K / 4) return false;
    for (int w = 0; w < n_words; w++)
        if (h_words[w].stride >= 65536u || (h_words[w].src >> 48) != 0 || (h_words[w].idx_col & ~FR_DESC_DENSE) >= (1u << 28)) return false;
    return true;
}

bool frk_fused_hk_takes_lp_rows(int K) {   // experiments build, opt-in: see frk_fused_hk_launch
#ifdef FR_EXPERIMENTS
    return FR_KNOB_ONCE("FUSED_LP_ROWS", 0) == 1 && K % 16 == 0 && K > 704 && K <= 880;
#else
    (void)K;
    return false;
#endif
}

template <int PREC, int KG, int KGS, int LW, int D, int R1D, int E2 = 0, int E3 = 0, int SRC = 0>
static int fused_hk_launch_inst(const FrFusedArgs &a, int n_cu, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    if (int rc_ = fr_allow_full_lds(&fr_fused_tile_hs_kernel<PREC, KG, KGS, LW, D, R1D, E2, E3, SRC>, lds_once)) return rc_;
    const size_t r1_rows = 2 * (HK_H1 / (PREC == 2 ? 32 : 16));
    const size_t x_bufs = PREC == 2 ? (KG + KGS - 1) / KGS : 2;   // fp8: the X image of a whole tile; bf16: a ring of two slices
    const size_t lds = (r1_rows * HK_LD + x_bufs * 2 * KGS * HK_LDX + (size_t)a.n_words) * 16;
    if (lds > 160 * 1024) FR_FAIL(FR_ERR_INVALID, "internal: the K-outer fused kernel needs %zu bytes of LDS", lds);
    const int tiles = a.n_batches * a.tiles_per_batch;
    fr_fused_tile_hs_kernel<PREC, KG, KGS, LW, D, R1D, E2, E3, SRC><<<dim3(tiles < n_cu ? tiles : n_cu), dim3(768), lds, s>>>(a);
    fr_note_kernel("fr_fused_tile_hs_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d>", PREC, KG, KGS, LW, D, R1D, E2, E3, SRC);   // as rocprofv3 prints it
    KCHECK();
    return FR_OK;
}

// a.w1q/w2q/w3q point at the bf16 q8 weights (+ a.wout the bf16 output vector) or at the e4m3 "q16h" weights (+ fp32 a.wout, a.e_w /
// a.e_act the exponents); a.blist the launch's batches in device memory; a.tiles_per_batch counts 64-item tiles; one persistent
// workgroup per CU
int frk_fused_hk_launch(const FrFusedArgs &a, int n_cu, int precision, hipStream_t s) {
    if (precision == FR_FC_FP8) {
        // The fp8 form is built into the EXPERIMENTS library only: it is correct (tests/test_gpu_lowprec.py::test_fp8_persistent_fused_kernel_many_tiles
        // under FR_LIB=libfleetrec_exp.so FR_FUSED_HK=1) and SLOWER than the chunked fr_fused_tile_f8_kernel (Model-B 1024: 340 vs
        // 398-400 M inf/s, profiles/archive/r03_fused_hs_fp8_ab.txt) -- with the consumers twice as fast as in bf16, the gather, which only runs
        // under FC1, is what a tile waits for.
#ifdef FR_EXPERIMENTS
        if (a.K == 880) {   // Model-B: K 880 -> 896 = 28 k-groups of 32, 14 slices of 2; 4 row sets in flight; 19 gather events per tile
            switch (FR_KNOB_ONCE("FUSED_F8_SCHED", 63)) {   // 10 E2 + E3: gather events under FC2 / FC3 (the rest under FC1)
                case 50: return fused_hk_launch_inst<2, 28, 2, 16, 4, 4, 5, 0>(a, n_cu, s);
                case 42: return fused_hk_launch_inst<2, 28, 2, 16, 4, 4, 4, 2>(a, n_cu, s);
                case 84: return fused_hk_launch_inst<2, 28, 2, 16, 4, 4, 8, 4>(a, n_cu, s);
                default: return fused_hk_launch_inst<2, 28, 2, 16, 4, 4, 6, 3>(a, n_cu, s);
            }
        }
        if (a.K == 352) return fused_hk_launch_inst<2, 11, 2, 16, 3, 4, 3, 1>(a, n_cu, s);   // Model-A: K 352 = 11 k-groups of 32, 6 slices of 2 (1); 10 events
#endif
        FR_FAIL(FR_ERR_INVALID, "no K-outer fp8 fused instantiation for K=%d in this build", a.K);
    }
#ifdef FR_EXPERIMENTS
    if (a.K == 880 && FR_KNOB_ONCE("FUSED_R1D", 6) == 4) return fused_hk_launch_inst<1, 55, 7, 32, 2, 4>(a, n_cu, s);
#endif
    if (a.src_lp) {
        // Rows already bf16 (fr_ctx::lp_arena, SRC = 1): 8-byte row words, so FOUR row sets ride in the producers' registers where two did -- 168
        // registers, no scratch: the depth round 5 could not reach (r05_experiments.md section 10).  MEASURED SLOWER (profiles/r06_experiments.md
        // section 3: Model-B 1024, per table 324 -> 319 M inf/s, per bank 357 -> 341 M; with two row sets of 8-byte words 314 / 336 M): the producers
        // are not short of rows in flight, and 8-byte row loads cost more than 16-byte ones.  EXPERIMENTS build only (FR_FUSED_LP_ROWS=1).
#ifdef FR_EXPERIMENTS
        if (a.K > 704 && a.K <= 880 && FR_KNOB_ONCE("FUSED_LP_D", 4) == 2) return fused_hk_launch_inst<1, 55, 7, 32, 2, 6, 0, 0, 1>(a, n_cu, s);   // the bytes alone: 8-byte rows, still two row sets
        if (a.K > 704 && a.K <= 880) return fused_hk_launch_inst<1, 55, 7, 32, 4, 6, 0, 0, 1>(a, n_cu, s);   // Model-B: 8 slices, 4 row sets
#endif
        FR_FAIL(FR_ERR_INVALID, "internal: no operand-type-rows instantiation of the K-outer fused kernel for K=%d in this build", a.K);
    }
    if (a.K % 16 == 0 && a.K >= 64) {
        if (a.K <= 352) return fused_hk_launch_inst<1, 22, 4, 16, 3, 4>(a, n_cu, s);   // Model-A: 6 slices of 4 (2) k-groups, 3 row sets in flight, 4 FC1 fragments
        if (a.K <= 528) return fused_hk_launch_inst<1, 33, 6, 32, 3, 4>(a, n_cu, s);   // 6 slices of 6 (3) k-groups, 3 row sets
        if (a.K <= 704) return fused_hk_launch_inst<1, 44, 6, 32, 2, 6>(a, n_cu, s);   // 8 slices of 6 (2) k-groups, 2 row sets
        if (a.K <= 880) return fused_hk_launch_inst<1, 55, 7, 32, 2, 6>(a, n_cu, s);   // Model-B: 8 slices of 7 (6) k-groups, 2 row sets in flight, 6 FC1 fragments
    }
    FR_FAIL(FR_ERR_INVALID, "no K-outer bf16 fused instantiation for K=%d", a.K);
}
