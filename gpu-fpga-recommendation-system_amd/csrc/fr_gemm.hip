#include <type_traits>

#include "fr_device.h"

// ===================================================================================================
// fc_lp_gemm_kernel<PREC>: LDS-tiled GEMM for low-precision layers that fill the chip on their own (Model-C FC1 at batch 4096).
// PREC 0 = fp32 (q4 elements, v_mfma_f32_32x32x2_f32: the same k-ordered exact-f32 sums as the other fp32 kernels, full K per
// output), PREC 1 = bf16 (q8 elements, v_mfma_f32_32x32x16_bf16), PREC 2 = fp8 (q16 elements, v_mfma_scale_f32_32x32x64_f8f6f4).
// Block tile 128 (n) x 256 (m), 8 waves as 2 x 4 with 64 x 64 wave tiles (2 x 2 MFMA tiles: 4 fragment reads per 4 MFMAs).
// A K step is 8 rows of 16-byte elements (64 k in bf16, 128 k in fp8): 16 KiB of W + 32 KiB of X, two steps in LDS (96 KiB).
// Operands go global -> LDS directly (buffer_load ... lds, 64 consecutive elements per wave-instruction, no VGPR staging) one
// step ahead of the MFMAs; one barrier per step.  Fragment reads are conflict-free ds_read_b128 (a b128 access is served 16
// lanes at a time, and 16 consecutive elements of a row are 256 contiguous bytes).  L2 -> CU traffic per output is 25 % below
// that of 128 x 128 tiles.
// ===================================================================================================
constexpr int FR_GN = 128, FR_GR = 8, FR_GSTAGES = 2;
// MU = m tiles per wave: 2 -> block tile 128 x 256 (64 x 64 wave tiles), 1 -> 128 x 128 (64 x 32 wave tiles) for layers with too few
// 128 x 256 tiles to cover the chip (Model-C FC2 / FC3 at batch 4096).  Two K steps in LDS (64 / 96 KiB): the loads of step s+1 are in
// flight during the MFMAs of step s, and a 32 KiB stage-pipeline workgroup of another stream still fits beside the workgroup.
// (Three steps, 144 KiB, were 10 % faster alone and slower overall: profiles/archive/r01_experiments.md.)

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// EPI = 1: the OUTPUT LAYER folded into the epilogue (SURVEY.md "FC-out (N = 1) folded into FC3 epilogue"; the reference runs it as a fourth
// GEMM with one output, cuda_server.c:486-491): a 256 (n) x 128 (m) tile holds ALL of FC3's outputs of its 128 items, so the tile rounds
// R3 once to the chain's type in registers, multiplies by wout and writes the items' scores -- R3 never reaches memory and the batch
// leaves the stage pipeline one launch earlier.
struct FrTailArgs {
    const void *wout;   // bf16 chain: bf16 w[n]; fp8 chain: fp32 w[n] (its output layer runs in fp32 on the de-quantised R3)
    float *scores;
    int batch;          // items past it (the padding up to ldm) are not stored
    float out_scale;    // fp8: 2^-e of R3's quantisation
};

template <int PREC, int MU, int GN, int S, int GR, int MF = 32, int EPI = 0>
__device__ __forceinline__ void lp_gemm_body(uint4 *glds, const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                             int N, int ldm, int sc_a, int sc_b, float oscale, const FrTailArgs tail = FrTailArgs{}) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int GM = 128 * MU, ROW = GN + GM, TN = GN / 64;  // elements per staged row: GN of W, GM of X; TN n tiles of 32 per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tn = N / GN, tm = ldm / GM;
    // XCD-aware tile map: workgroup b runs on XCD b % 8.  The XCDs form a 2 (n) x 4 (m) grid and each owns a tn/2 x tm/4 block
    // of tiles, so its L2 sees tn/2 weight panels + tm/4 activation panels instead of (with a linear map) two weight panels and
    // EVERY activation panel: half the traffic from beyond L2 for Model-C FC1.  The workgroups of an XCD walk K in step, so a
    // panel row is fetched once and hit by the others whatever the L2 capacity.
    int n_tile, m_tile;
    if (tn % 2 == 0 && tm % 4 == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int tnx = tn / 2;
        n_tile = (x & 1) * tnx + j % tnx;
        m_tile = (x >> 1) * (tm / 4) + j / tnx;
    } else {
        n_tile = blockIdx.x % tn;
        m_tile = blockIdx.x / tn;
    }
    const int n0 = n_tile * GN, m0 = m_tile * GM;
    const int r = lane & 31, h = lane >> 5;
    // Buffer resources built by hand (SGPR quads for the inline asm below): base, stride 0, bytes, gfx9 raw-buffer flags.
    auto make_rs = [](const void *p, unsigned bytes) {
        const unsigned long long a = (unsigned long long)p;
        i32x4_t rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        rs[3] = 0x00020000;
        return rs;
    };
    const i32x4_t rsW = make_rs(W, (unsigned)KE * (unsigned)N * 16u), rsX = make_rs(X, (unsigned)KE * (unsigned)ldm * 16u);
    auto As = [&](int st, int row) { return glds + ((size_t)st * GR + row) * ROW; };          // GN elements
    auto Bs = [&](int st, int row) { return glds + ((size_t)st * GR + row) * ROW + GN; };  // GM elements
    // global -> LDS without a VGPR round trip: lane i's 16 bytes land at M0 + 16 i.  Inline asm on purpose: through the builtin the
    // compiler treats every LDS read as a possible alias of the DMA write and waits for vmcnt(0) before each fragment read; the
    // s_waitcnt below is the only synchronisation these loads need.  M0 is written here without a clobber entry: hipcc rejects "m0" in a
    // clobber list (reserved register, the entry is ignored with a warning), and this kernel contains no other M0 user (no LDS-DMA
    // builtin, s_sendmsg, s_movrel or GWS op) whose M0 set-up the compiler could have hoisted across the asm.
    auto dma = [&](const i32x4_t &rs, const uint4 *dst, unsigned voff, unsigned soff) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr)dst);
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    // staging: wave-instructions of 64 elements; every wave issues 2 of W's 16 and 2 MU of X's 16 MU per step
    const unsigned vW = (unsigned)(n0 + (GN == 256 ? 64 * (wave & 3) : GN == 128 ? 64 * (wave & 1) : 0) + lane) * 16u, vX = (unsigned)(m0 + lane) * 16u;
    auto issue = [&](int step, int st) {
        if constexpr (GN == 256) {   // 256-wide n tile: four wave-instructions per row, wave w takes quarter w % 4 of rows w / 4, w / 4 + 2, ...
#pragma unroll
            for (int i = 0; i < GR / 2; i++) {
                const int row = (wave >> 2) + 2 * i;
                dma(rsW, As(st, row) + 64 * (wave & 3), vW, (unsigned)(step * GR + row) * (unsigned)N * 16u);
            }
        } else if constexpr (GN == 128) {
#pragma unroll
            for (int i = 0; i < GR / 4; i++) {
                const int row = (wave >> 1) + 4 * i;  // 8 waves x (GR / 4) = rows 0..GR-1 x 2 halves
                dma(rsW, As(st, row) + 64 * (wave & 1), vW, (unsigned)(step * GR + row) * (unsigned)N * 16u);
            }
        } else {  // 64-wide n tile: one wave-instruction per row, wave w takes rows w, w + 8, ...
#pragma unroll
            for (int rr = 0; rr < GR / 8; rr++) dma(rsW, As(st, wave + 8 * rr), vW, (unsigned)(step * GR + wave + 8 * rr) * (unsigned)N * 16u);
        }
#pragma unroll
        for (int rr = 0; rr < GR / 8; rr++)
#pragma unroll
            for (int i = 0; i < 2 * MU; i++)  // rows wave, wave + 8, ...; GM / 64 parts each
                dma(rsX, Bs(st, wave + 8 * rr) + 64 * i, vX + 64u * 16u * i, (unsigned)(step * GR + wave + 8 * rr) * (unsigned)ldm * 16u);
    };
    // MF = 16 (bf16 only): v_mfma_f32_16x16x32_bf16 on 16 x 16 output tiles -- same LDS bytes and MFMA cycles per FLOP as the 32x32x16 form, but
    // the chip holds a higher clock on it under load (see fc_gemm_pipe_kernel); the accumulators are the same registers seen as 4 x 4 tiles
    static_assert(MF == 32 || (MF == 16 && PREC == 1 && GR % 4 == 0), "16 x 16 MFMA tiles: bf16 only");
    f32x16 acc[TN][MU];
    f32x4_t acc16[MF == 16 ? 2 * TN : 1][MF == 16 ? 2 * MU : 1];
    if constexpr (MF == 16) {
#pragma unroll
        for (int t = 0; t < 2 * TN; t++)
#pragma unroll
            for (int u = 0; u < 2 * MU; u++) acc16[t][u] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
#pragma unroll
        for (int t = 0; t < TN; t++)
#pragma unroll
            for (int u = 0; u < MU; u++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
    }
    const int nsteps = KE / GR;
    // S stages in LDS: the loads of steps s + 1 .. s + S - 1 are in flight during the MFMAs of step s.  S = 2 is the round-1 form (one
    // step ahead, vmcnt(0)); the 64 x 128 tile of the narrow layers runs S = 4 with a counted wait -- its steps are only 128 MFMA
    // cycles per wave, far shorter than a load's round trip.
    constexpr int LPS = (GR / 8) * ((GN == 256 ? 4 : GN == 128 ? 2 : 1) + 2 * MU);      // DMA instructions per wave per step
    constexpr unsigned WAITN = (unsigned)LPS * (S - 2);    // may stay in flight when step s must have landed
    constexpr int IMM_WAIT = (int)((WAITN & 0xF) | ((WAITN >> 4) << 14) | 0x0F70u);
#pragma unroll
    for (int i = 0; i < S - 1; i++)
        if (i < nsteps) issue(i, i);
    for (int s = 0; s < nsteps; s++) {
        if (S == 2 || s + S - 2 >= nsteps) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's loads of step s have landed
        else __builtin_amdgcn_s_waitcnt(IMM_WAIT);                             // ... counted: the younger steps stay in flight
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();  // everyone's step-s data is in LDS, and everyone is done reading the buffer the next issue overwrites
        asm volatile("" ::: "memory");
        if (s + S - 1 < nsteps) issue(s + S - 1, (s + S - 1) % S);
        const int st = s % S;
        if constexpr (PREC == 0) {  // q4 fp32 elements: one element per lane feeds four v_mfma_f32_32x32x2_f32 (k = 8 kk + 4 h + c)
#pragma unroll
            for (int kk = 0; kk < GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 32 * TN + r, *br = Bs(st, 2 * kk + h) + wm * 32 * MU + r;
                uint4 a[TN], b[MU];
#pragma unroll
                for (int t = 0; t < TN; t++) a[t] = ar[32 * t];
#pragma unroll
                for (int u = 0; u < MU; u++) b[u] = br[32 * u];
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int u = 0; u < MU; u++)
#pragma unroll
                        for (int t = 0; t < TN; t++) {
                            const uint32_t av = c == 0 ? a[t].x : c == 1 ? a[t].y : c == 2 ? a[t].z : a[t].w;
                            const uint32_t bv = c == 0 ? b[u].x : c == 1 ? b[u].y : c == 2 ? b[u].z : b[u].w;
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av), __uint_as_float(bv), acc[t][u], 0, 0, 0);
                        }
            }
        } else if constexpr (PREC == 1 && MF == 16) {
#pragma unroll
            for (int kk = 0; kk < GR / 4; kk++) {   // 32 k: lane (g, j) = k group g = lane / 16 of the four, row j = lane % 16 of its tile
                const uint4 *ar = As(st, 4 * kk + (lane >> 4)) + wn * 32 * TN + (lane & 15), *br = Bs(st, 4 * kk + (lane >> 4)) + wm * 32 * MU + (lane & 15);
                uint4 a[2 * TN], b[2 * MU];
#pragma unroll
                for (int t = 0; t < 2 * TN; t++) a[t] = ar[16 * t];
#pragma unroll
                for (int u = 0; u < 2 * MU; u++) b[u] = br[16 * u];
#pragma unroll
                for (int u = 0; u < 2 * MU; u++)
#pragma unroll
                    for (int t = 0; t < 2 * TN; t++)
                        acc16[t][u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, b[u]), acc16[t][u], 0, 0, 0);
            }
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int kk = 0; kk < GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 32 * TN + r, *br = Bs(st, 2 * kk + h) + wm * 32 * MU + r;
                uint4 a[TN], b[MU];
#pragma unroll
                for (int t = 0; t < TN; t++) a[t] = ar[32 * t];
#pragma unroll
                for (int u = 0; u < MU; u++) b[u] = br[32 * u];
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int t = 0; t < TN; t++)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, b[u]), acc[t][u], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < GR / 4; kk++) {
                i32x8 a[TN], b[MU];
#pragma unroll
                for (int t = 0; t < TN; t++) {
                    const uint4 alo = As(st, 4 * kk + 2 * h)[wn * 32 * TN + 32 * t + r], ahi = As(st, 4 * kk + 2 * h + 1)[wn * 32 * TN + 32 * t + r];
                    a[t][0] = (int)alo.x; a[t][1] = (int)alo.y; a[t][2] = (int)alo.z; a[t][3] = (int)alo.w;
                    a[t][4] = (int)ahi.x; a[t][5] = (int)ahi.y; a[t][6] = (int)ahi.z; a[t][7] = (int)ahi.w;
                }
#pragma unroll
                for (int u = 0; u < MU; u++) {
                    const uint4 blo = Bs(st, 4 * kk + 2 * h)[wm * 32 * MU + 32 * u + r], bhi = Bs(st, 4 * kk + 2 * h + 1)[wm * 32 * MU + 32 * u + r];
                    b[u][0] = (int)blo.x; b[u][1] = (int)blo.y; b[u][2] = (int)blo.z; b[u][3] = (int)blo.w;
                    b[u][4] = (int)bhi.x; b[u][5] = (int)bhi.y; b[u][6] = (int)bhi.z; b[u][7] = (int)bhi.w;
                }
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int t = 0; t < TN; t++) acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], b[u], acc[t][u], 0, 0, 0, sc_a, 0, sc_b);
            }
        }
    }
    if constexpr (EPI == 1) {
        // The output layer, bit for bit as the stage pipeline's own out stage computes it (fc_out_h_body / fc_out_f_body, fr_pipeline.hip): the
        // N = 256 outputs of an item in eight slices of 32 n, each slice one fmaf chain over ASCENDING n from 0, the eight slice sums added
        // in slice order.  A slice is exactly one 32 x 32 MFMA tile of the accumulators, but a tile's 32 n of an item sit in TWO lanes (lane r:
        // n = 8 i + c, lane r + 32: n = 8 i + 4 + c): v_permlane32_swap on the accumulators of a PAIR of tiles hands lane r all 32 n of the
        // even tile and lane r + 32 all 32 n of the odd one, so every lane runs one whole chain and no partial sum crosses lanes.
        static_assert(PREC != 0 && MU == 1 && GN == 256 && MF == 32 && TN == 4, "the fused output layer: 256 x 128 tiles of the bf16 / fp8 chains");
        float *red = reinterpret_cast<float *>(glds);   // [8 slices][128 items]; the operand stages are dead after the K loop
        __syncthreads();                                // ... once every wave has read its last fragments
#pragma unroll
        for (int p = 0; p < TN / 2; p++) {
            f32x16 &E = acc[2 * p][0], &O = acc[2 * p + 1][0];
            float a_[16], b_[16];   // after the swap: a_ = the lane's tile, registers of the h = 0 half (n = 8 i + c), b_ = the h = 1 half (n = 8 i + 4 + c)
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(E[j]), __float_as_uint(O[j]), false, false);
                a_[j] = __uint_as_float(sw[0]);
                b_[j] = __uint_as_float(sw[1]);
            }
            const int slice = 4 * wn + 2 * p + h;       // = (n / 32) of the lane's tile: n0 == 0, the wave's tiles start at 128 wn
            float sacc = 0.0f;
            if constexpr (PREC == 1) {
                const uint4 *wv = reinterpret_cast<const uint4 *>(tail.wout) + 4 * slice;   // 8 bf16 per element
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const uint4 w4 = wv[i];
                    const uint32_t ww[4] = {w4.x, w4.y, w4.z, w4.w};
                    const uint32_t rr[4] = {pack_bf16x2(a_[4 * i + 0], a_[4 * i + 1]), pack_bf16x2(a_[4 * i + 2], a_[4 * i + 3]),
                                            pack_bf16x2(b_[4 * i + 0], b_[4 * i + 1]), pack_bf16x2(b_[4 * i + 2], b_[4 * i + 3])};
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        sacc = fmaf(__uint_as_float(ww[c] << 16), __uint_as_float(rr[c] << 16), sacc);
                        sacc = fmaf(__uint_as_float(ww[c] & 0xFFFF0000u), __uint_as_float(rr[c] & 0xFFFF0000u), sacc);
                    }
                }
            } else {
                const float4 *wf = reinterpret_cast<const float4 *>(tail.wout) + 8 * slice;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int q_[2] = {(int)pack_fp8x4(a_[4 * i + 0], a_[4 * i + 1], a_[4 * i + 2], a_[4 * i + 3], oscale),
                                       (int)pack_fp8x4(b_[4 * i + 0], b_[4 * i + 1], b_[4 * i + 2], b_[4 * i + 3], oscale)};
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        const float4 w4 = wf[2 * i + c];
                        sacc = fmaf(w4.x, __builtin_amdgcn_cvt_f32_fp8(q_[c], 0), sacc);
                        sacc = fmaf(w4.y, __builtin_amdgcn_cvt_f32_fp8(q_[c], 1), sacc);
                        sacc = fmaf(w4.z, __builtin_amdgcn_cvt_f32_fp8(q_[c], 2), sacc);
                        sacc = fmaf(w4.w, __builtin_amdgcn_cvt_f32_fp8(q_[c], 3), sacc);
                    }
                }
            }
            red[slice * 128 + wm * 32 + r] = sacc;
        }
        __syncthreads();
        if (tid < 128 && m0 + tid < tail.batch) {
            float t = red[tid];
#pragma unroll
            for (int q = 1; q < 8; q++) t += red[q * 128 + tid];
            tail.scores[m0 + tid] = PREC == 2 ? t * tail.out_scale : t;
        }
        return;
    }
    // epilogue: ONE rounding per output; registers 4i..4i+3 of a tile are 4 consecutive n
    if constexpr (MF == 16) {   // 16 x 16 tiles: lane holds m = lane % 16 and n = 4 (lane / 16) + c
#pragma unroll
        for (int t = 0; t < 2 * TN; t++)
#pragma unroll
            for (int u = 0; u < 2 * MU; u++) {
                const f32x4_t &c = acc16[t][u];
                const int m = m0 + wm * 32 * MU + 16 * u + (lane & 15);
                const int n = n0 + wn * 32 * TN + 16 * t + 4 * (lane >> 4);
                uint2 hv;
                hv.x = pack_bf16x2(c[0], c[1]);
                hv.y = pack_bf16x2(c[2], c[3]);
                reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
            }
        return;
    }
#pragma unroll
    for (int t = 0; t < TN; t++)
#pragma unroll
        for (int u = 0; u < MU; u++) {
            const f32x16 &c = acc[t][u];
            const int m = m0 + wm * 32 * MU + 32 * u + r;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int n = n0 + wn * 32 * TN + 32 * t + 8 * i + 4 * h;  // + c
                if constexpr (PREC == 0) {  // q4 element (n / 4) of the next layer's operand
                    reinterpret_cast<float4 *>(Y)[(size_t)(n >> 2) * ldm + m] = make_float4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3]);
                } else if constexpr (PREC == 1) {
                    uint2 hv;
                    hv.x = pack_bf16x2(c[4 * i + 0], c[4 * i + 1]);
                    hv.y = pack_bf16x2(c[4 * i + 2], c[4 * i + 3]);
                    reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
                } else {
                    reinterpret_cast<uint32_t *>(Y)[((size_t)(n >> 4) * ldm + m) * 4 + ((n & 15) >> 2)] =
                        pack_fp8x4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], oscale);
                }
            }
        }
}

template <int PREC, int MU, int GN, int S, int GR = FR_GR, int MF = 32>
__global__ void __launch_bounds__(512) fc_lp_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                         int N, int ldm, int sc_a, int sc_b, float oscale) {
    extern __shared__ uint4 glds[];
    lp_gemm_body<PREC, MU, GN, S, GR, MF>(glds, W, X, Y, KE, N, ldm, sc_a, sc_b, oscale);
}

// FC3 + the output layer: 256 (n) x 128 (m) tiles, EPI = 1 (see FrTailArgs)
template <int PREC, int S>
__global__ void __launch_bounds__(512) fc_lp_gemm_out_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, int KE, int N, int ldm, int sc_a, int sc_b, float oscale,
                                                             const FrTailArgs tail) {
    extern __shared__ uint4 glds[];
    lp_gemm_body<PREC, 1, 256, S, FR_GR, 32, 1>(glds, W, X, nullptr, KE, N, ldm, sc_a, sc_b, oscale, tail);
}

// ===================================================================================================
// fc_pp_gemm_kernel<PREC, D, NT>: the 256 (n) x 256 (m) tile of the bf16 / fp8 GEMM layers with the two waves of every SIMD in OPPOSITE phases.
// Same operand images, same 2 x 4 wave grid with 128 x 64 wave tiles, same MFMA instructions on the same k groups in the same order as
// fc_lp_gemm_kernel<PREC, 2, 256, ...> (scores are bit-identical), but where that kernel's eight waves all issue their DMAs, all read
// their fragments and only then all want the matrix pipe, here a K sub-step of 4 element rows (32 k in bf16, 64 k in fp8: 512 MFMA
// cycles per wave) is split into a MEMORY phase (the wave reads the sub-step's 12 fragments into registers and issues its 4 DMAs of
// sub-step s + D) and a MATRIX phase (its 32 / 8 MFMAs, from registers; fp8: at raised priority), the phases are fenced by s_barrier, and waves
// 4-7 run one phase behind waves 0-3: at any moment one wave of each SIMD multiplies while the other one fetches.  D + 1 sub-steps in
// LDS (bf16 D = 3: 128 KiB; fp8 D = 2: 96 KiB).  Hazards, with M(s) / C(s) the phases of sub-step s: group 0 runs M(s) in global phase 2 s, group 1 in 2 s + 1;
//   * landed before read: a wave waits (counted vmcnt: the D - 1 younger sub-steps stay in flight) for its own DMAs of sub-step s + 1 at
//     the end of M(s), i.e. before the barrier that ends phase 2 s + 1 at the latest; the first read of sub-step s + 1 is in phase 2 s + 2;
//   * read before overwritten: the DMAs of sub-step s + D, issued in M(s) (phase 2 s or later), overwrite the stage of sub-step s - 1,
//     whose last reads (group 1's M(s - 1), phase 2 s - 1) were retired by lgkmcnt(0) before that phase's closing barrier.
// ===================================================================================================
template <int PREC, int D, int NT, int GN, int PR>
__device__ __forceinline__ void pp_gemm_body(uint4 *glds, const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                             int N, int ldm, int sc_a, int sc_b, float oscale, int ablate) {
    static_assert(PREC == 1 || PREC == 2, "bf16 / fp8");
    static_assert((GN == 256 && PR == 4) || (GN == 128 && PR == 8), "256 x 256 tiles in sub-steps of 4 element rows, 128 x 256 tiles in sub-steps of 8");
#ifndef FR_EXPERIMENTS
    ablate = 0;   // timing ablations (wrong results: 1 = no DMA inside the loop, 2 = no fragment reads, 4 = no MFMAs, 8 = every sub-step re-reads the first
                  // one's rows) and schedule variants (16 = the other s_setprio choice, 32 = DMAs before the fragment reads, 64 = the fetching wave at raised priority) exist in the experiments build only
#endif
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int GM = 256, ROW = GN + GM, S = D + 1, STAGE = PR * ROW;
    constexpr int WN = GN / 2, KGS = PR / 4;   // n per wave (2 x 4 waves: WN x 64 wave tiles); k groups of 4 element rows per sub-step
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1, grp = wave >> 2;
    const int tn = NT ? NT : N / GN, tm = ldm / GM;   // NT: the layer's n tiles at compile time (8: Model-C FC1, 2: its FC2; 0 = any) -- the tile map's
                                                      // divisions fold, and the chain's layers carry distinct kernel names in a profile
    int n_tile, m_tile;  // XCD-aware 2 (n) x 4 (m) tile map, as in fc_lp_gemm_kernel
    if (tn % 2 == 0 && tm % 4 == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int tnx = tn / 2;
        n_tile = (x & 1) * tnx + j % tnx;
        m_tile = (x >> 1) * (tm / 4) + j / tnx;
    } else {
        n_tile = blockIdx.x % tn;
        m_tile = blockIdx.x / tn;
    }
    const int n0 = n_tile * GN, m0 = m_tile * GM;
    auto make_rs = [](const void *p, unsigned bytes) {
        const unsigned long long a = (unsigned long long)p;
        i32x4_t rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        rs[3] = 0x00020000;
        return rs;
    };
    // staging.  256 x 256, 4 rows: 32 wave-instructions of 64 elements per sub-step, 4 per wave -- waves 0-3 stage W's row w, waves 4-7 X's row
    // w - 4, the four quarters of the row each.  128 x 256, 8 rows: 48 per sub-step, 6 per wave -- wave w stages row w: W's two halves and X's four
    // quarters.  (M0 without a clobber: see fc_lp_gemm_kernel.)
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr)glds);
    // 256 x 256: a wave stages ONE operand (rsA = W for waves 0-3, X for waves 4-7); 128 x 256: both (rsA = W, rsB = X)
    const bool first_w = GN == 128 || grp == 0;
    const i32x4_t rsA = first_w ? make_rs(W, (unsigned)KE * (unsigned)N * 16u) : make_rs(X, (unsigned)KE * (unsigned)ldm * 16u);
    const unsigned voffA = (unsigned)((first_w ? n0 : m0) + lane) * 16u;
    const unsigned rowA = __builtin_amdgcn_readfirstlane((unsigned)(first_w ? N : ldm) * 16u);
    const int drow = GN == 256 ? (wave & 3) : wave;
    unsigned iss_lds = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(drow * ROW + (first_w ? 0 : GN)) * 16u);   // this wave's row (and operand) of the stage being filled
    unsigned iss_a = __builtin_amdgcn_readfirstlane((unsigned)drow * rowA);
    // (the second operand's state exists in the 128 x 256 form only)
    const i32x4_t rsB = GN == 128 ? make_rs(X, (unsigned)KE * (unsigned)ldm * 16u) : rsA;
    const unsigned voffB = (unsigned)(m0 + lane) * 16u;
    const unsigned rowB = __builtin_amdgcn_readfirstlane((unsigned)ldm * 16u);
    unsigned iss_b = __builtin_amdgcn_readfirstlane((unsigned)drow * rowB);
    int iss_stage = 0;
    auto dma = [&](const i32x4_t &rs, unsigned la, unsigned voff, unsigned so) {
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(la), "v"(voff), "s"(rs), "s"(so) : "memory");
    };
    auto issue_next = [&]() {
#pragma unroll
        for (int q = 0; q < (GN == 256 ? 4 : GN / 64); q++) dma(rsA, iss_lds + 1024u * q, voffA, iss_a + 1024u * q);
        if (!(ablate & 8)) iss_a += PR * rowA;
        if constexpr (GN == 128) {
#pragma unroll
            for (int q = 0; q < 4; q++) dma(rsB, iss_lds + (unsigned)GN * 16u + 1024u * q, voffB, iss_b + 1024u * q);
            if (!(ablate & 8)) iss_b += PR * rowB;
        }
        iss_stage++;
        iss_lds += (unsigned)STAGE * 16u;
        if (iss_stage == S) {
            iss_stage = 0;
            iss_lds -= (unsigned)(S * STAGE) * 16u;
        }
    };
    constexpr int NT16 = WN / 16, NT32 = WN / 32;   // the wave's n tiles: of 16 (bf16, v_mfma_f32_16x16x32_bf16), of 32 (fp8, v_mfma_scale_f32_32x32x64_f8f6f4)
    constexpr int NA = PREC == 1 ? NT16 : 2 * NT32, NB = 4;   // fragment registers (uint4) per k group: bf16 NT16 + 4 tiles of 16; fp8 NT32 + 2 tiles of 32, two elements each
    f32x4_t acc16[PREC == 1 ? 4 * NT16 : 1];
    f32x16 acc32[PREC == 1 ? 1 : 2 * NT32];
    if constexpr (PREC == 1) {
#pragma unroll
        for (int i = 0; i < 4 * NT16; i++) acc16[i] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
#pragma unroll
        for (int i = 0; i < 2 * NT32; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc32[i][e] = 0.0f;
    }
    uint4 fa[KGS][NA], fb[KGS][NB];
    const uint4 *rd_ptr = glds;
    int rd_stage = 0;
    auto read_frags = [&]() {
#pragma unroll
        for (int kg = 0; kg < KGS; kg++) {
            if constexpr (PREC == 1) {   // lane (g, j): k group g = lane / 16 of the four rows, row j = lane % 16 of its tile
                const uint4 *p = rd_ptr + (size_t)(4 * kg + (lane >> 4)) * ROW + (lane & 15);
#pragma unroll
                for (int t = 0; t < NT16; t++) fa[kg][t] = p[wn * WN + 16 * t];
#pragma unroll
                for (int u = 0; u < 4; u++) fb[kg][u] = p[GN + wm * 64 + 16 * u];
            } else {                     // k = 32 h + j: element rows 2 h, 2 h + 1
                const uint4 *p = rd_ptr + (size_t)(4 * kg + 2 * (lane >> 5)) * ROW + (lane & 31);
#pragma unroll
                for (int t = 0; t < NT32; t++) {
                    fa[kg][2 * t] = p[wn * WN + 32 * t];
                    fa[kg][2 * t + 1] = p[ROW + wn * WN + 32 * t];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    fb[kg][2 * u] = p[GN + wm * 64 + 32 * u];
                    fb[kg][2 * u + 1] = p[ROW + GN + wm * 64 + 32 * u];
                }
            }
        }
        rd_stage++;
        rd_ptr += STAGE;
        if (rd_stage == S) {
            rd_stage = 0;
            rd_ptr = glds;
        }
    };
    auto mfmas = [&]() {
        // raised priority for the multiplying wave pays in fp8 only (bf16 FC1, two launches side by side: 102.2 -> 99.7 us without it)
        constexpr int PRIO_BIT = PREC == 1 ? 16 : 0;   // experiments build, FR_PP_ABLATE=16: the other choice
        const bool prio = ((ablate & 16) != 0) == (PRIO_BIT != 0);
        if (prio) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kg = 0; kg < KGS; kg++) {
            if constexpr (PREC == 1) {
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int t = 0; t < NT16; t++)
                        acc16[4 * t + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[kg][t]), __builtin_bit_cast(bf16x8, fb[kg][u]), acc16[4 * t + u], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int t = 0; t < NT32; t++) {
                        const uint4 &a0 = fa[kg][2 * t], &a1 = fa[kg][2 * t + 1], &b0 = fb[kg][2 * u], &b1 = fb[kg][2 * u + 1];
                        i32x8 av, bv;
                        av[0] = (int)a0.x; av[1] = (int)a0.y; av[2] = (int)a0.z; av[3] = (int)a0.w;
                        av[4] = (int)a1.x; av[5] = (int)a1.y; av[6] = (int)a1.z; av[7] = (int)a1.w;
                        bv[0] = (int)b0.x; bv[1] = (int)b0.y; bv[2] = (int)b0.z; bv[3] = (int)b0.w;
                        bv[4] = (int)b1.x; bv[5] = (int)b1.y; bv[6] = (int)b1.z; bv[7] = (int)b1.w;
                        acc32[2 * t + u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc32[2 * t + u], 0, 0, 0, sc_a, 0, sc_b);
                    }
            }
        }
        if (prio) __builtin_amdgcn_s_setprio(0);
    };
    // sched_barrier(0): NOTHING is scheduled across -- MFMAs are register-only instructions, and without it hipcc moves most of a phase's
    // MFMAs behind the next barrier, into the other group's phase (the phases are the point of this kernel; measured: bf16 FC1 117 for 102 us)
    auto fence_barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
    };
    const int nsub = KE / PR;   // >= D + 1 (checked by the launcher)
    constexpr unsigned LPS = GN == 256 ? 4u : (unsigned)(GN / 64 + 4);   // DMAs per wave per sub-step
    constexpr unsigned INFL = LPS * (D - 1);   // DMAs of the D - 1 younger sub-steps may stay in flight
    constexpr int IMM_STEADY = (int)((INFL & 0xF) | ((INFL >> 4) << 14) | 0x0070u);   // ... and lgkmcnt(0): the fragment reads are done
    constexpr int IMM_FIRST = (int)((INFL & 0xF) | ((INFL >> 4) << 14) | 0x0F70u);
#pragma unroll
    for (int i = 0; i < D; i++) issue_next();
    __builtin_amdgcn_s_waitcnt(IMM_FIRST);   // sub-step 0 has landed
    fence_barrier();
    const int main_end = nsub - D;           // s < main_end: M(s) issues sub-step s + D
    auto memory_phase = [&](int s) {
        if (ablate & 64) __builtin_amdgcn_s_setprio(2);   // experiment: the FETCHING wave above the multiplying one
        if ((ablate & 32) && s < main_end) issue_next();
        if (!(ablate & 2) || s == 0) read_frags();
        if (s < main_end) {
            if (!(ablate & 33)) issue_next();
            __builtin_amdgcn_s_waitcnt(IMM_STEADY);
        } else {
            __builtin_amdgcn_s_waitcnt(0x0070);   // the pipeline's tail: vmcnt(0), lgkmcnt(0)
        }
        if (ablate & 64) __builtin_amdgcn_s_setprio(0);
    };
    if (grp == 1) fence_barrier();           // waves 4-7: one phase behind
    for (int s = 0; s < nsub; s++) {
        memory_phase(s);
        fence_barrier();
        if (!(ablate & 4)) mfmas();
        fence_barrier();
    }
    if (grp == 0) fence_barrier();
    // epilogue: ONE rounding per output (as fc_lp_gemm_kernel)
    if constexpr (PREC == 1) {   // 16 x 16 tiles: lane holds m = lane % 16 and n = 4 (lane / 16) + c
#pragma unroll
        for (int t = 0; t < NT16; t++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const f32x4_t &c = acc16[4 * t + u];
                const int m = m0 + wm * 64 + 16 * u + (lane & 15);
                const int n = n0 + wn * WN + 16 * t + 4 * (lane >> 4);
                uint2 hv;
                hv.x = pack_bf16x2(c[0], c[1]);
                hv.y = pack_bf16x2(c[2], c[3]);
                reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
            }
    } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int t = 0; t < NT32; t++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const f32x16 &c = acc32[2 * t + u];
                const int m = m0 + wm * 64 + 32 * u + r;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int n = n0 + wn * WN + 32 * t + 8 * i + 4 * h;
                    reinterpret_cast<uint32_t *>(Y)[((size_t)(n >> 4) * ldm + m) * 4 + ((n & 15) >> 2)] =
                        pack_fp8x4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], oscale);
                }
            }
    }
}

template <int PREC, int D, int NT>
__global__ void __launch_bounds__(512) fc_pp_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                         int N, int ldm, int sc_a, int sc_b, float oscale, int ablate) {
    extern __shared__ uint4 glds[];
    pp_gemm_body<PREC, D, NT, 256, 4>(glds, W, X, Y, KE, N, ldm, sc_a, sc_b, oscale, ablate);
}

// the same on 128 (n) x 256 (m) tiles (64 x 64 wave tiles) in sub-steps of 8 element rows (two k groups per phase: 32 / 8 MFMAs per wave again;
// D = 2: 144 KiB): layers whose 256 x 256 tiles would not cover their share of the chip -- Model-C FC1 for a lone worker
template <int PREC, int D>
__global__ void __launch_bounds__(512) fc_pp_gemm_n128_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                              int N, int ldm, int sc_a, int sc_b, float oscale, int ablate) {
    extern __shared__ uint4 glds[];
    pp_gemm_body<PREC, D, 0, 128, 8>(glds, W, X, Y, KE, N, ldm, sc_a, sc_b, oscale, ablate);
}

// ===================================================================================================
// fc_gemm_gather_kernel<PREC>: north_star's "fused concat + first FC" for the model whose record does not fit a CU (Model-C, batch 4096,
// bf16 / fp8): the FC1 GEMM of batch L - 1 (fc_lp_gemm_kernel's 128 x 256 tile body, 8 consumer waves) and the GATHER of batch L (4
// producer waves, one per SIMD) in ONE workgroup, so the two share every CU for the length of the kernel by construction -- the
// separately launched gather met another stream's gather, not a GEMM (profiles/archive/r04_experiments.md section 1).  A 128 x 256 tile cannot
// gather its own operand (16 n tiles share an item tile: 16 x the row fetches), hence the stage pipeline's shift by one batch: the
// producers write the q8 / q16 operand image of the NEXT launch's FC1 to HBM.
//   producers: the batch is cut into half-tiles of 16 items x 64 record words; producer wave p of workgroup b owns half-tiles
//   4 b + p, + 4 grid, ...; lanes run along the record words (a table row is read by adjacent lanes), 16 index loads and 16 row loads
//   per lane and half-tile stay in flight in registers; a half-tile is converted (fp32 -> bf16 / e4m3) into a wave-private LDS patch
//   laid out as operand elements and written out with lanes along items (256 contiguous bytes per 16 lanes).  No cross-wave hand-off:
//   the only synchronisation is the wave's own instruction order.
//   coupling: s_barrier counts every wave of the workgroup, so the producers pass one RAW s_barrier (no vmcnt / lgkmcnt wait) per K step
//   of the consumers; their n_ht + 2 events (event e = { convert half-tile e - 2, request the rows of e - 1, request the indices of e })
//   are spread evenly over the K steps, so an event's loads have ~ n_steps / n_events steps (several us) to land.
// ===================================================================================================
struct FrGatherJob {
    const FrWordDesc *words;
    const int32_t *idx;
    const float *dense;
    void *out;        // q8 (bf16) / q16 (e4m3) operand image of the batch, leading dimension ldm
    int *err_flag;
    int n_words, idx_stride, batch, ldm, K;
    float scale;      // fp8: 2^e_x
};
constexpr int FR_GG_LD = 17;                          // patch row stride in 16-byte elements (16 items + 1: conflict-free both ways)
constexpr int FR_GG_PATCH = 32 * FR_GG_LD;            // 16-byte elements of a wave's patch (32 q8 rows; fp8 uses 16 q16 rows of it)

template <int PREC>
__device__ __forceinline__ void gg_producer(const FrGatherJob &g, uint4 *patch, int pw, int nsteps) {
    int lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    const int n_wb = (g.n_words + 63) >> 6, n_mi = g.ldm >> 4;
    const int total = n_wb * n_mi, stride = (int)gridDim.x * 4, first = (int)blockIdx.x * 4 + pw;
    const int n_ht = (g.words && first < total) ? (total - first + stride - 1) / stride : 0;
    const int n_ev = n_ht ? n_ht + 2 : 0;
    const __amdgpu_buffer_rsrc_t rs_idx = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(g.idx), 0, (unsigned)g.batch * (unsigned)g.idx_stride * 4u, 0x00020000);
    uint32_t idxr[16];
    uint4 rows[16];
    // Word descriptors: fetched ONE EVENT AHEAD (raw, q0 / q1) and decoded when the half-tile's index event runs -- fetched inside the
    // event, the dependent chain descriptor -> index address put an s_waitcnt vmcnt(0) behind the 16 row loads the same event had just
    // issued (a wave's loads return in order), and every such wait held all 12 waves at the next s_barrier: the fused kernel took as
    // long as the GEMM and the gather one after the other (profiles/archive/r04_experiments.md section 1.6).
    uint4 q0 = make_uint4(0u, 0u, 0u, 0u), q1 = q0;
    uint64_t d_base = 0;   // decoded descriptor of the half-tile whose rows are requested next
    uint32_t d_stride = 0, d_rows = 0;
    bool d_dense = false;
    unsigned bad = 0u;
    auto ht_of = [&](int h) { return first + h * stride; };
    auto D_ld = [&](int h) {   // raw descriptor of half-tile h's word of this lane
        const int ht = ht_of(h), wb = ht / n_mi;
        int w = (wb << 6) + lane;
        w = w < g.n_words ? w : g.n_words - 1;   // a lane past the record repeats the last word (never stored)
        q0 = reinterpret_cast<const uint4 *>(g.words)[2 * w];
        q1 = reinterpret_cast<const uint4 *>(g.words)[2 * w + 1];
    };
    // the fetched descriptor, pinned into registers at the START of an event: the wait for it (everything older has long landed) must not sink
    // below the row loads the event issues
    uint32_t n_lo = 0, n_hi = 0, n_stride = 0, n_col = 0, n_rows = 0;
    auto D_pin = [&]() {
        n_lo = q0.x, n_hi = q0.y, n_stride = q0.z, n_col = q0.w, n_rows = q1.x;
        asm volatile("" : "+v"(n_lo), "+v"(n_hi), "+v"(n_stride), "+v"(n_col), "+v"(n_rows));
    };
    auto I_ev = [&](int h) {
        const int ht = ht_of(h), wb = ht / n_mi, m0 = (ht - wb * n_mi) << 4;
        d_dense = (n_col & FR_DESC_DENSE) != 0;
        d_base = (d_dense ? (uint64_t)reinterpret_cast<uintptr_t>(g.dense) : 0ull) + (((uint64_t)n_hi << 32) | n_lo);
        d_stride = n_stride, d_rows = n_rows;
        const unsigned icol = d_dense ? 0u : n_col * 4u;
#pragma unroll
        for (int i = 0; i < 16; i++)   // items past the batch: out of the resource's bounds, 0 comes back
            idxr[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs_idx, (unsigned)(m0 + i) * (unsigned)g.idx_stride * 4u + icol, 0, 0);
    };
    auto R_ev = [&](int h) {
        const int ht = ht_of(h), wb = ht / n_mi, m0 = (ht - wb * n_mi) << 4;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const unsigned m = (unsigned)(m0 + i);
            uint32_t r = idxr[i];
            const bool oob = !d_dense & (r >= d_rows);   // reference: silent out-of-bounds read (embedding_47_krnl.cpp:927-933); here reported
            bad |= oob ? 1u : 0u;
            r = oob ? 0u : r;
            r = d_dense ? (m < (unsigned)g.batch ? m : 0u) : r;
            typedef const u32x4_t __attribute__((address_space(1))) * gptr_t;
            const u32x4_t q = *(gptr_t)(d_base + (uint64_t)r * d_stride);
            rows[i] = make_uint4(q.x, q.y, q.z, q.w);
        }
    };
    auto W_ev = [&](int h) {
        const int ht = ht_of(h), wb = ht / n_mi, m0 = (ht - wb * n_mi) << 4, w0 = wb << 6;
        const uint32_t real = 0u - (uint32_t)(w0 + lane < g.n_words);   // words past the record are zeros (the fp8 image's pad up to 64 k)
        if constexpr (PREC == 1) {
            uint2 *p2 = reinterpret_cast<uint2 *>(patch);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t in = (0u - (uint32_t)(m0 + i < g.batch)) & real;
                uint2 hv;
                hv.x = pack_bf16x2(__uint_as_float(rows[i].x), __uint_as_float(rows[i].y)) & in;
                hv.y = pack_bf16x2(__uint_as_float(rows[i].z), __uint_as_float(rows[i].w)) & in;
                p2[((lane >> 1) * FR_GG_LD + i) * 2 + (lane & 1)] = hv;   // word = half (lane & 1) of q8 row lane / 2
            }
        } else {
            uint32_t *p1 = reinterpret_cast<uint32_t *>(patch);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const uint32_t in = (0u - (uint32_t)(m0 + i < g.batch)) & real;
                p1[((lane >> 2) * FR_GG_LD + i) * 4 + (lane & 3)] = pack_fp8_word(rows[i], g.scale) & in;   // word = dword (lane & 3) of q16 row lane / 4
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own stores have reached LDS (in-order per wave): read the patch back along items
        constexpr int NR = PREC == 1 ? 8 : 4;              // read passes: 4 element rows x 16 items per wave-instruction
        const int row0 = PREC == 1 ? (w0 >> 1) : (w0 >> 2), n_rows = PREC == 1 ? (g.n_words >> 1) : ((g.K + 63) / 64 * 4);
        uint4 *img = reinterpret_cast<uint4 *>(g.out);
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int j = (lane >> 4) + 4 * r, it = lane & 15;
            const uint4 v = patch[j * FR_GG_LD + it];
            if (row0 + j < n_rows) img[(size_t)(row0 + j) * g.ldm + m0 + it] = v;
        }
        asm volatile("" ::: "memory");
    };
    int e = 0;
    if (n_ht) D_ld(0);
    for (int s = 0; s < nsteps; s++) {
        while (e < n_ev && (long)e * nsteps <= (long)s * n_ev) {   // event e at step floor(e nsteps / n_ev)
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) HERE, where every outstanding operation is an event old: left to hipcc, counted waits land between the row loads
            asm volatile("" ::: "memory");
            if (e < n_ht) D_pin();
            if (e >= 2) W_ev(e - 2);
            if (e >= 1 && e - 1 < n_ht) R_ev(e - 1);
            if (e < n_ht) I_ev(e);
            if (e + 1 < n_ht) D_ld(e + 1);
            e++;
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();   // raw: no vmcnt / lgkmcnt wait -- the loads of the last events stay in flight across it
        asm volatile("" ::: "memory");
    }
    while (e < n_ev) {   // (fewer K steps than events: finish behind the loop)
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) HERE, where every outstanding operation is an event old: left to hipcc, counted waits land between the row loads
            asm volatile("" ::: "memory");
            if (e < n_ht) D_pin();
        if (e >= 2) W_ev(e - 2);
        if (e >= 1 && e - 1 < n_ht) R_ev(e - 1);
        if (e < n_ht) I_ev(e);
        if (e + 1 < n_ht) D_ld(e + 1);
        e++;
    }
    if (bad) atomicOr_system(g.err_flag, 1);
}

template <int PREC>
__global__ void __launch_bounds__(768) fc_gemm_gather_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE, int N, int ldm, int sc_a,
                                                              int sc_b, float oscale, const FrGatherJob g) {
    extern __shared__ uint4 glds[];
    constexpr int GEMM_LDS = FR_GSTAGES * FR_GR * (FR_GN + 256);   // 16-byte elements of the consumers' two K steps
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wave >= 8) {
        gg_producer<PREC>(g, glds + GEMM_LDS + (wave - 8) * FR_GG_PATCH, wave - 8, KE / FR_GR);
        return;
    }
    lp_gemm_body<PREC, 2, FR_GN, FR_GSTAGES, FR_GR>(glds, W, X, Y, KE, N, ldm, sc_a, sc_b, oscale);
}

// ===================================================================================================
// fc_gemm_pipe_kernel<PREC, NS>: the software-pipelined form of the bf16 / fp8 GEMM for the 128 (n) x 256 (m) block tile
// (Model-C FC1 at batch 4096: 256 tiles = one per CU).  Same operand images (q8 / q16 elements, global -> LDS DMA, conflict-free
// ds_read_b128 fragment reads), same 2 x 4 wave grid with 64 x 64 wave tiles, but:
//   * a K SUB-step is 4 element rows (32 k in bf16, 64 k in fp8 = 256 MFMA cycles per wave) and NS of them live in LDS (NS = 5:
//     120 KiB); the DMA loads of sub-step s + NS - 1 are issued in iteration s, and iteration s waits -- with a COUNTED vmcnt that
//     leaves the youngest NS - 3 sub-steps in flight -- only for sub-step s + 1: NS - 2 sub-steps (~0.75 us) of latency tolerance
//     instead of one K step, and no vmcnt(0) drain in the loop;
//   * the fragments of sub-step s + 1 are read into a second register set WHILE the MFMAs of sub-step s run, so the MFMAs after a
//     barrier start from registers (with one barrier per step and the reads behind it, both waves of a SIMD used to wait for LDS at
//     the same moment);
//   * raw s_barrier (no vmcnt(0) fence), s_setprio around the MFMA cluster;
//   * bf16 uses v_mfma_f32_16x16x32_bf16 (16 per sub-step per wave): same LDS bytes and cycles per FLOP as the 32x32x16 form, but the
//     chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS give-back item 7); fp8 keeps the block-scaled 32x32x64.
// Sums are over whole K in k order per output, as in fc_lp_gemm_kernel; results differ from it only through the MFMA's own internal
// summation inside one instruction (16x16x32 vs 32x32x16 group k differently) -- same tolerance class, tested against the oracle.
// ===================================================================================================
constexpr int FR_PR = 4;  // element rows per row group; a sub-step is G groups

template <int PREC>
struct FrPipeFrag {  // one row group's operands of one wave: bf16: 4 + 4 fragments of 16 rows x 32 k; fp8: 2 + 2 fragments of 32 rows x 64 k (two elements each)
    uint4 a[4], b[4];
};

template <int PREC>
__device__ __forceinline__ void pipe_read_frags(FrPipeFrag<PREC> &f, const uint4 *grp, int row_elems, int wn, int wm, int lane) {
    if constexpr (PREC == 1) {
        const uint4 *p = grp + (size_t)(lane >> 4) * row_elems + (lane & 15);  // k group = lane / 16, row = lane % 16
#pragma unroll
        for (int t = 0; t < 4; t++) f.a[t] = p[wn * 64 + 16 * t];
#pragma unroll
        for (int u = 0; u < 4; u++) f.b[u] = p[FR_GN + wm * 64 + 16 * u];
    } else {
        const uint4 *p = grp + (size_t)(2 * (lane >> 5)) * row_elems + (lane & 31);  // k = 32 h + j: element rows 2h, 2h + 1
#pragma unroll
        for (int t = 0; t < 2; t++) {
            f.a[2 * t] = p[wn * 64 + 32 * t];
            f.a[2 * t + 1] = p[row_elems + wn * 64 + 32 * t];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            f.b[2 * u] = p[FR_GN + wm * 64 + 32 * u];
            f.b[2 * u + 1] = p[row_elems + FR_GN + wm * 64 + 32 * u];
        }
    }
}

template <int PREC, int NS, int G>
__global__ void __launch_bounds__(512) fc_gemm_pipe_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                           int N, int ldm, int sc_a, int sc_b, float oscale, int ablate) {
#ifndef FR_EXPERIMENTS
    ablate = 0x110;  // product build: staggered issue order, wave priority 1, NO timing ablation can reach the kernel (the branches below fold away)
#endif
    extern __shared__ uint4 glds[];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int GM = 256, ROW = FR_GN + GM, STAGE = G * FR_PR * ROW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tn = N / FR_GN, tm = ldm / GM;
    int n_tile, m_tile;  // XCD-aware 2 (n) x 4 (m) tile map, as in fc_lp_gemm_kernel
    if (tn % 2 == 0 && tm % 4 == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        const int tnx = tn / 2;
        n_tile = (x & 1) * tnx + j % tnx;
        m_tile = (x >> 1) * (tm / 4) + j / tnx;
    } else {
        n_tile = blockIdx.x % tn;
        m_tile = blockIdx.x / tn;
    }
    const int n0 = n_tile * FR_GN, m0 = m_tile * GM;
    auto make_rs = [](const void *p, unsigned bytes) {
        const unsigned long long a = (unsigned long long)p;
        i32x4_t rs;
        rs[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
        rs[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
        rs[2] = __builtin_amdgcn_readfirstlane((int)bytes);
        rs[3] = 0x00020000;
        return rs;
    };
    const i32x4_t rsW = make_rs(W, (unsigned)KE * (unsigned)N * 16u), rsX = make_rs(X, (unsigned)KE * (unsigned)ldm * 16u);
    // global -> LDS DMA (see fc_lp_gemm_kernel for why this is inline asm and why M0 carries no clobber).  LDS addresses are kept as
    // plain wave-uniform integers and advanced incrementally: through pointers every DMA cost ~15 scalar instructions (stage modulo by
    // multiplication, address-space casts with null checks), and the issue of a sub-step's three DMAs is what a wave spends longest on
    // besides its MFMAs.
    auto dma = [&](const i32x4_t &rs, unsigned lds_addr, unsigned voff, unsigned soff) {
        const unsigned la = __builtin_amdgcn_readfirstlane(lds_addr), so = __builtin_amdgcn_readfirstlane(soff);  // wave-uniform by construction
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(la), "v"(voff), "s"(rs), "s"(so) : "memory");
    };
    // staging of one sub-step: 4 rows x (128 + 256) elements = 24 wave-instructions of 64 elements, 3 per wave: wave w takes row w / 2,
    // W's half w % 2 and X's quarters 2 (w % 2), 2 (w % 2) + 1.  Sub-steps are issued strictly in order, so the state is a running one.
    const int drow = wave >> 1, dpart = wave & 1;
    const unsigned vW = (unsigned)(n0 + 64 * dpart + lane) * 16u, vX = (unsigned)(m0 + 128 * dpart + lane) * 16u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr)glds);
    unsigned iss_lds = lds0 + (unsigned)(drow * ROW) * 16u;                               // this wave's row of the stage being filled
    unsigned iss_w = (unsigned)drow * (unsigned)N * 16u, iss_x = (unsigned)drow * (unsigned)ldm * 16u;  // SGPR offsets of that row in W / X
    // ablate (timing experiments only, results are wrong): 2 = every sub-step re-reads sub-step 0's rows (operands stay in L2 / L1)
    const unsigned grp_w = (unsigned)FR_PR * (unsigned)N * 16u, grp_x = (unsigned)FR_PR * (unsigned)ldm * 16u;  // one row group further in W / X
    const unsigned step_w = (ablate & 15) == 2 ? 0u : G * grp_w, step_x = (ablate & 15) == 2 ? 0u : G * grp_x;
    int iss_stage = 0, issued = 0;
    auto issue_next = [&]() {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const unsigned l = iss_lds + (unsigned)(g * FR_PR * ROW) * 16u;
            dma(rsW, l + (unsigned)(64 * dpart) * 16u, vW, iss_w + g * grp_w);
            dma(rsX, l + (unsigned)(FR_GN + 128 * dpart) * 16u, vX, iss_x + g * grp_x);
            dma(rsX, l + (unsigned)(FR_GN + 128 * dpart + 64) * 16u, vX + 64u * 16u, iss_x + g * grp_x);
        }
        iss_w += step_w;
        iss_x += step_x;
        issued++;
        iss_stage++;
        iss_lds += (unsigned)STAGE * 16u;
        if (iss_stage == NS) {
            iss_stage = 0;
            iss_lds -= (unsigned)(NS * STAGE) * 16u;
        }
    };
    constexpr int NACC = PREC == 1 ? 16 : 4;
    f32x4_t acc16[PREC == 1 ? 16 : 1];
    f32x16 acc32[PREC == 1 ? 1 : 4];
    if constexpr (PREC == 1) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc16[i] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
    } else {
#pragma unroll
        for (int i = 0; i < NACC; i++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc32[i][e] = 0.0f;
    }
    auto mfmas = [&](const FrPipeFrag<PREC> &f) {
        if (ablate & 256) __builtin_amdgcn_s_setprio(1);
        if constexpr (PREC == 1) {
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc16[4 * t + u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.a[t]), __builtin_bit_cast(bf16x8, f.b[u]), acc16[4 * t + u], 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint4 &a0 = f.a[2 * t], &a1 = f.a[2 * t + 1], &b0 = f.b[2 * u], &b1 = f.b[2 * u + 1];
                    i32x8 av, bv;
                    av[0] = (int)a0.x; av[1] = (int)a0.y; av[2] = (int)a0.z; av[3] = (int)a0.w;
                    av[4] = (int)a1.x; av[5] = (int)a1.y; av[6] = (int)a1.z; av[7] = (int)a1.w;
                    bv[0] = (int)b0.x; bv[1] = (int)b0.y; bv[2] = (int)b0.z; bv[3] = (int)b0.w;
                    bv[4] = (int)b1.x; bv[5] = (int)b1.y; bv[6] = (int)b1.z; bv[7] = (int)b1.w;
                    acc32[2 * t + u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc32[2 * t + u], 0, 0, 0, sc_a, 0, sc_b);
                }
        }
        if (ablate & 256) __builtin_amdgcn_s_setprio(0);
    };
    const int nsub = KE / (G * FR_PR);
    constexpr unsigned WAIT_LOOP = 3 * G * (NS - 3), WAIT_FIRST = 3 * G * (NS - 2);
    constexpr int IMM_LOOP = (int)((WAIT_LOOP & 0xF) | ((WAIT_LOOP >> 4) << 14) | 0x0F70u), IMM_FIRST = (int)((WAIT_FIRST & 0xF) | ((WAIT_FIRST >> 4) << 14) | 0x0F70u);
    // One iteration = one ROW GROUP (4 element rows): it is multiplied from `cur` (already in registers) while `nxt` receives the next
    // group.  Synchronisation is per SUB-STEP (G groups): its first group (HEAD) waits for the NEXT sub-step's loads and passes the
    // barrier, and the sub-step's DMAs are issued either there or after its last group (TAIL).  MAIN: the steady state (a sub-step is
    // issued, NS - 2 are in flight), branch-free around the waits so that hipcc counts its own lgkmcnt waits; otherwise the pipeline's
    // tail (plain vmcnt(0), nothing may be left to issue).  The fragment prefetch is unconditional: past the last group it reads a stage
    // nobody uses.
    // STAGGER (order 1, default): the two waves of a SIMD (w and w + 4) would otherwise do the same thing at the same time -- issue
    // DMAs (expensive beside other memory instructions), read fragments, and only then both want the matrix pipe.  Waves 0-3 issue and
    // read first and multiply afterwards, waves 4-7 multiply the moment the barrier opens and read / issue afterwards: one partner's
    // memory instructions hide behind the other's MFMAs (Model-C FC1 bf16: 57.5 -> 52.9 us; profiles/archive/r02_gemm_experiments.md).
    const int order = (ablate >> 4) & 15;  // experiment knob FR_GEMM_ORDER: 0 = every wave multiplies first, 1 = stagger, 2 = every wave issues + reads first
    const bool early = order == 2 || (order == 1 && wave < 4);
    int rd_grp = 1;                      // group to prefetch next, counted inside the ring of NS * G groups
    const uint4 *rd_ptr = glds + FR_PR * ROW;
    auto step = [&](const FrPipeFrag<PREC> &cur, FrPipeFrag<PREC> &nxt, auto main_tag, auto head_tag, auto tail_tag) {
        constexpr bool MAIN = decltype(main_tag)::value, HEAD = decltype(head_tag)::value, TAIL = decltype(tail_tag)::value;
        if constexpr (HEAD) {
            if constexpr (MAIN) __builtin_amdgcn_s_waitcnt(IMM_LOOP);  // this wave's loads of the NEXT sub-step have landed ...
            else __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... everybody's have, and everybody is past the fragment reads of the previous sub-step
            asm volatile("" ::: "memory");
        }
        const bool no_dma = (ablate & 15) == 1;  // timing experiment: no DMA inside the loop
        if (order == 3 || (order == 4 && wave < 4)) {  // issue, multiply, read
            if (HEAD && !no_dma && (MAIN || issued < nsub)) issue_next();
            mfmas(cur);
            pipe_read_frags<PREC>(nxt, rd_ptr, ROW, wn, wm, lane);
        } else if (early) {
            if (HEAD && !no_dma && (MAIN || issued < nsub)) issue_next();  // overwrites the stage the previous sub-step lived in
            pipe_read_frags<PREC>(nxt, rd_ptr, ROW, wn, wm, lane);
            mfmas(cur);
        } else {
            mfmas(cur);
            pipe_read_frags<PREC>(nxt, rd_ptr, ROW, wn, wm, lane);
            if (TAIL && !no_dma && (MAIN || issued < nsub)) issue_next();
        }
        rd_grp++;
        rd_ptr += FR_PR * ROW;
        if (rd_grp == NS * G) {
            rd_grp = 0;
            rd_ptr = glds;
        }
    };
#pragma unroll
    for (int i = 0; i < NS - 1; i++)
        if (i < nsub) issue_next();
    FrPipeFrag<PREC> fa, fb;
    if (nsub >= NS - 1) __builtin_amdgcn_s_waitcnt(IMM_FIRST);  // sub-step 0 has landed
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pipe_read_frags<PREC>(fa, glds, ROW, wn, wm, lane);
    using T = std::true_type;
    using F = std::false_type;
    const int main_end = nsub - (NS - 1);  // sub < main_end: an issue happens and the counted wait is valid
    int sub = 0;
    if constexpr (G == 1) {  // two sub-steps per trip (the register sets alternate)
        for (; sub + 1 < main_end; sub += 2) {
            step(fa, fb, T{}, T{}, T{});
            step(fb, fa, T{}, T{}, T{});
        }
        for (; sub < nsub; sub += 2) {  // nsub is even (checked by the launcher)
            step(fa, fb, F{}, T{}, T{});
            step(fb, fa, F{}, T{}, T{});
        }
    } else {  // G == 2: one sub-step per trip
        for (; sub < main_end; sub++) {
            step(fa, fb, T{}, T{}, F{});
            step(fb, fa, T{}, F{}, T{});
        }
        for (; sub < nsub; sub++) {
            step(fa, fb, F{}, T{}, F{});
            step(fb, fa, F{}, F{}, T{});
        }
    }
    // epilogue: ONE rounding per output
    if constexpr (PREC == 1) {  // 16 x 16 tiles: lane holds m = lane % 16 and n = 4 (lane / 16) + c
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const f32x4_t &c = acc16[4 * t + u];
                const int m = m0 + wm * 64 + 16 * u + (lane & 15);
                const int n = n0 + wn * 64 + 16 * t + 4 * (lane >> 4);
                uint2 hv;
                hv.x = pack_bf16x2(c[0], c[1]);
                hv.y = pack_bf16x2(c[2], c[3]);
                reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
            }
    } else {
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const f32x16 &c = acc32[2 * t + u];
                const int m = m0 + wm * 64 + 32 * u + r;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int n = n0 + wn * 64 + 32 * t + 8 * i + 4 * h;
                    reinterpret_cast<uint32_t *>(Y)[((size_t)(n >> 4) * ldm + m) * 4 + ((n & 15) >> 2)] =
                        pack_fp8x4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], oscale);
                }
            }
    }
}

// ===================================================================================================
// fc_splitk_gemm_kernel<PREC, NT, MT, PD>: the narrow layers of a large batch (Model-C FC2 1024 -> 512 and FC3 512 -> 256 at batch
// 4096: 4.3 + 1.1 GFLOP over 8 + 4 MB of activations).  Such a layer is a stream of operand bytes from L2 with a short K; an LDS-tiled
// block (fc_lp_gemm_kernel<., 1, 64, .>: 32 x 32 wave tiles) reads a 1 KiB fragment pair from LDS per MFMA -- twice what LDS
// delivers -- and pays a barrier every 64 k.  Here:
//   * a workgroup is 4 waves that SPLIT K: wave w takes element rows [w KE / 4, (w + 1) KE / 4) of the whole 32 NT (n) x 32 MT (m)
//     block tile, so every operand byte of the tile is loaded by exactly one wave, once;
//   * fragments go L2 -> registers directly in MFMA layout (the q8 / q16 images ARE fragment-major: lane (h, r) of k-step j reads
//     element row 2 j + h (bf16) / 4 j + 2 h, + 1 (fp8), column r of its 32-wide tile: one 16-byte buffer load, 512 contiguous bytes
//     per half wave) through a ring of PD k-steps per wave: no LDS staging, no barrier in the K loop, (NT + MT) KiB in flight per
//     k-step per wave;
//   * the four partial tiles are summed through LDS in wave (= k) order, each wave finishing NT MT / 4 of the tiles: one rounding per
//     output, as in the other GEMM kernels; fp32 sums of four in-order partial sums instead of one in-order sum (same tolerance class).
// EXPERIMENTS build only (see lp_gemm_mu: not faster end to end).  One workgroup per CU (256 tiles for FC2 as 128 x 64, for FC3 as 64 x 64), XCD-aware tile map: the workgroups of an XCD share its
// L2's copy of the weights and own a contiguous range of activation columns.
// ===================================================================================================
#ifdef FR_EXPERIMENTS
template <int PREC, int NT, int MT, int PD>
__global__ void __launch_bounds__(256) fc_splitk_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                             int N, int ldm, int sc_a, int sc_b, float oscale) {
    static_assert(PREC == 1 || PREC == 2, "bf16 / fp8 only: the fp32 chain keeps whole-K in-order sums");
    static_assert((NT * MT) % 4 == 0, "each wave finishes NT MT / 4 tiles");
    extern __shared__ float sk_red[];             // [tile][3 foreign waves][16 registers][64 lanes]
    constexpr int RPS = PREC == 2 ? 4 : 2;        // element rows per k-step (16 k of bf16 x 2 halves; 64 k of e4m3 = 2 halves x 2 rows)
    constexpr int FR = PREC == 2 ? 2 : 1;         // 16-byte loads per fragment
    constexpr int T = NT * MT;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = N / (32 * NT), tm = ldm / (32 * MT);
    int n_tile, m_tile;
    if (tm % 8 == 0) {  // workgroup b runs on XCD b % 8: XCD x owns the activation columns of m tiles [x tm / 8, (x + 1) tm / 8) and walks all n tiles over them
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        n_tile = j % tn;
        m_tile = x * (tm / 8) + j / tn;
    } else {
        n_tile = blockIdx.x % tn;
        m_tile = blockIdx.x / tn;
    }
    const int n0 = n_tile * 32 * NT, m0 = m_tile * 32 * MT;
    const int rows = KE / 4, k0 = wave * rows, nk = rows / RPS;   // this wave's element rows and k-steps
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(W), 0, (unsigned)KE * (unsigned)N * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(X), 0, (unsigned)KE * (unsigned)ldm * 16u, 0x00020000);
    const unsigned hrow = PREC == 2 ? 2u * h : (unsigned)h;
    const unsigned vW = (hrow * (unsigned)N + (unsigned)(n0 + r)) * 16u, vX = (hrow * (unsigned)ldm + (unsigned)(m0 + r)) * 16u;
    const unsigned rowW = (unsigned)N * 16u, rowX = (unsigned)ldm * 16u;   // byte step of one element row
    uint4 ra[PD][NT][FR], rb[PD][MT][FR];
    auto load_step = [&](int j, int slot) {   // k-step j of this wave -> ring slot
        const unsigned sW = (unsigned)(k0 + RPS * j) * rowW, sX = (unsigned)(k0 + RPS * j) * rowX;
#pragma unroll
        for (int f = 0; f < FR; f++) {
#pragma unroll
            for (int t = 0; t < NT; t++) ra[slot][t][f] = bload4u(rsW, vW + 512u * t, sW + f * rowW);
#pragma unroll
            for (int u = 0; u < MT; u++) rb[slot][u][f] = bload4u(rsX, vX + 512u * u, sX + f * rowX);
        }
    };
    f32x16 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int u = 0; u < MT; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < PD; i++)
        if (i < nk) load_step(i, i);
    for (int jb = 0; jb < nk; jb += PD) {
#pragma unroll
        for (int i = 0; i < PD; i++) {
            if (jb + i < nk) {
#pragma unroll
                for (int u = 0; u < MT; u++)
#pragma unroll
                    for (int t = 0; t < NT; t++) {
                        if constexpr (PREC == 1) {
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[i][t][0]), __builtin_bit_cast(bf16x8, rb[i][u][0]), acc[t][u], 0, 0, 0);
                        } else {
                            i32x8 a8, b8;
                            a8[0] = (int)ra[i][t][0].x; a8[1] = (int)ra[i][t][0].y; a8[2] = (int)ra[i][t][0].z; a8[3] = (int)ra[i][t][0].w;
                            a8[4] = (int)ra[i][t][FR - 1].x; a8[5] = (int)ra[i][t][FR - 1].y; a8[6] = (int)ra[i][t][FR - 1].z; a8[7] = (int)ra[i][t][FR - 1].w;
                            b8[0] = (int)rb[i][u][0].x; b8[1] = (int)rb[i][u][0].y; b8[2] = (int)rb[i][u][0].z; b8[3] = (int)rb[i][u][0].w;
                            b8[4] = (int)rb[i][u][FR - 1].x; b8[5] = (int)rb[i][u][FR - 1].y; b8[6] = (int)rb[i][u][FR - 1].z; b8[7] = (int)rb[i][u][FR - 1].w;
                            acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t][u], 0, 0, 0, sc_a, 0, sc_b);
                        }
                    }
                if (jb + i + PD < nk) load_step(jb + i + PD, i);
            }
        }
    }
    // ---- the four K partials -> one tile: tile q is finished by wave q % 4; the other three park their partial in LDS ----
#pragma unroll
    for (int q = 0; q < T; q++) {
        const int o = q % 4;
        if (wave != o) {
            const int p = wave < o ? wave : wave - 1;
            float *dst = sk_red + ((size_t)(q * 3 + p) * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < 16; i++) dst[i * 64] = acc[q / MT][q % MT][i];
        }
    }
    __syncthreads();
    auto finish = [&](auto wv) {
        constexpr int w = decltype(wv)::value;
#pragma unroll
        for (int q = w; q < T; q += 4) {
            const int t = q / MT, u = q % MT;
            float c[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                float sum = 0.0f;   // wave (= k) order: ((p0 + p1) + p2) + p3
#pragma unroll
                for (int src = 0; src < 4; src++) {
                    const int p = src < w ? src : src - 1;
                    const float v = src == w ? acc[t][u][i] : sk_red[((size_t)(q * 3 + p) * 16 + i) * 64 + lane];
                    sum = src == 0 ? v : sum + v;
                }
                c[i] = sum;
            }
            const int m = m0 + 32 * u + r;
#pragma unroll
            for (int i = 0; i < 4; i++) {   // registers 4 i .. 4 i + 3 of a tile are 4 consecutive n
                const int n = n0 + 32 * t + 8 * i + 4 * h;
                if constexpr (PREC == 1) {
                    uint2 hv;
                    hv.x = pack_bf16x2(c[4 * i + 0], c[4 * i + 1]);
                    hv.y = pack_bf16x2(c[4 * i + 2], c[4 * i + 3]);
                    reinterpret_cast<uint2 *>(Y)[((size_t)(n >> 3) * ldm + m) * 2 + ((n & 7) >> 2)] = hv;
                } else {
                    reinterpret_cast<uint32_t *>(Y)[((size_t)(n >> 4) * ldm + m) * 4 + ((n & 15) >> 2)] = pack_fp8x4(c[4 * i + 0], c[4 * i + 1], c[4 * i + 2], c[4 * i + 3], oscale);
                }
            }
        }
    };
    switch (wave) {
        case 0: finish(std::integral_constant<int, 0>{}); break;
        case 1: finish(std::integral_constant<int, 1>{}); break;
        case 2: finish(std::integral_constant<int, 2>{}); break;
        default: finish(std::integral_constant<int, 3>{}); break;
    }
}
#endif  // FR_EXPERIMENTS

// FC3 + the output layer as ONE launch (fc_lp_gemm_out_kernel): wherever FC3 would run as a GEMM launch of its own (large batches) in the
// bf16 / fp8 chains and ONE 256-wide n tile holds all of its outputs (the reference's 256, constant.h:26).  Scores are bit-identical to FC3's
// GEMM launch + the stage pipeline's out stage (the epilogue keeps that stage's summation order), so nothing else in the library can tell the
// difference -- except the clock: Model-C batch 4096 saves the output launch (6.7 us of chip time per batch at < 0.01 of the MFMA peak), the
// R3 image's round trip through memory, and one launch of pipeline depth (profiles/r05_fc3_out_epilogue_ab.txt).
// (Round 4's fc_tail_kernel -- whole-CU workgroups of 32 items streaming W3 from L2 into registers, no LDS staging -- was a different design
// and no faster inside the four-stream chain; it is gone.)
bool frk_fc_tail_ok(int precision, int K, int N, int ldm) {
    if (precision != FR_FC_BF16 && precision != FR_FC_FP8) return false;
    if (FR_KNOB_ONCE("FC_TAIL", 1) == 0) return false;   // experiments build: FR_FC_TAIL=0 = the two launches (A/B)
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : K / 8;
    if (N != 256 || ldm % 128 || (precision == FR_FC_BF16 && K % 8) || KE % FR_GR || KE / FR_GR < 2) return false;
    return true;
}

int frk_fc_tail(int precision, const void *W3, const void *R2, const void *wout, float *scores, int K, int N, int ldm, int batch, int e_w, int e_in, int e_r3, hipStream_t s) {
    if (!frk_fc_tail_ok(precision, K, N, ldm)) FR_FAIL(FR_ERR_INVALID, "internal: %d x %d x %d is not a fused-tail layer", K, N, ldm);
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : K / 8;
    const size_t lds = (size_t)FR_GSTAGES * FR_GR * (256 + 128) * 16;
    dim3 grid(ldm / 128);
    FrTailArgs t{wout, scores, batch, precision == FR_FC_FP8 ? ldexpf(1.0f, -e_r3) : 1.0f};
    if (precision == FR_FC_BF16) {
        static FrLdsAttrOnce once1;
        if (int rc_ = fr_allow_full_lds(&fc_lp_gemm_out_kernel<1, FR_GSTAGES>, once1)) return rc_;
        fc_lp_gemm_out_kernel<1, FR_GSTAGES><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(W3), reinterpret_cast<const uint4 *>(R2), KE, N, ldm, 0, 0, 1.0f, t);
        KCHECK();
        fr_note_kernel("fc_lp_gemm_out_kernel<1, %d>", FR_GSTAGES);
    } else {
        static FrLdsAttrOnce once2;
        if (int rc_ = fr_allow_full_lds(&fc_lp_gemm_out_kernel<2, FR_GSTAGES>, once2)) return rc_;
        fc_lp_gemm_out_kernel<2, FR_GSTAGES><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(W3), reinterpret_cast<const uint4 *>(R2), KE, N, ldm, 127 - e_w, 127 - e_in,
                                                                         ldexpf(1.0f, e_r3), t);
        KCHECK();
        fr_note_kernel("fc_lp_gemm_out_kernel<2, %d>", FR_GSTAGES);
    }
    return FR_OK;
}

// Pipeline shape (experiment knob FR_GEMM_PIPE = 10 * G + NS; 0 = fc_lp_gemm_kernel): G row groups of 4 element rows per sub-step, NS
// sub-steps in LDS.  Default: see pipe_shape().
static int pipe_shape() {
    return FR_KNOB_ONCE("GEMM_PIPE", 15);
}

template <int PREC, int NS, int G>
static int pipe_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;
    const size_t lds = (size_t)NS * G * FR_PR * (FR_GN + 256) * 16;
    if (int rc_ = fr_allow_full_lds(&fc_gemm_pipe_kernel<PREC, NS, G>, lds_once)) return rc_;
    dim3 grid((N / FR_GN) * (ldm / 256));
    // bits 0-3: timing-only ablations (wrong results; experiments build only), bit 4: issue order, bit 8: wave priority
    const int ablate = FR_KNOB_ONCE("GEMM_ABLATE", 0) | (FR_KNOB_ONCE("GEMM_ORDER", 1) << 4) | (FR_KNOB_ONCE("GEMM_PRIO", 1) << 8);
    fc_gemm_pipe_kernel<PREC, NS, G><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale, ablate);
    KCHECK();
    fr_note_kernel("fc_gemm_pipe_kernel<%d, %d, %d>", PREC, NS, G);
    return FR_OK;
}

// sub-steps must come in pairs (the loop alternates two fragment register sets) and there must be enough of them to fill the pipeline
static bool pipe_shape_ok(int shape, int KE) {
    const int g = shape / 10, ns = shape % 10;
    if (!((g == 1 && (ns == 4 || ns == 5 || ns == 6)) || (g == 2 && ns == 3))) return false;
    return KE % (2 * FR_PR) == 0 && KE / (g * FR_PR) >= 2 * ns;  // G == 1 consumes sub-steps in pairs, G == 2 one at a time (8 rows either way)
}

template <int PREC>
static int pipe_gemm_dispatch(int shape, const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    switch (shape) {
        case 14: return pipe_gemm_launch<PREC, 4, 1>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        case 16: return pipe_gemm_launch<PREC, 6, 1>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        case 23: return pipe_gemm_launch<PREC, 3, 2>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        default: return pipe_gemm_launch<PREC, 5, 1>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
    }
}

// Which block tile serves the layer: 2 = 128 (n) x 256 (m) when those tiles cover most of the chip, 1 = 128 x 128, 3 = 64 x 128 for layers
// with few outputs (Model-C FC2 / FC3 at batch 4096: 512 / 256 outputs -> 256 / 128 tiles instead of 128 / 64), 0 = not worth a GEMM
// launch (the stage pipeline's per-tile body takes it).
static int lp_gemm_mu(int precision, int K, int N, int ldm, int width = 1, bool minor = false) {
    const int forced = FR_KNOB_ONCE("LP_GEMM", -1);  // experiment knob: 0 = never, 1 / 2 / 3 = only that tile
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if ((precision == FR_FC_BF16 && K % 8) || (precision == FR_FC_FP32 && K % 4) || KE % FR_GR || KE / FR_GR < 2 || N % 64 || ldm % 128) return 0;
    if (forced == 0) return 0;
    const long t256 = (N % 128 || ldm % 256) ? 0 : (long)(N / 128) * (ldm / 256), t128 = N % 128 ? 0 : (long)(N / 128) * (ldm / 128),
               t64 = (long)(N / 64) * (ldm / 128);
    // split-K direct tiles of the narrow layers (bf16 / fp8, N <= 512, K <= 2048: what a 4-wave ring can stream in a few us): 4 = 128 (n) x 64 (m),
    // 5 = 64 x 64; the layer must give every wave whole k-steps and fill at least half the chip
    const int rps = precision == FR_FC_FP8 ? 4 : 2;
    const bool sk_ok = precision != FR_FC_FP32 && N <= 512 && K <= 2048 && KE % (4 * rps) == 0 && ldm % 64 == 0;
    const long s128 = (sk_ok && N % 128 == 0) ? (long)(N / 128) * (ldm / 64) : 0, s64 = sk_ok ? (long)(N / 64) * (ldm / 64) : 0;
#ifdef FR_EXPERIMENTS
    if ((forced == 4 && s128) || (forced == 5 && s64)) return forced;
#endif
    if (forced > 0) return ((forced == 2 && t256) || (forced == 1 && t128) || forced == 3) ? forced : 0;
    const int small_tile = FR_KNOB_ONCE("LP_GEMM_SMALL", 3);  // experiment knob: 1 = no 64 x 128 tiles
    // EXPERIMENTS build only (FR_LP_GEMM_SPLITK=1): measured on Model-C 4096 with four streams, the split-K kernels are no faster end to
    // end in bf16 (37.0 vs 37.2 M inf/s) and slower in fp8 (55 vs 58-60 M): alone on the chip they would stream a layer's bytes once, but
    // one wave per SIMD with 320-490 registers shares a CU with nothing, and this chain lives on four streams' kernels sharing CUs
    // (profiles/archive/r03_experiments.md section 4)
    if (FR_KNOB_ONCE("LP_GEMM_SPLITK", 0)) {
        if (s128 >= 192 && s128 <= 512) return 4;
        if (s64 >= 128 && s64 <= 512) return 5;
    }
    // 256 (n) x 256 (m) tiles when THEY cover the chip (batch 8192 of Model-C's FC1: 8 x 32): a third fewer operand bytes per output through
    // the CU's vector-memory return path, which is what the 128 x 256 kernel keeps busy (texture data return busy 0.73, MFMA busy 0.50:
    // profiles/archive/r04_pmc_gemm_bf16.json); bf16 / fp8 only (the fp32 kernel is MFMA-bound)
    const long t256sq = (N % 256 || ldm % 256) ? 0 : (long)(N / 256) * (ldm / 256);
    // `width` = the context's chain width W (fr_ctx_set_chain_width; frozen at min(live workers, 4) by the context's first low-precision
    // GEMM-layer launch otherwise).  The larger tile is taken as soon as it covers 1 / W
    // of the chip: a chain model's workers sit on their own hardware queues (fr_worker_create) and run their chains side by side, so W
    // part-chip launches of the cheaper tile (fewer operand bytes per output through the CU's vector-memory path) share the chip where W
    // full-chip launches time-share every CU.  Model-C batch 4096, four workers: FC1 8 x 16 tiles of 256 x 256 (for 256 of 128 x 256), FC2
    // 4 x 16 of 128 x 256 (for 256 of 64 x 128), FC3 2 x 32 of 128 x 128: bf16 38.4 -> 44.5 M inf/s, fp8 62.6 -> 69.0 M; a lone worker on
    // half-chip tiles would lose 15-18 % (profiles/archive/r04_C4096_half_chip_tiles_ab.txt).  bf16 / fp8 only (the fp32 kernel is MFMA-bound).
    const int part_knob = FR_KNOB_ONCE("LP_GEMM_PART", -1);   // experiment knob: the divisor (1 = full-chip tiles only, 2, 4), whatever the worker count
    const int part = precision == FR_FC_FP32 ? 1 : (part_knob > 0 ? part_knob : (width < 1 ? 1 : (width > 4 ? 4 : width)));
    const long full = 192 / part;
    // at W = 4 a MINOR layer of the chain (at most half the work of its heaviest layer: Model-C's FC2 beside FC1) takes 256 x 256 tiles from 24 of
    // them on (FC2 at batch 4096: 32 tiles on 32 CUs for 64 of 128 x 256 on 64): a launch takes longer (45 for 35 us) on half the CUs, the heavy
    // layers of the other chains fill the rest -- bf16 47.2 -> 48.2 M inf/s, fp8 69.6 -> 72.4 M.  The heaviest layer itself must cover its share:
    // with the low threshold Model-C FC1 at batch 1024 sat on 4 x 32 CUs -- bf16 34.3 -> 28.2 M inf/s (profiles/r05_experiments.md section 13)
    const long full256 = (part >= 4 && minor) ? FR_KNOB_ONCE("LP_GEMM_256_MIN", 24) : full;   // experiment knob: 48 = no low threshold
    if (precision != FR_FC_FP32 && t256sq >= full256 && FR_KNOB_ONCE("LP_GEMM_256", 1)) return 6;
    if (t256 >= full) return 2;
    if (t128 >= full) return 1;
    if (t64 >= 128 && small_tile == 3) return 3;
    if (t128 >= 64) return 1;
    return 0;
}

bool frk_fc_lp_gemm_ok(int precision, int K, int N, int ldm) { return lp_gemm_mu(precision, K, N, ldm) != 0; }

template <int PREC, int MU, int GN, int S, int GR = FR_GR, int MF = 32>
static int lp_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    const size_t lds = (size_t)S * GR * (GN + 128 * MU) * 16;
    if (int rc_ = fr_allow_full_lds(&fc_lp_gemm_kernel<PREC, MU, GN, S, GR, MF>, lds_once)) return rc_;
    dim3 grid((N / GN) * (ldm / (128 * MU)));
    fc_lp_gemm_kernel<PREC, MU, GN, S, GR, MF><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale);
    KCHECK();
    fr_note_kernel("fc_lp_gemm_kernel<%d, %d, %d, %d, %d, %d>", PREC, MU, GN, S, GR, MF);   // as rocprofv3 prints it
    return FR_OK;
}

template <int PREC, int D, int NT>
static int pp_gemm_launch_nt(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;
    const size_t lds = (size_t)(D + 1) * 4 * 512 * 16;
    if (int rc_ = fr_allow_full_lds(&fc_pp_gemm_kernel<PREC, D, NT>, lds_once)) return rc_;
    dim3 grid((N / 256) * (ldm / 256));
    fc_pp_gemm_kernel<PREC, D, NT><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale,
                                                                FR_KNOB_ONCE("PP_ABLATE", 0));
    KCHECK();
    fr_note_kernel("fc_pp_gemm_kernel<%d, %d, %d>", PREC, D, NT);
    return FR_OK;
}

template <int PREC, int D>
static int pp_gemm_n128_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;
    const size_t lds = (size_t)(D + 1) * 8 * (128 + 256) * 16;
    if (int rc_ = fr_allow_full_lds(&fc_pp_gemm_n128_kernel<PREC, D>, lds_once)) return rc_;
    dim3 grid((N / 128) * (ldm / 256));
    fc_pp_gemm_n128_kernel<PREC, D><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale,
                                                                 FR_KNOB_ONCE("PP_ABLATE", 0));
    KCHECK();
    fr_note_kernel("fc_pp_gemm_n128_kernel<%d, %d>", PREC, D);
    return FR_OK;
}

template <int PREC, int D>
static int pp_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    if (N == 2048) return pp_gemm_launch_nt<PREC, D, 8>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
    if (N == 512) return pp_gemm_launch_nt<PREC, D, 2>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
    return pp_gemm_launch_nt<PREC, D, 0>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
}

#ifdef FR_EXPERIMENTS
template <int PREC, int NT, int MT, int PD>
static int splitk_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;
    const size_t lds = (size_t)NT * MT * 3 * 16 * 64 * sizeof(float);
    if (int rc_ = fr_allow_full_lds(&fc_splitk_gemm_kernel<PREC, NT, MT, PD>, lds_once)) return rc_;
    dim3 grid((N / (32 * NT)) * (ldm / (32 * MT)));
    fc_splitk_gemm_kernel<PREC, NT, MT, PD><<<grid, dim3(256), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale);
    KCHECK();
    fr_note_kernel("fc_splitk_gemm_kernel<%d, %d, %d, %d>", PREC, NT, MT, PD);
    return FR_OK;
}
#endif

template <int PREC>
static int lp_gemm_tile(int mu, const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    if constexpr (PREC != 0) {
        if (mu == 6) {   // the 256 x 256 tile: fc_pp_gemm_kernel, the SIMD's two waves in opposite phases; 3 sub-steps ahead in bf16, 2 in fp8
                         // (profiles/r05_experiments.md section 13).  lp_gemm_mu guarantees KE % 8 == 0 and KE >= 16: whole sub-steps, a full pipeline.
#ifdef FR_EXPERIMENTS
            const int pp = FR_KNOB_ONCE("LP_GEMM_PP", (PREC == 1 ? 3 : 2));   // 0 = fc_lp_gemm_kernel's plain loop on the same tile (the A/B partner: same bits)
            if (pp == 3) return pp_gemm_launch<PREC, 3>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
            if (pp == 2) return pp_gemm_launch<PREC, 2>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
            if constexpr (PREC == 1) {
                if (FR_KNOB_ONCE("LP_GEMM_MF16", 1)) return lp_gemm_launch<1, 2, 256, FR_GSTAGES, FR_GR, 16>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
            }
            return lp_gemm_launch<PREC, 2, 256, FR_GSTAGES>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
#else
            return pp_gemm_launch<PREC, (PREC == 1 ? 3 : 2)>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
#endif
        }
    }
    if (mu == 2) return lp_gemm_launch<PREC, 2, 128, FR_GSTAGES>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
#ifdef FR_EXPERIMENTS
    if constexpr (PREC != 0) {   // ring depth: 4 k-steps of 16 k (bf16) / 2 of 64 k (fp8) = 24 / 48 KiB in flight per wave at 128 x 64
        if (mu == 4) return splitk_gemm_launch<PREC, 4, 2, (PREC == 1 ? 8 : 4)>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        if (mu == 5) return splitk_gemm_launch<PREC, 2, 2, (PREC == 1 ? 8 : 4)>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
    }
#endif
    if (mu == 3) {
        // 4 stages pay in bf16 only (Model-C FC2 19.8 -> 14.7 us, end to end +2.5 %); in fp8 / f32 the larger LDS footprint costs more
        // beside the other streams' kernels than the deeper prefetch gains (fp8 end to end 57.3 -> 52.8 M inf/s).  Knob: FR_LP_GEMM_STAGES
        const int deep = FR_KNOB_ONCE("LP_GEMM_STAGES", (PREC == 1 ? 4 : 2));
#ifdef FR_EXPERIMENTS
        const int rows = FR_KNOB_ONCE("LP_GEMM_ROWS", 8);   // element rows per K step of the narrow tile
        if (rows == 16 && KE % 16 == 0 && KE / 16 >= 2) {
            if (deep == 2) return lp_gemm_launch<PREC, 1, 64, 2, 16>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
            return lp_gemm_launch<PREC, 1, 64, 3, 16>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        }
        if (rows == 32 && KE % 32 == 0 && KE / 32 >= 2) return lp_gemm_launch<PREC, 1, 64, 2, 32>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        if (deep == 3) return lp_gemm_launch<PREC, 1, 64, 3>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        if (deep == 5) return lp_gemm_launch<PREC, 1, 64, 5>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
        if (deep == 6) return lp_gemm_launch<PREC, 1, 64, 6>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
#endif
        return deep == 2 ? lp_gemm_launch<PREC, 1, 64, 2>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s)
                         : lp_gemm_launch<PREC, 1, 64, 4>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
    }
    return lp_gemm_launch<PREC, 1, 128, FR_GSTAGES>(Wp, Xp, Yp, KE, N, ldm, sc_a, sc_b, oscale, s);
}

// FC1 of one batch + the gather of the next in one launch (fc_gemm_gather_kernel): for layers the 128 x 256 tile takes, bf16 / fp8
bool frk_fc_gemm_gather_ok(int precision, int K, int N, int ldm) {
    if (precision != FR_FC_BF16 && precision != FR_FC_FP8) return false;
    // EXPERIMENTS build only (FR_GEMM_GATHER=1).  Correct (tests/test_gpu_lowprec.py::test_model_c_streaming_gather_inside_fc1: operand image and
    // fp8 scores bit for bit) and SLOWER: alone on one stream the fused launch takes as long as the GEMM and the gather one after the other
    // (bf16 102.7 us against 63.1 + 37.7; fp8 66.3 against 32.8 + 35.8), in the four-stream chain bf16 38.1 -> 33.7 M inf/s, fp8 60.6 -> 50.1 M.
    // The two do not overlap inside a CU: the GEMM's operand DMA and the gather's row fetches go through the same vector-memory path, and a
    // wave-private producer still gates all 12 waves at every s_barrier (profiles/archive/r04_experiments.md section 1.2).
    if (FR_KNOB_ONCE("GEMM_GATHER", 0) == 0) return false;
    return lp_gemm_mu(precision, K, N, ldm) == 2;
}

int frk_fc_gemm_gather(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, const FrWordDesc *words, int n_words,
                       int idx_stride, const int32_t *idx, const float *dense, int g_batch, int g_ldm, int g_K, void *g_out, int g_e_x, int *err_flag, hipStream_t s) {
    if (!frk_fc_gemm_gather_ok(precision, K, N, ldm)) FR_FAIL(FR_ERR_INVALID, "internal: %d x %d x %d is not a gather + GEMM layer", K, N, ldm);
#ifndef FR_EXPERIMENTS
    FR_FAIL(FR_ERR_INVALID, "internal: fc_gemm_gather_kernel is built into the experiments library only");
#else
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : K / 8;
    FrGatherJob g{};
    g.words = words, g.idx = idx, g.dense = dense, g.out = g_out, g.err_flag = err_flag;
    g.n_words = n_words, g.idx_stride = idx_stride, g.batch = g_batch, g.ldm = g_ldm, g.K = g_K;
    g.scale = ldexpf(1.0f, g_e_x);
    const size_t lds = ((size_t)FR_GSTAGES * FR_GR * (FR_GN + 256) + 4 * (size_t)FR_GG_PATCH) * 16;
    dim3 grid((N / FR_GN) * (ldm / 256));
    if (precision == FR_FC_BF16) {
        static FrLdsAttrOnce once1;
        if (int rc_ = fr_allow_full_lds(&fc_gemm_gather_kernel<1>, once1)) return rc_;
        fc_gemm_gather_kernel<1><<<grid, dim3(768), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, 0, 0, 1.0f, g);
        KCHECK();
        fr_note_kernel("fc_gemm_gather_kernel<1>");
    } else {
        static FrLdsAttrOnce once2;
        if (int rc_ = fr_allow_full_lds(&fc_gemm_gather_kernel<2>, once2)) return rc_;
        fc_gemm_gather_kernel<2><<<grid, dim3(768), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, 127 - e_w, 127 - e_in,
                                                              ldexpf(1.0f, e_out), g);
        KCHECK();
        fr_note_kernel("fc_gemm_gather_kernel<2>");
    }
    return FR_OK;
#endif
}

int frk_fc_lp_gemm(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, int width, bool minor, hipStream_t s) {
    const int mu = lp_gemm_mu(precision, K, N, ldm, width, minor);
    if (mu == 0) FR_FAIL(FR_ERR_INVALID, "internal: layer %d x %d x %d is not a GEMM-kernel layer", K, N, ldm);
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if (precision == FR_FC_FP32) return lp_gemm_tile<0>(mu, Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    // bf16 128 x 256 layers run the software-pipelined kernel; fp8 stays on fc_lp_gemm_kernel, which measured faster there (29.4 vs
    // 32.9 us on Model-C FC1: its 8-row steps halve the barriers per 64-cycle MFMA) unless FR_GEMM_PIPE_FP8=1 asks for the experiment
    // 128 x 256 layers (a lone worker's Model-C FC1: chain width 1) on the phased-waves body in 8-row sub-steps (lp_gemm_mu guarantees KE % 8 == 0;
    // KE >= 24: a full pipeline): bit-identical to the kernels below, FC1 bf16 65.5 -> 59.6 us, fp8 32.5 -> 31.4 us (profiles/r05_experiments.md
    // section 13).  Experiment knob FR_LP_GEMM_PP128=0: the kernels below.
    if (mu == 2 && precision != FR_FC_FP32 && FR_KNOB_ONCE("LP_GEMM_PP128", 1) && KE / 8 >= 3) {
        if (precision == FR_FC_FP8) return pp_gemm_n128_launch<2, 2>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, ldexpf(1.0f, e_out), s);
        return pp_gemm_n128_launch<1, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    }
    const bool pipe_fp8 = FR_KNOB_ONCE("GEMM_PIPE_FP8", 0) != 0;
    if (mu == 2 && (precision == FR_FC_BF16 || (precision == FR_FC_FP8 && pipe_fp8)) && pipe_shape_ok(pipe_shape(), KE)) {
        if (precision == FR_FC_FP8) return pipe_gemm_dispatch<2>(pipe_shape(), Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, ldexpf(1.0f, e_out), s);
        return pipe_gemm_dispatch<1>(pipe_shape(), Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    }
    if (precision == FR_FC_FP8) return lp_gemm_tile<2>(mu, Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, ldexpf(1.0f, e_out), s);
    return lp_gemm_tile<1>(mu, Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
}
