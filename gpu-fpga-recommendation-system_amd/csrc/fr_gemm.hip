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
// (Three steps, 144 KiB, were 10 % faster alone and slower overall: profiles/r01_experiments.md.)

template <int PREC, int MU>
__global__ void __launch_bounds__(512) fc_lp_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                         int N, int ldm, int sc_a, int sc_b, float oscale) {
    extern __shared__ uint4 glds[];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int GM = 128 * MU, ROW = FR_GN + GM;  // elements per staged row: 128 of W, GM of X
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tn = N / FR_GN, tm = ldm / GM;
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
    const int n0 = n_tile * FR_GN, m0 = m_tile * GM;
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
    auto As = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * ROW; };          // 128 elements
    auto Bs = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * ROW + FR_GN; };  // GM elements
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
    const unsigned vW = (unsigned)(n0 + 64 * (wave & 1) + lane) * 16u, vX = (unsigned)(m0 + lane) * 16u;
    auto issue = [&](int step, int st) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (wave >> 1) + 4 * i;  // 8 waves x 2 = rows 0..7 x 2 halves
            dma(rsW, As(st, row) + 64 * (wave & 1), vW, (unsigned)(step * FR_GR + row) * (unsigned)N * 16u);
        }
#pragma unroll
        for (int i = 0; i < 2 * MU; i++)  // row = wave, GM / 64 parts
            dma(rsX, Bs(st, wave) + 64 * i, vX + 64u * 16u * i, (unsigned)(step * FR_GR + wave) * (unsigned)ldm * 16u);
    };
    f32x16 acc[2][MU];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < MU; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
    const int nsteps = KE / FR_GR;
    issue(0, 0);
    for (int s = 0; s < nsteps; s++) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's loads of step s have landed
        __syncthreads();  // everyone's step-s data is in LDS, and everyone is done reading the buffer the next issue overwrites
        if (s + 1 < nsteps) issue(s + 1, (s + 1) % FR_GSTAGES);
        const int st = s % FR_GSTAGES;
        if constexpr (PREC == 0) {  // q4 fp32 elements: one element per lane feeds four v_mfma_f32_32x32x2_f32 (k = 8 kk + 4 h + c)
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 32 * MU + r;
                uint4 a[2], b[MU];
#pragma unroll
                for (int t = 0; t < 2; t++) a[t] = ar[32 * t];
#pragma unroll
                for (int u = 0; u < MU; u++) b[u] = br[32 * u];
#pragma unroll
                for (int c = 0; c < 4; c++)
#pragma unroll
                    for (int u = 0; u < MU; u++)
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            const uint32_t av = c == 0 ? a[t].x : c == 1 ? a[t].y : c == 2 ? a[t].z : a[t].w;
                            const uint32_t bv = c == 0 ? b[u].x : c == 1 ? b[u].y : c == 2 ? b[u].z : b[u].w;
                            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(av), __uint_as_float(bv), acc[t][u], 0, 0, 0);
                        }
            }
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 32 * MU + r;
                uint4 a[2], b[MU];
#pragma unroll
                for (int t = 0; t < 2; t++) a[t] = ar[32 * t];
#pragma unroll
                for (int u = 0; u < MU; u++) b[u] = br[32 * u];
#pragma unroll
                for (int u = 0; u < MU; u++)
#pragma unroll
                    for (int t = 0; t < 2; t++)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]), __builtin_bit_cast(bf16x8, b[u]), acc[t][u], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 4; kk++) {
                i32x8 a[2], b[MU];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint4 alo = As(st, 4 * kk + 2 * h)[wn * 64 + 32 * t + r], ahi = As(st, 4 * kk + 2 * h + 1)[wn * 64 + 32 * t + r];
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
                    for (int t = 0; t < 2; t++) acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], b[u], acc[t][u], 0, 0, 0, sc_a, 0, sc_b);
            }
        }
    }
    // epilogue: ONE rounding per output; registers 4i..4i+3 of a tile are 4 consecutive n
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < MU; u++) {
            const f32x16 &c = acc[t][u];
            const int m = m0 + wm * 32 * MU + 32 * u + r;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int n = n0 + wn * 64 + 32 * t + 8 * i + 4 * h;  // + c
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

// Which block tile serves the layer: 2 (128 x 256) when those tiles cover most of the chip, 1 (128 x 128) when only the smaller ones
// reach a quarter of it, 0 = not worth a GEMM launch (the stage pipeline's per-tile body takes it).
static int lp_gemm_mu(int precision, int K, int N, int ldm) {
    static const int forced = getenv("FR_LP_GEMM") ? atoi(getenv("FR_LP_GEMM")) : -1;  // experiment knob: 0 = never
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if ((precision == FR_FC_BF16 && K % 8) || (precision == FR_FC_FP32 && K % 4) || KE % FR_GR || KE / FR_GR < 2 || N % FR_GN || ldm % 128) return 0;
    if (forced == 0) return 0;
    const long t256 = ldm % 256 ? 0 : (long)(N / FR_GN) * (ldm / 256), t128 = (long)(N / FR_GN) * (ldm / 128);
    if (t256 >= 192) return 2;
    if (t128 >= 64) return 1;
    return 0;
}

bool frk_fc_lp_gemm_ok(int precision, int K, int N, int ldm) { return lp_gemm_mu(precision, K, N, ldm) != 0; }

template <int PREC, int MU>
static int lp_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static FrLdsAttrOnce lds_once;  // per instantiation, per device
    const size_t lds = (size_t)FR_GSTAGES * FR_GR * (FR_GN + 128 * MU) * 16;
    if (int rc_ = fr_allow_full_lds(&fc_lp_gemm_kernel<PREC, MU>, lds_once)) return rc_;
    dim3 grid((N / FR_GN) * (ldm / (128 * MU)));
    fc_lp_gemm_kernel<PREC, MU><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale);
    KCHECK();
    return FR_OK;
}

int frk_fc_lp_gemm(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, hipStream_t s) {
    const int mu = lp_gemm_mu(precision, K, N, ldm);
    if (mu == 0) FR_FAIL(FR_ERR_INVALID, "internal: layer %d x %d x %d is not a GEMM-kernel layer", K, N, ldm);
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if (precision == FR_FC_FP32)
        return mu == 2 ? lp_gemm_launch<0, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<0, 1>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    if (precision == FR_FC_FP8) {
        const float os = ldexpf(1.0f, e_out);
        return mu == 2 ? lp_gemm_launch<2, 2>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s)
                       : lp_gemm_launch<2, 1>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s);
    }
    return mu == 2 ? lp_gemm_launch<1, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<1, 1>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
}
