#include "fr_device.h"

// ===================================================================================================
// fc_lp_gemm_kernel<PREC>: LDS-tiled GEMM for low-precision layers that fill the chip on their own (Model-C FC1 at batch 4096).
// PREC 0 = fp32 (q4 elements, v_mfma_f32_32x32x2_f32: the same k-ordered exact-f32 sums as the other fp32 kernels, full K per
// output), PREC 1 = bf16 (q8 elements, v_mfma_f32_32x32x16_bf16), PREC 2 = fp8 (q16 elements, v_mfma_scale_f32_32x32x64_f8f6f4).
// Block tile 128 (n) x 256 (m), 8 waves as 2 x 4 with 64 x 64 wave tiles (2 x 2 MFMA tiles: 4 fragment reads per 4 MFMAs).
// A K step is 8 rows of 16-byte elements (64 k in bf16, 128 k in fp8): 16 KiB of W + 32 KiB of X, three steps in LDS (144 KiB).
// Operands go global -> LDS directly (buffer_load ... lds, 64 consecutive elements per wave-instruction, no VGPR staging) two
// steps ahead of the MFMAs; one barrier per step.  Fragment reads are conflict-free ds_read_b128 (a b128 access is served 16
// lanes at a time, and 16 consecutive elements of a row are 256 contiguous bytes).  L2 -> CU traffic per output is 25 % below
// that of 128 x 128 tiles.
// ===================================================================================================
constexpr int FR_GN = 128, FR_GM = 256, FR_GR = 8;
// STAGES K steps in LDS: 3 = 144 KiB, loads two steps ahead (the workgroup owns its CU); 2 = 96 KiB, one step ahead, leaves room
// for a 32 KiB stage-pipeline workgroup of another stream on the same CU (FR_LP_GEMM_STAGES=2).

template <int PREC, int STAGES>
__global__ void __launch_bounds__(512) fc_lp_gemm_kernel(const uint4 *__restrict__ W, const uint4 *__restrict__ X, void *__restrict__ Y, int KE /* element rows */,
                                                         int N, int ldm, int sc_a, int sc_b, float oscale) {
    extern __shared__ uint4 glds[];
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 1, wm = wave >> 1;
    const int tn = N / FR_GN, tm = ldm / FR_GM;
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
    const int n0 = n_tile * FR_GN, m0 = m_tile * FR_GM;
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
    auto As = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * (FR_GN + FR_GM); };          // 128 elements
    auto Bs = [&](int st, int row) { return glds + ((size_t)st * FR_GR + row) * (FR_GN + FR_GM) + FR_GN; };  // 256 elements
    // global -> LDS without a VGPR round trip: lane i's 16 bytes land at M0 + 16 i.  Inline asm on purpose: through the builtin the
    // compiler treats every LDS read as a possible alias of the DMA write and waits for vmcnt(0) before each fragment read, which
    // removes the two-step prefetch distance; the counted s_waitcnt below are the only synchronisation these loads need.
    auto dma = [&](const i32x4_t &rs, const uint4 *dst, unsigned voff, unsigned soff) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(lds_ptr)dst);
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory");
    };
    // staging: 48 wave-instructions of 64 elements per step; every wave issues 2 of W's 16 and 4 of X's 32
    const unsigned vW = (unsigned)(n0 + 64 * (wave & 1) + lane) * 16u, vX = (unsigned)(m0 + lane) * 16u;
    auto issue = [&](int step, int st) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int row = (wave >> 1) + 4 * i;  // 8 waves x 2 = rows 0..7 x 2 halves
            dma(rsW, As(st, row) + 64 * (wave & 1), vW, (unsigned)(step * FR_GR + row) * (unsigned)N * 16u);
        }
#pragma unroll
        for (int i = 0; i < 4; i++)  // row = wave, four quarters
            dma(rsX, Bs(st, wave) + 64 * i, vX + 64u * 16u * i, (unsigned)(step * FR_GR + wave) * (unsigned)ldm * 16u);
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][u][i] = 0.0f;
    const int nsteps = KE / FR_GR;
    issue(0, 0);
    if (STAGES == 3 && nsteps > 1) issue(1, 1);
    for (int s = 0; s < nsteps; s++) {
        // this wave's loads of step s have landed once at most the 6 of step s+1 are outstanding
        if (STAGES == 3 && s + 1 < nsteps) __builtin_amdgcn_s_waitcnt(0x0F76);  // vmcnt(6)
        else __builtin_amdgcn_s_waitcnt(0x0F70);                                // vmcnt(0)
        __syncthreads();  // everyone's step-s data is in LDS, and everyone is done reading the buffer the next issue overwrites
        if (s + STAGES - 1 < nsteps) issue(s + STAGES - 1, (s + STAGES - 1) % STAGES);
        const int st = s % STAGES;
        if constexpr (PREC == 0) {  // q4 fp32 elements: one element per lane feeds four v_mfma_f32_32x32x2_f32 (k = 8 kk + 4 h + c)
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 64 + r;
                const uint4 a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
                const uint32_t a0c[4] = {a0.x, a0.y, a0.z, a0.w}, a1c[4] = {a1.x, a1.y, a1.z, a1.w};
                const uint32_t b0c[4] = {b0.x, b0.y, b0.z, b0.w}, b1c[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a0c[c]), __uint_as_float(b0c[c]), acc[0][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a1c[c]), __uint_as_float(b0c[c]), acc[1][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a0c[c]), __uint_as_float(b1c[c]), acc[0][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a1c[c]), __uint_as_float(b1c[c]), acc[1][1], 0, 0, 0);
                }
            }
        } else if constexpr (PREC == 1) {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 2; kk++) {
                const uint4 *ar = As(st, 2 * kk + h) + wn * 64 + r, *br = Bs(st, 2 * kk + h) + wm * 64 + r;
                const uint4 a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < FR_GR / 4; kk++) {
                i32x8 a[2], b[2];
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const uint4 alo = As(st, 4 * kk + 2 * h)[wn * 64 + 32 * t + r], ahi = As(st, 4 * kk + 2 * h + 1)[wn * 64 + 32 * t + r];
                    const uint4 blo = Bs(st, 4 * kk + 2 * h)[wm * 64 + 32 * t + r], bhi = Bs(st, 4 * kk + 2 * h + 1)[wm * 64 + 32 * t + r];
                    a[t][0] = (int)alo.x; a[t][1] = (int)alo.y; a[t][2] = (int)alo.z; a[t][3] = (int)alo.w;
                    a[t][4] = (int)ahi.x; a[t][5] = (int)ahi.y; a[t][6] = (int)ahi.z; a[t][7] = (int)ahi.w;
                    b[t][0] = (int)blo.x; b[t][1] = (int)blo.y; b[t][2] = (int)blo.z; b[t][3] = (int)blo.w;
                    b[t][4] = (int)bhi.x; b[t][5] = (int)bhi.y; b[t][6] = (int)bhi.z; b[t][7] = (int)bhi.w;
                }
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int t = 0; t < 2; t++) acc[t][u] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t], b[u], acc[t][u], 0, 0, 0, sc_a, 0, sc_b);
            }
        }
    }
    // epilogue: ONE rounding per output; registers 4i..4i+3 of a tile are 4 consecutive n
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const f32x16 &c = acc[t][u];
            const int m = m0 + wm * 64 + 32 * u + r;
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

// precision: FR_FC_BF16 (K % 64 == 0) or FR_FC_FP8 (K padded to 128 by the caller's layout: KE % 8 == 0)
bool frk_fc_lp_gemm_ok(int precision, int K, int N, int ldm) {
    static const int forced = getenv("FR_LP_GEMM") ? atoi(getenv("FR_LP_GEMM")) : -1;  // experiment knob: 0 = never, 1 = whenever legal
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if ((precision == FR_FC_BF16 && K % 8) || (precision == FR_FC_FP32 && K % 4) || KE % FR_GR || KE / FR_GR < 2 || N % FR_GN || ldm % FR_GM) return false;
    if (forced == 0) return false;
    if (forced == 1) return true;
    static const int min_tiles = getenv("FR_LP_GEMM_MIN_TILES") ? atoi(getenv("FR_LP_GEMM_MIN_TILES")) : 64;
    return (long)(N / FR_GN) * (ldm / FR_GM) >= min_tiles;  // 64 tiles = a quarter of the CUs (Model-C FC2 at batch 4096: runs beside another stream's gather)
}

template <int PREC, int STAGES>
static int lp_gemm_launch(const void *Wp, const void *Xp, void *Yp, int KE, int N, int ldm, int sc_a, int sc_b, float oscale, hipStream_t s) {
    static bool attr_set = false;
    const size_t lds = (size_t)STAGES * FR_GR * (FR_GN + FR_GM) * 16;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&fc_lp_gemm_kernel<PREC, STAGES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            FR_FAIL(FR_ERR_HIP, "hipFuncSetAttribute(max dynamic LDS) failed");
        attr_set = true;
    }
    dim3 grid((N / FR_GN) * (ldm / FR_GM));
    fc_lp_gemm_kernel<PREC, STAGES><<<grid, dim3(512), lds, s>>>(reinterpret_cast<const uint4 *>(Wp), reinterpret_cast<const uint4 *>(Xp), Yp, KE, N, ldm, sc_a, sc_b, oscale);
    KCHECK();
    return FR_OK;
}

int frk_fc_lp_gemm(int precision, const void *Wp, const void *Xp, void *Yp, int K, int N, int ldm, int e_w, int e_in, int e_out, hipStream_t s) {
    static const int stages = getenv("FR_LP_GEMM_STAGES") ? atoi(getenv("FR_LP_GEMM_STAGES")) : 2;  // 2: co-resident with other streams' stage kernels
    const int KE = precision == FR_FC_FP8 ? (K + 63) / 64 * 4 : (precision == FR_FC_BF16 ? K / 8 : K / 4);
    if (precision == FR_FC_FP32)
        return stages == 2 ? lp_gemm_launch<0, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<0, 3>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
    if (precision == FR_FC_FP8) {
        const float os = ldexpf(1.0f, e_out);
        return stages == 2 ? lp_gemm_launch<2, 2>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s)
                           : lp_gemm_launch<2, 3>(Wp, Xp, Yp, KE, N, ldm, 127 - e_w, 127 - e_in, os, s);
    }
    return stages == 2 ? lp_gemm_launch<1, 2>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s) : lp_gemm_launch<1, 3>(Wp, Xp, Yp, KE, N, ldm, 0, 0, 1.0f, s);
}

