// pafuse_hip.hip - host side of the C ABI declared in include/pafuse_hip.h: argument checks, kernel selection
// per width, the MixSTE2 layer schedule and the D3DP DDIM loop.  No allocation, no synchronisation.
#include "../../include/pafuse_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"
#include "hgemm.hpp"
#include "xgemm.hpp"
#include "sgemm.hpp"
#include "train_kernels.hpp"

using namespace pafuse;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PAFUSE_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return PAFUSE_OK;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// "first call on the current device" latch for per-device function attributes (benign race: the guarded call is
// idempotent).  A failing hipGetDevice reports "first" every time, which is merely slower.
struct DeviceOnce {
    static constexpr int MAX_DEV = 64;
    bool done[MAX_DEV] = {};
    bool first() {
        int d = -1;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) return true;
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// Makes the device of the caller's stream current for the duration of an entry point (kernel launches, events and
// function attributes go to the CURRENT device): a caller whose tensors live on cuda:1 while cuda:0 is current
// would otherwise record events on the wrong device.  The legacy null stream belongs to the current device.
struct StreamDevice {
    int prev = -1;
    bool switched = false;
    explicit StreamDevice(void* stream) {
        hipDevice_t d = -1;
        if (stream && hipStreamGetDevice((hipStream_t)stream, &d) == hipSuccess && hipGetDevice(&prev) == hipSuccess &&
            d >= 0 && d != prev)
            switched = hipSetDevice(d) == hipSuccess;
        (void)hipGetLastError();  // a failed query must not surface as the next launch's error
    }
    ~StreamDevice() {
        if (switched) (void)hipSetDevice(prev);
    }
    StreamDevice(const StreamDevice&) = delete;
    StreamDevice& operator=(const StreamDevice&) = delete;
};

// ------------------------------------------------------------------------------------------------ GEMM dispatch
template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW = 1, int TR = 0, int BF16 = 0>
int launch_gemm(const GemmParams& p, hipStream_t s) {
    using T = GemmTile<WM, WN, NT>;
    constexpr size_t stage_bytes = (size_t)NSTAGE * (BF16 == 2 ? T::STAGE_FLOATS_SPLIT : T::STAGE_FLOATS) * sizeof(float);
    if (BF16 == 2 && !p.Wsplit) return fail(PAFUSE_E_ARG, "split-precision GEMM without a pre-split weight image");
    static_assert(EPI == EPI_BIAS || 7 * T::BM * WN <= NSTAGE * T::STAGE_FLOATS, "cross-wave reduction scratch must fit");
    static_assert(stage_bytes <= 160 * 1024, "LDS budget");
    size_t lds = stage_bytes;
    auto k = gemm_kernel<WM, WN, NT, EPI, NSTAGE, MINW, TR, BF16>;
#ifdef PAFUSE_DIAG
    static const int dbg_pad = [] { const char* e = getenv("PAFUSE_DEBUG_LDS_PAD"); return e ? atoi(e) : 0; }();
    if (dbg_pad && BF16 == 2 && EPI == EPI_BIAS) {  // diagnostic: keep other kernels off this workgroup's CU
        lds = std::max(lds, (size_t)dbg_pad);
        hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    } else
#endif
    if (lds > 64 * 1024) {
        static DeviceOnce once;  // the attribute is per device (one-process multi-device callers: nn.DataParallel)
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t tiles_m = (p.M + T::BM - 1) / T::BM;
    const int64_t tiles = tiles_m * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, s, p);
    return check_launch("gemm_kernel");
}

// qkv projection + attention of one head per workgroup (kernels.hpp fqa_kernel)
template <int LP, int DP>
int launch_fqa(const FqaParams& f, hipStream_t s) {
    using FT = FqaTile<LP, DP>;
    static_assert(FT::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    auto k = fqa_kernel<LP, DP>;
    if (FT::LDS_BYTES > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FT::LDS_BYTES);
    }
    const int64_t ntiles = (f.nseq + f.nseq_tile - 1) / f.nseq_tile;
    const int64_t blocks = (ntiles + 7) / 8 * 8 * f.heads;
    if (blocks <= 0 || blocks > 0x7fffffff) return fail(PAFUSE_E_ARG, "fused qkv-attention grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), FT::LDS_BYTES, s, f);
    return check_launch("fqa_kernel");
}

// padded sizes of the fused kernel for a sequence length / head dim, 0 = no fused form (the caller keeps qkv GEMM + attention)
// (mode 3, f16x2, also has an 80-token form on five waves - 160-row tiles - for head dim <= 32: the face's 68 joints)
static int fqa_lp(int L, int bf16 = 2) { return L <= 32 ? 32 : (L <= 48 ? 48 : (bf16 >= 3 && L <= 80 ? 80 : 0)); }
static int fqa_dp(int d) { return (d % 4 || d > 48) ? 0 : (d <= 32 ? 32 : 48); }
static bool fqa_has(int L, int d, int bf16 = 2) {   // (48, 48) would need 80.4 KB of LDS: one workgroup per CU, not built
    const int lp = fqa_lp(L, bf16), dp = fqa_dp(d);
    return lp && dp && !(lp >= 48 && dp == 48);
}
// mode 4 (xgemm.hpp xfqa_kernel): the workgroup's columns come in whole 32-column blocks - two heads at head dim 33 .. 48
static bool xfqa_has(int L, int d, int heads) { return fqa_has(L, d, 4) && (fqa_dp(d) == 32 || heads % 2 == 0); }
// rows of the fused kernel's q | k | v tiles, and whole sequences per tile (the last one's LP-row key tile inside the buffer)
static int fqa_rows(int lp) { return lp == 80 ? 160 : 128 + (lp == 48 ? 4 : 0); }
static int fqa_tile_rows(int lp) { return lp == 80 ? 160 : 128; }
static int fqa_nseq_tile(int L, int lp) { return (fqa_rows(lp) - lp) / L + 1; }

// the same decomposition in the f16x2 H pipeline (hgemm.hpp hfqa_kernel): A and the head-major weight as H images, o as H image
template <int LP, int DP, int HPW>
int launch_hfqa(const FqaParams& f, hipStream_t s) {
    using FT = HfqaTile<LP, DP, HPW>;
    static_assert(FT::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    if (f.heads % HPW) return fail(PAFUSE_E_SHAPE, "fused qkv-attention: %d heads do not split into groups of %d", f.heads, HPW);
    auto k = hfqa_kernel<LP, DP, HPW>;
    if (FT::LDS_BYTES > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FT::LDS_BYTES);
    }
    if (!f.g.Ah || !f.g.Wh) return fail(PAFUSE_E_ARG, "fused qkv-attention (f16x2) without the H images of its operands");
    if (f.g.M >= (int64_t)1 << 31 || f.nseq >= (int64_t)1 << 31) return fail(PAFUSE_E_ARG, "fused qkv-attention: more than 2^31 tokens");
    const int64_t ntiles = (f.nseq + f.nseq_tile - 1) / f.nseq_tile;
    const int64_t blocks = (ntiles + 7) / 8 * 8 * (f.heads / HPW);
    if (blocks <= 0 || blocks > 0x7fffffff) return fail(PAFUSE_E_ARG, "fused qkv-attention grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FT::NTHR), FT::LDS_BYTES, s, f);
    return check_launch("hfqa_kernel");
}

// ... and in the bf16x3 X pipeline (xgemm.hpp xfqa_kernel): A and the head-major weight as X images, o as X image
template <int LP, int DP, int HPW>
int launch_xfqa(const FqaParams& f, hipStream_t s) {
    using FT = XfqaTile<LP, DP, HPW>;
    static_assert(2 * FT::LDS_BYTES <= 160 * 1024, "two workgroups per CU");
    if (f.heads % HPW) return fail(PAFUSE_E_SHAPE, "fused qkv-attention: %d heads do not split into groups of %d", f.heads, HPW);
    auto k = xfqa_kernel<LP, DP, HPW>;
    if (FT::LDS_BYTES > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FT::LDS_BYTES);
    }
    if (!f.g.Ah || !f.g.Wh) return fail(PAFUSE_E_ARG, "fused qkv-attention (bf16x3 images) without the X images of its operands");
    if (f.g.K % 32 || f.g.K <= 0) return fail(PAFUSE_E_SHAPE, "fused qkv-attention: K=%d must be a positive multiple of 32", f.g.K);
    if (f.g.M >= (int64_t)1 << 31 || f.nseq >= (int64_t)1 << 31) return fail(PAFUSE_E_ARG, "fused qkv-attention: more than 2^31 tokens");
    const int64_t ntiles = (f.nseq + f.nseq_tile - 1) / f.nseq_tile;
    const int64_t blocks = (ntiles + 7) / 8 * 8 * (f.heads / HPW);
    if (blocks <= 0 || blocks > 0x7fffffff) return fail(PAFUSE_E_ARG, "fused qkv-attention grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(FT::NTHR), FT::LDS_BYTES, s, f);
    return check_launch("xfqa_kernel");
}

int fused_qkv_attention(const FqaParams& f, hipStream_t s) {
    if (f.nseq <= 0) return PAFUSE_OK;
    const int lp = fqa_lp(f.L, f.g.bf16), dp = fqa_dp(f.d);
    if (f.g.bf16 == 4) {
        if (lp == 80 && dp == 32) return launch_xfqa<80, 32, 1>(f, s);
        if (lp == 32 && dp == 48) return launch_xfqa<32, 48, 2>(f, s);
        if (lp == 32 && dp == 32) return f.heads % 2 ? launch_xfqa<32, 32, 1>(f, s) : launch_xfqa<32, 32, 2>(f, s);
        if (lp == 48 && dp == 32) return f.heads % 2 ? launch_xfqa<48, 32, 1>(f, s) : launch_xfqa<48, 32, 2>(f, s);
        return fail(PAFUSE_E_SHAPE, "fused qkv-attention: no kernel for L=%d, d=%d", f.L, f.d);
    }
    if (f.g.bf16 == 3) {
        if (lp == 80 && dp == 32) return launch_hfqa<80, 32, 1>(f, s);
        // head dim <= 32: two heads per workgroup share the A stream (the ring grows to the 80 KB two workgroups per CU allow)
        if (lp == 32 && dp == 48) return launch_hfqa<32, 48, 1>(f, s);
        if (lp == 32 && dp == 32) return f.heads % 2 ? launch_hfqa<32, 32, 1>(f, s) : launch_hfqa<32, 32, 2>(f, s);
        if (lp == 48 && dp == 32) return f.heads % 2 ? launch_hfqa<48, 32, 1>(f, s) : launch_hfqa<48, 32, 2>(f, s);
        return fail(PAFUSE_E_SHAPE, "fused qkv-attention: no kernel for L=%d, d=%d", f.L, f.d);
    }
    if (lp == 32 && dp == 48) return launch_fqa<32, 48>(f, s);
    if (lp == 32 && dp == 32) return launch_fqa<32, 32>(f, s);
    if (lp == 48 && dp == 32) return launch_fqa<48, 32>(f, s);
    return fail(PAFUSE_E_SHAPE, "fused qkv-attention: no kernel for L=%d, d=%d", f.L, f.d);
}

// fc1 -> GELU -> fc2 -> whole-row epilogue in one kernel (hgemm.hpp hmlp_kernel): f16x2, folded LayerNorm, H-image residual only
static bool hmlp_has(int C, int hidden) { return hidden == 2 * C && (C == 224 || C == 256 || C == 384); }
template <int NT2, int MINW>
int launch_hmlp(const MlpParams& m, hipStream_t s) {
    using T = MlpTile<NT2>;
    static_assert(T::LDS_BYTES <= (MINW == 2 ? 80 : 160) * 1024, "LDS budget");
    auto k = hmlp_kernel<NT2, MINW>;
    {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES);
    }
    const int64_t tiles = (m.g.M + 127) / 128;
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "fused MLP grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(256), T::LDS_BYTES, s, m);
    return check_launch("hmlp_kernel");
}
int fused_mlp(const MlpParams& m, hipStream_t s) {
    if (m.g.M <= 0) return PAFUSE_OK;
    if (!m.g.Ah || !m.g.Wh || !m.W1h || !m.bias1 || !m.g.resid_h) return fail(PAFUSE_E_ARG, "fused MLP without its H images");
    if (m.g.K != 2 * m.g.N) return fail(PAFUSE_E_SHAPE, "fused MLP: hidden width %d is not twice the channel width %d", m.g.K, m.g.N);
    switch (m.g.N) {
        case 224: return launch_hmlp<7, 2>(m, s);
        case 256: return launch_hmlp<8, 2>(m, s);
        case 384: return launch_hmlp<12, 1>(m, s);
        default: return fail(PAFUSE_E_SHAPE, "fused MLP: no kernel for channel width %d", m.g.N);
    }
}

// the qkv layers' kernel on v_mfma_f32_16x16x32_bf16 (kernels.hpp gemm16_tile); the image must be in the M16 layout
template <int NB, int MINW>
int launch_gemm16(const GemmParams& p, hipStream_t s) {
    using T = Tile16<NB>;
    static_assert(T::STAGE_BYTES <= 64 * 1024, "LDS budget without the attribute");
    if (!p.Wsplit) return fail(PAFUSE_E_ARG, "split-precision GEMM without a pre-split weight image");
    const int64_t tiles = ((p.M + T::BM - 1) / T::BM) * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    hipLaunchKernelGGL((gemm16_kernel<NB, MINW>), dim3((unsigned)tiles), dim3(T::NTHR), T::STAGE_BYTES, s, p);
    return check_launch("gemm16_kernel");
}

// ---- the strip GEMM (sgemm.hpp, round 6): the plain split-precision layers (qkv, fc1) on the software-pipelined persistent kernel.
// A launch is STRIP_WGS_PER_CU workgroups per CU (two fit: 68 - 80 KB of LDS each) walking the tile stream; fewer tiles than slots
// = one workgroup per tile.  The CU count is the current device's (cached per device).
#ifndef PAFUSE_STRIP_WGS_PER_CU
#define PAFUSE_STRIP_WGS_PER_CU 2
#endif
int device_cus() {
    static int cus[DeviceOnce::MAX_DEV] = {};
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= DeviceOnce::MAX_DEV) return 256;
    if (!cus[d]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
        cus[d] = n;
    }
    return cus[d];
}

template <int NB, int FLAGS>
int launch_strip2(const GemmParams& p, hipStream_t s) {
    constexpr int RG = 2, NW = 4;
    using T = StripTile<NB, RG, NW, 2>;
    static_assert(T::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    if (!p.Wsplit) return fail(PAFUSE_E_ARG, "split-precision GEMM without a pre-split weight image");
    auto k = sgemm2_kernel<NB, RG, NW, SEPI_BIAS, 2, FLAGS>;
    static DeviceOnce once;
    if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES);
    const int64_t tiles = ((p.M + T::BM - 1) / T::BM) * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    int64_t grid = (int64_t)device_cus() * PAFUSE_STRIP_WGS_PER_CU / 8 * 8;   // a multiple of 8: virtual workgroup v = b + j grid keeps b's XCD
    if (grid >= tiles || grid <= 0) grid = tiles;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(T::NTHR), T::LDS_BYTES, s, p);
    return check_launch("sgemm2_kernel");
}
// which plain layers the strip kernel takes: an M16 image, at most one training epilogue option (alone), at least four 32-deep chunks (the deferred
// stores of a tile go out inside the first four chunks of the next one), a column count one of its tiles divides
bool strip2_ok(const GemmParams& p) {
    if ((p.out_act || p.dact_u) && (p.ln_in || p.act || (p.out_act && p.dact_u))) return false;   // training options: alone, on a plain product
    return p.bf16 == 2 && p.wlayout == 2 && p.K % 32 == 0 && p.K >= 128 && p.M > 0 &&
           p.M * (int64_t)p.N < (int64_t)1 << 31 && p.M * (int64_t)p.K < (int64_t)1 << 31 &&   // (32-bit float offsets of its rows)
           (p.N % 128 == 0 || p.N % 112 == 0 || p.N % 96 == 0);
}
template <int NB>
int launch_strip2_flags(const GemmParams& p, hipStream_t s) {
    const int flags = (p.ln_in ? 1 : 0) | (p.act ? 2 : 0) | (p.dact_u ? 4 : 0) | (p.out_act ? 8 : 0);
    switch (flags) {
        case 0: return launch_strip2<NB, 0>(p, s);
        case 1: return launch_strip2<NB, 1>(p, s);
        case 2: return launch_strip2<NB, 2>(p, s);
        case 3: return launch_strip2<NB, 3>(p, s);
        case 4: return launch_strip2<NB, 4>(p, s);   // training: dX of fc2
        case 8: return launch_strip2<NB, 8>(p, s);   // training: fc1 forward
        default: return fail(PAFUSE_E_ARG, "strip GEMM: epilogue options %d", flags);
    }
}
int strip2_bias(const GemmParams& p, hipStream_t s) {
    if (p.N % 128 == 0) return launch_strip2_flags<8>(p, s);    // body 1152 / 768, hands 768 / 512: 128 x 128
    if (p.N % 112 == 0) return launch_strip2_flags<7>(p, s);    // face 672 / 448: 128 x 112
    return launch_strip2_flags<6>(p, s);                         // 128 x 96 (the single-model variant: 864 = 9 x 96)
}

// split-precision GEMM, LDS-DMA pipelined form (one workgroup per CU, NSTAGE ring of 32-deep chunks)
// K-chunk depth of the pre-split image of a weight (the image format is a property of the weight, fixed when it is
// split, so every launch on it - any M - must use a kernel of that depth): the whole-row layers of widths 384, 288, 256
// and 224 run on the LDS-DMA kernel with 16-deep chunks, everything else on 32-deep chunks.
int wsplit_chunk(int N, bool whole_row) { return (whole_row && (N == 384 || N == 288 || N == 256 || N == 224)) ? 16 : 32; }

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW, int BKC = 32>
int launch_gemm_dma(const GemmParams& p, hipStream_t s) {
    using T = DmaTile<WM, WN, NT, BKC>;
    constexpr size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(EPI == EPI_BIAS || (size_t)7 * T::BM * WN * sizeof(float) <= lds, "cross-wave reduction scratch must fit");
    if (!p.Wsplit) return fail(PAFUSE_E_ARG, "split-precision GEMM without a pre-split weight image");
    if (p.K % BKC) return fail(PAFUSE_E_SHAPE, "split GEMM: K=%d is not a multiple of %d", p.K, BKC);
    auto k = gemm_dma_kernel<WM, WN, NT, EPI, NSTAGE, MINW, 0, BKC>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, s, p);
    return check_launch("gemm_dma_kernel");
}

// ---- f16x2 "H pipeline" (hgemm.hpp): both operands arrive as H images; one workgroup of WM x WN waves per CU
template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, int MINW, bool HRES = false>
int launch_hgemm(const GemmParams& p, hipStream_t s) {
    using T = HTile<WM, WN, NT, BKC>;
    constexpr size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(EPI == EPI_BIAS || (size_t)7 * T::BM * WN * sizeof(float) <= lds, "cross-wave reduction scratch must fit");
    if (!p.Ah || !p.Wh) return fail(PAFUSE_E_ARG, "f16x2 GEMM without the H images of its operands");
    if (p.K % BKC || p.K <= 0 || p.N % T::BN) return fail(PAFUSE_E_SHAPE, "f16x2 GEMM: N=%d, K=%d do not fit the %d-column tile", p.N, p.K, T::BN);
    if (HRES && !p.resid_h) return fail(PAFUSE_E_ARG, "f16x2 whole-row GEMM: no H-image residual");
    auto k = hgemm_kernel<WM, WN, NT, EPI, NSTAGE, BKC, MINW, HRES>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, s, p);
    return check_launch("hgemm_kernel");
}

// widths the f16x2 kernels serve: the PAFUSE parts' (whole-row layers), and every plain-layer width that tiles by 128 or 224
bool hgemm_width(int C) { return C == 384 || C == 256 || C == 224; }
bool hgemm_plain_n(int N) { return N > 0 && (N % 128 == 0 || N % 224 == 0); }

// tile choice per shape: tools/hgemm_bench.hip (profiles/r04_hgemm_bench_*.log).  With three products per k these layers sit
// between the matrix, the L2 -> LDS and the HBM bound (a whole-row launch moves 4 M C floats for 6 M C K MFMA-flops), so the
// winners are the tiles that keep MORE workgroups per CU in different phases, not the tallest ones.
int hgemm_bias(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return PAFUSE_OK;
    if (p.N % 128 == 0) return launch_hgemm<4, 2, 2, EPI_BIAS, 3, 16, 2>(p, s);   // 128 x 128, eight waves, three per CU
    if (p.N % 224 == 0) return launch_hgemm<4, 1, 7, EPI_BIAS, 3, 16, 2>(p, s);   // 128 x 224 (the face: 672 = 3 x 224, 448 = 2 x 224)
    return fail(PAFUSE_E_SHAPE, "f16x2 linear: N=%d must be a multiple of 128 or 224", p.N);
}

int hgemm_rowln(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return PAFUSE_OK;
    if (p.resid_h)   // the residual stream as its (centred) H image: the default of the folded-LayerNorm pipeline
        switch (p.N) {
            case 384: return launch_hgemm<4, 2, 6, EPI_ROWLN, 2, 32, 1, true>(p, s);
            case 256: return launch_hgemm<2, 2, 4, EPI_ROWLN, 2, 32, 2, true>(p, s);
            case 224: return launch_hgemm<4, 1, 7, EPI_ROWLN, 3, 16, 2, true>(p, s);
            default: break;
        }
    switch (p.N) {
        case 384: return launch_hgemm<4, 2, 6, EPI_ROWLN, 2, 32, 1>(p, s);   // 128 rows, eight waves, 128 KB ring: one per CU
        case 256: return launch_hgemm<2, 2, 4, EPI_ROWLN, 2, 32, 2>(p, s);   // 64 rows, four waves, 32-deep chunks, two per CU
        case 224: return launch_hgemm<4, 1, 7, EPI_ROWLN, 3, 16, 2>(p, s);   // 128 rows, four waves (7 column blocks do not split), two per CU
        default: return fail(PAFUSE_E_SHAPE, "no f16x2 whole-row kernel for channel width %d (have 224, 256, 384)", p.N);
    }
}

// ---- bf16x3 "X pipeline" (xgemm.hpp): both operands arrive as X images
template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, int MINW, bool HRES = false>
int launch_xgemm(const GemmParams& p, hipStream_t s) {
    using T = XTile<WM, WN, NT, BKC>;
    constexpr size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static_assert(EPI == EPI_BIAS || (size_t)7 * T::BM * WN * sizeof(float) <= lds, "cross-wave reduction scratch must fit");
    if (!p.Ah || !p.Wh) return fail(PAFUSE_E_ARG, "bf16x3 image GEMM without the X images of its operands");
    if (p.K % 32 || p.K <= 0 || p.N % T::BN) return fail(PAFUSE_E_SHAPE, "bf16x3 image GEMM: N=%d, K=%d do not fit the %d-column tile", p.N, p.K, T::BN);
    if (HRES && !p.resid_h) return fail(PAFUSE_E_ARG, "bf16x3 whole-row GEMM: no X-image residual");
    auto k = xgemm_kernel<WM, WN, NT, EPI, NSTAGE, BKC, MINW, HRES>;
    if (lds > 64 * 1024) {
        static DeviceOnce once;
        if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    if (tiles <= 0 || tiles > 0x7fffffff) return fail(PAFUSE_E_ARG, "gemm grid out of range");
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, s, p);
    return check_launch("xgemm_kernel");
}

// tile choice per shape: tools/xgemm_bench.hip (profiles/r05_xgemm_bench.log).  With six products per k the K loops of these
// tiles are balanced between the matrix pipe and the L2 -> LDS stream (a 128 x 128 tile asks for 24 KB per 768 matrix cycles,
// the stream gives 25 - 32 B per cycle and CU; in-kernel stamps: 70 - 80 % matrix-pipe use inside the loops).  What is left is per
// tile - first-chunk latency, epilogue, the last partial round of a launch -, so the winners are the tiles that keep MORE
// workgroups per CU in different phases (four-wave tiles on two-stage rings of 16-deep chunks: three per CU).
int xgemm_bias(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return PAFUSE_OK;
    if (p.N % 128 == 0) return launch_xgemm<4, 1, 4, EPI_BIAS, 2, 16, 3>(p, s);   // 128 x 128, four waves, 48 KB ring: three per CU
    if (p.N % 224 == 0) return launch_xgemm<4, 1, 7, EPI_BIAS, 2, 16, 2>(p, s);   // 128 x 224 (the face: 448 = 2 x 224), 66 KB: two per CU
    if (p.N % 96 == 0) return launch_xgemm<4, 1, 3, EPI_BIAS, 3, 16, 2>(p, s);    // 128 x 96 (unit tests: 672, 96)
    return fail(PAFUSE_E_SHAPE, "bf16x3 image linear: N=%d must be a multiple of 128, 224 or 96", p.N);
}

int xgemm_rowln(const GemmParams& p, hipStream_t s) {
    if (p.M <= 0) return PAFUSE_OK;
    if (p.resid_h)   // the residual stream as its (centred) X image: the default of the folded-LayerNorm pipeline
        switch (p.N) {
            case 384: return launch_xgemm<4, 2, 6, EPI_ROWLN, 3, 16, 1, true>(p, s);   // 128 rows, eight waves, 144 KB ring: one per CU
            case 256: return launch_xgemm<3, 2, 4, EPI_ROWLN, 3, 16, 1, true>(p, s);   // 96 rows, six waves, 99 KB: one per CU (473 tiles: 1.85 rounds)
            case 224: return p.K <= 256 ? launch_xgemm<3, 1, 7, EPI_ROWLN, 2, 16, 2, true>(p, s)     // proj: 96 rows, three waves, 60 KB: two per CU
                                        : launch_xgemm<3, 1, 7, EPI_ROWLN, 3, 16, 2, true>(p, s);    // fc2 (K = 448): a three-stage ring, one per CU
            default: break;
        }
    switch (p.N) {
        case 384: return launch_xgemm<4, 2, 6, EPI_ROWLN, 3, 16, 1>(p, s);
        case 256: return launch_xgemm<2, 2, 4, EPI_ROWLN, 2, 16, 2>(p, s);
        case 224: return launch_xgemm<4, 1, 7, EPI_ROWLN, 2, 16, 2>(p, s);
        default: return fail(PAFUSE_E_SHAPE, "no bf16x3 image whole-row kernel for channel width %d (have 224, 256, 384)", p.N);
    }
}

// diagnostic switch, compiled only into -DPAFUSE_DIAG builds (tools/): environment PAFUSE_DEBUG_F32_MASK, read once -
// bit 0 = plain linear layers, bit 1 = whole-row layers fall back to the fp32 matrix cores even in split-precision mode.
// The shipped library reads no environment variable: nothing outside its arguments changes which kernels it runs.
#ifdef PAFUSE_DIAG
int debug_f32_mask() {
    static const int mask = [] {
        const char* e = getenv("PAFUSE_DEBUG_F32_MASK");
        return e ? atoi(e) : 0;
    }();
    return mask;
}
#else
constexpr int debug_f32_mask() { return 0; }
#endif

// A/B build switch (tools/build_variant.py): workgroups per CU the 128 x 128 plain split tile is compiled for.  3 fits (152
// registers, 43 KB of LDS) and measured slower: 98.5 - 100.2 us per fc1 launch against 95.7 - 97.3 at 2 (round 5, same GPU call).
#ifndef PAFUSE_FC1_MINW
#define PAFUSE_FC1_MINW 2
#endif
int gemm_bias(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    if (p.bf16 == 2 && (debug_f32_mask() & 1)) p.bf16 = 0;
    if (p.bf16 == 2 && (debug_f32_mask() & 4) && !p.act) p.bf16 = 0;   // qkv only
    if (p.bf16 == 2 && (debug_f32_mask() & 8) && p.act) p.bf16 = 0;    // fc1 only
    if (p.bf16 == 2 && (debug_f32_mask() & 16)) {                        // split arithmetic, small 128x64 tiles only
        if (p.N % 64 == 0) return launch_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>(p, s);
    }
    if (p.M <= 0) return PAFUSE_OK;
    if (p.K % BK || p.N % 32 || p.K <= 0 || p.N <= 0)
        return fail(PAFUSE_E_SHAPE, "linear: N=%d K=%d must be positive multiples of 32", p.N, p.K);
    // the training epilogue options live in the slab epilogue of gemm_tile (every N % 64 == 0 route below ends there) and in gemm16_tile's
    if ((p.out_act || p.dact_u) && ((p.N % 64 && !(p.bf16 == 2 && p.wlayout == 2)) || p.bf16 > 2 || p.ln_in || p.act))
        return fail(PAFUSE_E_ARG, "linear: out_act / dact_u need N %% 64 == 0 and a plain fp32 or bf16x3 product");
    if (p.bf16 == 3) return hgemm_bias(p, s);   // f16x2: the H pipeline
    if (p.bf16 == 4) return xgemm_bias(p, s);   // bf16x3 on images: the X pipeline
    // small accumulators + single LDS stage = 4-5 independent workgroups per CU, which hides the per-tile
    // prologue/epilogue (measured with tools/gemm_bench.hip: 128x64 tiles reach 72-74 % of the f32 MFMA peak at the
    // qkv shape, 128x96/double-buffered 65-67 %, 128x128 58-60 %)
#ifdef PAFUSE_QKV_32X32   // A/B build (tools/): the qkv layers on the 32x32x16 tiles, their images in layout 0
    p.wlayout = 0;
#endif
#ifndef PAFUSE_NO_STRIP
    if (strip2_ok(p)) return strip2_bias(p, s);   // qkv, fc1 (+ GELU): the pipelined persistent strip kernel (sgemm.hpp; same bits as gemm16_tile)
#endif
    if (p.bf16 == 2 && p.wlayout == 2) {  // 16x16x32 MFMAs on the M16 image (kernels.hpp gemm16_tile): training's epilogue options, short K
        if (p.N % 128 == 0) return launch_gemm16<8, 3>(p, s);    // body 1152, hands 768: 128 x 128 tiles, 3 workgroups per CU
        if (p.N % 96 == 0) return launch_gemm16<6, 3>(p, s);     // face 672, single-model 864: 128 x 96
        if (p.N % 112 == 0) return launch_gemm16<7, 3>(p, s);    // face 224, 448 (training: every plain GEMM of the face): 128 x 112
        if (p.N % 64 == 0) return launch_gemm16<4, 4>(p, s);
        return launch_gemm16<2, 4>(p, s);
    }
    if (p.bf16 == 2) {  // split precision (bf16x3): fp32-equivalent products on the bf16 matrix cores
        // measured per shape with tools/gemm_bench.hip (profiles/r02_gemm_bench_split_v1.log): 128x128 tiles at two
        // workgroups per CU where N allows (159 TFLOP/s at the body qkv shape), 128x64 at four otherwise (148)
        if (p.N % 128 == 0 && p.M >= 4096) return launch_gemm<4, 1, 4, EPI_BIAS, 1, PAFUSE_FC1_MINW, 0, 2>(p, s);
        if (p.N % 64 == 0) return launch_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>(p, s);
        if (p.N % 96 == 0) return launch_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 2>(p, s);
        return launch_gemm<4, 1, 1, EPI_BIAS, 1, 1, 0, 2>(p, s);
    }
    if (p.bf16) {  // opt-in bf16-operand mode: same tiles, bf16 MFMA
        if (p.N % 64 == 0) return launch_gemm<4, 1, 2, EPI_BIAS, 1, 5, 0, 1>(p, s);
        if (p.N % 96 == 0) return launch_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 1>(p, s);
        return launch_gemm<4, 1, 1, EPI_BIAS, 1, 1, 0, 1>(p, s);
    }
    if (p.N % 64 == 0) return launch_gemm<4, 1, 2, EPI_BIAS, 1, 5>(p, s);   // pin 5 waves/SIMD (98 registers)
    if (p.N % 96 == 0) return launch_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1>(p, s);  // row-per-lane epilogue wins at NT=3
    return launch_gemm<4, 1, 1, EPI_BIAS, 1>(p, s);
}

template <int EPI>
int gemm_rowln_as(const GemmParams& p0, hipStream_t s) {
    GemmParams p = p0;
    if (p.bf16 == 2 && (debug_f32_mask() & 2)) p.bf16 = 0;
    if (p.M <= 0) return PAFUSE_OK;
    if (p.K % BK || p.K <= 0) return fail(PAFUSE_E_SHAPE, "rowln: K=%d must be a positive multiple of 32", p.K);
    if (p.bf16 == 3 || p.bf16 == 4) {   // f16x2: the H pipeline; bf16x3 on images: the X pipeline (inference only)
        if constexpr (EPI == EPI_ROWLN) return p.bf16 == 3 ? hgemm_rowln(p, s) : xgemm_rowln(p, s);
        else return fail(PAFUSE_E_ARG, "training has no image-pipeline kernels");
    }
    if constexpr (EPI == EPI_ROWLN_TRAIN) {   // training forward with split products: the three PAFUSE widths (train_host.inc)
        if (p.bf16 == 2) {
            switch (p.N) {
                case 384: return launch_gemm_dma<2, 2, 6, EPI, 2, 2, 16>(p, s);
                case 256: return launch_gemm_dma<2, 2, 4, EPI, 2, 2, 16>(p, s);
                case 224: if (p.M >= 4096) return launch_gemm_dma<4, 1, 7, EPI, 2, 2, 16>(p, s); break;
                default: break;
            }
            p.bf16 = 0;   // no split whole-row tile for this shape: the fp32 kernels below
        }
    }
    if constexpr (EPI == EPI_ROWLN) {
        if (p.bf16 == 2) {
            switch (p.N) {
                // picked per width with tools/gemm_bench.hip (profiles/r02_gemm_bench_*.log): the LDS-DMA kernel on 16-deep
                // chunks, four waves, a two-stage ring and TWO workgroups per CU - the LayerNorm epilogue of one tile
                // (a third of a tile's time when nothing overlaps it) runs under the K loop of the other.  The same
                // variants serve the grouped launches below, so both routes give bit-identical rows.
                // (round 5, with the slab epilogue: 128-row eight-wave tiles - half the weight re-reads - measured again: proj 62 vs 58 us,
                // fc2 104 vs 98: the two-workgroup tiles stay)
                case 384: return launch_gemm_dma<2, 2, 6, EPI, 2, 2, 16>(p, s);
                case 256: return launch_gemm_dma<2, 2, 4, EPI, 2, 2, 16>(p, s);
                case 288: return launch_gemm_dma<2, 3, 3, EPI, 2, 2, 16>(p, s);  // the single-model variant (cs = 288)
                case 224:
                    return p.M >= 4096 ? launch_gemm_dma<4, 1, 7, EPI, 2, 2, 16>(p, s) : launch_gemm_dma<2, 1, 7, EPI, 2, 2, 16>(p, s);
                case 128: return launch_gemm<1, 4, 1, EPI, 1, 1, 1, 2>(p, s);
                case 64: return launch_gemm<1, 2, 1, EPI, 1, 1, 1, 2>(p, s);
                default: break;
            }
        } else if (p.bf16) {
            switch (p.N) {
                case 384: return launch_gemm<1, 4, 3, EPI, 1, 1, 1, 1>(p, s);
                case 288: return launch_gemm<1, 3, 3, EPI, 1, 1, 1, 1>(p, s);
                case 256: return launch_gemm<2, 2, 4, EPI, 1, 2, 1, 1>(p, s);
                case 224: return launch_gemm<1, 7, 1, EPI, 1, 1, 1, 1>(p, s);
                case 128: return launch_gemm<1, 4, 1, EPI, 1, 1, 1, 1>(p, s);
                case 64: return launch_gemm<1, 2, 1, EPI, 1, 1, 1, 1>(p, s);
                default: break;
            }
        }
    }
    switch (p.N) {
        // row-per-lane accumulators (TR): LayerNorm statistics are in-lane sums + one shuffle + a tiny cross-wave
        // exchange, all global traffic is dwordx4, no LDS transposition (picked with tools/gemm_bench.hip)
        case 384: return launch_gemm<1, 4, 3, EPI, 1, 1, 1>(p, s);
        case 288: return launch_gemm<1, 3, 3, EPI, 1, 1, 1>(p, s);
        case 256: return launch_gemm<2, 2, 4, EPI, 1, 2, 1>(p, s);  // (MINW 3 capped it at 168 VGPRs: 6 spilled, -2 %)
        case 224: return launch_gemm<1, 7, 1, EPI, 1, 1, 1>(p, s);
        case 128: return launch_gemm<1, 4, 1, EPI, 1, 1, 1>(p, s);
        case 64: return launch_gemm<1, 2, 1, EPI, 1, 1, 1>(p, s);
        default: return fail(PAFUSE_E_SHAPE, "no whole-row kernel for channel width %d (have 64,128,224,256,288,384)", p.N);
    }
}

int gemm_rowln(const GemmParams& p, hipStream_t s) { return gemm_rowln_as<EPI_ROWLN>(p, s); }
// training forward: the same kernels with the DropPath row factor and the pre-norm sum as an extra output
int gemm_rowln_train(const GemmParams& p, hipStream_t s) { return gemm_rowln_as<EPI_ROWLN_TRAIN>(p, s); }

// ---- grouped launches: the same layer of several independent parts in one grid (kernels.hpp, grouped_*_kernel).
// Only the split-precision mode has them (the fp32 mode overlaps parts on streams instead); anything a grouped kernel
// has no variant for goes part by part through the dispatch above - same tiles, same arithmetic, same results.
bool group_ok_rowln(const GemmParams& p) {
    return p.bf16 == 2 && p.Wsplit && (p.N == 384 || p.N == 256 || (p.N == 224 && p.M >= 4096)) && p.K % 16 == 0 && p.K > 0 &&
           !(debug_f32_mask() & 2);
}
// (the plain layers - qkv, fc1 - have no shared grid since round 6: the persistent strip kernel fills the chip from one part's tiles;
// grouped_bias_kernel, the shared 32x32x16 grid of rounds 3 - 5, is gone)
// (whether the parts of a configuration share grids is the caller's choice per call: pafuse_d3dp_config.part_by_part_launches -
// same tiles, same arithmetic, same bits either way; there is no process-wide schedule state)

template <bool ROWLN>
int gemm_group(const GemmParams* ps, int n, hipStream_t s, bool shared_grids) {
    bool ok = ROWLN && shared_grids && n >= 2 && n <= GROUP_MAX;
    for (int i = 0; i < n && ok; ++i) ok = ps[i].M > 0 && group_ok_rowln(ps[i]);
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const int rc = ROWLN ? gemm_rowln(ps[i], s) : gemm_bias(ps[i], s);
            if (rc) return rc;
        }
        return PAFUSE_OK;
    }
    // most expensive tiles first (the tail of the grid is then made of the cheapest ones)
    int order[GROUP_MAX];
    double cost[GROUP_MAX];
    int64_t tiles[GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        const GemmParams& p = ps[i];
        int bm, bn;
        bm = p.N == 224 ? 128 : 64, bn = p.N;
        tiles[i] = (p.M + bm - 1) / bm * (p.N / bn);
        cost[i] = (double)bm * bn * p.K;
        order[i] = i;
    }
    std::sort(order, order + n, [&](int a, int b) { return cost[a] > cost[b]; });
    GroupedGemmParams g{};
    g.n = n;
    int64_t first = 0;
    for (int k = 0; k < n; ++k) {
        g.p[k] = ps[order[k]];
        g.first[k] = (int)first;
        first += (tiles[order[k]] + 7) / 8 * 8;
        if (first > 0x7fffffff) return fail(PAFUSE_E_ARG, "grouped gemm grid out of range");
    }
    g.first[n] = (int)first;
    for (int k = n + 1; k <= GROUP_MAX; ++k) g.first[k] = (int)first;
    constexpr size_t lds = 2 * DmaTile<2, 2, 6, 16>::STAGE_BYTES;  // the widest variant's ring
    static_assert(lds >= 2 * DmaTile<2, 2, 4, 16>::STAGE_BYTES && lds >= 2 * DmaTile<4, 1, 7, 16>::STAGE_BYTES && lds <= 80 * 1024, "LDS");
    auto k = grouped_rowln_kernel<EPI_ROWLN>;
    static DeviceOnce once;
    if (once.first()) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3((unsigned)first), dim3(256), lds, s, g);
    return check_launch("grouped_rowln_kernel");
}

bool width_supported(int C) { return C == 384 || C == 288 || C == 256 || C == 224 || C == 128 || C == 64; }

// ------------------------------------------------------------------------------------------- attention dispatch
template <int LP, int DP, int NW>
int launch_attn(const AttnParams& p, hipStream_t s) {
    constexpr int ITEMS = NW / (LP / 16);
    constexpr size_t lds = (size_t)2 * ITEMS * LP * (DP + 4) * sizeof(float);
    const int64_t nitems = p.nseq * p.heads;
    const int64_t grid = (nitems + ITEMS - 1) / ITEMS;
    if (grid > 0x7fffffff) return fail(PAFUSE_E_ARG, "attention grid out of range");
    hipLaunchKernelGGL((attn_kernel<LP, DP, NW>), dim3((unsigned)grid), dim3(NW * 64), lds, s, p);
    return check_launch("attn_kernel");
}

int attention(const AttnParams& p, hipStream_t s) {
    if (p.nseq <= 0) return PAFUSE_OK;
#ifdef PAFUSE_ABL_NO_ATTENTION   // timing ablation (tools/): what the loop costs with every attention launch removed (results wrong)
    return PAFUSE_OK;
#endif
    if (p.d % 4 || p.d > 48 || p.L > 144 || p.L <= 0)
        return fail(PAFUSE_E_SHAPE, "attention: head dim %d (need %%4, <=48) / length %d (need <=144)", p.d, p.L);
    const int dp = p.d <= 32 ? 32 : 48;
    const int lp = p.L <= 32 ? 32 : (p.L <= 48 ? 48 : (p.L <= 80 ? 80 : 144));
    if (lp == 144)  // the single-model variant: one sequence of all 134 keypoints (common/diffusionpose.py:150-153)
        return dp == 32 ? launch_attn<144, 32, 9>(p, s) : launch_attn<144, 48, 9>(p, s);
    if (dp == 32) {
        if (lp == 32) return launch_attn<32, 32, 4>(p, s);
        if (lp == 48) return launch_attn<48, 32, 6>(p, s);
        return launch_attn<80, 32, 5>(p, s);
    }
    if (lp == 32) return launch_attn<32, 48, 4>(p, s);
    if (lp == 48) return launch_attn<48, 48, 6>(p, s);
    return launch_attn<80, 48, 5>(p, s);
}

// --------------------------------------------------------------------------------------- per-part activations
constexpr size_t RANGE_FLAG_BYTES = 256;   // behind pafuse_d3dp_sample's workspace: [0] int32, set when a denoiser output is not finite

struct PartBuffers {
    float *x, *xn, *o, *wide;  // [M,C], [M,C], [M,C], [M,3C] (qkv, then the MLP hidden [M,2C])
    //                            f16x2 mode: xn, o and (as the hidden) wide hold H images of those tensors - same bytes;
    //                            bf16x3 on images (mode 4): X images, 6 bytes per element - xn and o are carved at that size
    //                            (`eb`), the [M,2C] hidden image is exactly the [M,3C] floats of `wide`
    int eb;                    // bytes per element of xn and o: 4, or 6 in mode 4
    float* temb;               // [B,C]
    float* pred;               // [M,3]
    float* stats;              // [M,2] (mean, rstd) of a row when the LayerNorm is folded into the consumer GEMM
};

static int image_elem_bytes(int operand_mode) { return operand_mode == 4 ? 6 : 4; }
size_t part_buffer_bytes(int64_t M, int C, int B, int eb = 4) {
    return align_up((size_t)M * C * 4) + 2 * align_up((size_t)M * C * eb) + align_up((size_t)M * 3 * C * 4) + align_up((size_t)B * C * 4) +
           align_up((size_t)M * 3 * 4) + align_up((size_t)M * 2 * 4);
}

char* carve_part(char* base, int64_t M, int C, int B, PartBuffers& pb, int eb = 4) {
    pb.eb = eb;
    pb.x = (float*)base;
    base += align_up((size_t)M * C * 4);
    pb.xn = (float*)base;
    base += align_up((size_t)M * C * eb);
    pb.o = (float*)base;
    base += align_up((size_t)M * C * eb);
    pb.wide = (float*)base;
    base += align_up((size_t)M * 3 * C * 4);
    pb.temb = (float*)base;
    base += align_up((size_t)B * C * 4);
    pb.pred = (float*)base;
    base += align_up((size_t)M * 3 * 4);
    pb.stats = (float*)base;
    base += align_up((size_t)M * 2 * 4);
    return base;
}

// the rows [row0, ...) of a part's buffers (hypotheses are independent, so a part can be cut into hypothesis groups
// that run on different streams; temb and the scratch behind `wide` stay shared / are offset like the rest)
PartBuffers offset_rows(const PartBuffers& pb, int64_t row0, int C) {
    PartBuffers o = pb;
    const int64_t img = row0 * C * pb.eb / 4;   // (floats; C is a multiple of 32)
    o.x = pb.x + row0 * C, o.xn = pb.xn + img, o.o = pb.o + img, o.wide = pb.wide + row0 * 3 * C;
    o.pred = pb.pred + row0 * 3;
    o.stats = pb.stats + row0 * 2;
    return o;
}

static bool ln_folded(const pafuse_mixste2_weights* w) { return w->ste[0].qkv_lt != nullptr; }   // (qkv_ls / fc1_ls: round 3's uncentred form, not read)
// f16x2 with the LayerNorm folded: the residual stream between the blocks lives in memory as its H image only
static bool h_residual_only(const pafuse_mixste2_weights* w) { return w->operand_bf16 >= 3 && ln_folded(w) && !w->keep_f32_residual; }

// `training`: the training entry points make the images of the weights they multiply themselves (weights change every step)
int check_weights(const pafuse_mixste2_weights* w, bool training = false) {
    if (!w) return fail(PAFUSE_E_ARG, "null weights");
    static_assert(EMBED_NV * 128 >= 384, "embed kernel covers the widest part");
    if (w->in_chans != 5) return fail(PAFUSE_E_SHAPE, "in_chans must be 5, got %d", w->in_chans);
    if (w->depth < 1 || w->depth > PAFUSE_MAX_DEPTH) return fail(PAFUSE_E_SHAPE, "depth %d out of range", w->depth);
    if (!width_supported(w->channels)) return fail(PAFUSE_E_SHAPE, "channel width %d has no kernel", w->channels);
    if (w->heads <= 0 || w->channels % w->heads) return fail(PAFUSE_E_SHAPE, "heads %d !| C %d", w->heads, w->channels);
    const int d = w->channels / w->heads;
    if (d % 4 || d > 48) return fail(PAFUSE_E_SHAPE, "head dim %d unsupported", d);
    if (w->joints < 1 || w->joints > 144 || w->frames < 1 || w->frames > 144)
        return fail(PAFUSE_E_SHAPE, "sequence lengths J=%d F=%d must be in 1..144", w->joints, w->frames);
    if (w->operand_bf16 < 0 || w->operand_bf16 > 4) return fail(PAFUSE_E_ARG, "matrix-product mode %d", w->operand_bf16);
    if (w->operand_bf16 >= 3) {   // the image pipelines: the PAFUSE part widths; plain layers that tile by 128 or 224 columns
        const int hidden = w->mlp_hidden > 0 ? w->mlp_hidden : 2 * w->channels;
        if (!hgemm_width(w->channels) || !hgemm_plain_n(3 * w->channels) || !hgemm_plain_n(hidden) || hidden % 32)
            return fail(PAFUSE_E_SHAPE, "the image pipelines (f16x2, bf16x3 on images) serve the widths 224 / 256 / 384 with an MLP hidden "
                                        "width that is a multiple of 128 or 224 (got C = %d, hidden = %d): use mode 2", w->channels, hidden);
    }
    if (w->operand_bf16 == 4 && !training) {   // bf16x3 on images: qkv + attention fused in every block (no unfused attention on X images)
        for (int i = 0; i < w->depth; ++i)
            for (const pafuse_block_weights* b : {&w->ste[i], &w->tte[i]})
                if (!b->qkv_hs || !b->qkv_hb) return fail(PAFUSE_E_ARG, "bf16x3 on images needs the head-major qkv image of every block (qkv_hs, qkv_hb)");
        if (!xfqa_has(w->joints, d, w->heads) || !xfqa_has(w->frames, d, w->heads))
            return fail(PAFUSE_E_SHAPE, "bf16x3 on images: no fused qkv + attention form for sequences of %d / %d tokens at head dim %d, %d heads: "
                                        "use mode 2", w->joints, w->frames, d, w->heads);
    }
    if (w->mlp_hidden < 0 || (w->mlp_hidden > 0 && (w->mlp_hidden % 32 || w->mlp_hidden > 3 * w->channels)))
        return fail(PAFUSE_E_SHAPE, "mlp hidden width %d must be a multiple of 32 and at most 3C = %d", w->mlp_hidden, 3 * w->channels);
    if (!(w->qk_scale >= 0.f)) return fail(PAFUSE_E_ARG, "qk_scale must be positive (0 = head_dim^-0.5)");
    if (training && (w->operand_bf16 == 1 || w->operand_bf16 >= 3))
        return fail(PAFUSE_E_ARG, "training runs fp32 ('f32') or split-precision ('bf16x3') products");
    if (w->operand_bf16 >= 2 && !training)
        for (int i = 0; i < w->depth; ++i)
            for (const pafuse_block_weights* b : {&w->ste[i], &w->tte[i]})
                if ((!b->qkv_ws && w->operand_bf16 != 4) || !b->proj_ws || !b->fc1_ws || !b->fc2_ws)   // (mode 4 multiplies qkv_hs only)
                    return fail(PAFUSE_E_ARG, "split-precision mode needs the pre-split image of every linear weight "
                                              "(pafuse_split_weights)");
    // a folded LayerNorm (pafuse_block_weights.qkv_lt, fc1_lt) is a property of the whole denoiser: the producer of a block's
    // statistics is the block before it
    const bool fold = ln_folded(w);
    for (int i = 0; i < w->depth; ++i)
        for (const pafuse_block_weights* b : {&w->ste[i], &w->tte[i]}) {
            const int set = (b->qkv_lt != nullptr) + (b->fc1_lt != nullptr);
            if (set != (fold ? 2 : 0)) return fail(PAFUSE_E_ARG, "folded-LayerNorm vectors (qkv_lt, fc1_lt) must be set in every block or in none");
        }
    if (fold && w->operand_bf16 < 2) return fail(PAFUSE_E_ARG, "the folded LayerNorm exists in the split-precision modes only");
    return PAFUSE_OK;
}

// one transformer block on the fixed-layout token matrix; `post`/`next` describe the epilogue of its fc2 GEMM
struct BlockTail {
    const float *post_w, *post_b;
    float post_eps;
    const float* pos;
    int posJ, posF;
    const float *next_w, *next_b;
    float next_eps;
    const float *head_w, *head_b;
    float* out_head;
};

// the five launches of one transformer block, as parameter blocks (run part by part, or grouped across parts)
struct BlockLaunch {
    GemmParams qkv, proj, fc1, fc2;
    AttnParams attn;
    FqaParams fqa;   // qkv + attention in one kernel (fused == true: replaces the qkv and attn launches)
    bool fused;
    MlpParams mlp;   // fc1 -> GELU -> fc2 in one kernel (mlp_fused == true: replaces the fc1 and fc2 launches)
    bool mlp_fused;
};

BlockLaunch make_block(const pafuse_block_weights& bw, const PartBuffers& pb, int64_t M, int C, int heads, int64_t nseq, int L,
                       int64_t group, int64_t group_stride, int64_t seq_stride, int64_t tok_stride, const BlockTail& tail,
                       int bf16, int hidden = 0, float qk_scale = 0.f, bool fold = false, bool keep_f32_residual = false) {
    if (hidden <= 0) hidden = 2 * C;  // mlp_ratio = 2, the PAFUSE configuration
    BlockLaunch b{};
    // fold: the LayerNorm in front of qkv / fc1 is applied inside those GEMMs (GemmParams::ln_in); the producer of a row
    // leaves its (mean, rstd) in the part's stats buffer
    float* const stats = pb.stats;
    // f16x2 (bf16 == 3): every GEMM operand is an H image - xn (of x with the fold, else of LN(x)), o, and the MLP hidden in
    // `wide` are written in that form by their producers (hgemm.hpp); the *_ws pointers are the weights' H images
    const bool hp = bf16 >= 3;   // (mode 4, bf16x3 on images: the same flow on X images - 6 bytes per element, exact)
    uint8_t* const xn_h = reinterpret_cast<uint8_t*>(pb.xn);
    const bool h_residual = hp && fold && !keep_f32_residual;
    // qkv = LN1(x) Wqkv^T + b        (xn already holds LN1(x))                         mixste.py:65
    GemmParams& g = b.qkv;
    g.A = pb.xn, g.W = bw.qkv_w, g.bias = bw.qkv_b, g.out = pb.wide, g.M = M, g.N = 3 * C, g.K = C, g.act = 0;
    g.bf16 = bf16, g.Wsplit = (const uint8_t*)bw.qkv_ws, g.wlayout = 2;   // qkv images are in the M16 layout (include/pafuse_hip.h)
    if (fold) g.A = pb.x, g.ln_in = stats, g.bias = bw.qkv_lt;
    if (hp) g.Ah = xn_h, g.Wh = (const uint8_t*)bw.qkv_ws;
    if (fold) g.ln_s = nullptr;   // x (its image; mode 2: its fp32 rows) is stored centred on the row mean: rstd acc + lt, no mean term
    AttnParams& a = b.attn;
    a.qkv = pb.wide, a.o = pb.o, a.nseq = nseq, a.L = L, a.C = C, a.heads = heads, a.d = C / heads;
    a.group = group, a.group_stride = group_stride, a.seq_stride = seq_stride, a.tok_stride = tok_stride;
    a.scale = qk_scale != 0.f ? qk_scale : 1.0f / sqrtf((float)(C / heads));  // qk_scale or head_dim ** -0.5  mixste.py:52
    if (hp) a.o_h = reinterpret_cast<uint8_t*>(pb.o);
    // the two in one kernel where a head-major image was supplied and the shape has a fused form
    b.fused = false;
    if (bf16 >= 2 && bw.qkv_hs && bw.qkv_hb && (bf16 == 4 ? xfqa_has(L, C / heads, heads) : fqa_has(L, C / heads, bf16))) {   // (qkv_hl, round 3's uncentred term, is not read: not required)
        const int lp = fqa_lp(L, bf16), dp = fqa_dp(C / heads);
        FqaParams& f = b.fqa;
        f.g = g;
        f.g.Wsplit = (const uint8_t*)bw.qkv_hs, f.g.bias = bw.qkv_hb, f.g.ln_s = nullptr;
        if (hp) f.g.Wh = (const uint8_t*)bw.qkv_hs, f.g.ln_s = nullptr;   // (Ah is the qkv launch's: centred; o is written as an H image)
        f.g.N = heads * 3 * dp;
        f.o = pb.o, f.nseq = nseq, f.L = L, f.C = C, f.heads = heads, f.d = C / heads;
        f.nseq_tile = fqa_nseq_tile(L, lp);   // whole sequences per tile, the last one's LP-row key tile inside the buffer
        f.group = group, f.group_stride = group_stride, f.seq_stride = seq_stride, f.tok_stride = tok_stride;
        f.scale = a.scale;
        b.fused = f.nseq_tile * L <= fqa_tile_rows(lp);
    }
    // x = x + o Wproj^T + b ; xn = LN2(x)                                               mixste.py:80,114-115
    GemmParams& pj = b.proj;
    pj.A = pb.o, pj.W = bw.proj_w, pj.bias = bw.proj_b, pj.M = M, pj.N = C, pj.K = C;
    pj.resid = pb.x, pj.out_x = pb.x, pj.out_n = pb.xn;
    pj.next_w = bw.norm2_w, pj.next_b = bw.norm2_b, pj.next_eps = 1e-6f;
    pj.bf16 = bf16, pj.Wsplit = (const uint8_t*)bw.proj_ws;
    if (fold) pj.out_n = nullptr, pj.ln_stats = stats;
    if (hp) {
        pj.Ah = reinterpret_cast<const uint8_t*>(pb.o), pj.Wh = (const uint8_t*)bw.proj_ws;
        // folded: the residual stream lives in memory as the H image of x ONLY - read as the residual, written back in place
        // (a workgroup owns its rows), and it IS the next GEMM's operand; no fp32 x is read or written inside the denoiser
        if (fold && h_residual) pj.out_xh = xn_h, pj.resid_h = xn_h, pj.resid = nullptr, pj.out_x = nullptr;
        else if (fold) pj.out_xh = xn_h;
        else pj.out_n = nullptr, pj.out_nh = xn_h;
    }
    // h = GELU(xn W1^T + b1)                                                            mixste.py:38-39
    GemmParams& f1 = b.fc1;
    f1.A = pb.xn, f1.W = bw.fc1_w, f1.bias = bw.fc1_b, f1.out = pb.wide, f1.M = M, f1.N = hidden, f1.K = C, f1.act = 1;
    f1.bf16 = bf16, f1.Wsplit = (const uint8_t*)bw.fc1_ws, f1.wlayout = 2;   // (round 6: fc1 images are in the M16 layout too)
    if (fold) f1.A = pb.x, f1.ln_in = stats, f1.bias = bw.fc1_lt;
    if (hp) f1.Ah = xn_h, f1.Wh = (const uint8_t*)bw.fc1_ws, f1.out_h = reinterpret_cast<uint8_t*>(pb.wide);
    if (fold) f1.ln_s = nullptr;
    // x = post(x + h W2^T + b2) [+ pos] ; xn = next(x) | head                           mixste.py:41,115,243,250,257
    GemmParams& f2 = b.fc2;
    f2.A = pb.wide, f2.W = bw.fc2_w, f2.bias = bw.fc2_b, f2.M = M, f2.N = C, f2.K = hidden;
    f2.resid = pb.x, f2.out_x = tail.out_head ? nullptr : pb.x, f2.out_n = tail.out_head ? nullptr : pb.xn;
    f2.post_w = tail.post_w, f2.post_b = tail.post_b, f2.post_eps = tail.post_eps;
    f2.pos = tail.pos, f2.posJ = tail.posJ, f2.posF = tail.posF;
    f2.next_w = tail.next_w, f2.next_b = tail.next_b, f2.next_eps = tail.next_eps;
    f2.head_w = tail.head_w, f2.head_b = tail.head_b, f2.out_head = tail.out_head;
    if (!tail.next_w) f2.out_n = nullptr;
    if (fold && tail.next_w && !tail.out_head) f2.out_n = nullptr, f2.ln_stats = stats;   // the head keeps its own LayerNorm
    f2.bf16 = bf16, f2.Wsplit = (const uint8_t*)bw.fc2_ws;
    if (hp) {
        f2.Ah = reinterpret_cast<const uint8_t*>(pb.wide), f2.Wh = (const uint8_t*)bw.fc2_ws;
        if (f2.ln_stats) f2.out_xh = xn_h;                       // folded: the next block's qkv reads the H image of x
        else if (f2.out_n) f2.out_n = nullptr, f2.out_nh = xn_h;  // not folded: the H image of the next LayerNorm's output
        if (fold && h_residual) {
            f2.resid_h = xn_h, f2.resid = nullptr;
            if (f2.out_xh) f2.out_x = nullptr;                    // (the head's block writes neither)
        }
    }
    // the two MLP launches as one kernel: the hidden activations stay in registers (hgemm.hpp hmlp_kernel); fc2_hp is the H
    // image of fc2.weight with the columns of each group of 16 in the order the kernel's accumulators hand them over
    b.mlp_fused = false;
    if (h_residual && bf16 == 3 && bw.fc2_hp && hmlp_has(C, hidden)) {
        MlpParams& m = b.mlp;
        m.g = f2;
        m.g.Ah = xn_h, m.g.Wh = (const uint8_t*)bw.fc2_hp;
        m.W1h = (const uint8_t*)bw.fc1_ws, m.bias1 = f1.bias, m.ln_in = f1.ln_in;
        b.mlp_fused = true;
    }
    return b;
}

// ---- PAFUSE_DIAG builds only: a hash of every intermediate tensor of a denoiser pass, written by a tiny kernel queued right
// behind its producer (tests/cabi/queue_concurrency.c compares the hashes of a concurrent run with the single-stream ones:
// the FIRST slot that differs names the kernel that went wrong on right inputs)
#ifdef PAFUSE_DIAG
thread_local unsigned long long* g_trace = nullptr;   // device array of slots, zeroed by the caller
thread_local int g_trace_n = 0, g_trace_i = 0;
thread_local float* g_snap = nullptr;                 // device buffer [M*C (+ B*C)]: copy of x right behind embed_kernel,
//                                                       then of temb taken between time_embed_kernel and embed_kernel
__global__ void trace_hash_kernel(const uint32_t* data, int64_t words, unsigned long long* slot) {
    unsigned long long h = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256)
        h += (unsigned long long)data[i] * (2 * (unsigned long long)i + 1);      // order-independent sum
    for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(slot, h);
}
void trace(const void* buf, size_t bytes, hipStream_t s) {
    if (!g_trace || g_trace_i >= g_trace_n) return;
    hipLaunchKernelGGL(trace_hash_kernel, dim3(64), dim3(256), 0, s, (const uint32_t*)buf, (int64_t)(bytes / 4), g_trace + g_trace_i++);
}
#define PAFUSE_TRACE(buf, bytes, s) trace(buf, bytes, s)
#else
#define PAFUSE_TRACE(buf, bytes, s)
#endif

// one block of n independent parts: the same layer of every part in one grid where the grouped kernels apply
// (n == 1: the plain per-part launches)
int run_blocks(const BlockLaunch* bl, int n, hipStream_t s, bool gemms_only = false, double* flops = nullptr,
               int* launches = nullptr, int layer_mask = 15, bool shared_grids = true) {
    int rc;
    GemmParams g[GROUP_MAX];
    auto count = [&](int layers) {
        if (!flops) return;
        for (int i = 0; i < n; ++i) *flops += 2.0 * g[i].M * g[i].N * g[i].K;
        *launches += layers;
    };
    auto layer_of = [&](const BlockLaunch* bl, int n, GemmParams BlockLaunch::*which, bool rowln, int bit) -> int {
        if (!(layer_mask & bit)) return PAFUSE_OK;
        const bool grouped = shared_grids && n >= 2;
        bool all = grouped;
        for (int i = 0; i < n; ++i) {
            g[i] = bl[i].*which;
            all = all && rowln && g[i].M > 0 && group_ok_rowln(g[i]);
        }
        const int r = rowln ? gemm_group<true>(g, n, s, shared_grids) : gemm_group<false>(g, n, s, shared_grids);
        if (flops) {
            for (int i = 0; i < n; ++i) *flops += 2.0 * g[i].M * g[i].N * g[i].K;
            *launches += all ? 1 : n;
        }
        for (int i = 0; i < n; ++i) {
            if (rowln) {
                if (g[i].out_x) PAFUSE_TRACE(g[i].out_x, (size_t)g[i].M * g[i].N * 4, s);
                if (g[i].out_n) PAFUSE_TRACE(g[i].out_n, (size_t)g[i].M * g[i].N * 4, s);
            } else {
                PAFUSE_TRACE(g[i].out, (size_t)g[i].M * g[i].N * 4, s);
            }
        }
        return r;
    };
    // qkv -> attention stays part by part: a part's qkv rows (119 - 197 MB at P = 20) are read back by its attention
    // while they are still in the 256 MB Infinity Cache; written for all parts first (455 MB) they are not, and the six
    // attention launches of a block pair get 21 us slower, more than the shared qkv grid saves (9 us; rocprofv3, r02)
    for (int i = 0; i < n; ++i) {
        g[i] = bl[i].qkv;
        if (bl[i].fused) {   // one kernel: the GEMM replay (gemms_only) cannot leave its attention phase out
            if ((layer_mask & 1) && (rc = fused_qkv_attention(bl[i].fqa, s))) return rc;
            PAFUSE_TRACE(bl[i].attn.o, (size_t)g[i].M * bl[i].attn.C * 4, s);
            continue;
        }
        if (g[i].bf16 == 4) return fail(PAFUSE_E_SHAPE, "bf16x3 on images: this block has no fused qkv + attention form");
        if ((layer_mask & 1) && (rc = gemm_bias(g[i], s))) return rc;
        PAFUSE_TRACE(g[i].out, (size_t)g[i].M * g[i].N * 4, s);
        if (!gemms_only && (rc = attention(bl[i].attn, s))) return rc;
        PAFUSE_TRACE(bl[i].attn.o, (size_t)g[i].M * bl[i].attn.C * 4, s);
    }
    if (layer_mask & 1) count(n);
    if ((rc = layer_of(bl, n, &BlockLaunch::proj, true, 2))) return rc;
    // the MLP: one kernel where the part's block has the fused form (a replay of the fc1 layer alone runs it, of fc2 alone
    // nothing of that part), the fc1 and fc2 launches for the other parts
    BlockLaunch rest[GROUP_MAX];
    int nrest = 0;
    for (int i = 0; i < n; ++i) {
        if (!bl[i].mlp_fused) {
            rest[nrest++] = bl[i];
            continue;
        }
        if (!(layer_mask & 4)) continue;
        if ((rc = fused_mlp(bl[i].mlp, s))) return rc;
        if (flops) *flops += 2.0 * (2.0 * bl[i].fc1.M * bl[i].fc1.N * bl[i].fc1.K), *launches += 1;
        if (bl[i].mlp.g.out_xh) PAFUSE_TRACE(bl[i].mlp.g.out_xh, (size_t)bl[i].fc2.M * bl[i].fc2.N * 4, s);
    }
    if (!nrest) return PAFUSE_OK;
    if ((rc = layer_of(rest, nrest, &BlockLaunch::fc1, false, 4))) return rc;
    return layer_of(rest, nrest, &BlockLaunch::fc2, true, 8);
}

int run_block(const pafuse_block_weights& bw, const PartBuffers& pb, int64_t M, int C, int heads, int64_t nseq, int L,
              int64_t group, int64_t group_stride, int64_t seq_stride, int64_t tok_stride, const BlockTail& tail,
              hipStream_t s, bool gemms_only = false, double* flops = nullptr, int* launches = nullptr, int bf16 = 0) {
    const BlockLaunch b = make_block(bw, pb, M, C, heads, nseq, L, group, group_stride, seq_stride, tok_stride, tail, bf16);
    return run_blocks(&b, 1, s, gemms_only, flops, launches);
}

// the 2*depth blocks + head of n independent MixSTE2 denoisers of equal depth on rows already embedded in pb[i].x /
// pb[i].xn; results in pb[i].pred [M_i,3].  Block k of every part is issued together (run_blocks).
int run_mixste_layers_n(const pafuse_mixste2_weights* const* ws, const PartBuffers* pbs, const int64_t* Rs, int n,
                        hipStream_t s, bool gemms_only = false, double* flops = nullptr, int* launches = nullptr,
                        int layer_mask = 15, bool shared_grids = true) {
    if (n < 1 || n > GROUP_MAX) return fail(PAFUSE_E_ARG, "run_mixste_layers_n: %d parts", n);
    for (int i = 1; i < n; ++i)
        if (ws[i]->depth != ws[0]->depth) return fail(PAFUSE_E_ARG, "run_mixste_layers_n: parts of unequal depth");
    int rc;
    BlockLaunch bl[GROUP_MAX];
    for (int i = 0; i < ws[0]->depth; ++i) {
        for (int k = 0; k < n; ++k) {
            const pafuse_mixste2_weights* w = ws[k];
            const int F = w->frames, J = w->joints, C = w->channels;
            BlockTail t{};
            // spatial block i: sequences = the J joints of one (r, f); then Spatial_norm; TTE block 0 first adds the
            // temporal position embedding; the next LayerNorm is norm1 of temporal block i.
            t.post_w = w->snorm_w, t.post_b = w->snorm_b, t.post_eps = 1e-6f;
            if (i == 0) t.pos = w->pos_temporal, t.posJ = J, t.posF = F;
            t.next_w = w->tte[i].norm1_w, t.next_b = w->tte[i].norm1_b, t.next_eps = 1e-6f;
            bl[k] = make_block(w->ste[i], pbs[k], Rs[k] * F * J, C, w->heads, Rs[k] * F, J, 1, J, 0, 1, t, w->operand_bf16,
                               w->mlp_hidden, w->qk_scale, ln_folded(w), w->keep_f32_residual != 0);
        }
        if ((rc = run_blocks(bl, n, s, gemms_only, flops, launches, layer_mask, shared_grids))) return rc;
        for (int k = 0; k < n; ++k) {
            const pafuse_mixste2_weights* w = ws[k];
            const int F = w->frames, J = w->joints, C = w->channels;
            // temporal block i: sequences = the F frames of one (r, j); then Temporal_norm; next is norm1 of spatial
            // block i+1, or the head (LayerNorm eps 1e-5 + Linear(C,3)) after the last block.
            BlockTail t{};
            t.post_w = w->tnorm_w, t.post_b = w->tnorm_b, t.post_eps = 1e-6f;
            if (i + 1 < w->depth) {
                t.next_w = w->ste[i + 1].norm1_w, t.next_b = w->ste[i + 1].norm1_b, t.next_eps = 1e-6f;
            } else {
                t.next_w = w->hnorm_w, t.next_b = w->hnorm_b, t.next_eps = 1e-5f;
                t.head_w = w->head_w, t.head_b = w->head_b, t.out_head = pbs[k].pred;
            }
            bl[k] = make_block(w->tte[i], pbs[k], Rs[k] * F * J, C, w->heads, Rs[k] * J, F, J, (int64_t)F * J, 1, J, t,
                               w->operand_bf16, w->mlp_hidden, w->qk_scale, ln_folded(w), w->keep_f32_residual != 0);
        }
        if ((rc = run_blocks(bl, n, s, gemms_only, flops, launches, layer_mask, shared_grids))) return rc;
    }
    return PAFUSE_OK;
}

int run_mixste_layers(const pafuse_mixste2_weights* w, const PartBuffers& pb, int64_t R, hipStream_t s,
                      bool gemms_only = false, double* flops = nullptr, int* launches = nullptr, int layer_mask = 15) {
    return run_mixste_layers_n(&w, &pb, &R, 1, s, gemms_only, flops, launches, layer_mask);
}

// parts of one configuration may share grids when they all run split-precision products and have the same depth
bool parts_groupable(const pafuse_d3dp_config* cfg) {
    if (cfg->part_by_part_launches || cfg->num_parts < 2 || cfg->num_parts > GROUP_MAX) return false;
    for (int i = 0; i < cfg->num_parts; ++i)
        if (cfg->part[i].operand_bf16 != 2 || cfg->part[i].depth != cfg->part[0].depth) return false;
    return true;
}

int launch_time_embed(const pafuse_mixste2_weights* w, const int64_t* t, int64_t t_scalar, int B, float* out,
                      float* hid_scratch, hipStream_t s) {
    TimeEmbedParams p{};
    const int C = w->channels;
    p.t = t, p.t_scalar = t_scalar, p.freqs = w->freqs;
    p.w1 = w->tm1_w, p.b1 = w->tm1_b, p.w3 = w->tm3_w, p.b3 = w->tm3_b, p.hid = hid_scratch, p.out = out, p.C = C;
    hipLaunchKernelGGL(time_embed_kernel<0>, dim3((2 * C + 7) / 8, B), dim3(512), (size_t)C * sizeof(float), s, p);
    hipLaunchKernelGGL(time_embed_kernel<1>, dim3((C + 7) / 8, B), dim3(512), (size_t)2 * C * sizeof(float), s, p);
    return check_launch("time_embed_kernel");
}

__global__ void copy_kernel(const float* src, float* dst, int64_t n) {
    PAFUSE_XQ_GUARD();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

#include "train_host.inc"

}  // namespace

// ================================================================================================== C ABI
extern "C" {

const char* pafuse_version(void) {
#ifdef PAFUSE_NO_PACKED_F32
    return "pafuse_hip 0.5 (gfx950, f32 MFMA + split bf16x3 / f16x2 MFMA, no packed-fp32 VALU)";
#else
    return "pafuse_hip 0.5 (gfx950, f32 MFMA + split bf16x3 / f16x2 MFMA, packed-fp32 VALU: 16-bit MFMA modes on one stream)";
#endif
}
const char* pafuse_last_error(void) { return g_err; }
int pafuse_abi_version(void) { return PAFUSE_ABI_VERSION; }

#ifdef PAFUSE_DIAG
/* diagnostic builds only (not declared in include/pafuse_hip.h): hash every intermediate of the following passes of this
 * thread into slots[0 .. n) (device memory, zeroed by the caller); returns how many slots the previous passes used */
int pafuse_diag_trace(unsigned long long* slots, int32_t n) {
    const int used = g_trace_i;
    g_trace = slots, g_trace_n = n, g_trace_i = 0;
    return used;
}
void pafuse_diag_snapshot(float* x_copy) { g_snap = x_copy; }
#endif

int pafuse_linear(const float* A, const float* W, const float* bias, float* out, int64_t M, int32_t N, int32_t K,
                  int32_t act, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!A || !W || !bias || !out || M < 0) return fail(PAFUSE_E_ARG, "linear: null pointer or negative M");
    GemmParams g{};
    g.A = A, g.W = W, g.bias = bias, g.out = out, g.M = M, g.N = N, g.K = K, g.act = act & 1, g.bf16 = (act >> 1) & 1;
    return gemm_bias(g, (hipStream_t)stream);
}

size_t pafuse_split_weights_bytes(int64_t N, int64_t K) { return (N > 0 && K > 0) ? wsplit_bytes(N, K) : 0; }
size_t pafuse_split_image_bytes(int64_t N, int64_t K, int32_t layout) {
    if (N <= 0 || K <= 0) return 0;
    return (layout & PAFUSE_SPLIT_F16X2) && !(layout & PAFUSE_SPLIT_X) ? hsplit_bytes(N, K) : wsplit_bytes(N, K);
}

int pafuse_mode_supported(int32_t mode, int32_t C, int32_t hidden, int32_t heads, int32_t joints, int32_t frames) {
    if (mode < 0 || mode > 4 || heads <= 0 || C <= 0 || C % heads) return 0;
    if (hidden <= 0) hidden = 2 * C;
    const int d = C / heads;
    if (!width_supported(C) || d % 4 || d > 48 || joints < 1 || joints > 144 || frames < 1 || frames > 144) return 0;
    if (mode >= 3 && (!hgemm_width(C) || !hgemm_plain_n(3 * C) || !hgemm_plain_n(hidden) || hidden % 32)) return 0;
    if (mode == 4 && (!xfqa_has(joints, d, heads) || !xfqa_has(frames, d, heads))) return 0;
    return 1;
}

int pafuse_mixste2_fused_blocks(const pafuse_mixste2_weights* w) {
    int rc = check_weights(w);
    if (rc) return rc;
    const int d = w->channels / w->heads;
    const bool fold = ln_folded(w);
    int n = 0;
    for (int i = 0; i < w->depth; ++i) {
        const pafuse_block_weights* pair[2] = {&w->ste[i], &w->tte[i]};
        const int len[2] = {w->joints, w->frames};
        for (int k = 0; k < 2; ++k)   // the condition of make_block, plus the tile-capacity check it makes
            if (w->operand_bf16 >= 2 && pair[k]->qkv_hs && pair[k]->qkv_hb &&
                (w->operand_bf16 == 4 ? xfqa_has(len[k], d, w->heads) : fqa_has(len[k], d, w->operand_bf16)) &&
                true) {   // (no qkv_hl requirement: nothing reads it since the centred fold of round 5)
                const int lp = fqa_lp(len[k], w->operand_bf16);
                n += (fqa_nseq_tile(len[k], lp) * len[k] <= fqa_tile_rows(lp)) ? 1 : 0;
            }
    }
    return n;
}

int pafuse_split_weights(const float* W, int32_t N, int32_t K, int32_t layout, void* out, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!W || !out) return fail(PAFUSE_E_ARG, "split_weights: null pointer");
    if (N <= 0 || K <= 0 || K % BK) return fail(PAFUSE_E_SHAPE, "split_weights: N=%d, K=%d (K must be a positive multiple of 32)", N, K);
    const int64_t n = (int64_t)N * (K / 8);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (layout & PAFUSE_SPLIT_X) {   // the X image (bf16x3 on the LDS-DMA pipeline): one geometry for every layer, no scaling, no tail
        hipLaunchKernelGGL(xsplit_weights_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, (uint8_t*)out, N, K);
        return check_launch("xsplit_weights_kernel");
    }
    if (layout & PAFUSE_SPLIT_F16X2) {   // the f16x2 H image (one geometry for every layer): largest |W| first, then the slices
        hipStream_t st = (hipStream_t)stream;
        uint8_t* const tail = (uint8_t*)out + (size_t)N * K * 4;
        if (hipMemsetAsync(tail, 0, HSPLIT_TAIL_BYTES, st) != hipSuccess) return fail(PAFUSE_E_HIP, "split_weights: memset failed");
        const int64_t elems = (int64_t)N * K;
        hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)std::min<int64_t>((elems + 255) / 256, 1024)), dim3(256), 0, st, W, elems,
                           reinterpret_cast<uint32_t*>(tail) + 1);
        hipLaunchKernelGGL(hsplit_weights_kernel, grid, dim3(256), 0, st, W, (uint8_t*)out, N, K);
        return check_launch("hsplit_weights_kernel");
    }
    if (layout < 0 || layout > 2) return fail(PAFUSE_E_ARG, "split_weights: layout %d (0 plain, 1 whole-row, 2 qkv; or PAFUSE_SPLIT_F16X2)", layout);
#ifdef PAFUSE_QKV_32X32
    if (layout == 2) layout = 0;
#endif
    if (layout == 2)
        hipLaunchKernelGGL((split_weights_kernel<32, 1>), grid, dim3(256), 0, (hipStream_t)stream, W, (uint8_t*)out, N, K);
    else if (wsplit_chunk(N, layout == 1) == 16)
        hipLaunchKernelGGL(split_weights_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, W, (uint8_t*)out, N, K);
    else
        hipLaunchKernelGGL(split_weights_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, W, (uint8_t*)out, N, K);
    return check_launch("split_weights_kernel");
}

int pafuse_linear_split(const float* A, const void* Wsplit, const float* bias, float* out, int64_t M, int32_t N, int32_t K,
                        int32_t act, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!A || !Wsplit || !bias || !out || M < 0) return fail(PAFUSE_E_ARG, "linear_split: null pointer or negative M");
    GemmParams g{};
    g.A = A, g.Wsplit = (const uint8_t*)Wsplit, g.bias = bias, g.out = out, g.M = M, g.N = N, g.K = K, g.act = act & 1;
    g.bf16 = 2, g.wlayout = (act & 2) ? 2 : 0;
    return gemm_bias(g, (hipStream_t)stream);
}

int pafuse_hsplit_rows(const float* X, int64_t R, int32_t K, void* out, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!X || !out || R < 0) return fail(PAFUSE_E_ARG, "hsplit_rows: null pointer or negative row count");
    if (K <= 0 || K % 8) return fail(PAFUSE_E_SHAPE, "hsplit_rows: K=%d must be a positive multiple of 8", K);
    if (R == 0) return PAFUSE_OK;
    const int64_t n = R * (K / 8);
    if ((n + 255) / 256 > 0x7fffffff) return fail(PAFUSE_E_ARG, "hsplit_rows: grid out of range");
    hipLaunchKernelGGL(hsplit_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, (uint8_t*)out, R, K);
    return check_launch("hsplit_rows_kernel");
}

int pafuse_xsplit_rows(const float* X, int64_t R, int32_t K, void* out, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!X || !out || R < 0) return fail(PAFUSE_E_ARG, "xsplit_rows: null pointer or negative row count");
    if (K <= 0 || K % 32) return fail(PAFUSE_E_SHAPE, "xsplit_rows: K=%d must be a positive multiple of 32", K);
    if (R == 0) return PAFUSE_OK;
    const int64_t n = R * (K / 8);
    if ((n + 255) / 256 > 0x7fffffff) return fail(PAFUSE_E_ARG, "xsplit_rows: grid out of range");
    hipLaunchKernelGGL(xsplit_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, (uint8_t*)out, R, K);
    return check_launch("xsplit_rows_kernel");
}

int pafuse_linear_x(const void* Ax, const void* Wx, const float* bias, float* out, void* out_x, int64_t M, int32_t N, int32_t K,
                    int32_t act, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!Ax || !Wx || !bias || (!out && !out_x) || M < 0) return fail(PAFUSE_E_ARG, "linear_x: null pointer or negative M");
    GemmParams g{};
    g.Ah = (const uint8_t*)Ax, g.Wh = (const uint8_t*)Wx, g.bias = bias, g.out = out, g.out_h = (uint8_t*)out_x;
    g.M = M, g.N = N, g.K = K, g.act = act & 1, g.bf16 = 4;
    return gemm_bias(g, (hipStream_t)stream);
}

int pafuse_linear_h(const void* Ah, const void* Wh, const float* bias, float* out, void* out_h, int64_t M, int32_t N, int32_t K,
                    int32_t act, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!Ah || !Wh || !bias || (!out && !out_h) || M < 0) return fail(PAFUSE_E_ARG, "linear_h: null pointer or negative M");
    GemmParams g{};
    g.Ah = (const uint8_t*)Ah, g.Wh = (const uint8_t*)Wh, g.bias = bias, g.out = out, g.out_h = (uint8_t*)out_h;
    g.M = M, g.N = N, g.K = K, g.act = act & 1, g.bf16 = 3;
    return gemm_bias(g, (hipStream_t)stream);
}

int pafuse_qkv_attention_image(int32_t scheme, const void* x_img, const float* stats, const void* qkv_hs, const float* qkv_hb, void* o_img,
                               int64_t M, int64_t nseq, int32_t L, int32_t C, int32_t heads, int64_t group, int64_t group_stride,
                               int64_t seq_stride, int64_t tok_stride, float qk_scale, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!x_img || !qkv_hs || !qkv_hb || !o_img || M < 0 || nseq < 0 || group <= 0) return fail(PAFUSE_E_ARG, "qkv_attention_image: bad argument");
    if (scheme != PAFUSE_SPLIT_F16X2 && scheme != PAFUSE_SPLIT_X) return fail(PAFUSE_E_ARG, "qkv_attention_image: scheme must be PAFUSE_SPLIT_F16X2 or PAFUSE_SPLIT_X");
    if (heads <= 0 || C <= 0 || C % heads || C % 32) return fail(PAFUSE_E_SHAPE, "qkv_attention_image: C=%d, heads=%d", C, heads);
    const int mode = scheme == PAFUSE_SPLIT_X ? 4 : 3, d = C / heads;
    if (L <= 0 || !(mode == 4 ? xfqa_has(L, d, heads) : fqa_has(L, d, 3)))
        return fail(PAFUSE_E_SHAPE, "qkv_attention_image: no fused form for sequences of %d tokens at head dim %d", L, d);
    if (nseq == 0) return PAFUSE_OK;
    const int lp = fqa_lp(L, mode), dp = fqa_dp(d);
    FqaParams f{};
    f.g.Ah = (const uint8_t*)x_img, f.g.Wh = (const uint8_t*)qkv_hs, f.g.bias = qkv_hb, f.g.ln_in = stats;
    f.g.M = M, f.g.N = heads * 3 * dp, f.g.K = C, f.g.bf16 = mode;
    f.o = (float*)o_img, f.nseq = nseq, f.L = L, f.C = C, f.heads = heads, f.d = d;
    f.nseq_tile = fqa_nseq_tile(L, lp);
    if (f.nseq_tile * L > fqa_tile_rows(lp)) return fail(PAFUSE_E_SHAPE, "qkv_attention_image: %d sequences of %d tokens do not fit a tile", f.nseq_tile, L);
    f.group = group, f.group_stride = group_stride, f.seq_stride = seq_stride, f.tok_stride = tok_stride;
    f.scale = qk_scale != 0.f ? qk_scale : 1.0f / sqrtf((float)d);
    return fused_qkv_attention(f, (hipStream_t)stream);
}

int pafuse_mlp_h(const void* xh, const float* stats_in, const void* W1h, const float* bias1, const void* W2hp, const float* bias2,
                 void* out_xh, float* stats_out, int64_t M, int32_t C, float eps, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!xh || !W1h || !bias1 || !W2hp || !bias2 || !out_xh || !stats_out || M < 0) return fail(PAFUSE_E_ARG, "mlp_h: null pointer or negative M");
    if (!hmlp_has(C, 2 * C)) return fail(PAFUSE_E_SHAPE, "mlp_h: no fused MLP kernel for channel width %d (have 224, 256, 384)", C);
    MlpParams m{};
    m.g.Ah = (const uint8_t*)xh, m.g.Wh = (const uint8_t*)W2hp, m.g.bias = bias2, m.g.resid_h = (const uint8_t*)xh;
    m.g.out_xh = (uint8_t*)out_xh, m.g.ln_stats = stats_out, m.g.next_w = bias2, m.g.next_b = bias2, m.g.next_eps = eps;   // (folded: next_* only say "a LayerNorm follows")
    m.g.M = M, m.g.N = C, m.g.K = 2 * C, m.g.bf16 = 3;
    m.W1h = (const uint8_t*)W1h, m.bias1 = bias1, m.ln_in = stats_in;
    return fused_mlp(m, (hipStream_t)stream);
}

int pafuse_layernorm(const float* x, const float* w, const float* b, float* out, int64_t M, int32_t C, float eps,
                     void* stream) {
    StreamDevice on_stream_device(stream);
    if (!x || !w || !b || !out || M < 0) return fail(PAFUSE_E_ARG, "layernorm: null pointer or negative M");
    if (C <= 0 || C > 64 * LN_MAX_PER_LANE) return fail(PAFUSE_E_SHAPE, "layernorm: C=%d out of range", C);
    if (M == 0) return PAFUSE_OK;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, out,
                       M, C, eps);
    return check_launch("layernorm_kernel");
}

int pafuse_attention(const float* qkv, float* o, int64_t nseq, int32_t L, int32_t C, int32_t heads, int64_t group,
                     int64_t group_stride, int64_t seq_stride, int64_t tok_stride, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!qkv || !o || nseq < 0 || heads <= 0 || C % heads || group <= 0)
        return fail(PAFUSE_E_ARG, "attention: bad argument");
    AttnParams a{};
    a.qkv = qkv, a.o = o, a.nseq = nseq, a.L = L, a.C = C, a.heads = heads, a.d = C / heads;
    a.group = group, a.group_stride = group_stride, a.seq_stride = seq_stride, a.tok_stride = tok_stride;
    a.scale = 1.0f / sqrtf((float)(C / heads));
    return attention(a, (hipStream_t)stream);
}

size_t pafuse_block_workspace_bytes(int64_t rows, int32_t C) { return part_buffer_bytes(rows, C, 1, 6); }

int pafuse_block_forward(const pafuse_block_weights* w, float* x, int64_t S, int32_t L, int32_t C, int32_t heads,
                         int32_t operand_bf16, void* workspace, size_t workspace_bytes, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!w || !x || !workspace || S < 0) return fail(PAFUSE_E_ARG, "block_forward: bad argument");
    if (operand_bf16 < 0 || operand_bf16 > 4) return fail(PAFUSE_E_ARG, "block_forward: matrix-product mode %d", operand_bf16);
    if (operand_bf16 >= 3 && !hgemm_width(C)) return fail(PAFUSE_E_SHAPE, "block_forward: the image pipelines serve the widths 224 / 256 / 384");
    if (operand_bf16 == 4 && (heads <= 0 || C % heads || !w->qkv_hs || !w->qkv_hb || !xfqa_has(L, C / heads, heads)))
        return fail(PAFUSE_E_ARG, "block_forward: bf16x3 on images needs the head-major qkv image (qkv_hs, qkv_hb) and a sequence length / head "
                                  "dim with a fused qkv + attention form");
    if (operand_bf16 >= 2 && ((!w->qkv_ws && operand_bf16 != 4) || !w->proj_ws || !w->fc1_ws || !w->fc2_ws))
        return fail(PAFUSE_E_ARG, "block_forward: split-precision mode needs the pre-split image of every linear weight");
    if (!width_supported(C)) return fail(PAFUSE_E_SHAPE, "channel width %d has no kernel", C);
    if (heads <= 0 || C % heads) return fail(PAFUSE_E_SHAPE, "block_forward: heads %d must divide C %d", heads, C);
    if (L <= 0) return fail(PAFUSE_E_SHAPE, "block_forward: sequence length %d", L);
    const int64_t M = S * L;
    if (workspace_bytes < pafuse_block_workspace_bytes(M, C)) return fail(PAFUSE_E_WORKSPACE, "workspace too small");
    if (M == 0) return PAFUSE_OK;
    hipStream_t s = (hipStream_t)stream;
    PartBuffers pb;
    carve_part((char*)workspace, M, C, 1, pb, 6);
    pb.x = x;
    // image pipelines: the qkv GEMM reads the image of LN1(x) - the fp32 rows pass through `wide` (free until qkv writes it)
    int rc = pafuse_layernorm(x, w->norm1_w, w->norm1_b, operand_bf16 >= 3 ? pb.wide : pb.xn, M, C, 1e-6f, stream);
    if (rc) return rc;
    if (operand_bf16 == 3 && (rc = pafuse_hsplit_rows(pb.wide, M, C, pb.xn, stream))) return rc;
    if (operand_bf16 == 4 && (rc = pafuse_xsplit_rows(pb.wide, M, C, pb.xn, stream))) return rc;
    BlockTail t{};  // plain Block.forward: no post norm, nothing after
    return run_block(*w, pb, M, C, heads, S, L, 1, L, 0, 1, t, s, false, nullptr, nullptr, operand_bf16);
}

int pafuse_time_embed(const pafuse_mixste2_weights* w, const int64_t* t, int32_t B, float* temb, float* hid_scratch,
                      void* stream) {
    StreamDevice on_stream_device(stream);
    if (!w || !t || !temb || B <= 0) return fail(PAFUSE_E_ARG, "time_embed: bad argument");
    if (w->channels % 2 || w->channels > 1024) return fail(PAFUSE_E_SHAPE, "time_embed: C=%d", w->channels);
    if (!hid_scratch) return fail(PAFUSE_E_ARG, "time_embed: null scratch");
    return launch_time_embed(w, t, 0, B, temb, hid_scratch, (hipStream_t)stream);
}

size_t pafuse_mixste2_workspace_bytes(const pafuse_mixste2_weights* w, int32_t B, int32_t P) {
    if (!w) return 0;
    return part_buffer_bytes((int64_t)B * P * w->frames * w->joints, w->channels, B, image_elem_bytes(w->operand_bf16));
}

int pafuse_mixste2_forward(const pafuse_mixste2_weights* w, const float* x2d, const float* x3d, const int64_t* t,
                           int32_t B, int32_t P, float* out, void* workspace, size_t workspace_bytes, void* stream) {
    StreamDevice on_stream_device(stream);
    int rc = check_weights(w);
    if (rc) return rc;
    if (!x2d || !x3d || !t || !out || !workspace || B <= 0 || P <= 0)
        return fail(PAFUSE_E_ARG, "mixste2_forward: bad argument");
    if (workspace_bytes < pafuse_mixste2_workspace_bytes(w, B, P)) return fail(PAFUSE_E_WORKSPACE, "workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int64_t R = (int64_t)B * P, M = R * w->frames * w->joints;
    PartBuffers pb;
    carve_part((char*)workspace, M, w->channels, B, pb, image_elem_bytes(w->operand_bf16));
    if ((rc = launch_time_embed(w, t, 0, B, pb.temb, pb.wide, s))) return rc;
#ifdef PAFUSE_DIAG
    if (g_snap) (void)hipMemcpyAsync(g_snap + M * w->channels, pb.temb, (size_t)B * w->channels * 4, hipMemcpyDeviceToDevice, s);
#endif
    EmbedParams e{};
    e.x3d = x3d, e.x2d = x2d, e.x2d_flip = nullptr, e.joints = nullptr, e.perm = nullptr;
    e.pw = w->patch_w, e.pb = w->patch_b, e.pos = w->pos_spatial, e.temb = pb.temb;
    e.n_w = w->ste[0].norm1_w, e.n_b = w->ste[0].norm1_b, e.n_eps = 1e-6f;
    e.x = pb.x, e.xn = pb.xn, e.stats = ln_folded(w) ? pb.stats : nullptr;
    e.xh = w->operand_bf16 >= 3 ? reinterpret_cast<uint8_t*>(pb.xn) : nullptr;
    e.x_image = w->operand_bf16 == 4;
    e.centre_x = w->operand_bf16 == 2 && ln_folded(w);
    if (h_residual_only(w)) e.x = nullptr;
    e.B = B, e.P = P, e.F = w->frames, e.J = w->joints, e.J3 = w->joints, e.C = w->channels, e.nflip = 1;
    e.do_clamp = 0, e.scale = 1.f, e.lim = 1.1f, e.row0 = 0, e.nrows = M;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((M + EMBED_ROWS_PER_BLOCK - 1) / EMBED_ROWS_PER_BLOCK)), dim3(256), 0, s, e);
    if ((rc = check_launch("embed_kernel"))) return rc;
    PAFUSE_TRACE(pb.temb, (size_t)B * w->channels * 4, s);
    PAFUSE_TRACE(pb.x, (size_t)M * w->channels * 4, s);
    PAFUSE_TRACE(pb.xn, (size_t)M * w->channels * 4, s);
#ifdef PAFUSE_DIAG
    if (g_snap) (void)hipMemcpyAsync(g_snap, pb.x, (size_t)M * w->channels * 4, hipMemcpyDeviceToDevice, s);
#endif
    if ((rc = run_mixste_layers(w, pb, R, s, (debug_f32_mask() & 32) != 0))) return rc;
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((M * 3 + 255) / 256)), dim3(256), 0, s, pb.pred, out, M * 3);
    return check_launch("copy_kernel");
}

size_t pafuse_d3dp_workspace_bytes(const pafuse_d3dp_config* cfg, int32_t B, int32_t P) {
    if (!cfg) return 0;
    const int nflip = cfg->flip ? 2 : 1;
    size_t total = align_up((size_t)B * P * cfg->frames * cfg->num_kps * 3 * 4);  // img
    for (int i = 0; i < cfg->num_parts; ++i) {
        const pafuse_mixste2_weights& w = cfg->part[i];
        total += part_buffer_bytes((int64_t)nflip * B * P * w.frames * w.joints, w.channels, B, image_elem_bytes(w.operand_bf16));
    }
    return total + RANGE_FLAG_BYTES;   // the range flag word sits behind everything else
}

size_t pafuse_d3dp_range_flag_offset(const pafuse_d3dp_config* cfg, int32_t B, int32_t P) {
    const size_t total = pafuse_d3dp_workspace_bytes(cfg, B, P);
    return total ? total - RANGE_FLAG_BYTES : 0;
}

int pafuse_d3dp_check_range(const pafuse_d3dp_config* cfg, int32_t B, int32_t P, const void* workspace, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!cfg || !workspace || B <= 0 || P <= 0) return fail(PAFUSE_E_ARG, "d3dp_check_range: bad argument");
    int32_t flag = 0;
    const char* at = (const char*)workspace + pafuse_d3dp_range_flag_offset(cfg, B, P);
    hipError_t e = hipMemcpyAsync(&flag, at, sizeof flag, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return fail(PAFUSE_E_HIP, "d3dp_check_range: %s", hipGetErrorString(e));
    if (flag)
        return fail(PAFUSE_E_RANGE, "a denoiser output was not finite: in 'f16x2' an activation left the fp16 range (|a| >= 65504 makes "
                                    "its high slice inf); run this model in 'bf16x3' or 'f32' (the predictions of the affected rows are NaN)");
    return PAFUSE_OK;
}

// Side streams beside the split-precision kernels need a library WITHOUT packed-fp32 VALU instructions: on MI355X a
// v_pk_{add,mul,fma}_f32 whose src1 takes the high register of its pair for the low result (op_sel:[0,1,..]) returns wrong
// lanes while a wave of another kernel on the same SIMD issues v_mfma_f32_32x32x16_bf16 (two-kernel reproducer:
// tools/mfma_queue_isolate.hip V9.3 / V9.9 / V9.10 / V9.12; profiles/r03_bf16_mfma_concurrency.md).  hipcc's SLP vectoriser
// emits that form in the VALU-only kernels of this file (embed, LayerNorm, epilogues); __graft_entry__.build() therefore
// compiles the device code with the packed-fp32-ops target feature off and says so with -DPAFUSE_NO_PACKED_F32.  A build
// without that promise keeps the bf16-MFMA modes on the caller's stream.
static bool lanes_allowed(const pafuse_d3dp_config* cfg) {
#if defined(PAFUSE_NO_PACKED_F32) || defined(PAFUSE_ALLOW_BF16_LANES)
    (void)cfg;
    return true;
#else
    for (int i = 0; i < cfg->num_parts; ++i)
        if (cfg->part[i].operand_bf16 != 0) return false;
    return true;
#endif
}

static int d3dp_check(const pafuse_d3dp_config* cfg, int B, int P) {
    if (!cfg || B <= 0 || P <= 0) return fail(PAFUSE_E_ARG, "d3dp: bad argument");
    if (cfg->num_parts < 1 || cfg->num_parts > PAFUSE_MAX_PARTS) return fail(PAFUSE_E_SHAPE, "num_parts %d", cfg->num_parts);
    int total = 0;
    for (int i = 0; i < cfg->num_parts; ++i) {
        int rc = check_weights(&cfg->part[i]);
        if (rc) return rc;
        if (cfg->part[i].frames != cfg->frames) return fail(PAFUSE_E_SHAPE, "part %d: frames mismatch", i);
        if (!cfg->part_joints[i]) return fail(PAFUSE_E_ARG, "part %d: null joint list", i);
        total += cfg->part[i].joints;
    }
    if (total != cfg->num_kps) return fail(PAFUSE_E_SHAPE, "parts cover %d joints, expected %d", total, cfg->num_kps);
    if (!cfg->joint_part || !cfg->joint_local || (cfg->flip && !cfg->flip_perm))
        return fail(PAFUSE_E_ARG, "d3dp: null index table");
    return PAFUSE_OK;
}

int pafuse_d3dp_sample(const pafuse_d3dp_config* cfg, const pafuse_ddim_step* steps, int32_t nsteps, const float* x2d,
                       const float* x2d_flip, const float* noise, int32_t n_draws, int32_t B, int32_t P, float* out,
                       void* workspace, size_t workspace_bytes, void* stream, void* const* aux_streams, int32_t n_aux) {
    StreamDevice on_stream_device(stream);
    int rc = d3dp_check(cfg, B, P);
    if (rc) return rc;
    if (!steps || nsteps <= 0 || !x2d || !noise || !out || !workspace || (cfg->flip && !x2d_flip) || n_draws < 1)
        return fail(PAFUSE_E_ARG, "d3dp_sample: bad argument");
    int need = 1;
    for (int k = 0; k < nsteps; ++k) need += steps[k].last ? 0 : 1;
    if (n_draws < need) return fail(PAFUSE_E_ARG, "d3dp_sample: %d noise draws given, %d needed", n_draws, need);
    if (workspace_bytes < pafuse_d3dp_workspace_bytes(cfg, B, P)) return fail(PAFUSE_E_WORKSPACE, "workspace too small");

    hipStream_t s0 = (hipStream_t)stream;
    const int nflip = cfg->flip ? 2 : 1, F = cfg->frames, J = cfg->num_kps, NP = cfg->num_parts;
    const int64_t R = (int64_t)nflip * B * P;
    const int64_t img_elems = (int64_t)B * P * F * J * 3;
    char* base = (char*)workspace;
    float* img = (float*)base;
    base += align_up((size_t)img_elems * 4);
    PartBuffers pb[PAFUSE_MAX_PARTS];
    for (int i = 0; i < NP; ++i)
        base = carve_part(base, R * F * cfg->part[i].joints, cfg->part[i].channels, B, pb[i], image_elem_bytes(cfg->part[i].operand_bf16));
    int32_t* const range_flag = reinterpret_cast<int32_t*>((char*)workspace + pafuse_d3dp_range_flag_offset(cfg, B, P));
    if (hipMemsetAsync(range_flag, 0, sizeof(int32_t), s0) != hipSuccess) return fail(PAFUSE_E_HIP, "d3dp_sample: memset failed");

    // Parts are independent inside a step, and so are hypotheses: with aux streams the work of a step is cut into
    // (part, hypothesis-group) lanes, each a chain of small launches on its own stream, so that the ramp-up and
    // tail of one lane's kernels are filled by the other lanes (groups = (n_aux + 1) / parts, at least 1).
    // Without aux streams (or in a build that may hold packed-fp32 instructions, see lanes_allowed) everything runs on
    // `stream`, block k of every part in shared grids.
    n_aux = lanes_allowed(cfg) ? n_aux : 0;
    const int n_lanes_max = 1 + (n_aux > 0 ? n_aux : 0);
    int groups = n_lanes_max / NP;
    if (groups < 1) groups = 1;
    if (groups > R) groups = (int)R;
    const int lanes = NP * groups;
    auto lane_stream = [&](int lane) -> hipStream_t {
        if (n_aux <= 0 || lane == 0) return s0;
        return (hipStream_t)aux_streams[(lane - 1) % n_aux];
    };
    const bool multi = n_aux > 0 && lanes > 1;
    constexpr int MAX_LANES = 64;
    if (lanes > MAX_LANES) return fail(PAFUSE_E_ARG, "too many stream lanes (%d)", lanes);
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)((img_elems + 255) / 256)), dim3(256), 0, s0, noise, img, img_elems);
    if ((rc = check_launch("copy_kernel"))) return rc;
    hipEvent_t ev_fork = nullptr, ev_join[MAX_LANES] = {};
    if (multi) {
        hipError_t e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming);
        for (int i = 1; i < lanes && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            if (ev_fork) hipEventDestroy(ev_fork);
            for (int i = 1; i < lanes; ++i)
                if (ev_join[i]) hipEventDestroy(ev_join[i]);
            return fail(PAFUSE_E_HIP, "hipEventCreate: %s", hipGetErrorString(e));
        }
    }
    bool forked = false;  // lanes hold work the main stream has not waited for yet
    hipError_t sync_err = hipSuccess;  // first failing event call: a missed fork/join would be a silent race
    auto note = [&](hipError_t e) {
        if (e != hipSuccess && sync_err == hipSuccess) sync_err = e;
    };
    auto join = [&]() {
        for (int i = 1; i < lanes; ++i) {
            note(hipEventRecord(ev_join[i], lane_stream(i)));
            note(hipStreamWaitEvent(s0, ev_join[i], 0));
        }
        forked = false;
    };
    int draw = 1;
    for (int k = 0; k < nsteps && rc == PAFUSE_OK; ++k) {
        const pafuse_ddim_step& st = steps[k];
        // the time embedding of every part once, on the main stream, before the fork
        for (int i = 0; i < NP && rc == PAFUSE_OK; ++i)
            rc = launch_time_embed(&cfg->part[i], nullptr, st.time, B, pb[i].temb, pb[i].wide, s0);
        if (rc) break;
        if (multi) {
            note(hipEventRecord(ev_fork, s0));
            for (int i = 1; i < lanes; ++i) note(hipStreamWaitEvent(lane_stream(i), ev_fork, 0));
            forked = true;
            if (sync_err != hipSuccess) {
                rc = fail(PAFUSE_E_HIP, "d3dp_sample: stream fork failed: %s", hipGetErrorString(sync_err));
                break;
            }
        }
        const bool together = !multi && parts_groupable(cfg);  // one stream: block k of every part in shared grids
        for (int lane = 0; lane < lanes && rc == PAFUSE_OK; ++lane) {
            const int i = lane % NP, gi = lane / NP;
            const pafuse_mixste2_weights* w = &cfg->part[i];
            hipStream_t ls = lane_stream(lane);
            const int64_t r0 = R * gi / groups, r1 = R * (gi + 1) / groups;
            const int64_t rows_per_r = (int64_t)F * w->joints, row0 = r0 * rows_per_r, nrows = (r1 - r0) * rows_per_r;
            EmbedParams e{};
            e.x3d = img, e.x2d = x2d, e.x2d_flip = x2d_flip, e.joints = cfg->part_joints[i], e.perm = cfg->flip_perm;
            e.pw = w->patch_w, e.pb = w->patch_b, e.pos = w->pos_spatial, e.temb = pb[i].temb;
            e.n_w = w->ste[0].norm1_w, e.n_b = w->ste[0].norm1_b, e.n_eps = 1e-6f;
            e.x = pb[i].x, e.xn = pb[i].xn;
            e.stats = ln_folded(w) ? pb[i].stats + row0 * 2 : nullptr;   // where offset_rows puts this group's statistics
            e.xh = w->operand_bf16 >= 3 ? reinterpret_cast<uint8_t*>(pb[i].xn) : nullptr;   // (indexed by the absolute row)
            e.x_image = w->operand_bf16 == 4;
            e.centre_x = w->operand_bf16 == 2 && ln_folded(w);
            if (h_residual_only(w)) e.x = nullptr;
            e.B = B, e.P = P, e.F = F, e.J = w->joints, e.J3 = J, e.C = w->channels, e.nflip = nflip;
            e.do_clamp = 1, e.scale = (float)cfg->scale, e.lim = (float)(1.1 * cfg->scale), e.row0 = row0, e.nrows = nrows;
            hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((nrows + EMBED_ROWS_PER_BLOCK - 1) / EMBED_ROWS_PER_BLOCK)),
                               dim3(256), 0, ls, e);
            if ((rc = check_launch("embed_kernel"))) break;
            if (!together) rc = run_mixste_layers(w, offset_rows(pb[i], row0, w->channels), r1 - r0, ls);
        }
        if (rc) break;
        if (together) {
            const pafuse_mixste2_weights* ws[PAFUSE_MAX_PARTS];
            int64_t Rs[PAFUSE_MAX_PARTS];
            for (int i = 0; i < NP; ++i) ws[i] = &cfg->part[i], Rs[i] = R;
            if ((rc = run_mixste_layers_n(ws, pb, Rs, NP, s0))) break;
        }
        if (multi) {
            join();
            if (sync_err != hipSuccess) {
                rc = fail(PAFUSE_E_HIP, "d3dp_sample: stream join failed: %s", hipGetErrorString(sync_err));
                break;
            }
        }
        FinalizeParams f{};
        for (int i = 0; i < NP; ++i) f.pred[i] = pb[i].pred, f.Jp[i] = cfg->part[i].joints;
        f.joint_part = cfg->joint_part, f.joint_local = cfg->joint_local, f.perm = cfg->flip_perm;
        f.img = img, f.noise = st.last ? nullptr : noise + (int64_t)draw * img_elems, f.out = out;
        f.B = B, f.P = P, f.F = F, f.J = J, f.T = nsteps, f.step = k, f.flip = cfg->flip, f.last = st.last;
        f.scale = (float)cfg->scale, f.lim = (float)(1.1 * cfg->scale), f.sr = st.sqrt_recip_acp, f.srm1 = st.sqrt_recipm1_acp, f.c = st.c;
        f.an_f = (float)st.sqrt_alpha_next, f.c_f = (float)st.c, f.sigma_f = (float)st.sigma;
        f.range_flag = range_flag;
        const int64_t n = (int64_t)B * P * F * J;
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s0, f);
        rc = check_launch("finalize_kernel");
        if (!st.last) ++draw;
    }
    if (multi) {
        if (forked) join();  // error exit mid-step: the caller's stream still orders after everything launched
        // aux streams beyond the lanes in use did no work: nothing to join (callers that capture the call into a graph
        // must fork only the first pafuse_d3dp_lanes(...) - 1 aux streams, see include/pafuse_hip.h)
        hipEventDestroy(ev_fork);
        for (int i = 1; i < lanes; ++i) hipEventDestroy(ev_join[i]);
        if (rc == PAFUSE_OK && sync_err != hipSuccess)
            rc = fail(PAFUSE_E_HIP, "d3dp_sample: stream join failed: %s", hipGetErrorString(sync_err));
    }
    return rc;
}

int pafuse_d3dp_lanes(const pafuse_d3dp_config* cfg, int32_t B, int32_t P, int32_t n_aux) {
    if (!cfg || B <= 0 || P <= 0 || cfg->num_parts < 1) return fail(PAFUSE_E_ARG, "d3dp_lanes: bad argument");
    if (!lanes_allowed(cfg)) return 1;
    const int64_t R = (int64_t)(cfg->flip ? 2 : 1) * B * P;
    int groups = (1 + (n_aux > 0 ? n_aux : 0)) / cfg->num_parts;
    if (groups < 1) groups = 1;
    if (groups > R) groups = (int)R;
    const int lanes = cfg->num_parts * groups;
    if (n_aux <= 0) return 1;
    return lanes < 1 + n_aux ? lanes : 1 + n_aux;  // distinct streams in use (lane 0 = the caller's stream)
}

int pafuse_embed(const float* x3d, const float* x2d, const float* x2d_flip, const int32_t* joints, const int32_t* perm,
                 const float* patch_w, const float* patch_b, const float* pos_spatial, const float* temb,
                 const float* norm_w, const float* norm_b, float norm_eps, int32_t B, int32_t P, int32_t F, int32_t J,
                 int32_t J3, int32_t C, int32_t nflip, int32_t do_clamp, double scale, float* x, float* xn, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!x3d || !x2d || !patch_w || !patch_b || !pos_spatial || !temb || !norm_w || !norm_b || !x || !xn)
        return fail(PAFUSE_E_ARG, "embed: null pointer");
    if (B <= 0 || P <= 0 || F <= 0 || J <= 0 || J3 < J || (nflip != 1 && nflip != 2))
        return fail(PAFUSE_E_ARG, "embed: bad size (B=%d P=%d F=%d J=%d J3=%d nflip=%d)", B, P, F, J, J3, nflip);
    if (nflip == 2 && (!x2d_flip || !perm)) return fail(PAFUSE_E_ARG, "embed: the flipped half needs x2d_flip and perm");
    if (J3 != J && !joints) return fail(PAFUSE_E_ARG, "embed: a part of %d joints out of %d needs its joint list", J, J3);
    if (C % 4 || C <= 0 || C > EMBED_NV * 128) return fail(PAFUSE_E_SHAPE, "embed: C=%d (need %%4, <= %d)", C, EMBED_NV * 128);
    if (do_clamp && !(scale > 0.0)) return fail(PAFUSE_E_ARG, "embed: scale must be positive");
    EmbedParams e{};
    e.x3d = x3d, e.x2d = x2d, e.x2d_flip = x2d_flip, e.joints = joints, e.perm = perm;
    e.pw = patch_w, e.pb = patch_b, e.pos = pos_spatial, e.temb = temb;
    e.n_w = norm_w, e.n_b = norm_b, e.n_eps = norm_eps, e.x = x, e.xn = xn;
    e.B = B, e.P = P, e.F = F, e.J = J, e.J3 = J3, e.C = C, e.nflip = nflip, e.do_clamp = do_clamp;
    e.scale = (float)scale, e.lim = (float)(1.1 * scale);
    e.row0 = 0, e.nrows = (int64_t)nflip * B * P * F * J;
    const int64_t blocks = (e.nrows + EMBED_ROWS_PER_BLOCK - 1) / EMBED_ROWS_PER_BLOCK;
    if (blocks > 0x7fffffff) return fail(PAFUSE_E_ARG, "embed: grid out of range");
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, e);
    return check_launch("embed_kernel");
}

int pafuse_ddim_finalize(const float* const* pred, const int32_t* part_joints, int32_t num_parts, const int32_t* joint_part,
                         const int32_t* joint_local, const int32_t* flip_perm, float* img, const float* noise, float* out,
                         int32_t B, int32_t P, int32_t F, int32_t J, int32_t T, int32_t step, int32_t flip, double scale,
                         const pafuse_ddim_step* st, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!pred || !part_joints || !joint_part || !joint_local || !img || !out || !st)
        return fail(PAFUSE_E_ARG, "ddim_finalize: null pointer");
    if (num_parts < 1 || num_parts > PAFUSE_MAX_PARTS) return fail(PAFUSE_E_SHAPE, "ddim_finalize: num_parts %d", num_parts);
    if (B <= 0 || P <= 0 || F <= 0 || J <= 0 || T <= 0 || step < 0 || step >= T)
        return fail(PAFUSE_E_ARG, "ddim_finalize: bad size");
    if (flip && !flip_perm) return fail(PAFUSE_E_ARG, "ddim_finalize: flip needs the permutation");
    if (!st->last && !noise) return fail(PAFUSE_E_ARG, "ddim_finalize: an update step needs its noise draw");
    FinalizeParams f{};
    int total = 0;
    for (int i = 0; i < num_parts; ++i) {
        if (!pred[i] || part_joints[i] <= 0) return fail(PAFUSE_E_ARG, "ddim_finalize: part %d", i);
        f.pred[i] = pred[i], f.Jp[i] = part_joints[i], total += part_joints[i];
    }
    if (total != J) return fail(PAFUSE_E_SHAPE, "ddim_finalize: parts cover %d joints, expected %d", total, J);
    f.joint_part = joint_part, f.joint_local = joint_local, f.perm = flip_perm;
    f.img = img, f.noise = st->last ? nullptr : noise, f.out = out;
    f.B = B, f.P = P, f.F = F, f.J = J, f.T = T, f.step = step, f.flip = flip ? 1 : 0, f.last = st->last;
    f.scale = (float)scale, f.lim = (float)(1.1 * scale), f.sr = st->sqrt_recip_acp, f.srm1 = st->sqrt_recipm1_acp, f.c = st->c;
    f.an_f = (float)st->sqrt_alpha_next, f.c_f = (float)st->c, f.sigma_f = (float)st->sigma;
    const int64_t n = (int64_t)B * P * F * J;
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f);
    return check_launch("finalize_kernel");
}

int pafuse_hypothesis_errors(const float* pred, const float* gt, const float* x2d, const float* traj, const float* cam,
                             const int32_t* conn, const int32_t* pbroot, int32_t B, int32_t T, int32_t P, int32_t F,
                             int32_t J, float* e3, float* epb, float* jbest, float* pagg, float* jagg, float* paggpb,
                             void* stream) {
    StreamDevice on_stream_device(stream);
    if (!pred || !gt || !x2d || !traj || !cam || !conn || !pbroot || !e3 || !epb || !jbest || !pagg || !jagg || !paggpb)
        return fail(PAFUSE_E_ARG, "hypothesis_errors: null pointer");
    if (B <= 0 || T <= 0 || P <= 0 || F <= 0 || J <= 0) return fail(PAFUSE_E_ARG, "hypothesis_errors: bad size");
    MetricsParams m{};
    m.pred = pred, m.gt = gt, m.x2d = x2d, m.traj = traj, m.cam = cam, m.conn = conn, m.pbroot = pbroot;
    m.e3 = e3, m.epb = epb, m.jbest = jbest, m.pagg = pagg, m.jagg = jagg, m.paggpb = paggpb;
    m.B = B, m.T = T, m.P = P, m.F = F, m.J = J;
    const int64_t n = (int64_t)B * T * F * J;
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m);
    return check_launch("metrics_kernel");
}

int pafuse_d3dp_replay_gemms(const pafuse_d3dp_config* cfg, int32_t B, int32_t P, void* workspace,
                             size_t workspace_bytes, void* stream, double* flops) {
    return pafuse_d3dp_replay_layers(cfg, B, P, workspace, workspace_bytes, stream, 15, flops);
}

int pafuse_d3dp_replay_layers(const pafuse_d3dp_config* cfg, int32_t B, int32_t P, void* workspace,
                              size_t workspace_bytes, void* stream, int32_t layer_mask, double* flops) {
    StreamDevice on_stream_device(stream);
    if (layer_mask < 1 || layer_mask > 15) return fail(PAFUSE_E_ARG, "replay_layers: layer_mask %d", layer_mask);
    int rc = d3dp_check(cfg, B, P);
    if (rc) return rc;
    if (!workspace || workspace_bytes < pafuse_d3dp_workspace_bytes(cfg, B, P))
        return fail(PAFUSE_E_WORKSPACE, "workspace too small");
    const int nflip = cfg->flip ? 2 : 1;
    const int64_t R = (int64_t)nflip * B * P;
    char* base = (char*)workspace + align_up((size_t)B * P * cfg->frames * cfg->num_kps * 3 * 4);
    int launches = 0;
    double fl = 0.0;
    PartBuffers pb[PAFUSE_MAX_PARTS];
    const pafuse_mixste2_weights* ws[PAFUSE_MAX_PARTS];
    int64_t Rs[PAFUSE_MAX_PARTS];
    for (int i = 0; i < cfg->num_parts; ++i) {
        base = carve_part(base, R * cfg->frames * cfg->part[i].joints, cfg->part[i].channels, B, pb[i], image_elem_bytes(cfg->part[i].operand_bf16));
        ws[i] = &cfg->part[i], Rs[i] = R;
    }
    if (parts_groupable(cfg)) {  // the schedule pafuse_d3dp_sample runs: block k of every part in shared grids
        if ((rc = run_mixste_layers_n(ws, pb, Rs, cfg->num_parts, (hipStream_t)stream, true, &fl, &launches, layer_mask)))
            return rc;
    } else {
        for (int i = 0; i < cfg->num_parts; ++i)
            if ((rc = run_mixste_layers(&cfg->part[i], pb[i], R, (hipStream_t)stream, true, &fl, &launches, layer_mask)))
                return rc;
    }
    if (flops) *flops += fl;
    return launches;
}

// ------------------------------------------------------------------------------------------------- training
int pafuse_attention_backward(const float* qkv, const float* d_o, float* dqkv, int64_t nseq, int32_t L, int32_t C, int32_t heads,
                              int64_t group, int64_t group_stride, int64_t seq_stride, int64_t tok_stride, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!qkv || !d_o || !dqkv || nseq < 0 || heads <= 0 || C % heads || group <= 0 || L <= 0)
        return fail(PAFUSE_E_ARG, "attention_backward: bad argument");
    if ((C / heads) % 4) return fail(PAFUSE_E_SHAPE, "attention_backward: head width %d must be a multiple of 4", C / heads);
    if (nseq == 0) return PAFUSE_OK;
    AttnBackwardParams ab{};
    ab.qkv = qkv, ab.d_o = d_o, ab.dqkv = dqkv, ab.nseq = nseq, ab.group = group, ab.group_stride = group_stride;
    ab.seq_stride = seq_stride, ab.tok_stride = tok_stride, ab.L = L, ab.C = C, ab.heads = heads, ab.d = C / heads;
    ab.scale = 1.0f / sqrtf((float)(C / heads));
    return attention_backward(ab, (hipStream_t)stream);
}

size_t pafuse_linear_weight_grad_bytes(void) { return TRAIN_PARTIAL_FLOATS * sizeof(float); }

int pafuse_linear_weight_grad(const float* dY, const float* X, float* dW, float* db, int64_t M, int32_t N, int32_t K,
                              int32_t operand_bf16, void* workspace, size_t workspace_bytes, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!dY || !X || !dW || !workspace || M < 0 || N <= 0 || K <= 0) return fail(PAFUSE_E_ARG, "linear_weight_grad: bad argument");
    if (operand_bf16 != 0 && operand_bf16 != 2) return fail(PAFUSE_E_ARG, "linear_weight_grad: matrix-product mode %d (0 or 2)", operand_bf16);
    if (N % 4 || K % 4) return fail(PAFUSE_E_SHAPE, "linear_weight_grad: N=%d K=%d must be multiples of 4", N, K);
    if (workspace_bytes < pafuse_linear_weight_grad_bytes()) return fail(PAFUSE_E_WORKSPACE, "workspace too small");
    if (M == 0) return PAFUSE_OK;
    TrainBuffers tb{};
    tb.split = operand_bf16 == 2;
    tb.partial = tb.partial_side = (float*)workspace;
    tb.partial_floats = TRAIN_PARTIAL_FLOATS;
    return weight_grad(dY, X, dW, M, N, K, tb, (hipStream_t)stream, false, db);
}

size_t pafuse_mixste2_train_bytes(const pafuse_mixste2_weights* w, int32_t B) {
    if (!w || B <= 0 || check_weights(w, true)) return 0;
    return train_bytes(w, B);
}

int pafuse_mixste2_train_forward(const pafuse_mixste2_weights* w, const float* x2d, const float* x3d, const int64_t* t,
                                 int32_t B, const float* drop_path, float* out, void* saved, size_t saved_bytes,
                                 void* stream) {
    StreamDevice on_stream_device(stream);
    int rc = check_weights(w, true);
    if (rc) return rc;
    if (!x2d || !x3d || !t || !out || !saved || B <= 0) return fail(PAFUSE_E_ARG, "mixste2_train_forward: bad argument");
    if ((w->mlp_hidden && w->mlp_hidden != 2 * w->channels) || w->qk_scale != 0.f)
        return fail(PAFUSE_E_SHAPE, "training implements the PAFUSE configuration only (mlp_ratio = 2, qk_scale = None)");
    if (w->joints > 80 || w->frames > 80)  // attn_backward_kernel keeps one item in LDS (train_kernels.hpp)
        return fail(PAFUSE_E_SHAPE, "training: sequence lengths J=%d F=%d must be <= 80 (the single-model variant is inference only)",
                    w->joints, w->frames);
    if (saved_bytes < train_bytes(w, B)) return fail(PAFUSE_E_WORKSPACE, "mixste2_train_forward: buffer too small");
    TrainBuffers tb;
    carve_train((char*)saved, w, B, tb);
    return train_forward(w, x2d, x3d, t, B, drop_path, out, tb, (hipStream_t)stream);
}

int pafuse_mixste2_train_backward(const pafuse_mixste2_weights* w, const pafuse_mixste2_weights* grads, const float* dout,
                                  int32_t B, const float* drop_path, void* saved, size_t saved_bytes, void* stream,
                                  void* side_stream) {
    StreamDevice on_stream_device(stream);
    int rc = check_weights(w, true);
    if (rc) return rc;
    if (!grads || !dout || !saved || B <= 0) return fail(PAFUSE_E_ARG, "mixste2_train_backward: bad argument");
    if (saved_bytes < train_bytes(w, B)) return fail(PAFUSE_E_WORKSPACE, "mixste2_train_backward: buffer too small");
    TrainBuffers tb;
    carve_train((char*)saved, w, B, tb);
    return train_backward(w, grads, dout, B, drop_path, tb, (hipStream_t)stream, (hipStream_t)side_stream);
}

int pafuse_d3dp_qsample(const float* x0, const float* noise, const int64_t* t, const double* sqrt_alphas_cumprod,
                        const double* sqrt_one_minus_alphas_cumprod, double scale, float* out, int32_t B,
                        int64_t per_sample, void* stream) {
    StreamDevice on_stream_device(stream);
    if (!x0 || !noise || !t || !sqrt_alphas_cumprod || !sqrt_one_minus_alphas_cumprod || !out || B <= 0 || per_sample <= 0)
        return fail(PAFUSE_E_ARG, "d3dp_qsample: bad argument");
    const int64_t n = (int64_t)B * per_sample;
    hipLaunchKernelGGL(qsample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, noise, t,
                       sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, scale, out, per_sample, n);
    return check_launch("qsample_kernel");
}

}  // extern "C"
