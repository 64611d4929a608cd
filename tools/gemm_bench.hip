// gemm_bench.hip - stand-alone timing of the linear-layer kernels at the hot-path shapes (dev tool).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_bench.hip -o gpurun_out/gemm_bench && ./gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "kernels_r3_experiments.hpp"
using namespace pafuse;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static float* g_ref = nullptr;      // output of the last f32 run of a shape (compared with the split runs that follow)
static float* g_out_host = nullptr;
static size_t g_out_elems = 0;

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW = 1, int TR = 0, int MODE = 0>
float time_gemm(const char* tag, GemmParams p, int reps = 20) {
    using T = GemmTile<WM, WN, NT>;
    if (const char* f = getenv("GB_FILTER")) { if (!strstr(tag, f)) return 0.f; }
    if (const char* r = getenv("GB_REPS")) reps = atoi(r);
    size_t lds = (size_t)NSTAGE * (MODE == 2 ? T::STAGE_FLOATS_SPLIT : T::STAGE_FLOATS) * 4;
    if (EPI == EPI_ROWLN && !TR && (size_t)(32 * (T::BN + 4) + 5 * T::BN) * 4 > lds) lds = (size_t)(32 * (T::BN + 4) + 5 * T::BN) * 4;
    p.bf16 = MODE;
    if (MODE == 2) {
        hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W,
                           (uint8_t*)p.Wsplit, p.N, p.K);
    }
    auto k = gemm_kernel<WM, WN, NT, EPI, NSTAGE, MINW, TR, MODE>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    // keep the f32 result of a shape as the reference, report how far a split run is from it (whole-row kernels: out_n)
    double maxd = -1.0, meand = 0.0;
    const float* res = EPI == EPI_BIAS ? p.out : p.out_n;
    if (res) {
        const size_t n = (size_t)p.M * p.N;
        if (n > g_out_elems) { free(g_ref); free(g_out_host); g_ref = (float*)malloc(n * 4); g_out_host = (float*)malloc(n * 4); g_out_elems = n; }
        if (MODE == 0) {
            CK(hipMemcpy(g_ref, res, n * 4, hipMemcpyDeviceToHost));
        } else {
            CK(hipMemcpy(g_out_host, res, n * 4, hipMemcpyDeviceToHost));
            maxd = 0.0;
            for (size_t i = 0; i < n; ++i) { double d = fabs((double)g_out_host[i] - g_ref[i]); meand += d; if (d > maxd) maxd = d; }
            meand /= n;
        }
    }
    printf("%-34s M=%6ld N=%4d K=%3d tiles=%5ld lds=%6zu : %8.1f us  %6.1f TF/s (%.1f%% of f32 peak)", tag, (long)p.M, p.N, p.K,
           (long)tiles, lds, us, tf, tf / 157.3 * 100);
    if (maxd >= 0) printf("  |d vs f32| max %.2e mean %.2e", maxd, meand);
    printf("\n");
    fflush(stdout);
    return us;
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW, int ABL = 0, int BKC = 32>
float time_dma(const char* tag, GemmParams p, int reps = 20) {
    using T = DmaTile<WM, WN, NT, BKC>;
    if (const char* f = getenv("GB_FILTER")) { if (!strstr(tag, f)) return 0.f; }
    if (const char* r = getenv("GB_REPS")) reps = atoi(r);
    size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    p.bf16 = 2;
    hipLaunchKernelGGL(split_weights_kernel<BKC>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W,
                       (uint8_t*)p.Wsplit, p.N, p.K);
    auto k = gemm_dma_kernel<WM, WN, NT, EPI, NSTAGE, MINW, ABL, BKC>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    double maxd = -1.0, meand = 0.0;
    const float* res = EPI == EPI_BIAS ? p.out : p.out_n;
    if (res && g_ref) {
        const size_t n = (size_t)p.M * p.N;
        CK(hipMemcpy(g_out_host, res, n * 4, hipMemcpyDeviceToHost));
        maxd = 0.0;
        for (size_t i = 0; i < n; ++i) { double d = fabs((double)g_out_host[i] - g_ref[i]); meand += d; if (d > maxd) maxd = d; }
        meand /= n;
    }
    printf("%-34s M=%6ld N=%4d K=%3d tiles=%5ld lds=%6zu : %8.1f us  %6.1f TF/s (%.1f%% of f32 peak)", tag, (long)p.M, p.N, p.K,
           (long)tiles, lds, us, tf, tf / 157.3 * 100);
    if (maxd >= 0) printf("  |d vs f32| max %.2e mean %.2e", maxd, meand);
    printf("\n");
    fflush(stdout);
    return us;
}

// ---- round 3: tall whole-row tiles, ONE 8-wave workgroup per CU, all parts in one grid (most expensive tiles first).
// 384 -> 128 x 384 (4 x 2 waves), 256 -> 256 x 256, 224 -> 256 x 224 (8 x 1 waves): the W' stream per row is 2 - 4x
// smaller than with the 64 / 128-row two-per-CU tiles of grouped_rowln_kernel.
template <int EPI, int NST, int HANDS_BM>
__global__ void __launch_bounds__(512, 2) grouped_tall_kernel(const GroupedGemmParams g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    int s = 0;
#pragma unroll
    for (int i = 1; i < GROUP_MAX; ++i)
        if (i < g.n && b >= g.first[i]) s = i;
    const GemmParams& p = g.p[s];
    const int lb = b - g.first[s], nb = g.first[s + 1] - g.first[s];
    switch (p.N) {
        case 384: gemm_dma_tile<4, 2, 6, EPI, NST, 0, 16>(p, lb, nb, smem); break;
        case 256:
            if constexpr (HANDS_BM == 256) gemm_dma_tile<8, 1, 8, EPI, NST, 0, 16>(p, lb, nb, smem);
            else gemm_dma_tile<4, 2, 4, EPI, NST, 0, 16>(p, lb, nb, smem);
            break;
        case 224: gemm_dma_tile<8, 1, 7, EPI, NST, 0, 16>(p, lb, nb, smem); break;
        default: break;
    }
}

// both operands pre-split (round 3): A' made here from p.A (split_rows_kernel<16>, padded rows), W' 16-deep
static uint8_t* g_aimg = nullptr;   // activation image, [768/16][Mpad][96]
static uint8_t* g_oimg = nullptr;   // output image (timing of the split-store epilogue)
template <int WM, int WN, int MT, int NT, int EPI, int NSTAGE, int MINW>
float time_pre(const char* tag, GemmParams p, bool split_out = false, int reps = 20) {
    using T = PreTile<WM, WN, MT, NT>;
    if (const char* f = getenv("GB_FILTER")) { if (!strstr(tag, f)) return 0.f; }
    if (const char* r = getenv("GB_REPS")) reps = atoi(r);
    const size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    const int64_t Mpad = (p.M + PRE_ROW_PAD - 1) / PRE_ROW_PAD * PRE_ROW_PAD;
    p.bf16 = 2;
    hipLaunchKernelGGL(split_weights_kernel<16>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W,
                       (uint8_t*)p.Wsplit, p.N, p.K);
    hipLaunchKernelGGL(split_rows_kernel<16>, dim3((unsigned)((p.M * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.A, g_aimg, p.M, p.K, Mpad);
    p.Asplit = g_aimg, p.A_pad = Mpad;
    if (split_out) { p.out_pad = Mpad; if (EPI == EPI_BIAS) p.out_s = g_oimg; else p.out_n_s = g_oimg; }
    auto k = gemm_pre_kernel<WM, WN, MT, NT, EPI, NSTAGE, MINW>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    double maxd = -1.0, meand = 0.0;
    const float* res = EPI == EPI_BIAS ? p.out : p.out_n;
    if (res && g_ref && !split_out) {
        const size_t n = (size_t)p.M * p.N;
        CK(hipMemcpy(g_out_host, res, n * 4, hipMemcpyDeviceToHost));
        maxd = 0.0;
        for (size_t i = 0; i < n; ++i) { double d = fabs((double)g_out_host[i] - g_ref[i]); meand += d; if (d > maxd) maxd = d; }
        meand /= n;
    }
    printf("%-40s M=%6ld N=%4d K=%3d tiles=%5ld lds=%6zu : %8.1f us  %6.1f TF/s (%.3f of 416.7)", tag, (long)p.M, p.N, p.K,
           (long)tiles, lds, us, tf, tf / 416.7);
    if (maxd >= 0) printf("  |d vs f32| max %.2e mean %.2e", maxd, meand);
    printf("\n");
    fflush(stdout);
    return us;
}

// wave-specialised tile (round 3): 4 producer waves + WM x WN consumer waves, A fp32, W' 16-deep
template <int WM, int WN, int NT, int EPI>
float time_ws(const char* tag, GemmParams p, int reps = 20) {
    using T = WsTile<WM, WN, NT>;
    if (const char* f = getenv("GB_FILTER")) { if (!strstr(tag, f)) return 0.f; }
    if (const char* r = getenv("GB_REPS")) reps = atoi(r);
    const size_t lds = (size_t)T::NSTAGE * T::STAGE_BYTES;
    p.bf16 = 2;
    hipLaunchKernelGGL(split_weights_kernel<16>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W,
                       (uint8_t*)p.Wsplit, p.N, p.K);
    auto k = gemm_ws_kernel<WM, WN, NT, EPI>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    double maxd = -1.0, meand = 0.0;
    const float* res = EPI == EPI_BIAS ? p.out : p.out_n;
    if (res && g_ref) {
        const size_t n = (size_t)p.M * p.N;
        CK(hipMemcpy(g_out_host, res, n * 4, hipMemcpyDeviceToHost));
        maxd = 0.0;
        for (size_t i = 0; i < n; ++i) { double d = fabs((double)g_out_host[i] - g_ref[i]); meand += d; if (d > maxd) maxd = d; }
        meand /= n;
    }
    printf("%-44s M=%6ld N=%4d K=%3d tiles=%5ld lds=%6zu : %8.1f us  %6.1f TF/s (%.3f of 416.7)", tag, (long)p.M, p.N, p.K,
           (long)tiles, lds, us, tf, tf / 416.7);
    if (maxd >= 0) printf("  |d vs f32| max %.2e mean %.2e", maxd, meand);
    printf("\n");
    fflush(stdout);
    return us;
}

// plain split layer on the 16x16x32 MFMA shape (round 3)
template <int NB, int MINW>
float time_16(const char* tag, GemmParams p, int reps = 20) {
    using T = Tile16<NB>;
    if (const char* f = getenv("GB_FILTER")) { if (!strstr(tag, f)) return 0.f; }
    if (const char* r = getenv("GB_REPS")) reps = atoi(r);
    const size_t lds = T::STAGE_BYTES;
    p.bf16 = 2;
    hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W,
                       (uint8_t*)p.Wsplit, p.N, p.K);
    auto k = gemm16_kernel<NB, MINW>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    double maxd = -1.0, meand = 0.0;
    if (p.out && g_ref) {
        const size_t n = (size_t)p.M * p.N;
        CK(hipMemcpy(g_out_host, p.out, n * 4, hipMemcpyDeviceToHost));
        maxd = 0.0;
        for (size_t i = 0; i < n; ++i) { double d = fabs((double)g_out_host[i] - g_ref[i]); meand += d; if (d > maxd) maxd = d; }
        meand /= n;
    }
    printf("%-44s M=%6ld N=%4d K=%3d tiles=%5ld lds=%6zu : %8.1f us  %6.1f TF/s (%.3f of 416.7)", tag, (long)p.M, p.N, p.K,
           (long)tiles, lds, us, tf, tf / 416.7);
    if (maxd >= 0) printf("  |d vs f32| max %.2e mean %.2e", maxd, meand);
    printf("\n");
    fflush(stdout);
    return us;
}

int main() {
    const int64_t Mmax = 73440;
    float *A, *W, *bias, *out, *x, *xn, *vec, *xo;
    uint8_t* Wsp;
    CK(hipMalloc(&Wsp, 1152 * 768 * 6));
    CK(hipMalloc(&A, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&bias, 1152 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4)); CK(hipMalloc(&x, Mmax * 384 * 4)); CK(hipMalloc(&xn, Mmax * 384 * 4));
    CK(hipMalloc(&vec, 1152 * 4));
    CK(hipMalloc(&xo, Mmax * 384 * 4));
    CK(hipMalloc(&g_aimg, (size_t)(Mmax + 256) * 768 * 6)); CK(hipMalloc(&g_oimg, (size_t)(Mmax + 256) * 1152 * 6));
    std::vector<float> h(Mmax * 768);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(A, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, h.data() + 5000, 1152 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, h.data(), Mmax * 384 * 4, hipMemcpyHostToDevice));
    GemmParams p{};
    p.A = A, p.W = W, p.bias = bias, p.out = out, p.Wsplit = Wsp;
    p.M = 25920, p.N = 1152, p.K = 384, p.act = 0;
    { const char* f = getenv("GB_FILTER"); if (!f) time_gemm<4, 1, 4, EPI_BIAS, 2>("(warm-up, ignore)", p, 200); }
    // ---- round 3: plain layers on the 16x16x32 shape (tag "m16"); each shape's f32 line first (reference result)
    for (int rep = 0; rep < 2; ++rep) {   // twice: the second pass shows the run-to-run spread inside one process
    p.M = 25920, p.N = 1152, p.K = 384, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("m16-ref body qkv  f32   <4,1,2> s1 minw5", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("m16-ref body qkv  split <4,1,4> s1 minw2 (r2 production)", p);
    time_16<8, 2>("m16 body qkv  128x128 minw2", p);
    time_16<8, 3>("m16 body qkv  128x128 minw3", p);
    time_16<4, 4>("m16 body qkv  128x64 minw4", p);
    time_16<6, 3>("m16 body qkv  128x96 minw3", p);
    p.N = 768, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("m16-ref body fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("m16-ref body fc1+gelu split <4,1,4> s1 (r2 production)", p);
    time_16<8, 2>("m16 body fc1+gelu 128x128 minw2", p);
    time_16<8, 3>("m16 body fc1+gelu 128x128 minw3", p);
    time_16<4, 4>("m16 body fc1+gelu 128x64 minw4", p);
    p.M = 73440, p.N = 672, p.K = 224, p.act = 0;
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1>("m16-ref face qkv  f32   <4,1,3> s1 TR", p);
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 2>("m16-ref face qkv  split <4,1,3> s1 TR (r2 production)", p);
    time_16<6, 3>("m16 face qkv  128x96 minw3", p);
    time_16<6, 4>("m16 face qkv  128x96 minw4", p);
    time_16<14, 2>("m16 face qkv  128x224 minw2", p);
    p.N = 448, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("m16-ref face fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("m16-ref face fc1+gelu split <4,1,2> s1 (r2 production)", p);
    time_16<4, 4>("m16 face fc1+gelu 128x64 minw4", p);
    time_16<14, 2>("m16 face fc1+gelu 128x224 minw2", p);
    p.M = 45360, p.N = 768, p.K = 256, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("m16-ref hands qkv f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("m16-ref hands qkv split <4,1,2> s1 (r2 production)", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("m16-ref hands qkv split <4,1,4> s1 minw2", p);
    time_16<8, 2>("m16 hands qkv 128x128 minw2", p);
    time_16<8, 3>("m16 hands qkv 128x128 minw3", p);
    time_16<4, 4>("m16 hands qkv 128x64 minw4", p);
    }
    // ---- round 3: wave-specialised tiles (tag "ws"); each shape's f32 line first (reference result)
    p.M = 25920, p.N = 1152, p.K = 384, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("ws-ref body qkv  f32   <4,1,2> s1 minw5", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("ws-ref body qkv  split <4,1,4> s1 minw2 (r2 production)", p);
    time_ws<4, 2, 6, EPI_BIAS>("ws body qkv  <4,2,6> 128x384", p);
    time_ws<4, 2, 4, EPI_BIAS>("ws body qkv  <4,2,4> 128x256 (N % 256 != 0: timing only)", p);
    time_ws<8, 1, 4, EPI_BIAS>("ws body qkv  <8,1,4> 256x128", p);
    p.N = 768, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("ws-ref body fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("ws-ref body fc1+gelu split <4,1,4> s1 (r2 production)", p);
    time_ws<4, 2, 6, EPI_BIAS>("ws body fc1+gelu <4,2,6> 128x384", p);
    time_ws<4, 2, 4, EPI_BIAS>("ws body fc1+gelu <4,2,4> 128x256", p);
    p.M = 73440, p.N = 672, p.K = 224, p.act = 0;
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1>("ws-ref face qkv  f32   <4,1,3> s1 TR", p);
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 2>("ws-ref face qkv  split <4,1,3> s1 TR (r2 production)", p);
    time_ws<8, 1, 7, EPI_BIAS>("ws face qkv  <8,1,7> 256x224", p);
    time_ws<4, 2, 3, EPI_BIAS>("ws face qkv  <4,2,3> 128x192 (N % 192 != 0: timing only)", p);
    p.N = 448, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("ws-ref face fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("ws-ref face fc1+gelu split <4,1,2> s1 (r2 production)", p);
    time_ws<8, 1, 7, EPI_BIAS>("ws face fc1+gelu <8,1,7> 256x224", p);
    p.M = 45360, p.N = 768, p.K = 256, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("ws-ref hands qkv f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("ws-ref hands qkv split <4,1,2> s1 (r2 production)", p);
    time_ws<4, 2, 4, EPI_BIAS>("ws hands qkv <4,2,4> 128x256", p);
    time_ws<4, 2, 6, EPI_BIAS>("ws hands qkv <4,2,6> 128x384", p);
    {   // lnfold: what the whole-row kernels save when the next LayerNorm is left to the consumer (stats only, no xn)
        GemmParams q{};
        q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = xo, q.out_n = xn, q.Wsplit = Wsp;
        q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f, q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
        GemmParams f = q;
        f.out_n = nullptr, f.ln_stats = g_oimg ? (float*)g_oimg : xn;
        for (int rep = 0; rep < 2; ++rep) {
            q.M = f.M = 25920, q.N = f.N = 384, q.K = f.K = 768;
            time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("lnfold body fc2  <2,2,6> st2 production (x and xn)", q);
            time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("lnfold body fc2  <2,2,6> st2 stats only (x, mean, rstd)", f);
            q.K = f.K = 384;
            time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("lnfold body proj <2,2,6> st2 production (x and xn)", q);
            time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("lnfold body proj <2,2,6> st2 stats only (x, mean, rstd)", f);
            q.M = f.M = 45360, q.N = f.N = 256, q.K = f.K = 512;
            time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("lnfold hands fc2 <2,2,4> st2 production (x and xn)", q);
            time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("lnfold hands fc2 <2,2,4> st2 stats only (x, mean, rstd)", f);
            q.K = f.K = 256;
            time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("lnfold hands proj <2,2,4> st2 production (x and xn)", q);
            time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("lnfold hands proj <2,2,4> st2 stats only (x, mean, rstd)", f);
            q.M = f.M = 73440, q.N = f.N = 224, q.K = f.K = 448;
            time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("lnfold face fc2  <4,1,7> st2 production (x and xn)", q);
            time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("lnfold face fc2  <4,1,7> st2 stats only (x, mean, rstd)", f);
            q.K = f.K = 224;
            time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("lnfold face proj <4,1,7> st2 production (x and xn)", q);
            time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("lnfold face proj <4,1,7> st2 stats only (x, mean, rstd)", f);
        }
    }
    {
        GemmParams q{};
        q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = xo, q.out_n = xn, q.Wsplit = Wsp;
        q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f, q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
        q.M = 45360, q.N = 256, q.K = 512;
        time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1>("ws-ref hands fc2 rowln f32   <2,2,4> minw2", q);
        time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("ws-ref hands fc2 rowln dma16 <2,2,4> st2 (r2 production)", q);
        time_ws<4, 2, 4, EPI_ROWLN>("ws hands fc2 rowln <4,2,4> 128x256", q);
        q.K = 256;
        time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1>("ws-ref hands proj rowln f32   <2,2,4> minw2", q);
        time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("ws-ref hands proj rowln dma16 <2,2,4> st2 (r2 production)", q);
        time_ws<4, 2, 4, EPI_ROWLN>("ws hands proj rowln <4,2,4> 128x256", q);
        q.M = 73440, q.N = 224, q.K = 448;
        time_gemm<1, 7, 1, EPI_ROWLN, 1, 1, 1>("ws-ref face fc2  rowln f32   <1,7,1>", q);
        time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("ws-ref face fc2 rowln dma16 <4,1,7> st2 (r2 production)", q);
        time_ws<8, 1, 7, EPI_ROWLN>("ws face fc2 rowln <8,1,7> 256x224", q);
        q.M = 25920, q.N = 384, q.K = 768;
        time_gemm<1, 4, 3, EPI_ROWLN, 1, 1, 1>("ws-ref body fc2  rowln f32   <1,4,3>", q);
        time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("ws-ref body fc2 rowln dma16 <2,2,6> st2 (r2 production)", q);
        time_ws<4, 2, 6, EPI_ROWLN>("ws body fc2 rowln <4,2,6> 128x384", q);
        q.K = 384;
        time_gemm<1, 4, 3, EPI_ROWLN, 1, 1, 1>("ws-ref body proj rowln f32   <1,4,3>", q);
        time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("ws-ref body proj rowln dma16 <2,2,6> st2 (r2 production)", q);
        time_ws<4, 2, 6, EPI_ROWLN>("ws body proj rowln <4,2,6> 128x384", q);
    }
    // ---- round 3: both operands pre-split (tag "pre"); each shape's f32 line first (reference result)
    p.M = 25920, p.N = 1152, p.K = 384, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("pre-ref body qkv  f32   <4,1,2> s1 minw5", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("pre-ref body qkv  split <4,1,4> s1 minw2 (r2 production)", p);
    time_pre<4, 1, 2, 4, EPI_BIAS, 2, 2>("pre body qkv  <4,1,2,4> 256x128 st2", p);
    time_pre<4, 1, 2, 4, EPI_BIAS, 3, 2>("pre body qkv  <4,1,2,4> 256x128 st3", p);
    time_pre<2, 2, 2, 4, EPI_BIAS, 2, 2>("pre body qkv  <2,2,2,4> 128x256 st2 (N%256!=0: skip)", p);
    time_pre<4, 1, 1, 4, EPI_BIAS, 2, 2>("pre body qkv  <4,1,1,4> 128x128 st2", p);
    time_pre<4, 1, 1, 4, EPI_BIAS, 2, 3>("pre body qkv  <4,1,1,4> 128x128 st2 minw3", p);
    time_pre<2, 1, 2, 4, EPI_BIAS, 2, 2>("pre body qkv  <2,1,2,4> 128x128 2 waves st2", p);
    p.N = 768, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("pre-ref body fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("pre-ref body fc1+gelu split <4,1,4> s1 (r2 production)", p);
    time_pre<4, 1, 2, 4, EPI_BIAS, 2, 2>("pre body fc1+gelu <4,1,2,4> 256x128 st2 f32 out", p);
    time_pre<4, 1, 2, 4, EPI_BIAS, 2, 2>("pre body fc1+gelu <4,1,2,4> 256x128 st2 split out", p, true);
    time_pre<2, 2, 2, 4, EPI_BIAS, 2, 2>("pre body fc1+gelu <2,2,2,4> 128x256 st2 split out", p, true);
    time_pre<4, 1, 1, 4, EPI_BIAS, 2, 3>("pre body fc1+gelu <4,1,1,4> 128x128 st2 minw3 split out", p, true);
    p.M = 73440, p.N = 672, p.K = 224, p.act = 0;
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1>("pre-ref face qkv  f32   <4,1,3> s1 TR", p);
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 2>("pre-ref face qkv  split <4,1,3> s1 TR (r2 production)", p);
    time_pre<4, 1, 2, 3, EPI_BIAS, 2, 2>("pre face qkv  <4,1,2,3> 256x96 st2", p);
    time_pre<4, 1, 1, 7, EPI_BIAS, 2, 2>("pre face qkv  <4,1,1,7> 128x224 st2", p);
    time_pre<2, 1, 2, 7, EPI_BIAS, 2, 1>("pre face qkv  <2,1,2,7> 128x224 2 waves st2", p);
    p.N = 448, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("pre-ref face fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("pre-ref face fc1+gelu split <4,1,2> s1 (r2 production)", p);
    time_pre<4, 1, 2, 2, EPI_BIAS, 2, 2>("pre face fc1+gelu <4,1,2,2> 256x64 st2 split out", p, true);
    time_pre<4, 1, 1, 7, EPI_BIAS, 2, 2>("pre face fc1+gelu <4,1,1,7> 128x224 st2 split out", p, true);
    time_pre<4, 1, 1, 7, EPI_BIAS, 2, 2>("pre face fc1+gelu <4,1,1,7> 128x224 st2 f32 out", p);
    p.M = 45360, p.N = 768, p.K = 256, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("pre-ref hands qkv f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("pre-ref hands qkv split <4,1,2> s1 (r2 production)", p);
    time_pre<4, 1, 2, 4, EPI_BIAS, 2, 2>("pre hands qkv <4,1,2,4> 256x128 st2", p);
    time_pre<2, 2, 2, 4, EPI_BIAS, 2, 2>("pre hands qkv <2,2,2,4> 128x256 st2", p);
    {
        GemmParams q{};
        q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = xo, q.out_n = xn, q.Wsplit = Wsp;
        q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f, q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
        q.M = 45360, q.N = 256, q.K = 512;
        time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1>("pre-ref hands fc2 rowln f32   <2,2,4> minw2", q);
        time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("pre-ref hands fc2 rowln dma16 <2,2,4> st2 (r2 production)", q);
        time_pre<2, 2, 2, 4, EPI_ROWLN, 2, 2>("pre hands fc2 rowln <2,2,2,4> 128x256 st2", q);
        time_pre<2, 2, 2, 4, EPI_ROWLN, 2, 2>("pre hands fc2 rowln <2,2,2,4> 128x256 st2 split out", q, true);
        time_pre<2, 2, 1, 4, EPI_ROWLN, 2, 2>("pre hands fc2 rowln <2,2,1,4> 64x256 st2", q);
        time_pre<2, 2, 1, 4, EPI_ROWLN, 2, 3>("pre hands fc2 rowln <2,2,1,4> 64x256 st2 minw3", q);
        q.M = 73440, q.N = 224, q.K = 448;
        time_gemm<1, 7, 1, EPI_ROWLN, 1, 1, 1>("pre-ref face fc2  rowln f32   <1,7,1>", q);
        time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("pre-ref face fc2 rowln dma16 <4,1,7> st2 (r2 production)", q);
        time_pre<4, 1, 1, 7, EPI_ROWLN, 2, 2>("pre face fc2 rowln <4,1,1,7> 128x224 st2", q);
        time_pre<4, 1, 1, 7, EPI_ROWLN, 2, 2>("pre face fc2 rowln <4,1,1,7> 128x224 st2 split out", q, true);
        time_pre<2, 1, 1, 7, EPI_ROWLN, 2, 2>("pre face fc2 rowln <2,1,1,7> 64x224 st2", q);
        q.M = 25920, q.N = 384, q.K = 768;
        time_gemm<1, 4, 3, EPI_ROWLN, 1, 1, 1>("pre-ref body fc2  rowln f32   <1,4,3>", q);
        time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("pre-ref body fc2 rowln dma16 <2,2,6> st2 (r2 production)", q);
        time_pre<4, 2, 1, 6, EPI_ROWLN, 3, 2>("pre body fc2 rowln <4,2,1,6> 128x384 8 waves st3", q);
        time_pre<4, 2, 1, 6, EPI_ROWLN, 2, 2>("pre body fc2 rowln <4,2,1,6> 128x384 8 waves st2", q);
    }
    // ---- f32 production tile (reference result) / best register-staged split tile / LDS-DMA pipelined split kernel
    p.M = 25920, p.N = 1152, p.K = 384, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("body qkv  f32   <4,1,2> s1 minw5", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("body qkv  split <4,1,4> s1 minw2", p);
    time_dma<8, 1, 4, EPI_BIAS, 3, 2, 0, 16>("body qkv  dma16p <8,1,4> st3", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 3, 0, 16>("body qkv  dma16 <4,1,4> st2 minw3 (3/CU)", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2, 0, 16>("body qkv  dma16 <4,1,4> st2 minw2", p);
    time_dma<4, 1, 2, EPI_BIAS, 2, 4, 0, 16>("body qkv  dma16 <4,1,2> st2 minw4 (4/CU)", p);
    time_dma<4, 1, 3, EPI_BIAS, 2, 3, 0, 16>("body qkv  dma16 <4,1,3> st2 minw3 (3/CU)", p);
    time_dma<2, 2, 4, EPI_BIAS, 2, 3, 0, 16>("body qkv  dma16 <2,2,4> st2 minw3 (64x256)", p);
    time_dma<4, 1, 4, EPI_BIAS, 3, 2, 0, 16>("body qkv  dma16p <4,1,4> st3 (2/CU)", p);
    time_dma<4, 1, 2, EPI_BIAS, 3, 2, 0, 16>("body qkv  dma16p <4,1,2> st3 (3/CU)", p);
    time_dma<8, 1, 2, EPI_BIAS, 3, 2, 0, 16>("body qkv  dma16p <8,1,2> st3", p);
    time_dma<8, 1, 3, EPI_BIAS, 3, 2, 0, 16>("body qkv  dma16p <8,1,3> st3", p);
    time_dma<8, 1, 4, EPI_BIAS, 3, 2, 2, 16>("body qkv  dma16p <8,1,4> st3 ABL2 compute", p);
    time_dma<8, 1, 4, EPI_BIAS, 3, 2, 4, 16>("body qkv  dma16p <8,1,4> st3 ABL4 lds+mfma no epi", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2>("body qkv  dma <8,1,4> st2", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2, 1>("body qkv  dma <8,1,4> st2 ABL1 stream only", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2, 2>("body qkv  dma <8,1,4> st2 ABL2 compute only", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2, 3>("body qkv  dma <8,1,4> st2 ABL3 lds+mfma only", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2, 4>("body qkv  dma <8,1,4> st2 ABL4 lds+mfma, no epilogue", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2, 3>("body qkv  dma <4,1,4> st2 ABL3 lds+mfma only", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2, 4>("body qkv  dma <4,1,4> st2 ABL4 lds+mfma, no epilogue", p);
    time_dma<4, 1, 2, EPI_BIAS, 2, 2, 4>("body qkv  dma <4,1,2> st2 ABL4 lds+mfma, no epilogue", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2, 1>("body qkv  dma <4,1,4> st2 ABL1 stream only", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2, 2>("body qkv  dma <4,1,4> st2 ABL2 compute only", p);
    time_dma<4, 1, 4, EPI_BIAS, 3, 2>("body qkv  dma <4,1,4> st3", p);
    time_dma<4, 1, 4, EPI_BIAS, 2, 2>("body qkv  dma <4,1,4> st2", p);
    time_dma<4, 1, 2, EPI_BIAS, 2, 2>("body qkv  dma <4,1,2> st2", p);
    time_dma<4, 1, 2, EPI_BIAS, 3, 2>("body qkv  dma <4,1,2> st3", p);
    time_dma<8, 1, 2, EPI_BIAS, 3, 2>("body qkv  dma <8,1,2> st3", p);
    time_dma<8, 1, 3, EPI_BIAS, 2, 2>("body qkv  dma <8,1,3> st2", p);
    time_dma<8, 1, 3, EPI_BIAS, 3, 2>("body qkv  dma <8,1,3> st3", p);
    time_dma<4, 2, 2, EPI_BIAS, 3, 2>("body qkv  dma <4,2,2> st3", p);
    time_dma<8, 2, 2, EPI_BIAS, 2, 4>("body qkv  dma <8,2,2> st2 (16 waves)", p);
    p.N = 768, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("body fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("body fc1+gelu split <4,1,4> s1", p);
    { GemmParams pn = p; pn.act = 0; time_gemm<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("body fc1 NOgelu split <4,1,4> s1", pn); }
    time_dma<8, 1, 4, EPI_BIAS, 2, 2>("body fc1+gelu dma <8,1,4> st2", p);
    time_dma<4, 1, 4, EPI_BIAS, 3, 1>("body fc1+gelu dma <4,1,4> st3", p);
    time_dma<8, 1, 2, EPI_BIAS, 3, 2>("body fc1+gelu dma <8,1,2> st3", p);
    p.M = 73440, p.N = 672, p.K = 224, p.act = 0;
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1>("face qkv  f32   <4,1,3> s1 TR", p);
    time_gemm<4, 1, 3, EPI_BIAS, 1, 1, 1, 2>("face qkv  split <4,1,3> s1 TR", p);
    time_dma<8, 1, 3, EPI_BIAS, 2, 2>("face qkv  dma <8,1,3> st2", p);
    time_dma<8, 1, 3, EPI_BIAS, 3, 2>("face qkv  dma <8,1,3> st3", p);
    time_dma<4, 1, 3, EPI_BIAS, 3, 1>("face qkv  dma <4,1,3> st3", p);
    time_dma<8, 1, 7, EPI_BIAS, 2, 2>("face qkv  dma <8,1,7> st2", p);
    p.N = 448, p.act = 1;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("face fc1+gelu f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("face fc1+gelu split <4,1,2> s1", p);
    time_dma<8, 1, 2, EPI_BIAS, 3, 2>("face fc1+gelu dma <8,1,2> st3", p);
    time_dma<8, 1, 7, EPI_BIAS, 2, 2>("face fc1+gelu dma <8,1,7> st2", p);
    p.M = 45360, p.N = 768, p.K = 256, p.act = 0;
    time_gemm<4, 1, 2, EPI_BIAS, 1, 5>("hands qkv f32   <4,1,2> s1", p);
    time_gemm<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("hands qkv split <4,1,2> s1", p);
    time_dma<8, 1, 4, EPI_BIAS, 2, 2>("hands qkv dma <8,1,4> st2", p);
    time_dma<4, 1, 4, EPI_BIAS, 3, 1>("hands qkv dma <4,1,4> st3", p);
    // whole-row kernels
    GemmParams q{};
    q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = xo, q.out_n = xn, q.Wsplit = Wsp;
    q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f, q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
    q.M = 25920, q.N = 384, q.K = 768;
    time_gemm<1, 4, 3, EPI_ROWLN, 1, 1, 1>("body fc2  rowln f32   <1,4,3>", q);
    time_gemm<4, 4, 3, EPI_ROWLN, 1, 1, 1, 2>("body fc2  rowln split <4,4,3>", q);
    time_dma<2, 4, 3, EPI_ROWLN, 2, 2>("body fc2  rowln dma32 <2,4,3> st2", q);
    time_dma<4, 4, 3, EPI_ROWLN, 3, 4, 0, 16>("body fc2  rowln dma16 <4,4,3> st3", q);
    time_dma<4, 4, 3, EPI_ROWLN, 2, 4, 0, 16>("body fc2  rowln dma16 <4,4,3> st2", q);
    time_dma<2, 4, 3, EPI_ROWLN, 3, 2, 0, 16>("body fc2  rowln dma16 <2,4,3> st3", q);
    time_dma<2, 4, 3, EPI_ROWLN, 2, 2, 0, 16>("body fc2  rowln dma16 <2,4,3> st2 (2/CU)", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("body fc2  rowln dma16 <4,2,6> st3", q);
    time_dma<2, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("body fc2  rowln dma16 <2,2,6> st3", q);
    time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("body fc2  rowln dma16 <2,2,6> st2 (2/CU)", q);
    time_dma<6, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("body fc2  rowln dma16 <6,2,6> st3", q);
    time_dma<4, 3, 4, EPI_ROWLN, 3, 2, 0, 16>("body fc2  rowln dma16 <4,3,4> st3", q);
    time_dma<4, 4, 3, EPI_ROWLN, 3, 4, 1, 16>("body fc2  rowln dma16 <4,4,3> st3 ABL1 stream", q);
    time_dma<4, 4, 3, EPI_ROWLN, 3, 4, 2, 16>("body fc2  rowln dma16 <4,4,3> st3 ABL2 compute", q);
    q.K = 384;
    time_gemm<1, 4, 3, EPI_ROWLN, 1, 1, 1>("body proj rowln f32   <1,4,3>", q);
    time_gemm<4, 4, 3, EPI_ROWLN, 1, 1, 1, 2>("body proj rowln split <4,4,3>", q);
    time_dma<4, 4, 3, EPI_ROWLN, 3, 4, 0, 16>("body proj rowln dma16 <4,4,3> st3", q);
    time_dma<2, 4, 3, EPI_ROWLN, 3, 2, 0, 16>("body proj rowln dma16 <2,4,3> st3", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("body proj rowln dma16 <4,2,6> st3", q);
    time_dma<4, 3, 4, EPI_ROWLN, 3, 2, 0, 16>("body proj rowln dma16 <4,3,4> st3", q);
    q.M = 45360, q.N = 256, q.K = 512;
    time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1>("hands fc2 rowln f32   <2,2,4> minw2", q);
    time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1, 2>("hands fc2 rowln split <2,2,4>", q);
    time_dma<4, 4, 2, EPI_ROWLN, 3, 4, 0, 16>("hands fc2 rowln dma16 <4,4,2> st3", q);
    time_dma<4, 2, 4, EPI_ROWLN, 3, 2, 0, 16>("hands fc2 rowln dma16 <4,2,4> st3", q);
    time_dma<4, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("hands fc2 rowln dma16 <4,2,4> st2", q);
    time_dma<2, 4, 2, EPI_ROWLN, 3, 2, 0, 16>("hands fc2 rowln dma16 <2,4,2> st3", q);
    time_dma<2, 2, 4, EPI_ROWLN, 3, 2, 0, 16>("hands fc2 rowln dma16 <2,2,4> st3", q);
    time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("hands fc2 rowln dma16 <2,2,4> st2 (2/CU)", q);
    time_dma<8, 2, 4, EPI_ROWLN, 3, 4, 0, 16>("hands fc2 rowln dma16 <8,2,4> st3", q);
    q.M = 73440, q.N = 224, q.K = 448;
    time_gemm<1, 7, 1, EPI_ROWLN, 1, 1, 1>("face fc2  rowln f32   <1,7,1>", q);
    time_gemm<2, 1, 7, EPI_ROWLN, 1, 1, 1, 2>("face fc2  rowln split <2,1,7>", q);
    time_dma<2, 7, 1, EPI_ROWLN, 3, 3, 0, 16>("face fc2  rowln dma16 <2,7,1> st3", q);
    time_dma<4, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("face fc2  rowln dma16 <4,1,7> st3", q);
    time_dma<8, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("face fc2  rowln dma16 <8,1,7> st3", q);
    time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("face fc2  rowln dma16 <4,1,7> st2", q);
    time_dma<2, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("face fc2  rowln dma16 <2,1,7> st3", q);
    time_dma<2, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("face fc2  rowln dma16 <2,1,7> st2", q);
    time_dma<6, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("face fc2  rowln dma16 <6,1,7> st2", q);
    q.K = 224;
    time_dma<8, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("face proj rowln dma16 <8,1,7> st3", q);
    time_gemm<1, 7, 1, EPI_ROWLN, 1, 1, 1>("face proj rowln f32   <1,7,1>", q);
    time_gemm<2, 1, 7, EPI_ROWLN, 1, 1, 1, 2>("face proj rowln split <2,1,7>", q);
    time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("face proj rowln dma16 <4,1,7> st2", q);
    time_dma<4, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("face proj rowln dma16 <4,1,7> st3", q);
    q.M = 45360, q.N = 256, q.K = 256;
    time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1>("hands proj rowln f32   <2,2,4> minw2", q);
    time_gemm<2, 2, 4, EPI_ROWLN, 1, 2, 1, 2>("hands proj rowln split <2,2,4>", q);
    time_dma<2, 2, 4, EPI_ROWLN, 3, 2, 0, 16>("hands proj rowln dma16 <2,2,4> st3", q);
    time_dma<4, 2, 4, EPI_ROWLN, 3, 2, 0, 16>("hands proj rowln dma16 <4,2,4> st3", q);
    time_dma<2, 4, 2, EPI_ROWLN, 3, 2, 0, 16>("hands proj rowln dma16 <2,4,2> st3", q);
    // ---- round 3: what one tall tile costs (1 round: every CU runs at most one tile, the launch time IS a tile's time)
    q.M = 25920, q.N = 384, q.K = 768;
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("tallabl body fc2 <4,2,6> st3 full", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 1, 16>("tallabl body fc2 <4,2,6> st3 ABL1 stream+epilogue", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 2, 16>("tallabl body fc2 <4,2,6> st3 ABL2 compute+epilogue", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 4, 16>("tallabl body fc2 <4,2,6> st3 ABL4 lds+mfma only", q);
    time_dma<4, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("tallabl body fc2 <4,2,6> st2 full", q);
    time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("tallabl body fc2 <2,2,6> st2 (2/CU) full", q);
    time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 1, 16>("tallabl body fc2 <2,2,6> st2 (2/CU) ABL1 stream+epilogue", q);
    time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 4, 16>("tallabl body fc2 <2,2,6> st2 (2/CU) ABL4 lds+mfma only", q);
    q.K = 384;
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 0, 16>("tallabl body proj <4,2,6> st3 full", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 1, 16>("tallabl body proj <4,2,6> st3 ABL1 stream+epilogue", q);
    time_dma<4, 2, 6, EPI_ROWLN, 3, 2, 4, 16>("tallabl body proj <4,2,6> st3 ABL4 lds+mfma only", q);
    time_dma<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>("tallabl body proj <2,2,6> st2 (2/CU) full", q);
    q.M = 45360, q.N = 256, q.K = 512;
    time_dma<8, 1, 8, EPI_ROWLN, 3, 2, 0, 16>("tallabl hands fc2 <8,1,8> st3 full", q);
    time_dma<8, 1, 8, EPI_ROWLN, 3, 2, 1, 16>("tallabl hands fc2 <8,1,8> st3 ABL1 stream+epilogue", q);
    time_dma<8, 1, 8, EPI_ROWLN, 3, 2, 4, 16>("tallabl hands fc2 <8,1,8> st3 ABL4 lds+mfma only", q);
    time_dma<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>("tallabl hands fc2 <2,2,4> st2 (2/CU) full", q);
    q.M = 73440, q.N = 224, q.K = 448;
    time_dma<8, 1, 7, EPI_ROWLN, 3, 2, 0, 16>("tallabl face fc2 <8,1,7> st3 full", q);
    time_dma<8, 1, 7, EPI_ROWLN, 3, 2, 1, 16>("tallabl face fc2 <8,1,7> st3 ABL1 stream+epilogue", q);
    time_dma<8, 1, 7, EPI_ROWLN, 3, 2, 4, 16>("tallabl face fc2 <8,1,7> st3 ABL4 lds+mfma only", q);
    time_dma<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>("tallabl face fc2 <4,1,7> st2 (2/CU) full", q);
    // ---- grouped whole-row launch (body + face + hands in one grid) against the per-part launches
    if (!getenv("GB_FILTER") || strstr("grouped", getenv("GB_FILTER"))) {
        struct Part { int64_t M; int N, K; uint8_t* ws; float *xo, *xn; };
        for (int fc2 = 1; fc2 >= 0; --fc2) {
            Part pt[3] = {{25920, 384, fc2 ? 768 : 384}, {73440, 224, fc2 ? 448 : 224}, {45360, 256, fc2 ? 512 : 256}};
            GemmParams gp[3];
            for (int i = 0; i < 3; ++i) {
                CK(hipMalloc(&pt[i].ws, (size_t)pt[i].N * pt[i].K * 6));
                CK(hipMalloc(&pt[i].xo, pt[i].M * pt[i].N * 4)); CK(hipMalloc(&pt[i].xn, pt[i].M * pt[i].N * 4));
                hipLaunchKernelGGL(split_weights_kernel<16>, dim3((unsigned)(((int64_t)pt[i].N * (pt[i].K / 8) + 255) / 256)), dim3(256), 0, 0,
                                   W + i * 1000, pt[i].ws, pt[i].N, pt[i].K);
                gp[i] = q;
                gp[i].W = W + i * 1000, gp[i].Wsplit = pt[i].ws, gp[i].M = pt[i].M, gp[i].N = pt[i].N, gp[i].K = pt[i].K;
                gp[i].out_x = pt[i].xo, gp[i].out_n = pt[i].xn, gp[i].bf16 = 2;
            }
            auto k0 = gemm_dma_kernel<2, 2, 6, EPI_ROWLN, 2, 2, 0, 16>;
            auto k1 = gemm_dma_kernel<4, 1, 7, EPI_ROWLN, 2, 2, 0, 16>;
            auto k2 = gemm_dma_kernel<2, 2, 4, EPI_ROWLN, 2, 2, 0, 16>;
            auto kp0 = gemm_dma_kernel<4, 2, 6, EPI_ROWLN, 3, 2, 0, 16>;   // production, round-2 profile
            auto kp2 = gemm_kernel<2, 2, 4, EPI_ROWLN, 1, 2, 1, 2>;
            auto kg = grouped_rowln_kernel<EPI_ROWLN>;
            const size_t l0 = 2 * DmaTile<2, 2, 6, 16>::STAGE_BYTES, l1 = 2 * DmaTile<4, 1, 7, 16>::STAGE_BYTES, l2 = 2 * DmaTile<2, 2, 4, 16>::STAGE_BYTES;
            const size_t lp0 = 3 * DmaTile<4, 2, 6, 16>::STAGE_BYTES, lp2 = GemmTile<2, 2, 4>::STAGE_FLOATS_SPLIT * 4;
            CK(hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0));
            CK(hipFuncSetAttribute((const void*)kp0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp0));
            CK(hipFuncSetAttribute((const void*)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0));
            const int t0 = (25920 + 63) / 64, t1 = (73440 + 127) / 128, t2 = (45360 + 63) / 64;
            auto pad8 = [](int t) { return (t + 7) / 8 * 8; };
            GroupedGemmParams g{};
            g.p[0] = gp[0], g.p[1] = gp[1], g.p[2] = gp[2], g.n = 3;
            g.first[0] = 0, g.first[1] = pad8(t0), g.first[2] = g.first[1] + pad8(t1), g.first[3] = g.first[2] + pad8(t2);
            uint8_t* ws32; CK(hipMalloc(&ws32, (size_t)256 * 512 * 6));   // hands image in the 32-deep format of the production kernel
            hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)256 * (pt[2].K / 8) + 255) / 256)), dim3(256), 0, 0, gp[2].W, ws32, 256, pt[2].K);
            GemmParams gh = gp[2]; gh.Wsplit = ws32;
            auto production = [&]() {
                hipLaunchKernelGGL(kp0, dim3((25920 + 127) / 128), dim3(512), lp0, 0, gp[0]);
                hipLaunchKernelGGL(k1, dim3(t1), dim3(256), l1, 0, gp[1]);
                hipLaunchKernelGGL(kp2, dim3(t2), dim3(256), lp2, 0, gh);
            };
            auto separate = [&]() {
                hipLaunchKernelGGL(k0, dim3(t0), dim3(256), l0, 0, gp[0]);
                hipLaunchKernelGGL(k1, dim3(t1), dim3(256), l1, 0, gp[1]);
                hipLaunchKernelGGL(k2, dim3(t2), dim3(256), l2, 0, gp[2]);
            };
            auto grouped = [&]() { hipLaunchKernelGGL(kg, dim3(g.first[3]), dim3(256), l0, 0, g); };
            auto timeit = [&](const char* tag, auto&& f) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 3; ++i) f();
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < 20; ++i) f();
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                double fl = 0; for (int i = 0; i < 3; ++i) fl += 2.0 * pt[i].M * pt[i].N * pt[i].K;
                printf("grouped %s %-40s : %8.1f us  %6.1f TF/s\n", fc2 ? "fc2 " : "proj", tag, ms * 1e3 / 20, fl / (ms * 1e-3 / 20) / 1e12);
                fflush(stdout);
            };
            // results: grouped == separate launches of the same variants, bit for bit
            std::vector<float> ref[3], got;
            separate(); CK(hipDeviceSynchronize());
            for (int i = 0; i < 3; ++i) { ref[i].resize(pt[i].M * pt[i].N); CK(hipMemcpy(ref[i].data(), pt[i].xn, ref[i].size() * 4, hipMemcpyDeviceToHost)); CK(hipMemset(pt[i].xn, 0, ref[i].size() * 4)); }
            grouped(); CK(hipDeviceSynchronize());
            for (int i = 0; i < 3; ++i) {
                got.resize(ref[i].size()); CK(hipMemcpy(got.data(), pt[i].xn, got.size() * 4, hipMemcpyDeviceToHost));
                printf("grouped %s part %d: %s\n", fc2 ? "fc2 " : "proj", i, memcmp(got.data(), ref[i].data(), got.size() * 4) ? "DIFFERENT" : "bit-identical");
            }
            timeit("three launches, round-2 production kernels", production);
            timeit("three launches, 4-wave st2 variants (2/CU)", separate);
            timeit("one grouped launch", grouped);
            {   // tall tiles, one 8-wave workgroup per CU
                auto tall = [&](auto kt, const char* tag, int hands_bm, size_t lds) {
                    CK(hipFuncSetAttribute((const void*)kt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    const int u0 = (25920 + 127) / 128, u1 = (73440 + 255) / 256, u2 = (45360 + hands_bm - 1) / hands_bm;
                    GroupedGemmParams gt{};
                    // most expensive tiles first: body 128 x 384, hands, face
                    gt.p[0] = gp[0], gt.p[1] = gp[2], gt.p[2] = gp[1], gt.n = 3;
                    gt.first[0] = 0, gt.first[1] = pad8(u0), gt.first[2] = gt.first[1] + pad8(u2), gt.first[3] = gt.first[2] + pad8(u1);
                    gt.first[4] = gt.first[3];
                    for (int i = 0; i < 3; ++i) CK(hipMemset(pt[i].xn, 0, ref[i].size() * 4));
                    hipLaunchKernelGGL(kt, dim3(gt.first[3]), dim3(512), lds, 0, gt); CK(hipDeviceSynchronize());
                    for (int i = 0; i < 3; ++i) {
                        got.resize(ref[i].size()); CK(hipMemcpy(got.data(), pt[i].xn, got.size() * 4, hipMemcpyDeviceToHost));
                        double md = 0; for (size_t q = 0; q < got.size(); ++q) { double d = fabs((double)got[q] - ref[i][q]); if (d > md) md = d; }
                        printf("tall %s part %d: max |d vs two-per-CU tiles| %.3e %s\n", tag, i, md, memcmp(got.data(), ref[i].data(), got.size() * 4) ? "" : "(bit-identical)");
                    }
                    timeit(tag, [&]() { hipLaunchKernelGGL(kt, dim3(gt.first[3]), dim3(512), lds, 0, gt); });
                };
                constexpr size_t st = DmaTile<4, 2, 6, 16>::STAGE_BYTES;
                static_assert(st >= DmaTile<8, 1, 8, 16>::STAGE_BYTES && st >= DmaTile<8, 1, 7, 16>::STAGE_BYTES, "stage");
                tall(grouped_tall_kernel<EPI_ROWLN, 3, 256>, "tall st3 (body 128, hands 256, face 256 rows)", 256, 3 * st);
                tall(grouped_tall_kernel<EPI_ROWLN, 2, 256>, "tall st2 (body 128, hands 256, face 256 rows)", 256, 2 * st);
                tall(grouped_tall_kernel<EPI_ROWLN, 3, 128>, "tall st3 (body 128, hands 128, face 256 rows)", 128, 3 * st);
            }
            for (int i = 0; i < 3; ++i) { CK(hipFree(pt[i].ws)); CK(hipFree(pt[i].xo)); CK(hipFree(pt[i].xn)); }
        }
    }
    return 0;
}
