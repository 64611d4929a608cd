// sgemm_bench.hip - round 6: the strip GEMM (pafuse_amd/csrc/sgemm.hpp) against the production kernels at the hot path's
// layer shapes (P = 20 flip-TTA: M = 25 920 / 73 440 / 45 360), every variant in one process: us per launch (HIP events, 20
// launches back to back), TFLOP/s of fp32-equivalent work, and a BITWISE compare of every variant's output with the production
// kernel's (same products in the same order: equal bits).
//   hipcc <library flags> tools/sgemm_bench.hip -o tools/bin/sgemm_bench ;  SB_FILTER=<substring> SB_REPS=<n> ./sgemm_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/sgemm.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const char* g_filter = nullptr;
static int g_reps = 20;
static float* g_ref = nullptr;      // the production kernel's output of the current shape
static unsigned long long* g_cnt = nullptr;

__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, int64_t n, unsigned long long* cnt) {
    unsigned long long d = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) d += a[i] != b[i];
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if ((threadIdx.x & 63) == 0 && d) atomicAdd(cnt, d);
}
static unsigned long long differing(const float* a, const float* b, int64_t n) {
    CK(hipMemset(g_cnt, 0, 8));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, n, g_cnt);
    unsigned long long h;
    CK(hipMemcpy(&h, g_cnt, 8, hipMemcpyDeviceToHost));
    return h;
}

template <class F>
static double time_us(F&& launch) {
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < g_reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms * 1e3 / g_reps;
}

static void report(const char* tag, const GemmParams& p, double us, long tiles, int occ, size_t lds, long grid, const float* out, bool is_ref) {
    const double tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    const int64_t n = p.M * p.N;
    char cmp[64] = "(reference)";
    if (!is_ref) snprintf(cmp, sizeof cmp, "differing words %llu", differing(out, g_ref, n));
    printf("%-58s tiles %5ld grid %5ld (%d/CU, %3zu KB) %7.2f us %6.1f TF (%.3f of 417)  %s\n", tag, tiles, grid, occ, lds / 1024, us, tf,
           tf / 416.7, cmp);
    fflush(stdout);
}

template <int NB, int MINW>
void run_ref(const char* shape, GemmParams p) {   // production gemm16_kernel into g_ref
    using T = Tile16<NB>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s gemm16<%d,%d> 128x%d (production)", shape, NB, MINW, T::BN);
    p.out = g_ref;
    const long tiles = ((p.M + T::BM - 1) / T::BM) * (p.N / T::BN);
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm16_kernel<NB, MINW>, 256, T::STAGE_BYTES));
    const double us = time_us([&] { hipLaunchKernelGGL((gemm16_kernel<NB, MINW>), dim3((unsigned)tiles), dim3(256), T::STAGE_BYTES, 0, p); });
    if (g_filter && !strstr(tag, g_filter) && !strstr("production", g_filter)) {}
    report(tag, p, us, tiles, occ, T::STAGE_BYTES, tiles, g_ref, true);
}

// PERSIST: 0 = one workgroup per tile, 1 = occupancy x 256 workgroups over the tile stream
template <int NB, int RG, int NW, int NSTAGE, int MINW, int PERSIST>
void run(const char* shape, GemmParams p) {
    using T = StripTile<NB, RG, NW, NSTAGE>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s strip<NB%d,RG%d,NW%d,ST%d> %dx%d minw%d %s", shape, NB, RG, NW, NSTAGE, T::BM, T::BN, MINW,
             PERSIST ? "persistent" : "per-tile");
    if (g_filter && !strstr(tag, g_filter)) return;
    if (p.N % T::BN) { printf("%s: N %% BN != 0, skipped\n", tag); return; }
    auto k = p.act ? sgemm_kernel<NB, RG, NW, NSTAGE, SEPI_BIAS, MINW, 3> : sgemm_kernel<NB, RG, NW, NSTAGE, SEPI_BIAS, MINW, 1>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T::NTHR, T::LDS_BYTES));
    if (occ < 1) { printf("%s: does not fit a CU\n", tag); return; }
    const long tiles = ((p.M + T::BM - 1) / T::BM) * (p.N / T::BN);
    long grid = tiles;
    if (PERSIST) grid = std::min<long>(tiles, 256L * occ);
    if (grid < tiles) grid = grid / 8 * 8;
    CK(hipMemset(p.out, 0xff, (size_t)p.M * p.N * 4));
#ifdef SGEMM_STAMPS
    unsigned long long* st;
    const size_t nw = (size_t)grid * NW;
    CK(hipMalloc(&st, nw * 64)); CK(hipMemset(st, 0, nw * 64));
    p.stamps = st;
#endif
    const double us = time_us([&] { hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(T::NTHR), T::LDS_BYTES, 0, p); });
    report(tag, p, us, tiles, occ, T::LDS_BYTES, grid, p.out, false);
#ifdef SGEMM_STAMPS
    {
        std::vector<unsigned long long> h(nw * 8);
        CK(hipMemcpy(h.data(), st, nw * 64, hipMemcpyDeviceToHost));
        double a[6] = {0, 0, 0, 0, 0, 0};
        size_t n = 0;
        for (size_t w = 0; w < nw; ++w) {
            if (!h[w * 8 + 4]) continue;
            for (int i = 0; i < 6; ++i) a[i] += (double)h[w * 8 + i];
            ++n;
        }
        for (int i = 0; i < 6; ++i) a[i] /= (double)std::max<size_t>(n, 1);
        const double chunks = (double)tiles * (p.K / 32) / (double)grid;       // per workgroup (average)
        const double mfma = chunks * RG * NB * 6 * 16;                          // this wave's own MFMA cycles
        printf("      per wave (cycles): top wait %7.0f  reads+split %7.0f  MFMA groups %7.0f (own MFMA issue %7.0f)  epilogues %7.0f  | lifetime %7.0f = %.1f us at %.0f MHz; per chunk: wait %.0f split %.0f groups %.0f (MFMA %d)\n",
               a[0], a[1], a[2], mfma, a[3], a[4], a[5] / 100.0, a[4] / (a[5] / 100.0), a[0] / chunks, a[1] / chunks, a[2] / chunks, RG * NB * 96);
        CK(hipFree(st));
    }
#endif
}

#ifdef SGEMM_STAMPS
static void stamp_report(unsigned long long* st, size_t nw, double chunks, int own_mfma_per_chunk) {
    std::vector<unsigned long long> h(nw * 8);
    CK(hipMemcpy(h.data(), st, nw * 64, hipMemcpyDeviceToHost));
    double a[6] = {0, 0, 0, 0, 0, 0};
    size_t n = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * 8 + 4]) continue;
        for (int i = 0; i < 6; ++i) a[i] += (double)h[w * 8 + i];
        ++n;
    }
    for (int i = 0; i < 6; ++i) a[i] /= (double)std::max<size_t>(n, 1);
    printf("      per wave (cycles): top wait %7.0f  reads+split %7.0f  MFMA groups %7.0f (own MFMA issue %7.0f)  epilogues %7.0f  | lifetime %7.0f = %.1f us at %.0f MHz; per chunk: wait %.0f split %.0f groups %.0f (MFMA %d)\n",
           a[0], a[1], a[2], chunks * own_mfma_per_chunk, a[3], a[4], a[5] / 100.0, a[4] / (a[5] / 100.0), a[0] / chunks, a[1] / chunks, a[2] / chunks, own_mfma_per_chunk);
}
#endif

// the software-pipelined form (sgemm2_kernel): always the persistent tile stream, two-stage ring
template <int NB, int RG, int NW, int MINW>
void run2(const char* shape, GemmParams p) {
    using T = StripTile<NB, RG, NW, 2>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s strip2<NB%d,RG%d,NW%d> %dx%d minw%d pipelined", shape, NB, RG, NW, T::BM, T::BN, MINW);
    if (g_filter && !strstr(tag, g_filter)) return;
    if (p.N % T::BN) { printf("%s: N %% BN != 0, skipped\n", tag); return; }
    auto k = p.act ? sgemm2_kernel<NB, RG, NW, SEPI_BIAS, MINW, 3> : sgemm2_kernel<NB, RG, NW, SEPI_BIAS, MINW, 1>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T::NTHR, T::LDS_BYTES));
    if (occ < 1) { printf("%s: does not fit a CU\n", tag); return; }
    const long tiles = ((p.M + T::BM - 1) / T::BM) * (p.N / T::BN);
    long grid = std::min<long>(tiles, 256L * occ);
    if (grid < tiles) grid = grid / 8 * 8;
    CK(hipMemset(p.out, 0xff, (size_t)p.M * p.N * 4));
#ifdef SGEMM_STAMPS
    unsigned long long* st;
    const size_t nw = (size_t)grid * NW;
    CK(hipMalloc(&st, nw * 64)); CK(hipMemset(st, 0, nw * 64));
    p.stamps = st;
#endif
    const double us = time_us([&] { hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(T::NTHR), T::LDS_BYTES, 0, p); });
    report(tag, p, us, tiles, occ, T::LDS_BYTES, grid, p.out, false);
#ifdef SGEMM_STAMPS
    stamp_report(st, nw, (double)tiles * (p.K / 32) / (double)grid, RG * NB * 96);
    CK(hipFree(st));
#endif
}

int main() {
    g_filter = getenv("SB_FILTER");
    if (getenv("SB_REPS")) g_reps = atoi(getenv("SB_REPS"));
    const int64_t Mmax = 73440;
    float *X, *W, *vec, *out, *stats;
    uint8_t* Wi;
    CK(hipMalloc(&X, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&vec, 4096 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4)); CK(hipMalloc(&g_ref, Mmax * 1152 * 4)); CK(hipMalloc(&stats, Mmax * 8));
    CK(hipMalloc(&Wi, 1152 * 768 * 6)); CK(hipMalloc(&g_cnt, 8));
    std::vector<float> h(Mmax * 768);
    srand(7);
    for (auto& v : h) v = (float)(rand() % 200001 - 100000) * 1e-5f;
    CK(hipMemcpy(X, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < 1152 * 768; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, h.data() + 999, 4096 * 4, hipMemcpyHostToDevice));
    std::vector<float> sth(Mmax * 2);
    for (int64_t i = 0; i < Mmax; ++i) sth[2 * i] = 0.01f, sth[2 * i + 1] = 1.3f + 1e-3f * (i % 97);
    CK(hipMemcpy(stats, sth.data(), Mmax * 8, hipMemcpyHostToDevice));

    struct Part { const char* name; int64_t M; int C; };
    const Part parts[3] = {{"body", 25920, 384}, {"face", 73440, 224}, {"hands", 45360, 256}};
    for (const Part& pt : parts) {
        const int C = pt.C;
        for (int layer = 0; layer < 2; ++layer) {   // 0 = qkv (N = 3C), 1 = fc1 + GELU (N = 2C); both with the LayerNorm folded
            const int N = layer == 0 ? 3 * C : 2 * C, K = C;
            hipLaunchKernelGGL((split_weights_kernel<32, 1>), dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, W, Wi, N, K);
            GemmParams p{};
            p.A = X, p.Wsplit = Wi, p.bias = vec, p.ln_in = stats, p.out = out, p.M = pt.M, p.N = N, p.K = K, p.bf16 = 2, p.wlayout = 2;
            p.act = layer;
            char shape[64];
            snprintf(shape, sizeof shape, "%s %s", pt.name, layer == 0 ? "qkv" : "fc1");
            if (N % 128 == 0) {
                run_ref<8, 3>(shape, p);
                run<8, 2, 4, 2, 2, 0>(shape, p);
                run<8, 2, 4, 2, 2, 1>(shape, p);
                run2<8, 2, 4, 2>(shape, p);
                run2<8, 2, 8, 2>(shape, p);
                run<8, 2, 8, 2, 2, 1>(shape, p);
                run<8, 1, 8, 3, 2, 1>(shape, p);
                if (N % 256 == 0) {
                    run<16, 1, 8, 2, 2, 1>(shape, p);
                    run<16, 2, 4, 2, 1, 1>(shape, p);
                }
                if (N % 192 == 0) run<12, 2, 8, 2, 2, 1>(shape, p);
            } else {
                if (N % 96 == 0) run_ref<6, 3>(shape, p);
                else run_ref<7, 3>(shape, p);
                if (N % 96 == 0) {
                    run<6, 2, 4, 2, 2, 0>(shape, p);
                    run<6, 2, 4, 2, 2, 1>(shape, p);
                    run2<6, 2, 4, 2>(shape, p);
                    run<6, 2, 8, 2, 2, 1>(shape, p);
                }
                run<7, 2, 4, 2, 2, 0>(shape, p);
                run<7, 2, 4, 2, 2, 1>(shape, p);
                run2<7, 2, 4, 2>(shape, p);
                run<7, 2, 8, 2, 2, 1>(shape, p);
                run<14, 2, 4, 2, 1, 1>(shape, p);
                run<14, 1, 8, 2, 2, 1>(shape, p);
                run<14, 2, 8, 2, 2, 1>(shape, p);
            }
        }
    }
    return 0;
}
