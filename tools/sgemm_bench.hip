// sgemm_bench.hip - round 6: the strip GEMM (pafuse_amd/csrc/sgemm.hpp; its first form: tools/sgemm_v1.hpp) against gemm16_kernel at the hot path's
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
#include "sgemm_v1.hpp"                    // (includes pafuse_amd/csrc/sgemm.hpp: the shipped sgemm2_kernel)
#include "../pafuse_amd/csrc/hgemm.hpp"   // epilogue_rows_h: the whole-row epilogue of the production gemm_dma_kernel
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

// ---- whole-row layers (proj, fc2 + residual + LayerNorms): the production LDS-DMA kernel per part (slab epilogue) and in the shared grid
// (direct epilogue): equal bits.  (The strip kernel's whole-row form that was measured against them in round 6 is
// profiles/r06_whole_row_strip_experiment.patch; it ties them and is not in the tree.)
#ifndef PAFUSE_DMA_APIPE   // only defined with profiles/r06_dma_apipe_experiment.patch applied
#define PAFUSE_DMA_APIPE 0
#endif
static float *g_x0 = nullptr, *g_xa = nullptr, *g_xb = nullptr, *g_sa = nullptr, *g_sb = nullptr;
static void compare_rows(const char* tag, int64_t M, int C) {
    std::vector<float> a((size_t)M * C), b((size_t)M * C), sa((size_t)M * 2), sb((size_t)M * 2);
    CK(hipMemcpy(a.data(), g_xa, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), g_xb, b.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sa.data(), g_sa, sa.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), g_sb, sb.size() * 4, hipMemcpyDeviceToHost));
    double dx = 0, mx = 0, ds = 0;
    size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        if (!(std::fabs(a[i] - b[i]) <= 1e-3)) ++bad;
        dx = std::max(dx, (double)std::fabs(a[i] - b[i])), mx = std::max(mx, (double)std::fabs(a[i]));
    }
    for (size_t i = 0; i < sa.size(); ++i) ds = std::max(ds, (double)std::fabs(sa[i] - sb[i]) / std::max(1.0, (double)std::fabs(sa[i])));
    printf("      %s: centred rows max |d| %.3e (max |x| %.3f, elements off by > 1e-3: %zu), statistics max rel d %.3e\n", tag, dx, mx, bad, ds);
}
// ABL (gemm_dma_tile's diagnostic builds, results wrong by design): 2 = no DMA (the compute side alone), 3 = 2 without the split arithmetic
template <int WM, int WN, int NT, int ABL = 0, int MINW = 2>
void run_rowln_ref(const char* shape, GemmParams p) {   // production: gemm_dma_kernel<.., EPI_ROWLN, 2 stages, 2 per CU, 16-deep chunks>
    using T = DmaTile<WM, WN, NT, 16>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s gemm_dma<%d,%d,%d> %dx%d (production%s)", shape, WM, WN, NT, T::BM, T::BN,
             ABL == 2 ? ", ABL 2: no operand stream" : ABL == 3 ? ", ABL 3: no operand stream, no split" : "");
    constexpr size_t lds = 2 * T::STAGE_BYTES;
    auto k = gemm_dma_kernel<WM, WN, NT, EPI_ROWLN, 2, MINW, ABL, 16>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long tiles = (p.M + T::BM - 1) / T::BM;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T::NTHR, lds));
    float* const xdst = (WM == 3) ? g_xb : g_xa;   // (the 96-row experiment writes the second buffer and is compared with the production result)
    p.resid = p.out_x = xdst, p.ln_stats = (WM == 3) ? g_sb : g_sa;
    CK(hipMemcpy(xdst, g_x0, (size_t)p.M * p.N * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, 0, p);   // the compared result: ONE launch on fresh rows
    CK(hipDeviceSynchronize());
    if (ABL == 0) {   // a digest of the compared result: equal across builds (-DPAFUSE_DMA_APIPE=0 / 1) = equal bits
        std::vector<uint32_t> hx((size_t)p.M * p.N), hs((size_t)p.M * 2);
        CK(hipMemcpy(hx.data(), xdst, hx.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), (WM == 3) ? g_sb : g_sa, hs.size() * 4, hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull;
        for (uint32_t v : hx) h = (h ^ v) * 1099511628211ull;
        for (uint32_t v : hs) h = (h ^ v) * 1099511628211ull;
        printf("      digest of rows + statistics (APIPE %d): %016llx\n", (int)PAFUSE_DMA_APIPE, h);
    }
    p.resid = p.out_x = g_ref, p.ln_stats = g_sb + 0;   // timing on scratch rows (in place: the values drift, the work does not)
    float* scratch_stats; CK(hipMalloc(&scratch_stats, (size_t)p.M * 8)); p.ln_stats = scratch_stats;
    CK(hipMemcpy(g_ref, g_x0, (size_t)p.M * p.N * 4, hipMemcpyDeviceToDevice));
    const double us = time_us([&] { hipLaunchKernelGGL(k, dim3((unsigned)tiles), dim3(T::NTHR), lds, 0, p); });
    CK(hipFree(scratch_stats));
    const double tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    printf("%-58s tiles %5ld grid %5ld (%d/CU, %3zu KB) %7.2f us %6.1f TF (%.3f of 417)  (reference)\n", tag, tiles, tiles, occ, lds / 1024, us, tf, tf / 416.7);
    fflush(stdout);
}
// the shared-grid form (grouped_rowln_kernel, direct epilogue) of ONE part against the per-part launch (slab epilogue): equal bits?
static void run_rowln_grouped_bits(const char* shape, GemmParams p, int bm) {
    GroupedGemmParams g{};
    g.n = 1;
    p.resid = p.out_x = g_xb, p.ln_stats = g_sb;
    g.p[0] = p;
    const long tiles = (p.M + bm - 1) / bm;
    g.first[0] = 0;
    for (int k = 1; k <= GROUP_MAX; ++k) g.first[k] = (int)((tiles + 7) / 8 * 8);
    constexpr size_t lds = 2 * DmaTile<2, 2, 6, 16>::STAGE_BYTES;
    auto k = grouped_rowln_kernel<EPI_ROWLN>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipMemcpy(g_xb, g_x0, (size_t)p.M * p.N * 4, hipMemcpyDeviceToDevice));
    hipLaunchKernelGGL(k, dim3((unsigned)g.first[1]), dim3(256), lds, 0, g);
    CK(hipDeviceSynchronize());
    printf("      %s shared-grid kernel vs per-part launch: differing words rows %llu, statistics %llu\n", shape,
           differing(g_xa, g_xb, p.M * p.N), differing(g_sa, g_sb, p.M * 2));
    {
        const size_t n = (size_t)512 * p.N;
        std::vector<float> a(n), b(n), x0(n), st(1024);
        CK(hipMemcpy(a.data(), g_xa, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), g_xb, n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(st.data(), g_sa, 4096, hipMemcpyDeviceToHost));
        int shown = 0;
        for (size_t i = 0; i < n && shown < 12; ++i)
            if (a[i] != b[i]) {
                printf("        row %zu col %zu: per-part %.9g shared %.9g (d %.3g) mean %.9g\n", i / p.N, i % p.N, a[i], b[i], a[i] - b[i], st[2 * (i / p.N)]);
                ++shown;
            }
    }
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
    // ---- whole-row layers of the face and the hands (SB_ROWLN=0 skips them)
    if (!getenv("SB_ROWLN") || atoi(getenv("SB_ROWLN"))) {
        uint8_t* Wi1;
        float* postv;
        CK(hipMalloc(&Wi1, 1152 * 768 * 6)); CK(hipMalloc(&g_x0, Mmax * 256 * 4)); CK(hipMalloc(&g_xa, Mmax * 256 * 4)); CK(hipMalloc(&g_xb, Mmax * 256 * 4));
        CK(hipMalloc(&g_sa, Mmax * 8)); CK(hipMalloc(&g_sb, Mmax * 8)); CK(hipMalloc(&postv, 4096 * 4));
        {
            std::vector<float> pv(4096);
            for (int i = 0; i < 4096; ++i) pv[i] = 1.0f + 0.1f * (float)((i * 37) % 19 - 9) / 9.0f;
            CK(hipMemcpy(postv, pv.data(), 4096 * 4, hipMemcpyHostToDevice));
        }
        CK(hipMemcpy(g_x0, X + 12345, Mmax * 256 * 4, hipMemcpyDeviceToDevice));
        for (int pi = 1; pi < 3; ++pi) {
            const Part& pt = parts[pi];
            const int C = pt.C;
            for (int layer = 0; layer < 2; ++layer) {   // 0 = proj (K = C), 1 = fc2 (K = 2C, with the post LayerNorm)
                const int N = C, K = layer == 0 ? C : 2 * C;
                hipLaunchKernelGGL((split_weights_kernel<16, 0>), dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, W, Wi1, N, K);
                hipLaunchKernelGGL((split_weights_kernel<32, 1>), dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, W, Wi, N, K);
                GemmParams p{};
                p.A = X, p.bias = vec, p.M = pt.M, p.N = N, p.K = K, p.bf16 = 2;
                p.next_w = vec, p.next_b = vec, p.next_eps = 1e-6f;
                if (layer == 1) p.post_w = postv, p.post_b = vec + 512, p.post_eps = 1e-6f;
                char shape[64];
                snprintf(shape, sizeof shape, "%s %s", pt.name, layer == 0 ? "proj" : "fc2");
                p.Wsplit = Wi1;
                if (C == 256) run_rowln_ref<2, 2, 4>(shape, p);
                else run_rowln_ref<4, 1, 7>(shape, p);
                run_rowln_grouped_bits(shape, p, C == 256 ? 64 : 128);
                if (C == 256) {   // 96-row tiles on six waves, two workgroups (twelve waves) per CU: 473 tiles = 0.92 rounds, 26 B per clock of operands
                    run_rowln_ref<3, 2, 4, 0, 3>(shape, p);
                    printf("      96-row tile vs production: differing words rows %llu, statistics %llu\n", differing(g_xa, g_xb, p.M * p.N), differing(g_sa, g_sb, p.M * 2));
                }
                if (getenv("SB_ROWLN_ABL")) {   // what the operand stream and the in-register split cost the whole-row tiles
                    if (C == 256) run_rowln_ref<2, 2, 4, 2>(shape, p), run_rowln_ref<2, 2, 4, 3>(shape, p);
                    else run_rowln_ref<4, 1, 7, 2>(shape, p), run_rowln_ref<4, 1, 7, 3>(shape, p);
                }
            }
        }
    }
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
                if (N % 96 == 0) run2<6, 2, 4, 2>(shape, p);   // more, narrower tiles: a better last round on 512 workgroups?
                if (N % 64 == 0) run2<4, 2, 4, 2>(shape, p);
                run2<8, 2, 8, 2>(shape, p);
                if (N % 256 == 0) run2<16, 1, 8, 2>(shape, p);
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
                run2<14, 1, 8, 2>(shape, p);
                run<7, 2, 8, 2, 2, 1>(shape, p);
                run<14, 2, 4, 2, 1, 1>(shape, p);
                run<14, 1, 8, 2, 2, 1>(shape, p);
                run<14, 2, 8, 2, 2, 1>(shape, p);
            }
        }
    }
    return 0;
}
