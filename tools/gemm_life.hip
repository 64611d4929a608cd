// gemm_life.hip - wave lifetimes of the production linear-layer kernels (diagnostic build with stamps).
#define PAFUSE_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
#include "../pafuse_amd/csrc/hgemm.hpp"   // the slab whole-row epilogue (epilogue_rows_h) the default path uses since round 5
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW = 1, int TR = 0, int MODE = 0>
void life(const char* tag, GemmParams p) {
    using T = GemmTile<WM, WN, NT>;
    size_t lds = (size_t)NSTAGE * (MODE == 2 ? T::STAGE_FLOATS_SPLIT : T::STAGE_FLOATS) * 4;
    p.bf16 = MODE;
    if (MODE == 2) {  // split-precision products: the pre-split image of W
        uint8_t* ws; CK(hipMalloc(&ws, (size_t)p.N * p.K * 6));
        hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W, ws, p.N, p.K);
        p.Wsplit = ws;
    }
    auto k = gemm_kernel<WM, WN, NT, EPI, NSTAGE, MINW, TR, MODE>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    size_t nw = tiles * (T::NTHR / 64);
    unsigned long long* st; CK(hipMalloc(&st, nw * 32));
    p.stamps = st;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw * 4);
    CK(hipMemcpy(h.data(), st, nw * 32, hipMemcpyDeviceToHost));
    double loop = 0, epi = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t w = 0; w < nw; ++w) {
        loop += h[w * 4 + 1] - h[w * 4]; epi += h[w * 4 + 2] - h[w * 4 + 1];
        t0 = std::min(t0, h[w * 4]); t1 = std::max(t1, h[w * 4 + 2]);
    }
    // concurrency profile: how many waves are alive over time (20 bins)
    const int NB = 20; double alive[NB] = {0};
    for (size_t w = 0; w < nw; ++w) {
        double a = (double)(h[w * 4] - t0) / (t1 - t0) * NB, b = (double)(h[w * 4 + 2] - t0) / (t1 - t0) * NB;
        for (int i = 0; i < NB; ++i) { double lo = std::max(a, (double)i), hi = std::min(b, (double)i + 1); if (hi > lo) alive[i] += hi - lo; }
    }
    double mfma = MODE == 2 ? (double)(p.K / 16) * 6 * NT * 32 : (double)(p.K / 32) * 16 * NT * 64;
    printf("%s: tiles=%ld waves=%zu span=%.0f cyc | per wave: prologue+loop %.0f  epilogue %.0f  (own MFMA issue %.0f) | MFMA-pipe busy over span: %.1f%%\n  waves alive per CU over time:", tag, (long)tiles, nw,
           (double)(t1 - t0), loop / nw, epi / nw, mfma, mfma * nw / 1024.0 / (double)(t1 - t0) * 100);
    for (int i = 0; i < NB; ++i) printf(" %.1f", alive[i] / 256.0);
    printf("\n");
    CK(hipFree(st));
}

// the LDS-DMA tiles: stamps 0 = tile start, 3 = first chunk landed and visible, 1 = K loop done, 2 = epilogue done
template <int WM, int WN, int NT, int EPI, int NSTAGE, int MINW, int BKC>
void life_dma(const char* tag, GemmParams p) {
    using T = DmaTile<WM, WN, NT, BKC>;
    const size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    p.bf16 = 2;
    uint8_t* ws; CK(hipMalloc(&ws, (size_t)p.N * p.K * 6));
    hipLaunchKernelGGL(split_weights_kernel<BKC>, dim3((unsigned)(((int64_t)p.N * (p.K / 8) + 255) / 256)), dim3(256), 0, 0, p.W, ws, p.N, p.K);
    p.Wsplit = ws;
    auto k = gemm_dma_kernel<WM, WN, NT, EPI, NSTAGE, MINW, 0, BKC>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    const size_t nw = tiles * (T::NTHR / 64);
    unsigned long long* st; CK(hipMalloc(&st, nw * 32)); CK(hipMemset(st, 0, nw * 32));
    p.stamps = st;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw * 4);
    CK(hipMemcpy(h.data(), st, nw * 32, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; size_t n = 0;
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * 4] || !h[w * 4 + 2]) continue;
        pro += h[w * 4 + 3] - h[w * 4]; loop += h[w * 4 + 1] - h[w * 4 + 3]; epi += h[w * 4 + 2] - h[w * 4 + 1]; ++n;
        t0 = std::min(t0, h[w * 4]); t1 = std::max(t1, h[w * 4 + 2]);
    }
    const double mfma = (double)(p.K / 16) * 6 * NT * 32;
    printf("%s: tiles=%ld waves=%zu span=%.0f cyc | per wave: prologue (first chunk) %.0f  K loop %.0f  epilogue %.0f  (own MFMA issue %.0f)\n", tag,
           (long)tiles, n, (double)(t1 - t0), pro / n, loop / n, epi / n, mfma);
    CK(hipFree(st)); CK(hipFree(ws));
}

int main() {
    const int64_t Mmax = 73440;
    float *A, *W, *bias, *out, *x, *xn, *vec;
    CK(hipMalloc(&A, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&bias, 1152 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4)); CK(hipMalloc(&x, Mmax * 384 * 4)); CK(hipMalloc(&xn, Mmax * 384 * 4));
    CK(hipMalloc(&vec, 1152 * 4));
    std::vector<float> h(Mmax * 768);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(A, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data() + 777, 1152 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, h.data() + 5000, 1152 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(x, h.data(), Mmax * 384 * 4, hipMemcpyHostToDevice));
    float* stats; CK(hipMalloc(&stats, Mmax * 2 * 4));
    { std::vector<float> st(Mmax * 2); for (int64_t i = 0; i < Mmax; ++i) st[2 * i] = 0.f, st[2 * i + 1] = 1.f; CK(hipMemcpy(stats, st.data(), Mmax * 2 * 4, hipMemcpyHostToDevice)); }
    // ---- round 5: the default path as it runs (LayerNorm folded: the plain GEMMs read (mean, rstd), the whole-row GEMMs write them and
    // store the centred row through the slab epilogue), the three parts at P = 20 flip-TTA
    struct Part { const char* name; int64_t M; int C; } parts[3] = {{"body", 25920, 384}, {"face", 73440, 224}, {"hands", 45360, 256}};
    for (const Part& pt : parts) {
        char tag[160];
        GemmParams p{};
        p.A = A, p.W = W, p.bias = bias, p.out = out, p.ln_in = stats, p.act = 1;
        p.M = pt.M, p.N = 2 * pt.C, p.K = pt.C;
        snprintf(tag, sizeof tag, "%s fc1 + GELU, folded LN, 128 x 128 SPLIT <4,1,4> (face: 128 x 64 <4,1,2>)", pt.name);
        if (p.N % 128 == 0) life<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>(tag, p);
        else life<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>(tag, p);
        for (int which = 0; which < 2; ++which) {   // proj (K = C, next LayerNorm folded), fc2 (K = 2 C, post norm + next folded)
            GemmParams q{};
            q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = x, q.ln_stats = stats;
            q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
            if (which) q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f;
            q.M = pt.M, q.N = pt.C, q.K = which ? 2 * pt.C : pt.C;
            snprintf(tag, sizeof tag, "%s %s whole-row, slab epilogue, centred store", pt.name, which ? "fc2" : "proj");
            if (pt.C == 384) life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>(tag, q);
            else if (pt.C == 256) life_dma<2, 2, 4, EPI_ROWLN, 2, 2, 16>(tag, q);
            else life_dma<4, 1, 7, EPI_ROWLN, 2, 2, 16>(tag, q);
        }
    }
    return 0;
}
