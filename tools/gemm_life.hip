// gemm_life.hip - wave lifetimes of the production linear-layer kernels (diagnostic build with stamps).
#define PAFUSE_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
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
    GemmParams p{};
    p.A = A, p.W = W, p.bias = bias, p.out = out;
    p.M = 25920, p.N = 1152, p.K = 384;
    life<4, 1, 2, EPI_BIAS, 1>("body qkv <4,1,2> s1", p);
    life<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("body qkv SPLIT <4,1,4> s1", p);
    life<4, 1, 2, EPI_BIAS, 1, 4, 0, 2>("body qkv SPLIT <4,1,2> s1", p);
    life<4, 1, 3, EPI_BIAS, 1, 1, 1>("body qkv <4,1,3> s1 TR", p);
    p.N = 768, p.act = 1;
    life<4, 1, 2, EPI_BIAS, 1>("body fc1+gelu <4,1,2> s1", p);
    life<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>("body fc1+gelu SPLIT <4,1,4> s1", p);
    GemmParams q{};
    q.A = A, q.W = W, q.bias = bias, q.resid = x, q.out_x = x, q.out_n = xn;
    q.post_w = vec, q.post_b = vec, q.post_eps = 1e-6f, q.next_w = vec, q.next_b = vec, q.next_eps = 1e-6f;
    q.M = 25920, q.N = 384, q.K = 768;
    life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body fc2 rowln SPLIT dma16 <2,2,6> st2", q);
    life<1, 4, 3, EPI_ROWLN, 1, 1, 1>("body fc2 rowln <1,4,3> s1 TR", q);
    q.K = 384;
    life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body proj rowln SPLIT dma16 <2,2,6> st2", q);
    { GemmParams a = q; a.out_x = nullptr; a.out_n = nullptr; life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body proj rowln SPLIT dma16 <2,2,6> st2 ABL no stores", a); }
    { GemmParams a = q; a.post_w = nullptr; life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body proj rowln SPLIT dma16 <2,2,6> st2 ABL no post LN", a); }
    { GemmParams a = q; a.post_w = nullptr; a.next_w = nullptr; a.out_n = nullptr; life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body proj rowln SPLIT dma16 <2,2,6> st2 ABL no LN at all", a); }
    { GemmParams a = q; a.post_w = nullptr; a.next_w = nullptr; a.out_n = nullptr; a.out_x = nullptr; life_dma<2, 2, 6, EPI_ROWLN, 2, 2, 16>("body proj rowln SPLIT dma16 <2,2,6> st2 ABL resid only", a); }
    life<1, 4, 3, EPI_ROWLN, 1, 1, 1>("body proj rowln <1,4,3> s1 TR", q);
    q.M = 73440, q.N = 224, q.K = 448;
    life_dma<4, 1, 7, EPI_ROWLN, 2, 2, 16>("face fc2 rowln SPLIT dma16 <4,1,7> st2", q);
    life<1, 7, 1, EPI_ROWLN, 1, 1, 1>("face fc2 rowln <1,7,1> s1 TR", q);
    return 0;
}
