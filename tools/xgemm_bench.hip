// xgemm_bench.hip - the bf16x3 X-pipeline tiles (pafuse_amd/csrc/xgemm.hpp) at the hot path's layer shapes, several tile
// configurations per shape in one process: us per launch (HIP events), TFLOP/s of fp32-equivalent work, and the per-wave
// lifetime split by in-kernel stamps (prologue = until the first chunk is visible, K loop, epilogue), in shader cycles.
//   hipcc <library flags> [-DPAFUSE_X_ABL=n] tools/xgemm_bench.hip -o tools/bin/xgemm_bench ;  XB_FILTER=<substring> ./xgemm_bench
// PAFUSE_X_ABL (diagnostic, results wrong by design): 1 = no operand stream (compute on whatever the LDS holds), 2 = no MFMAs
// (the stream, the fragment reads and the barriers alone), 3 = neither.
#define PAFUSE_STAMPS 1
#define PAFUSE_STAMP_SLOTS 8
#define SL PAFUSE_STAMP_SLOTS
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/xgemm.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const char* g_filter = nullptr;
static int g_reps = 20;

struct Life { double pro = 0, loop = 0, epi = 0; size_t n = 0; };
static Life lifetimes(unsigned long long* st, size_t nw) {
    std::vector<unsigned long long> h(nw * SL);
    CK(hipMemcpy(h.data(), st, nw * SL * 8, hipMemcpyDeviceToHost));
    Life l;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * SL] || !h[w * SL + 2]) continue;
        l.pro += h[w * SL + 3] - h[w * SL], l.loop += h[w * SL + 1] - h[w * SL + 3], l.epi += h[w * SL + 2] - h[w * SL + 1], ++l.n;
    }
    if (l.n) l.pro /= l.n, l.loop /= l.n, l.epi /= l.n;
    return l;
}

template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, int MINW, bool HRES = (EPI == EPI_ROWLN)>
void run(const char* shape, GemmParams p) {
    using T = XTile<WM, WN, NT, BKC>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s <%d,%d,%d> %dx%d st%d bkc%d minw%d", shape, WM, WN, NT, T::BM, T::BN, NSTAGE, BKC, MINW);
    if (g_filter && !strstr(tag, g_filter)) return;
    if (p.N % T::BN) { printf("%s: N %% BN != 0, skipped\n", tag); return; }
    const size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    auto k = xgemm_kernel<WM, WN, NT, EPI, NSTAGE, BKC, MINW, HRES>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    const size_t nw = tiles * (T::NTHR / 64);
    unsigned long long* st; CK(hipMalloc(&st, nw * SL * 8)); CK(hipMemset(st, 0, nw * SL * 8));
    p.stamps = st;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T::NTHR, lds));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const Life l = lifetimes(st, nw);
    const double us = ms * 1e3 / g_reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    const double mfma = (double)(p.K / 16) * 6 * NT * 32;   // this wave's own MFMA issue cycles
    printf("%-56s tiles %5ld (%.2f rounds at %d/CU, %3zu KB) %7.2f us %6.1f TF (%.3f of 417) | per wave: prologue %5.0f  K loop %6.0f (MFMA %5.0f)  epilogue %6.0f\n",
           tag, (long)tiles, (double)tiles / (256.0 * occ), occ, lds / 1024, us, tf, tf / 416.7, l.pro, l.loop, mfma, l.epi);
    fflush(stdout);
    CK(hipFree(st));
}

template <int LP, int DP, int HPW>
void run_fqa(const char* shape, FqaParams f) {
    using FT = XfqaTile<LP, DP, HPW>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s qkv+attn fused <%d,%d,%d>", shape, LP, DP, HPW);
    if (g_filter && !strstr(tag, g_filter)) return;
    auto k = xfqa_kernel<LP, DP, HPW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FT::LDS_BYTES));
    const int64_t ntiles = (f.nseq + f.nseq_tile - 1) / f.nseq_tile;
    const int64_t blocks = (ntiles + 7) / 8 * 8 * (f.heads / HPW);
    const size_t nw = blocks * FT::NWV;
    unsigned long long* st; CK(hipMalloc(&st, nw * SL * 8)); CK(hipMemset(st, 0, nw * SL * 8));
    f.g.stamps = st;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, FT::NTHR, FT::LDS_BYTES));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(FT::NTHR), FT::LDS_BYTES, 0, f);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(FT::NTHR), FT::LDS_BYTES, 0, f);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const Life l = lifetimes(st, nw);
    const double us = ms * 1e3 / g_reps;
    const double mfma = (double)(f.g.K / 16) * 6 * FT::NT * 32;
    printf("%-56s wgs %6ld (%.2f rounds at %d/CU, %3d KB) %7.2f us | per wave: prologue %5.0f  projection loop %6.0f (MFMA %5.0f)  attention phases %6.0f\n",
           tag, (long)blocks, (double)blocks / (256.0 * occ), occ, FT::LDS_BYTES / 1024, us, l.pro, l.loop, mfma, l.epi);
    fflush(stdout);
    CK(hipFree(st));
}

int main() {
    g_filter = getenv("XB_FILTER");
    if (getenv("XB_REPS")) g_reps = atoi(getenv("XB_REPS"));
    const int64_t Mmax = 73440;
    float *X, *W, *vec, *out, *stats;
    uint8_t *Ax, *Wx, *outx, *xx;
    CK(hipMalloc(&X, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&vec, 4096 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4)); CK(hipMalloc(&stats, Mmax * 8));
    CK(hipMalloc(&Ax, Mmax * 768 * 6)); CK(hipMalloc(&Wx, 1152 * 768 * 6)); CK(hipMalloc(&outx, Mmax * 768 * 6)); CK(hipMalloc(&xx, Mmax * 384 * 6));
    std::vector<float> h(Mmax * 768);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(X, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < 1152 * 768; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(vec, h.data() + 999, 4096 * 4, hipMemcpyHostToDevice));
    std::vector<float> sth(Mmax * 2);
    for (int64_t i = 0; i < Mmax; ++i) sth[2 * i] = 0.01f, sth[2 * i + 1] = 1.3f;
    CK(hipMemcpy(stats, sth.data(), Mmax * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(xsplit_rows_kernel, dim3((unsigned)((Mmax * 48 + 255) / 256)), dim3(256), 0, 0, X, xx, Mmax, 384);

    struct Part { const char* name; int64_t M; int C; };
    const Part parts[3] = {{"body", 25920, 384}, {"face", 73440, 224}, {"hands", 45360, 256}};
    for (const Part& pt : parts) {
        const int C = pt.C;
        auto images = [&](int N, int K) {   // A [M,K] and W [N,K] as X images
            hipLaunchKernelGGL(xsplit_rows_kernel, dim3((unsigned)((pt.M * (K / 8) + 255) / 256)), dim3(256), 0, 0, X, Ax, pt.M, K);
            hipLaunchKernelGGL(xsplit_weights_kernel, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, W, Wx, N, K);
        };
        char shape[64];
        {   // ---- fc1 (+ GELU, image out)
            const int N = 2 * C;
            images(N, C);
            GemmParams p{};
            p.Ah = Ax, p.Wh = Wx, p.bias = vec, p.ln_in = stats, p.M = pt.M, p.N = N, p.K = C, p.bf16 = 4, p.out_h = outx, p.act = 1;
            snprintf(shape, sizeof shape, "%s fc1", pt.name);
            if (C != 224) {
                run<4, 2, 2, EPI_BIAS, 3, 16, 2>(shape, p);
                run<4, 2, 2, EPI_BIAS, 4, 16, 2>(shape, p);
                run<4, 2, 2, EPI_BIAS, 2, 32, 2>(shape, p);
                run<4, 2, 2, EPI_BIAS, 3, 32, 1>(shape, p);
                run<4, 1, 4, EPI_BIAS, 3, 16, 2>(shape, p);
                run<4, 1, 4, EPI_BIAS, 2, 32, 2>(shape, p);
                run<4, 2, 4, EPI_BIAS, 3, 16, 1>(shape, p);
                run<4, 2, 4, EPI_BIAS, 2, 32, 1>(shape, p);
                run<8, 1, 4, EPI_BIAS, 3, 16, 1>(shape, p);
                run<4, 1, 4, EPI_BIAS, 2, 16, 3>(shape, p);
                run<5, 2, 2, EPI_BIAS, 2, 16, 2>(shape, p);
                run<5, 1, 4, EPI_BIAS, 2, 16, 3>(shape, p);
                run<3, 2, 2, EPI_BIAS, 3, 16, 3>(shape, p);
            } else {
                run<4, 1, 7, EPI_BIAS, 2, 16, 2>(shape, p);
                run<4, 1, 7, EPI_BIAS, 3, 16, 1>(shape, p);
                run<4, 1, 7, EPI_BIAS, 2, 32, 1>(shape, p);
                run<8, 1, 7, EPI_BIAS, 2, 16, 1>(shape, p);
                run<8, 1, 7, EPI_BIAS, 3, 16, 1>(shape, p);
                run<5, 1, 7, EPI_BIAS, 2, 16, 2>(shape, p);
                run<3, 1, 7, EPI_BIAS, 2, 16, 2>(shape, p);
                run<3, 1, 7, EPI_BIAS, 3, 16, 2>(shape, p);
            }
        }
        {   // ---- qkv + attention as one kernel: the part's spatial (L = joints) and temporal (L = 27 frames) blocks
            const int heads = 8, d = C / heads, dp = d <= 32 ? 32 : 48, J = (int)(pt.M / 27 / 40);
            images(heads * 3 * dp, C);
            for (int temporal = 0; temporal < 2; ++temporal) {
                const int L = temporal ? 27 : J, lp = L <= 32 ? 32 : (L <= 48 ? 48 : 80);
                FqaParams f{};
                f.g.Ah = Ax, f.g.Wh = Wx, f.g.bias = vec, f.g.ln_in = stats, f.g.M = pt.M, f.g.N = heads * 3 * dp, f.g.K = C, f.g.bf16 = 4;
                f.o = reinterpret_cast<float*>(outx), f.L = L, f.C = C, f.heads = heads, f.d = d;
                f.nseq = temporal ? 40 * J : 40 * 27;
                f.nseq_tile = ((lp == 80 ? 160 : 128 + (lp == 48 ? 4 : 0)) - lp) / L + 1;
                if (temporal) f.group = J, f.group_stride = 27 * J, f.seq_stride = 1, f.tok_stride = J;
                else f.group = 1, f.group_stride = J, f.seq_stride = 0, f.tok_stride = 1;
                f.scale = 1.0f / sqrtf((float)d);
                snprintf(shape, sizeof shape, "%s %s L=%d", pt.name, temporal ? "temporal" : "spatial", L);
                if (lp == 32 && dp == 48) run_fqa<32, 48, 2>(shape, f);
                if (lp == 32 && dp == 32) { run_fqa<32, 32, 1>(shape, f); run_fqa<32, 32, 2>(shape, f); }
                if (lp == 48 && dp == 32) { run_fqa<48, 32, 1>(shape, f); run_fqa<48, 32, 2>(shape, f); }
                if (lp == 80 && dp == 32) { run_fqa<80, 32, 1>(shape, f); run_fqa<80, 32, 2>(shape, f); }
            }
        }
        // ---- whole-row layers
        for (int layer = 0; layer < 2; ++layer) {
            const int K = layer == 0 ? C : 2 * C;
            images(C, K);
            GemmParams q{};
            q.Ah = Ax, q.Wh = Wx, q.bias = vec, q.resid_h = xx, q.out_xh = xx, q.ln_stats = stats;   // the production form: image residual in place
            q.post_w = layer ? vec + 400 : nullptr, q.post_b = vec + 800, q.post_eps = 1e-6f, q.next_w = vec + 1200, q.next_b = vec + 1600, q.next_eps = 1e-6f;
            q.M = pt.M, q.N = C, q.K = K, q.bf16 = 4;
            snprintf(shape, sizeof shape, "%s %s", pt.name, layer == 0 ? "proj" : "fc2");
            if (getenv("XB_ABLATE")) {   // where the whole-row epilogue's cycles go (results wrong by design)
                char sh2[96];
                auto one = [&](const char* sh, GemmParams a) {
                    if (C == 384) run<4, 2, 6, EPI_ROWLN, 3, 16, 1>(sh, a);
                    else if (C == 256) run<2, 2, 4, EPI_ROWLN, 2, 16, 2>(sh, a);
                    else run<4, 1, 7, EPI_ROWLN, 2, 16, 2>(sh, a);
                };
                { GemmParams a = q; a.out_xh = nullptr; snprintf(sh2, sizeof sh2, "%s ABL no store", shape); one(sh2, a); }
                { GemmParams a = q; a.out_xh = nullptr; a.next_w = nullptr; a.post_w = nullptr; a.ln_stats = nullptr; snprintf(sh2, sizeof sh2, "%s ABL resid only", shape); one(sh2, a); }
                { GemmParams a = q; a.next_w = nullptr; a.post_w = nullptr; a.ln_stats = nullptr; snprintf(sh2, sizeof sh2, "%s ABL resid + store", shape); one(sh2, a); }
                one(shape, q);
                continue;
            }
            if (C == 384) {
                run<4, 2, 6, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<4, 2, 6, EPI_ROWLN, 2, 16, 1>(shape, q);
                run<2, 2, 6, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 2, 6, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<2, 4, 3, EPI_ROWLN, 2, 16, 1>(shape, q);
                run<3, 2, 6, EPI_ROWLN, 2, 16, 1>(shape, q);
                run<3, 2, 6, EPI_ROWLN, 3, 16, 1>(shape, q);
            } else if (C == 256) {
                run<2, 2, 4, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 2, 4, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<4, 2, 4, EPI_ROWLN, 2, 16, 1>(shape, q);
                run<4, 2, 4, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<2, 2, 4, EPI_ROWLN, 2, 32, 1>(shape, q);
                run<3, 2, 4, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<3, 2, 4, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<3, 1, 8, EPI_ROWLN, 2, 16, 2>(shape, q);
            } else {
                run<4, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<4, 1, 7, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<2, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 1, 7, EPI_ROWLN, 3, 16, 3>(shape, q);
                run<4, 1, 7, EPI_ROWLN, 2, 32, 1>(shape, q);
                run<5, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<3, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<3, 1, 7, EPI_ROWLN, 3, 16, 2>(shape, q);
            }
        }
    }
    return 0;
}
