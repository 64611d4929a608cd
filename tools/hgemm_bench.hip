// hgemm_bench.hip - the f16x2 H-pipeline tiles (pafuse_amd/csrc/hgemm.hpp) at the hot path's layer shapes, several tile
// configurations per shape in one process: us per launch (HIP events), TFLOP/s of fp32-equivalent work, and the per-wave
// lifetime split by in-kernel stamps (prologue = until the first chunk is visible, K loop, epilogue), in shader cycles.
//   hipcc <library flags> -DPAFUSE_STAMPS tools/hgemm_bench.hip -o tools/bin/hgemm_bench ;  HB_FILTER=<substring> ./hgemm_bench
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
#include "../pafuse_amd/csrc/hgemm.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

static const char* g_filter = nullptr;
static int g_reps = 20;
static std::vector<float> g_hX, g_hW, g_hvec;
static float g_rstd = 1.3f;   // host copies of the activations, the weights and the vector pool (value checks)

template <int WM, int WN, int NT, int EPI, int NSTAGE, int BKC, int MINW, bool HRES = (EPI == EPI_ROWLN)>
void run(const char* shape, GemmParams p) {
    using T = HTile<WM, WN, NT, BKC>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s <%d,%d,%d> %dx%d st%d bkc%d minw%d", shape, WM, WN, NT, T::BM, T::BN, NSTAGE, BKC, MINW);
    if (g_filter && !strstr(tag, g_filter)) return;
    if (p.N % T::BN) { printf("%s: N %% BN != 0, skipped\n", tag); return; }
    const size_t lds = (size_t)NSTAGE * T::STAGE_BYTES;
    auto k = hgemm_kernel<WM, WN, NT, EPI, NSTAGE, BKC, MINW, HRES>;
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
    std::vector<unsigned long long> h(nw * SL);
    CK(hipMemcpy(h.data(), st, nw * SL * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; size_t n = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * SL] || !h[w * SL + 2]) continue;
        pro += h[w * SL + 3] - h[w * SL]; loop += h[w * SL + 1] - h[w * SL + 3]; epi += h[w * SL + 2] - h[w * SL + 1]; ++n;
    }
    const double us = ms * 1e3 / g_reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    const double mfma = (double)(p.K / 16) * 3 * NT * 32;   // this wave's own MFMA issue cycles
    printf("%-58s tiles %5ld (%.2f rounds at %d/CU) %7.2f us %6.1f TF (%.3f of 833) | per wave: prologue %5.0f  K loop %6.0f (MFMA %5.0f)  epilogue %6.0f\n",
           tag, (long)tiles, (double)tiles / (256.0 * occ), occ, us, tf, tf / 833.3, n ? pro / n : 0, n ? loop / n : 0, mfma, n ? epi / n : 0);
    fflush(stdout);
    CK(hipFree(st));
}

// the fused MLP kernel (hmlp_kernel): fc1 -> GELU -> fc2 -> whole-row epilogue of one part
template <int NT2, int MINW>
void run_mlp(const char* part, MlpParams m) {
    using T = MlpTile<NT2>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s mlp fused <%d> minw%d", part, NT2, MINW);
    if (g_filter && !strstr(tag, g_filter)) return;
    auto k = hmlp_kernel<NT2, MINW>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    const int64_t tiles = (m.g.M + 127) / 128;
    const size_t nw = tiles * 4;
    unsigned long long* st; CK(hipMalloc(&st, nw * SL * 8)); CK(hipMemset(st, 0, nw * SL * 8));
    m.g.stamps = st;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 256, T::LDS_BYTES));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), T::LDS_BYTES, 0, m);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(256), T::LDS_BYTES, 0, m);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nw * SL);
    CK(hipMemcpy(h.data(), st, nw * SL * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; size_t n = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * SL] || !h[w * SL + 2]) continue;
        pro += h[w * SL + 3] - h[w * SL]; loop += h[w * SL + 1] - h[w * SL + 3]; epi += h[w * SL + 2] - h[w * SL + 1]; ++n;
    }
    const double us = ms * 1e3 / g_reps, tf = 2.0 * 2.0 * m.g.M * T::C * T::HID / (us * 1e-6) / 1e12;
    const double mfma = (double)T::NSLAB * (T::NK1 * 12 + 4 * NT2 * 3) * 32;
    printf("%-58s tiles %5ld (%.2f rounds at %d/CU) %7.2f us %6.1f TF (%.3f of 833) | per wave: prologue %5.0f  loop %6.0f (MFMA %5.0f)  epilogue %6.0f\n",
           tag, (long)tiles, (double)tiles / (256.0 * occ), occ, us, tf, tf / 833.3, n ? pro / n : 0, n ? loop / n : 0, mfma, n ? epi / n : 0);
    fflush(stdout);
    CK(hipFree(st));
}

// the fused qkv + attention kernel (hfqa_kernel) of one block kind: L tokens per sequence, head dim d, temporal or spatial addressing
template <int LP, int DP, int HPW>
void run_fqa(const char* shape, FqaParams f) {
    using FT = HfqaTile<LP, DP, HPW>;
    char tag[160];
    snprintf(tag, sizeof tag, "%s qkv+attn fused <%d,%d,%d>", shape, LP, DP, HPW);
    if (g_filter && !strstr(tag, g_filter)) return;
    auto k = hfqa_kernel<LP, DP, HPW>;
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
    std::vector<unsigned long long> h(nw * SL);
    CK(hipMemcpy(h.data(), st, nw * SL * 8, hipMemcpyDeviceToHost));
    double pro = 0, loop = 0, epi = 0; size_t n = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * SL] || !h[w * SL + 2]) continue;
        pro += h[w * SL + 3] - h[w * SL]; loop += h[w * SL + 1] - h[w * SL + 3]; epi += h[w * SL + 2] - h[w * SL + 1]; ++n;
    }
    double p2[2] = {0, 0}, p3[2] = {0, 0};
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * SL] || !h[w * SL + 2]) continue;
        unsigned long long t = h[w * SL + 1];
        for (int hh = 0; hh < HPW; ++hh) { p2[hh] += h[w * SL + 4 + 2 * hh] - t; p3[hh] += h[w * SL + 5 + 2 * hh] - h[w * SL + 4 + 2 * hh]; t = h[w * SL + 5 + 2 * hh]; }
    }
    {   // values: a few sequences against an fp64 evaluation on the host (q | k | v over the DP padded columns, as the kernel)
        const int C = f.g.K, d = f.d, L = f.L;
        std::vector<uint8_t> oh((size_t)f.g.M * C * 4);
        CK(hipMemcpy(oh.data(), f.o, oh.size(), hipMemcpyDeviceToHost));
        auto tok = [&](int64_t sq, int t) { return (sq / f.group) * f.group_stride + (sq % f.group) * f.seq_stride + t * f.tok_stride; };
        double emax = 0, esum = 0; size_t en = 0;
        const int64_t picks[6] = {0, 1, 5, f.nseq / 2, f.nseq - 2, f.nseq - 1};
        for (int64_t sq : picks)
            for (int head = 0; head < f.heads; head += 3) {
                std::vector<double> q(L * DP), k(L * DP), v(L * DP);
                for (int t = 0; t < L; ++t)
                    for (int n = 0; n < 3 * DP; ++n) {
                        const float* x = &g_hX[(size_t)tok(sq, t) * C];
                        const float* w = &g_hW[(size_t)(head * 3 * DP + n) * C];
                        double a = 0;
                        for (int kk = 0; kk < C; ++kk) a += (double)x[kk] * w[kk];
                        a = (double)g_rstd * a + g_hvec[head * 3 * DP + n];
                        (n < DP ? q : (n < 2 * DP ? k : v))[t * DP + n % DP] = a;
                    }
                for (int t = 0; t < L; ++t) {
                    std::vector<double> pr(L);
                    double mx = -1e300, sum = 0;
                    for (int u = 0; u < L; ++u) { double a = 0; for (int e = 0; e < DP; ++e) a += q[t * DP + e] * k[u * DP + e]; pr[u] = a * f.scale; mx = std::max(mx, pr[u]); }
                    for (int u = 0; u < L; ++u) { pr[u] = exp(pr[u] - mx); sum += pr[u]; }
                    for (int ch = 0; ch < d; ++ch) {
                        double o = 0;
                        for (int u = 0; u < L; ++u) o += pr[u] / sum * v[u * DP + ch];
                        const size_t col = (size_t)head * d + ch;
                        const _Float16* sb = reinterpret_cast<const _Float16*>(&oh[((size_t)tok(sq, t) * C + (col & ~(size_t)7)) * 4]);
                        const double got = (double)(float)sb[col & 7] + (double)(float)sb[8 + (col & 7)] * 0.00048828125;
                        const double e = fabs(got - o);
                        emax = std::max(emax, e); esum += e; ++en;
                    }
                }
            }
        printf("    values vs fp64 on %zu outputs: max |err| %.3e  mean %.3e\n", en, emax, esum / en);
    }
    const double us = ms * 1e3 / g_reps;
    const double mfma = (double)(f.g.K / 32) * 2 * FT::NB * 3 * 16;
    if (n) printf("    heads: phase 2 %6.0f / phase 3 %6.0f", p2[0] / n, p3[0] / n);
    if (n && HPW == 2) printf("  |  phase 2 %6.0f / phase 3 %6.0f", p2[1] / n, p3[1] / n);
    if (n) printf("\n");
    printf("%-58s wgs %6ld (%.2f rounds at %d/CU) %7.2f us | per wave: prologue %5.0f  projection loop %6.0f (MFMA %5.0f)  attention phases %6.0f\n",
           tag, (long)blocks, (double)blocks / (256.0 * occ), occ, us, n ? pro / n : 0, n ? loop / n : 0, mfma, n ? epi / n : 0);
    fflush(stdout);
    CK(hipFree(st));
}

int main() {
    g_filter = getenv("HB_FILTER");
    if (getenv("HB_REPS")) g_reps = atoi(getenv("HB_REPS"));
    if (getenv("HB_RSTD")) g_rstd = (float)atof(getenv("HB_RSTD"));
    const int64_t Mmax = 73440;
    float *X, *W, *vec, *out, *x, *stats;
    uint8_t *Ah, *Wh, *outh, *xh;
    CK(hipMalloc(&X, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&vec, 4096 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4)); CK(hipMalloc(&x, Mmax * 384 * 4)); CK(hipMalloc(&stats, Mmax * 8));
    CK(hipMalloc(&Ah, Mmax * 768 * 4)); CK(hipMalloc(&Wh, 1152 * 768 * 4 + 256)); CK(hipMalloc(&outh, Mmax * 1152 * 4)); CK(hipMalloc(&xh, Mmax * 384 * 4));
    std::vector<float> h(Mmax * 768);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(X, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    g_hX = h;
    for (size_t i = 0; i < 1152 * 768; ++i) h[i] *= 0.05f;
    CK(hipMemcpy(W, h.data(), 1152 * 768 * 4, hipMemcpyHostToDevice));
    g_hW.assign(h.begin(), h.begin() + 1152 * 768);
    CK(hipMemcpy(vec, h.data() + 999, 4096 * 4, hipMemcpyHostToDevice));
    g_hvec.assign(h.begin() + 999, h.begin() + 999 + 4096);
    CK(hipMemcpy(x, h.data() + 5, Mmax * 384 * 4, hipMemcpyHostToDevice));
    std::vector<float> sth(Mmax * 2);
    for (int64_t i = 0; i < Mmax; ++i) sth[2 * i] = 0.01f, sth[2 * i + 1] = g_rstd;
    CK(hipMemcpy(stats, sth.data(), Mmax * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hsplit_rows_kernel, dim3((unsigned)((Mmax * 96 + 255) / 256)), dim3(256), 0, 0, X, Ah, Mmax, 768);   // (row stride 4 K: re-made per K below)

    struct Part { const char* name; int64_t M; int C; };
    const Part parts[3] = {{"body", 25920, 384}, {"face", 73440, 224}, {"hands", 45360, 256}};
    for (const Part& pt : parts) {
        const int C = pt.C;
        auto images = [&](int N, int K) {   // A [M,K] and W [N,K] as H images
            hipLaunchKernelGGL(hsplit_rows_kernel, dim3((unsigned)((pt.M * (K / 8) + 255) / 256)), dim3(256), 0, 0, X, Ah, pt.M, K);
            CK(hipMemset(Wh + (size_t)N * K * 4, 0, 256));
            hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, 0, W, (int64_t)N * K, reinterpret_cast<uint32_t*>(Wh + (size_t)N * K * 4) + 1);
            hipLaunchKernelGGL(hsplit_weights_kernel, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, W, Wh, N, K);
        };
        char shape[64];
        // ---- plain layers
        for (int layer = 0; layer < 2; ++layer) {
            const int N = layer == 0 ? 3 * C : 2 * C;
            images(N, C);
            GemmParams p{};
            p.Ah = Ah, p.Wh = Wh, p.bias = vec, p.ln_in = stats, p.ln_s = vec + 1500, p.M = pt.M, p.N = N, p.K = C, p.bf16 = 3;
            if (layer == 0) p.out = out; else p.out_h = outh, p.act = 1;
            snprintf(shape, sizeof shape, "%s %s", pt.name, layer == 0 ? "qkv" : "fc1");
            if (C != 224) {
                run<8, 1, 4, EPI_BIAS, 3, 32, 1>(shape, p);
                run<8, 1, 4, EPI_BIAS, 2, 32, 1>(shape, p);
                run<4, 1, 4, EPI_BIAS, 2, 32, 2>(shape, p);
                run<4, 1, 4, EPI_BIAS, 3, 16, 2>(shape, p);
                run<4, 1, 4, EPI_BIAS, 2, 16, 3>(shape, p);
                run<4, 2, 2, EPI_BIAS, 2, 32, 1>(shape, p);
                run<4, 2, 2, EPI_BIAS, 3, 16, 2>(shape, p);
                run<8, 1, 2, EPI_BIAS, 2, 32, 2>(shape, p);
            } else {
                run<8, 1, 7, EPI_BIAS, 2, 32, 1>(shape, p);
                run<4, 1, 7, EPI_BIAS, 2, 32, 1>(shape, p);
                run<4, 1, 7, EPI_BIAS, 3, 16, 2>(shape, p);
                run<4, 1, 7, EPI_BIAS, 2, 16, 2>(shape, p);
                run<8, 1, 7, EPI_BIAS, 3, 16, 1>(shape, p);
            }
        }
        {   // ---- qkv + attention as one kernel: the part's spatial (L = joints) and temporal (L = 27 frames) blocks
            const int heads = 8, d = C / heads, dp = d <= 32 ? 32 : 48, J = (int)(pt.M / 27 / 40);
            images(heads * 3 * dp, C);
            for (int temporal = 0; temporal < 2; ++temporal) {
                const int L = temporal ? 27 : J, lp = L <= 32 ? 32 : (L <= 48 ? 48 : 80);
                FqaParams f{};
                f.g.Ah = Ah, f.g.Wh = Wh, f.g.bias = vec, f.g.ln_in = stats, f.g.M = pt.M, f.g.N = heads * 3 * dp, f.g.K = C, f.g.bf16 = 3;
                f.o = reinterpret_cast<float*>(outh), f.L = L, f.C = C, f.heads = heads, f.d = d;
                f.nseq = temporal ? 40 * J : 40 * 27;
                f.nseq_tile = ((lp == 80 ? 160 : 128 + (lp == 48 ? 4 : 0)) - lp) / L + 1;
                if (temporal) f.group = J, f.group_stride = 27 * J, f.seq_stride = 1, f.tok_stride = J;
                else f.group = 1, f.group_stride = J, f.seq_stride = 0, f.tok_stride = 1;
                f.scale = 1.0f / sqrtf((float)d);
                snprintf(shape, sizeof shape, "%s %s L=%d", pt.name, temporal ? "temporal" : "spatial", L);
                if (lp == 32 && dp == 48) run_fqa<32, 48, 1>(shape, f);
                if (lp == 32 && dp == 32) { run_fqa<32, 32, 1>(shape, f); run_fqa<32, 32, 2>(shape, f); }
                if (lp == 48 && dp == 32) { run_fqa<48, 32, 1>(shape, f); run_fqa<48, 32, 2>(shape, f); }
                if (lp == 80 && dp == 32) run_fqa<80, 32, 1>(shape, f);
            }
        }
        {   // ---- the MLP as one kernel (weights in natural column order: timing only)
            images(2 * C, C);
            uint8_t* W2h;
            CK(hipMalloc(&W2h, (size_t)C * 2 * C * 4 + 256));
            CK(hipMemset(W2h + (size_t)C * 2 * C * 4, 0, 256));
            hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, 0, W, (int64_t)C * 2 * C, reinterpret_cast<uint32_t*>(W2h + (size_t)C * 2 * C * 4) + 1);
            hipLaunchKernelGGL(hsplit_weights_kernel, dim3((unsigned)(((int64_t)C * (2 * C / 8) + 255) / 256)), dim3(256), 0, 0, W, W2h, C, 2 * C);
            MlpParams m{};
            m.g.Ah = xh, m.g.Wh = W2h, m.g.bias = vec, m.g.resid_h = xh, m.g.out_xh = xh, m.g.ln_stats = stats;
            m.g.post_w = vec + 400, m.g.post_b = vec + 800, m.g.post_eps = 1e-6f, m.g.next_w = vec + 1200, m.g.next_b = vec + 1600, m.g.next_eps = 1e-6f;
            m.g.M = pt.M, m.g.N = C, m.g.K = 2 * C, m.g.bf16 = 3;
            m.W1h = Wh, m.bias1 = vec + 2000, m.ln_in = stats;
            if (C == 224) { run_mlp<7, 2>(pt.name, m); }
            if (C == 256) { run_mlp<8, 2>(pt.name, m); }
            if (C == 384) { run_mlp<12, 1>(pt.name, m); }
            CK(hipDeviceSynchronize());
            CK(hipFree(W2h));
        }
        // ---- whole-row layers
        for (int layer = 0; layer < 2; ++layer) {
            const int K = layer == 0 ? C : 2 * C;
            images(C, K);
            GemmParams q{};
            q.Ah = Ah, q.Wh = Wh, q.bias = vec, q.resid_h = xh, q.out_xh = xh, q.ln_stats = stats;   // the production form: H-image residual in place
            q.post_w = layer ? vec + 400 : nullptr, q.post_b = vec + 800, q.post_eps = 1e-6f, q.next_w = vec + 1200, q.next_b = vec + 1600, q.next_eps = 1e-6f;
            q.M = pt.M, q.N = C, q.K = K, q.bf16 = 3;
            snprintf(shape, sizeof shape, "%s %s", pt.name, layer == 0 ? "proj" : "fc2");
            if (C == 384 && getenv("HB_ABLATE")) {   // where the whole-row epilogue's cycles go (results wrong by design)
                char sh2[96];
                { GemmParams a = q; a.out_xh = nullptr; snprintf(sh2, sizeof sh2, "%s ABL no store", shape); run<4, 2, 6, EPI_ROWLN, 2, 32, 1>(sh2, a); }
                { GemmParams a = q; a.out_xh = nullptr; a.next_w = nullptr; a.post_w = nullptr; snprintf(sh2, sizeof sh2, "%s ABL resid only", shape); run<4, 2, 6, EPI_ROWLN, 2, 32, 1>(sh2, a); }
            }
            if (C == 384) {
                run<4, 2, 6, EPI_ROWLN, 2, 32, 1>(shape, q);
                run<4, 2, 6, EPI_ROWLN, 3, 16, 1>(shape, q);
                run<2, 2, 6, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 2, 6, EPI_ROWLN, 2, 32, 1>(shape, q);
                run<2, 2, 6, EPI_ROWLN, 3, 16, 2>(shape, q);
                run<4, 2, 6, EPI_ROWLN, 2, 16, 2>(shape, q);
            } else if (C == 256) {
                run<4, 2, 4, EPI_ROWLN, 3, 32, 1>(shape, q);
                run<4, 2, 4, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 2, 4, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 2, 4, EPI_ROWLN, 3, 16, 3>(shape, q);
                run<2, 2, 4, EPI_ROWLN, 2, 32, 2>(shape, q);
                run<4, 2, 4, EPI_ROWLN, 2, 16, 2>(shape, q);
            } else {
                run<4, 1, 7, EPI_ROWLN, 3, 16, 2>(shape, q);
                run<4, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 1, 7, EPI_ROWLN, 2, 16, 2>(shape, q);
                run<2, 1, 7, EPI_ROWLN, 3, 16, 3>(shape, q);
                run<4, 1, 7, EPI_ROWLN, 2, 32, 1>(shape, q);
            }
        }
    }
    return 0;
}
