// attn_bench.hip - stand-alone timing of the attention kernels at the hot-path shapes (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LP, int DP, int NW>
void run(const char* tag, AttnParams p, int64_t M) {
    constexpr int ITEMS = NW / (LP / 16);
    size_t lds = (size_t)2 * ITEMS * LP * (DP + 4) * 4;
    int64_t nitems = p.nseq * p.heads, grid = (nitems + ITEMS - 1) / ITEMS;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((attn_kernel<LP, DP, NW>), dim3(grid), dim3(NW * 64), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((attn_kernel<LP, DP, NW>), dim3(grid), dim3(NW * 64), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, bytes = (double)M * p.C * 16;
    printf("%-28s seqs=%6ld L=%2d d=%2d grid=%6ld : %7.1f us   %.2f TB/s of qkv+o (floor at 5 TB/s: %.1f us)\n", tag,
           (long)p.nseq, p.L, p.d, (long)grid, us, bytes / us / 1e6, bytes / 5e6);
}

int main() {
    const int64_t Mmax = 73440;
    float *qkv, *o;
    CK(hipMalloc(&qkv, Mmax * 1152 * 4)); CK(hipMalloc(&o, Mmax * 384 * 4));
    std::vector<float> h(Mmax * 1152);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(qkv, h.data(), Mmax * 1152 * 4, hipMemcpyHostToDevice));
    const int R = 40, F = 27;
    struct Part { const char* n; int J, C; } parts[3] = {{"body", 24, 384}, {"face", 68, 224}, {"hands", 42, 256}};
    for (auto& pt : parts) {
        AttnParams a{};
        a.qkv = qkv, a.o = o, a.C = pt.C, a.heads = 8, a.d = pt.C / 8, a.scale = 1.f / sqrtf((float)a.d);
        int64_t M = (int64_t)R * F * pt.J;
        char tag[64];
        // spatial
        a.nseq = R * F, a.L = pt.J, a.group = 1, a.group_stride = pt.J, a.seq_stride = 0, a.tok_stride = 1;
        snprintf(tag, 64, "%s spatial", pt.n);
        if (pt.J <= 32) { if (a.d <= 32) run<32, 32, 4>(tag, a, M); else run<32, 48, 4>(tag, a, M); }
        else if (pt.J <= 48) run<48, 32, 6>(tag, a, M);
        else run<80, 32, 5>(tag, a, M);
        // temporal
        a.nseq = (int64_t)R * pt.J, a.L = F, a.group = pt.J, a.group_stride = (int64_t)F * pt.J, a.seq_stride = 1, a.tok_stride = pt.J;
        snprintf(tag, 64, "%s temporal", pt.n);
        if (a.d <= 32) run<32, 32, 4>(tag, a, M); else run<32, 48, 4>(tag, a, M);
    }
    return 0;
}
