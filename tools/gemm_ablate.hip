// gemm_ablate.hip - ablations of the linear-layer kernel's main loop (dev tool; results are WRONG by design).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ABL bits: 1 = no epilogue stores (one dummy store), 2 = no global loads in the loop, 4 = no barrier in the loop,
//           8 = no LDS writes in loop, 16 = setprio around MFMA
template <int WM, int WN, int NT, int NSTAGE, int ABL>
__global__ void __launch_bounds__(WM* WN * 64) gemm_abl(const GemmParams p) {
    using T = GemmTile<WM, WN, NT>;
    constexpr int NTHR = T::NTHR, BM = T::BM, BN = T::BN;
    constexpr int A_LD = BM * 8 / NTHR, W_LD = BN * 8 / NTHR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + NSTAGE * BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    const int tiles_n = p.N / BN;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int64_t m0 = (int64_t)tile_m * BM;
    const int n0 = tile_n * BN;
    const int K = p.K;
    if (ABL & 32) {  // de-phase the first residency round: workgroup b of the first 256*5 waits ((b/256)%5) * act * 64 clocks
        if (blockIdx.x < 1280) for (int i = 0; i < (int)((blockIdx.x >> 8) % 5) * p.act; ++i) __builtin_amdgcn_s_sleep(1);
    }
    const float* a_src[A_LD]; int a_dst[A_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
        const int idx = tid + i * NTHR, row = idx >> 3, c4 = idx & 7;
        int64_t gm = m0 + row; gm = gm < p.M ? gm : p.M - 1;
        a_src[i] = p.A + gm * K + c4 * 4; a_dst[i] = row * LDK + c4 * 4;
    }
    const float* w_src[W_LD]; int w_dst[W_LD];
#pragma unroll
    for (int i = 0; i < W_LD; ++i) {
        const int idx = tid + i * NTHR, row = idx >> 3, c4 = idx & 7;
        w_src[i] = p.W + (int64_t)(n0 + row) * K + c4 * 4; w_dst[i] = row * LDK + c4 * 4;
    }
    f32x4 a_reg[A_LD], w_reg[W_LD];
#pragma unroll
    for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_src[i]);
#pragma unroll
    for (int i = 0; i < W_LD; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(w_src[i]);
#pragma unroll
    for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(As + a_dst[i]) = a_reg[i];
#pragma unroll
    for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Ws + w_dst[i]) = w_reg[i];
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
    const int a_frag = (wm * 32 + r) * LDK + 4 * h;
    const int w_frag = (wn * NT * 32 + r) * LDK + 4 * h;
    const int nk = K / BK;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = (NSTAGE == 2) ? (kc & 1) : 0;
        const bool more = kc + 1 < nk;
        if (more && !(ABL & 2)) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (kc + 1) * BK);
#pragma unroll
            for (int i = 0; i < W_LD; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(w_src[i] + (kc + 1) * BK);
        }
        const float* Ac = As + cur * BM * LDK + a_frag;
        const float* Wc = Ws + cur * BN * LDK + w_frag;
        if (ABL & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(Ac + 8 * g);
            f32x4 wf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[nt] = *reinterpret_cast<const f32x4*>(Wc + nt * 32 * LDK + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
        }
        if (ABL & 16) __builtin_amdgcn_s_setprio(0);
        if (NSTAGE == 1 && !(ABL & 4)) __syncthreads();
        if (more && !(ABL & 8)) {
            const int nxt = (NSTAGE == 2) ? (cur ^ 1) : 0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(As + nxt * BM * LDK + a_dst[i]) = a_reg[i];
#pragma unroll
            for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Ws + nxt * BN * LDK + w_dst[i]) = w_reg[i];
        }
        if (!(ABL & 4)) __syncthreads();
    }
    if (ABL & 1) {
        float s = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) s += acc[nt][reg];
        if (s == 123.456f) p.out[0] = s;
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + r;
        const float bv = p.bias[n];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int64_t m = m0 + wm * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            if (m < p.M) p.out[m * p.N + n] = acc[nt][reg] + bv;
        }
    }
}

template <int WM, int WN, int NT, int NSTAGE, int ABL>
void run(const char* tag, GemmParams p, int reps = 20) {
    using T = GemmTile<WM, WN, NT>;
    size_t lds = (size_t)NSTAGE * T::STAGE_FLOATS * 4;
    auto k = gemm_abl<WM, WN, NT, NSTAGE, ABL>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / reps, tf = 2.0 * p.M * p.N * p.K / (us * 1e-6) / 1e12;
    printf("%-44s tiles=%5ld : %8.1f us  %6.1f TF/s (%.1f%%)\n", tag, (long)tiles, us, tf, tf / 157.3 * 100);
}

int main() {
    const int64_t Mmax = 73440;
    float *A, *W, *bias, *out;
    CK(hipMalloc(&A, Mmax * 768 * 4)); CK(hipMalloc(&W, 1152 * 768 * 4)); CK(hipMalloc(&bias, 1152 * 4));
    CK(hipMalloc(&out, Mmax * 1152 * 4));
    std::vector<float> h(Mmax * 768);
    for (auto& v : h) v = (float)(rand() % 2001 - 1000) * 1e-3f;
    CK(hipMemcpy(A, h.data(), Mmax * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data() + 777, 1152 * 768 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, h.data(), 1152 * 4, hipMemcpyHostToDevice));
    GemmParams p{};
    p.A = A, p.W = W, p.bias = bias, p.out = out;
    p.M = 25920, p.N = 1152, p.K = 384;
    run<4, 1, 4, 2, 0>("body qkv s2 baseline", p);
    run<4, 1, 4, 2, 1>("body qkv s2 no-epilogue-stores", p);
    run<4, 1, 4, 2, 2>("body qkv s2 no-global-loads", p);
    run<4, 1, 4, 2, 4>("body qkv s2 no-barrier", p);
    run<4, 1, 4, 2, 8>("body qkv s2 no-lds-writes", p);
    run<4, 1, 4, 2, 16>("body qkv s2 setprio", p);
    run<4, 1, 4, 2, 1 | 2 | 8>("body qkv s2 no-epi no-loads no-ldswrites", p);
    run<4, 1, 4, 2, 1 | 2 | 4 | 8>("body qkv s2 MFMA+ds_read only", p);
    run<4, 1, 4, 1, 0>("body qkv s1 baseline", p);
    run<4, 1, 4, 1, 1>("body qkv s1 no-epilogue-stores", p);
    run<4, 1, 4, 1, 1 | 2 | 4 | 8>("body qkv s1 MFMA+ds_read only", p);
    {
        GemmParams big = p; big.M = 73440; big.N = 1152; big.K = 384;   // 574 x 18 = 10332 tiles of 128x64: ~8 rounds
        run<4, 1, 2, 1, 0>("big qkv <4,1,2> s1 baseline", big);
        for (int d : {100, 200, 300, 400}) { big.act = d; char t[64]; snprintf(t, 64, "big qkv <4,1,2> s1 stagger %d", d); run<4, 1, 2, 1, 32>(t, big); }
        big.act = 0;
    }
    p.M = 25600;   // 200 M-tiles * 9 = 1800 tiles
    run<4, 1, 4, 2, 0>("M=25600 (1800 tiles) s2 baseline", p);
    p.M = 128 * 256 * 2 / 9 * 9;  // dummy
    p.M = 32768; p.N = 1024;  // 256 x 8 = 2048 tiles = 4 full rounds at 2 WG/CU
    run<4, 1, 4, 2, 0>("M=32768 N=1024 (2048 tiles) s2 baseline", p);
    run<4, 1, 4, 2, 1>("M=32768 N=1024 (2048 tiles) s2 no-epilogue", p);
    return 0;
}
