// gemm_stamps.hip - s_memtime stamps inside the linear-layer main loop: where a chunk's cycles go (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

// segments: 0 prologue, 1 global-load issue, 2 frag reads + MFMA issue, 3 vmcnt wait + LDS write, 4 barrier, 5 epilogue
template <int WM, int WN, int NT, int NSTAGE>
__global__ void __launch_bounds__(WM* WN * 64) gemm_st(const GemmParams p, unsigned long long* st) {
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
    unsigned long long seg[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t0 = stamp(), t1;
    const unsigned long long tstart = t0;
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
    t1 = stamp(); seg[0] += t1 - t0; t0 = t1;
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
        if (more) {
#pragma unroll
            for (int i = 0; i < A_LD; ++i) a_reg[i] = *reinterpret_cast<const f32x4*>(a_src[i] + (kc + 1) * BK);
#pragma unroll
            for (int i = 0; i < W_LD; ++i) w_reg[i] = *reinterpret_cast<const f32x4*>(w_src[i] + (kc + 1) * BK);
        }
        t1 = stamp(); seg[1] += t1 - t0; t0 = t1;
        const float* Ac = As + cur * BM * LDK + a_frag;
        const float* Wc = Ws + cur * BN * LDK + w_frag;
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
        t1 = stamp(); seg[2] += t1 - t0; t0 = t1;
        if (NSTAGE == 1) __syncthreads();
        if (more) {
            const int nxt = (NSTAGE == 2) ? (cur ^ 1) : 0;
#pragma unroll
            for (int i = 0; i < A_LD; ++i) *reinterpret_cast<f32x4*>(As + nxt * BM * LDK + a_dst[i]) = a_reg[i];
#pragma unroll
            for (int i = 0; i < W_LD; ++i) *reinterpret_cast<f32x4*>(Ws + nxt * BN * LDK + w_dst[i]) = w_reg[i];
        }
        t1 = stamp(); seg[3] += t1 - t0; t0 = t1;
        __syncthreads();
        t1 = stamp(); seg[4] += t1 - t0; t0 = t1;
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t1 = stamp(); seg[5] += t1 - t0;
    if (lane == 0) {
        unsigned long long* o = st + ((size_t)blockIdx.x * (NTHR / 64) + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = seg[i];
        o[6] = t1 - tstart;
        o[7] = tstart;
    }
}

template <int WM, int WN, int NT, int NSTAGE>
void run(const char* tag, GemmParams p) {
    using T = GemmTile<WM, WN, NT>;
    size_t lds = (size_t)NSTAGE * T::STAGE_FLOATS * 4;
    auto k = gemm_st<WM, WN, NT, NSTAGE>;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int64_t tiles = (p.M + T::BM - 1) / T::BM * (p.N / T::BN);
    size_t nw = tiles * (T::NTHR / 64);
    unsigned long long* st; CK(hipMalloc(&st, nw * 64));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(tiles), dim3(T::NTHR), lds, 0, p, st);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(nw * 8);
    CK(hipMemcpy(h.data(), st, nw * 64, hipMemcpyDeviceToHost));
    double s[7] = {0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t w = 0; w < nw; ++w) {
        for (int i = 0; i < 7; ++i) s[i] += h[w * 8 + i];
        tmin = std::min(tmin, h[w * 8 + 7]); tmax = std::max(tmax, h[w * 8 + 7] + h[w * 8 + 6]);
    }
    int nk = p.K / 32;
    printf("%s: waves=%zu chunks=%d  kernel span %.0f cyc\n  per wave: prologue %.0f | per chunk: load-issue %.0f  frag+MFMA %.0f  vmcnt+ldswrite %.0f  barrier %.0f | epilogue %.0f | lifetime %.0f (MFMA ideal %d)\n",
           tag, nw, nk, (double)(tmax - tmin), s[0] / nw, s[1] / nw / nk, s[2] / nw / nk, s[3] / nw / nk, s[4] / nw / nk, s[5] / nw, s[6] / nw, nk * 16 * NT * 64);
    CK(hipFree(st));
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
    run<4, 1, 4, 2>("body qkv <4,1,4> s2", p);
    run<4, 1, 4, 1>("body qkv <4,1,4> s1", p);
    p.M = 25920, p.N = 384, p.K = 768;
    run<2, 4, 3, 2>("body fc2 tile <2,4,3> s2 (plain epilogue)", p);
    run<1, 4, 3, 1>("body fc2 tile <1,4,3> s1 (plain epilogue)", p);
    return 0;
}
