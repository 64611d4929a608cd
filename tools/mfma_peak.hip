// mfma_peak.hip - what the f32-input matrix pipe delivers on this device: pure register MFMA loops (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float seed, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float a = seed * (threadIdx.x % 17 - 8), b = seed * (threadIdx.x % 13 - 6);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[n], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
// random fp32 operands (8 per lane, rotated) instead of one constant pair: what the chip sustains on real data
template <int NACC>
__global__ void __launch_bounds__(256) k32r(float* out, int iters, const float* rnd, unsigned long long* clk) {
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd[(threadIdx.x * 8 + i + blockIdx.x * 7) % 65536]; b[i] = rnd[(threadIdx.x * 8 + i + 32768 + blockIdx.x * 13) % 65536]; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + n) & 7], acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}
template <int NACC>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) acc[n][i] = 0.f;
    float a = seed * (threadIdx.x % 17 - 8), b = seed * (threadIdx.x % 13 - 6);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[n], 0, 0, 0);
        a += 1e-6f;
    }
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* out; unsigned long long* clk;
    const int maxb = 256 * 8;
    CK(hipMalloc(&out, maxb * 256 * 4)); CK(hipMalloc(&clk, maxb * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpc : {1, 2, 4}) {          // workgroups (4 waves) per CU
        int blocks = 256 * wpc, iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k32<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f, clk);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
        unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        printf("32x32x2 f32, 4 acc, %d WG/CU: %.2f ms  %.1f TF/s   clock %.0f MHz (s_memtime/s_memrealtime)\n", wpc, ms,
               flops / ms / 1e9, (double)h[0] / h[1] * 100.0);
    }
    {
        float* rnd; CK(hipMalloc(&rnd, 65536 * 4));
        float* hr = (float*)malloc(65536 * 4);
        for (int i = 0; i < 65536; ++i) hr[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
        CK(hipMemcpy(rnd, hr, 65536 * 4, hipMemcpyHostToDevice));
        for (int wpc : {1, 2}) {
            int blocks = 256 * wpc, iters = 40000;   // ~60 ms: long enough for DVFS to settle
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k32r<4>, dim3(blocks), dim3(256), 0, 0, out, iters, rnd, clk);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double flops = (double)blocks * 4 * iters * 8 * 4 * 4096.0;
            unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            printf("32x32x2 f32 RANDOM operands, 4 acc, %d WG/CU: %.2f ms  %.1f TF/s   clock %.0f MHz\n", wpc, ms,
                   flops / ms / 1e9, (double)h[0] / h[1] * 100.0);
        }
    }
    for (int wpc : {1, 2}) {
        int blocks = 256 * wpc, iters = 4000;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k16<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)blocks * 4 * iters * 8 * 4 * 2048.0;
        printf("16x16x4 f32, 4 acc, %d WG/CU: %.2f ms  %.1f TF/s\n", wpc, ms, flops / ms / 1e9);
    }
    return 0;
}
