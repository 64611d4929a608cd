// mfma_bf16_peak.hip - what the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) sustains on this device, from registers and
// fed from LDS at the split-precision kernels' read ratio (dev tool, round 2).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_bf16_peak.hip -o build/mfma_bf16_peak
// Variants:  ORDER 0 = six dependent MFMAs on one accumulator, then the next accumulator (the production order)
//            ORDER 1 = consecutive MFMAs on different accumulators (no back-to-back dependence)
//            LDSR    = ds_read_b128 per six MFMAs (0 = operands stay in registers; 3 = one W' fragment per group, as in
//                      the split kernels; 5 = plus an A fragment per group, the NT = 1 worst case)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NACC, int ORDER, int LDSR, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64, 2) kern(float* out, int iters, const uint32_t* rnd, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += NWAVE * 64) lds[i] = rnd[i];  // 64 KB of random bf16 pairs
    __syncthreads();
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    u32x4 a[3], w[3];
    for (int s = 0; s < 3; ++s) for (int i = 0; i < 4; ++i) { a[s][i] = rnd[(tid * 12 + s * 4 + i) & 16383]; w[s][i] = rnd[(tid * 12 + s * 4 + i + 7777) & 16383]; }
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds + (tid & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t off = 0;
    for (int it = 0; it < iters; ++it) {
        if constexpr (ORDER == 0) {
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                u32x4 nw[3], na[2];
                if constexpr (LDSR >= 3) {  // the NEXT group's fragment, in flight during this group's MFMAs
                    const uint32_t ad = base + ((off + n * 3072) & 0xffff & ~1023u);
                    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048"
                                 : "=v"(nw[0]), "=v"(nw[1]), "=v"(nw[2]), "+v"(acc[n]) : "v"(ad));
                    if constexpr (LDSR >= 5)
                        asm volatile("ds_read_b128 %0, %2 offset:3072\n ds_read_b128 %1, %2 offset:4096" : "=v"(na[0]), "=v"(na[1]) : "v"(ad));
                }
                const int ws[6] = {0, 2, 1, 0, 1, 0}, as[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ws[j]]), __builtin_bit_cast(bf16x8, a[as[j]]), acc[n], 0, 0, 0);
                if constexpr (LDSR >= 3) {
                    if constexpr (LDSR >= 5) {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nw[0]), "+v"(nw[1]), "+v"(nw[2]), "+v"(na[0]), "+v"(na[1]), "+v"(acc[n]));
                        a[1] = na[0], a[2] = na[1];
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nw[0]), "+v"(nw[1]), "+v"(nw[2]), "+v"(acc[n]));
                    }
                    w[0] = nw[0], w[1] = nw[1], w[2] = nw[2];
                }
            }
        } else {
            const int ws[6] = {0, 2, 1, 0, 1, 0}, as[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int n = 0; n < NACC; ++n)
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[ws[j]]), __builtin_bit_cast(bf16x8, a[as[j]]), acc[n], 0, 0, 0);
        }
        off += 3072 * NACC;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * NWAVE * 64 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int ORDER, int LDSR, int NWAVE>
void run(const char* tag, int wg_per_cu, float* out, const uint32_t* rnd, unsigned long long* clk) {
    const int blocks = 256 * wg_per_cu, iters = 6000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = kern<NACC, ORDER, LDSR, NWAVE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NWAVE * 64), 65536, 0, out, iters, rnd, clk);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flops = (double)blocks * NWAVE * iters * NACC * 6 * 32768.0;
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-52s %d WG/CU x %d waves: %7.2f ms  %7.1f TF/s = %.3f of 2500   clock %.0f MHz\n", tag, wg_per_cu, NWAVE, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 2500.0, (double)h[0] / h[1] * 100.0);
    fflush(stdout);
}

// The same work on v_mfma_f32_16x16x32_bf16 (MI355X_MICROARCH.md "DVFS give-back" item 7): one 32 x 32
// output tile per accumulator set = four 16 x 16 blocks, a 32-deep step = 6 products x 4 blocks = 24 MFMAs of 16 cycles (the
// 32x32x16 loop: 12 MFMAs of 32 cycles for the same 32-deep step).  LDSR = 6: the six W' fragments (2 row halves x 3 slices)
// of the next group are read while this group multiplies - the same LDS bytes per FLOP as LDSR = 3 above.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int NACC, int LDSR, int NWAVE>
__global__ void __launch_bounds__(NWAVE * 64, 2) kern16(float* out, int iters, const uint32_t* rnd, unsigned long long* clk) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += NWAVE * 64) lds[i] = rnd[i];
    __syncthreads();
    f32x4_t acc[NACC][4];
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) acc[n][q][i] = 0.f;
    u32x4 a[2][3], w[2][3];
    for (int hh = 0; hh < 2; ++hh) for (int s = 0; s < 3; ++s) for (int i = 0; i < 4; ++i) {
        a[hh][s][i] = rnd[(tid * 24 + hh * 12 + s * 4 + i) & 16383]; w[hh][s][i] = rnd[(tid * 24 + hh * 12 + s * 4 + i + 7777) & 16383]; }
    const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds + (tid & 63) * 16;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t off = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            u32x4 nw[2][3];
            if constexpr (LDSR >= 6) {
                const uint32_t ad = base + ((off + n * 6144) & 0xffff & ~1023u);
                asm volatile("ds_read_b128 %0, %6\n ds_read_b128 %1, %6 offset:1024\n ds_read_b128 %2, %6 offset:2048\n"
                             "ds_read_b128 %3, %6 offset:3072\n ds_read_b128 %4, %6 offset:4096\n ds_read_b128 %5, %6 offset:5120"
                             : "=v"(nw[0][0]), "=v"(nw[0][1]), "=v"(nw[0][2]), "=v"(nw[1][0]), "=v"(nw[1][1]), "=v"(nw[1][2]) : "v"(ad));
            }
            const int ws[6] = {0, 2, 1, 0, 1, 0}, as[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[n][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[q >> 1][ws[j]]), __builtin_bit_cast(bf16x8, a[q & 1][as[j]]), acc[n][q], 0, 0, 0);
            if constexpr (LDSR >= 6) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nw[0][0]), "+v"(nw[0][1]), "+v"(nw[0][2]), "+v"(nw[1][0]), "+v"(nw[1][1]), "+v"(nw[1][2]));
                for (int hh = 0; hh < 2; ++hh) for (int s = 0; s < 3; ++s) w[hh][s] = nw[hh][s];
            }
        }
        off += 6144 * NACC;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int n = 0; n < NACC; ++n) for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) s += acc[n][q][i];
    out[blockIdx.x * NWAVE * 64 + tid] = s;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC, int LDSR, int NWAVE>
void run16(const char* tag, int wg_per_cu, float* out, const uint32_t* rnd, unsigned long long* clk) {
    const int blocks = 256 * wg_per_cu, iters = 3000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = kern16<NACC, LDSR, NWAVE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NWAVE * 64), 65536, 0, out, iters, rnd, clk);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double flops = (double)blocks * NWAVE * iters * NACC * 24 * 16384.0;
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    printf("%-52s %d WG/CU x %d waves: %7.2f ms  %7.1f TF/s = %.3f of 2500   clock %.0f MHz\n", tag, wg_per_cu, NWAVE, ms,
           flops / ms / 1e9, flops / ms / 1e9 / 2500.0, (double)h[0] / h[1] * 100.0);
    fflush(stdout);
}

int main() {
    float* out; unsigned long long* clk; uint32_t* rnd;
    CK(hipMalloc(&out, 512 * 512 * 4)); CK(hipMalloc(&clk, 512 * 16)); CK(hipMalloc(&rnd, 16384 * 4));
    uint32_t* h = (uint32_t*)malloc(16384 * 4);
    for (int i = 0; i < 16384; ++i) {  // two random bf16 in [-1, 1) per word
        auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u >> 16; };
        h[i] = bf((float)rand() / RAND_MAX * 2.f - 1.f) | (bf((float)rand() / RAND_MAX * 2.f - 1.f) << 16);
    }
    CK(hipMemcpy(rnd, h, 16384 * 4, hipMemcpyHostToDevice));
    run<4, 0, 0, 4>("registers, chains of 6 on one accumulator, 4 acc", 1, out, rnd, clk);
    run<4, 0, 0, 4>("registers, chains of 6 on one accumulator, 4 acc", 2, out, rnd, clk);
    run<4, 1, 0, 4>("registers, round-robin over 4 acc", 1, out, rnd, clk);
    run<4, 1, 0, 4>("registers, round-robin over 4 acc", 2, out, rnd, clk);
    run<4, 0, 3, 4>("3 ds_read_b128 per 6 MFMAs (prefetched one group ahead)", 1, out, rnd, clk);
    run<4, 0, 3, 4>("3 ds_read_b128 per 6 MFMAs (prefetched one group ahead)", 2, out, rnd, clk);
    run<4, 0, 5, 4>("5 ds_read_b128 per 6 MFMAs", 1, out, rnd, clk);
    run<4, 0, 5, 4>("5 ds_read_b128 per 6 MFMAs", 2, out, rnd, clk);
    run<6, 0, 3, 8>("3 reads per 6 MFMAs, 8-wave workgroup, 6 acc", 1, out, rnd, clk);
    // the 16x16x32 shape at the same output tile per wave, same LDS bytes per FLOP
    run16<4, 0, 4>("16x16x32: registers, 4 tiles of 32x32", 1, out, rnd, clk);
    run16<4, 0, 4>("16x16x32: registers, 4 tiles of 32x32", 2, out, rnd, clk);
    run16<4, 6, 4>("16x16x32: 6 ds_read_b128 per 24 MFMAs", 1, out, rnd, clk);
    run16<4, 6, 4>("16x16x32: 6 ds_read_b128 per 24 MFMAs", 2, out, rnd, clk);
    run<4, 0, 3, 4>("32x32x16 again: 3 ds_read_b128 per 6 MFMAs", 2, out, rnd, clk);
    run16<4, 6, 4>("16x16x32 again: 6 ds_read_b128 per 24 MFMAs", 2, out, rnd, clk);
    return 0;
}
