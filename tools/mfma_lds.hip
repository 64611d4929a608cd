// mfma_lds.hip - what caps the K loop of the linear-layer kernel?  The production loop body (per 32-wide K chunk and
// wave: 12 ds_read_b128 fragment reads feeding 32 v_mfma_f32_32x32x2_f32, 6 ds_write_b128, 2 barriers) rebuilt piece
// by piece on LDS-resident data (no global traffic), 5 workgroups of 4 waves per CU like production:
//   V0 MFMAs on register operands      V1 + fragment reads from LDS      V2 + the two barriers per chunk
//   V3 + the LDS refill writes         V4 = V1 with the accumulators in AGPRs (inline asm)
//   V5 + the 6 global_load_dwordx4 of the register-staged refill (V6 half of them, V8 no s_setprio, V12 prefetch
//   distance 2)   V7/V9/V10/V11 the refill as LDS-DMA (scratch target / after the cluster / double-buffered / 16-wide)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lds.hip -o build/mfma_lds ; gpurun -- ./build/mfma_lds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36, BM = 128, BN = 64;

template <int V>
__global__ void __launch_bounds__(256, 5) loop_kernel(float* out, int chunks, const float* src, const float* big,
                                                      int rows, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (BM + BN) * LDK; i += 256) smem[i] = src[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const float* Ac = As + (wave * 32 + r) * LDK + 4 * h;
    const float* Wc = Ws + r * LDK + 4 * h;
    f32x4 st[6];
    for (int i = 0; i < 6; ++i) st[i] = *reinterpret_cast<const f32x4*>(smem + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4);
    float ra[4] = {src[tid], src[tid + 1], src[tid + 2], src[tid + 3]};
    // V5/V6: the production staging - every thread fetches 6 float4 of the NEXT chunk from an L2-resident matrix
    // (row (tid>>3) + 32 i of a tile that moves with the workgroup, K-chunk kc) before the MFMA cluster
    const int tile_row0 = (blockIdx.x * 192) % (rows - 192);
    const float* gbase = big + (size_t)(tile_row0 + (tid >> 3)) * K + (tid & 7) * 4;
    const int nkc = K / 32;
    for (int kc = 0; kc < chunks; ++kc) {
        const float* gp = gbase + (kc % nkc) * 32;
        if (V == 5 || V == 8) {
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)32 * i * K);
        }
        if (V == 6) {  // half the bytes
#pragma unroll
            for (int i = 0; i < 3; ++i) st[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)32 * i * K);
        }
        if (V == 7) {  // LDS-DMA: same bytes, no VGPR write-back (lands in a scratch LDS area behind the tiles)
#pragma unroll
            for (int i = 0; i < 6; ++i)
                __builtin_amdgcn_global_load_lds(gp + (size_t)32 * i * K,
                                                 (__attribute__((address_space(3))) void*)(smem + (BM + BN) * LDK + (i * 4 + wave) * 256),
                                                 16, 0, 0);
        }
        __builtin_amdgcn_s_setprio(V == 8 ? 0 : 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 af, wf[2];
            if (V == 0) {
                af = f32x4{ra[0], ra[1], ra[2], ra[3]};
                wf[0] = f32x4{ra[1], ra[2], ra[3], ra[0]};
                wf[1] = f32x4{ra[2], ra[3], ra[0], ra[1]};
            } else {
                af = *reinterpret_cast<const f32x4*>(Ac + 8 * g);
                wf[0] = *reinterpret_cast<const f32x4*>(Wc + 8 * g);
                wf[1] = *reinterpret_cast<const f32x4*>(Wc + 32 * LDK + 8 * g);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    if (V == 4)
                        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[nt]) : "v"(af[j]), "v"(wf[nt][j]));
                    else
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
                }
        }
        __builtin_amdgcn_s_setprio(0);
        if (V == 2 || V == 3 || V >= 5) __syncthreads();
        if (V == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (V == 3 || V == 5 || V == 6 || V == 8) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                *reinterpret_cast<f32x4*>(smem + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = st[i];
        }
        if (V == 2 || V == 3 || V >= 5) __syncthreads();
        if (V == 0) ra[kc & 3] += 1e-6f;
    }
    float s = 0;
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + tid] = s;
}

// V9: the candidate production loop - unpadded [row][32] tiles with the 16-byte chunk c of row r stored at position
// c ^ (r & 7) (conflict-free b128 fragment reads without padding), refilled by LDS-DMA AFTER the MFMA cluster
// (single LDS stage, no staging registers): cluster -> barrier -> 6 global_load_lds_dwordx4 -> vmcnt(0) -> barrier.
template <int MINW>
__global__ void __launch_bounds__(256, MINW) dma_kernel(float* out, int chunks, const float* big, int rows, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;             // [128][32]
    float* Ws = smem + BM * 32;   // [64][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (BM + BN) * 32; i += 256) smem[i] = big[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    // staging: instruction i of wave w fills rows 8*(4i + w) .. +7 (1 KB); lane -> (row = lane>>3, slot = lane&7),
    // which must hold global chunk slot ^ (row & 7)
    const int srow = lane >> 3, slot = lane & 7;
    const int tile_row0 = (blockIdx.x * 192) % (rows - 192);
    const float* gbase = big + (size_t)(tile_row0 + 8 * wave + srow) * K + ((slot ^ srow) & 7) * 4;
    const int nkc = K / 32;
    const int sw = r & 7;
    for (int kc = 0; kc < chunks; ++kc) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int pos = ((2 * g + h) ^ sw) * 4;
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + (wave * 32 + r) * 32 + pos);
            f32x4 wf[2];
            wf[0] = *reinterpret_cast<const f32x4*>(Ws + r * 32 + pos);
            wf[1] = *reinterpret_cast<const f32x4*>(Ws + (32 + r) * 32 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        const float* gp = gbase + (kc % nkc) * 32;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds(gp + (size_t)32 * i * K,
                                             (__attribute__((address_space(3))) void*)(smem + (4 * i + wave) * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float s = 0;
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + tid] = s;
}

// V10: double-buffered LDS-DMA - the refill of the other buffer is issued BEFORE the MFMA cluster and waited for after
// it; one barrier per chunk.  2 x 24 KB of LDS per workgroup = 3 workgroups per CU.
__global__ void __launch_bounds__(256, 3) dma2_kernel(float* out, int chunks, const float* big, int rows, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = (BM + BN) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = big[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const int srow = lane >> 3, slot = lane & 7;
    const int tile_row0 = (blockIdx.x * 192) % (rows - 192);
    const float* gbase = big + (size_t)(tile_row0 + 8 * wave + srow) * K + ((slot ^ srow) & 7) * 4;
    const int nkc = K / 32;
    const int sw = r & 7;
    for (int kc = 0; kc < chunks; ++kc) {
        const int cur = kc & 1;
        const float* gp = gbase + ((kc + 1) % nkc) * 32;
        float* nxt = smem + (cur ^ 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds(gp + (size_t)32 * i * K,
                                             (__attribute__((address_space(3))) void*)(nxt + (4 * i + wave) * 256), 16, 0, 0);
        const float* As = smem + cur * STAGE;
        const float* Ws = As + BM * 32;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int pos = ((2 * g + h) ^ sw) * 4;
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + (wave * 32 + r) * 32 + pos);
            f32x4 wf[2];
            wf[0] = *reinterpret_cast<const f32x4*>(Ws + r * 32 + pos);
            wf[1] = *reinterpret_cast<const f32x4*>(Ws + (32 + r) * 32 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float s = 0;
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + tid] = s;
}

// V11: V10 with 16-wide K chunks: half the LDS per buffer (2 x 12 KB -> 6 workgroups per CU), one barrier per 16 MFMAs.
// 64-byte rows: chunk c (0..3) of row r sits at position c ^ ((r >> 1) & 3).
__global__ void __launch_bounds__(256, 6) dma16_kernel(float* out, int chunks, const float* big, int rows, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int STAGE = (BM + BN) * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * STAGE; i += 256) smem[i] = big[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const int srow = lane >> 2, slot = lane & 3;   // 16 rows x 4 chunks per instruction
    const int tile_row0 = (blockIdx.x * 192) % (rows - 192);
    const float* gbase = big + (size_t)(tile_row0 + 16 * wave + srow) * K + ((slot ^ (srow >> 1)) & 3) * 4;
    const int nkc = K / 16;
    const int sw = (r >> 1) & 3;
    for (int kc = 0; kc < chunks; ++kc) {
        const int cur = kc & 1;
        const float* gp = gbase + ((kc + 1) % nkc) * 16;
        float* nxt = smem + (cur ^ 1) * STAGE;
#pragma unroll
        for (int i = 0; i < 3; ++i)   // 192 rows = 12 groups of 16 rows, 3 per wave
            __builtin_amdgcn_global_load_lds(gp + (size_t)64 * i * K,
                                             (__attribute__((address_space(3))) void*)(nxt + (4 * i + wave) * 256), 16, 0, 0);
        const float* As = smem + cur * STAGE;
        const float* Ws = As + BM * 16;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int pos = ((2 * g + h) ^ sw) * 4;
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + (wave * 32 + r) * 16 + pos);
            f32x4 wf[2];
            wf[0] = *reinterpret_cast<const f32x4*>(Ws + r * 16 + pos);
            wf[1] = *reinterpret_cast<const f32x4*>(Ws + (32 + r) * 16 + pos);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float s = 0;
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + tid] = s;
}

// V12: register-staged refill with a prefetch distance of TWO chunks (two staging register sets, ping-pong), so a
// load has two MFMA clusters of its own workgroup (plus everybody else's) to land.  4 workgroups per CU.
__global__ void __launch_bounds__(256, 4) pre2_kernel(float* out, int chunks, const float* big, int rows, int K) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;
    float* Ws = smem + BM * LDK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < (BM + BN) * LDK; i += 256) smem[i] = big[(i * 7 + blockIdx.x) & 65535];
    __syncthreads();
    f32x16 acc[2];
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
    const float* Ac = As + (wave * 32 + r) * LDK + 4 * h;
    const float* Wc = Ws + r * LDK + 4 * h;
    const int tile_row0 = (blockIdx.x * 192) % (rows - 192);
    const float* gbase = big + (size_t)(tile_row0 + (tid >> 3)) * K + (tid & 7) * 4;
    const int nkc = K / 32;
    f32x4 sa[6], sb[6];
    auto load = [&](f32x4* st, int kc) {
        const float* gp = gbase + (kc % nkc) * 32;
#pragma unroll
        for (int i = 0; i < 6; ++i) st[i] = *reinterpret_cast<const f32x4*>(gp + (size_t)32 * i * K);
    };
    auto store = [&](const f32x4* st) {
#pragma unroll
        for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(smem + ((tid >> 3) + 32 * i) * LDK + (tid & 7) * 4) = st[i];
    };
    auto cluster = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(Ac + 8 * g);
            f32x4 wf[2];
            wf[0] = *reinterpret_cast<const f32x4*>(Wc + 8 * g);
            wf[1] = *reinterpret_cast<const f32x4*>(Wc + 32 * LDK + 8 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], wf[nt][j], acc[nt], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
    };
    load(sa, 1);
    for (int kc = 0; kc < chunks; kc += 2) {
        load(sb, kc + 2);   // two ahead
        cluster();          // chunk kc (in LDS)
        __syncthreads();
        store(sa);          // chunk kc+1
        __syncthreads();
        load(sa, kc + 3);
        cluster();          // chunk kc+1
        __syncthreads();
        store(sb);          // chunk kc+2
        __syncthreads();
    }
    float s = 0;
    for (int n = 0; n < 2; ++n)
        for (int i = 0; i < 16; ++i) s += acc[n][i];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MINW>
void run_dma(const char* what, float* out, const float* big, int rows, int wgs_per_cu) {
    const int blocks = 256 * wgs_per_cu, chunks = 3000;
    const size_t lds = (size_t)(BM + BN) * 32 * 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(dma_kernel<MINW>, dim3(blocks), dim3(256), lds, 0, out, chunks, big, rows, 384);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * chunks * 32 * 4096.0;
    printf("%-46s %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
}

template <int V>
void run(const char* what, float* out, const float* src, const float* big = nullptr, int rows = 0, int K = 384) {
    const int blocks = 256 * 5, chunks = 3000;
    const size_t lds = (size_t)(BM + BN) * LDK * 4 + (V == 7 ? 24 * 1024 : 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(loop_kernel<V>, dim3(blocks), dim3(256), lds, 0, out, chunks, src, big, rows, K);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 4 * chunks * 32 * 4096.0;
    printf("%-46s %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", what, ms, flops / ms / 1e9, flops / ms / 1e9 / 1.573);
}

int main() {
    float *out, *src;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&src, 65536 * 4 + 64);
    float* h = (float*)malloc(65536 * 4 + 64);
    for (int i = 0; i < 65536 + 16; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(src, h, 65536 * 4 + 64, hipMemcpyHostToDevice);
    run<0>("V0 register operands", out, src);
    run<1>("V1 + 12 ds_read_b128 fragments per chunk", out, src);
    run<2>("V2 + 2 barriers per chunk", out, src);
    run<3>("V3 + 6 ds_write_b128 refill per chunk", out, src);
    run<4>("V4 = V1, accumulators in AGPRs", out, src);
    // V5: + global->register staging of the next chunk, from a matrix of `rows` x 384 floats
    for (int rows : {4096, 26112, 200000}) {
        float* big;
        hipMalloc(&big, (size_t)rows * 384 * 4);
        hipMemset(big, 0, (size_t)rows * 384 * 4);
        char name[96];
        snprintf(name, sizeof name, "V5 + 6 global_load_dwordx4 / chunk, %d-row matrix (%.0f MB)", rows, rows * 384 * 4 / 1e6);
        run<5>(name, out, src, big, rows, 384);
        hipFree(big);
    }
    float* big;
    hipMalloc(&big, (size_t)26112 * 384 * 4);
    hipMemset(big, 0, (size_t)26112 * 384 * 4);
    run<6>("V6 = V5 with 3 loads per chunk (half the bytes)", out, src, big, 26112, 384);
    run<7>("V7 = V5 as LDS-DMA (global_load_lds_dwordx4)", out, src, big, 26112, 384);
    run<8>("V8 = V5 without s_setprio(1) around the MFMAs", out, src, big, 26112, 384);
    run_dma<5>("V9 DMA after cluster, swizzled tiles, 5 WG/CU", out, big, 26112, 5);
    run_dma<6>("V9 DMA after cluster, swizzled tiles, 6 WG/CU", out, big, 26112, 6);
    {
        const int blocks = 256 * 3, chunks = 3000;
        const size_t lds = (size_t)2 * (BM + BN) * 32 * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(dma2_kernel, dim3(blocks), dim3(256), lds, 0, out, chunks, big, 26112, 384);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 4 * chunks * 32 * 4096.0;
        printf("%-46s %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", "V10 double-buffered DMA, swizzled, 3 WG/CU", ms,
               flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    }
    for (int wpc : {3, 4}) {
        const int blocks = 256 * wpc, chunks = 3000;
        const size_t lds = (size_t)(BM + BN) * LDK * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(pre2_kernel, dim3(blocks), dim3(256), lds, 0, out, chunks, big, 26112, 384);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 4 * chunks * 32 * 4096.0;
        printf("V12 register staging, prefetch distance 2, %d WG/CU  %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", wpc, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    }
    for (int wpc : {4, 6}) {
        const int blocks = 256 * wpc, chunks = 6000;
        const size_t lds = (size_t)2 * (BM + BN) * 16 * 4;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(dma16_kernel, dim3(blocks), dim3(256), lds, 0, out, chunks, big, 26112, 384);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double flops = (double)blocks * 4 * chunks * 16 * 4096.0;
        printf("V11 double-buffered DMA, 16-wide chunks, %d WG/CU  %.2f ms  %.1f TFLOP/s (%.0f %% of 157.3)\n", wpc, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 1.573);
    }
    return 0;
}
