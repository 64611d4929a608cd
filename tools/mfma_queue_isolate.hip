// mfma_queue_isolate.hip - torch-free isolation of the multi-queue hazard of the bf16-MFMA kernels
// (profiles/r02_bf16_mfma_concurrency.md).  A "victim" kernel runs on stream A while a "partner"
// kernel runs on stream B (another hardware queue); the victim's output is compared bit for bit with its own result from
// a run with the partner absent.  Nothing is shared between the two but the GPU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_queue_isolate.hip -o tools/bin/mfma_queue_isolate
//   ./tools/bin/mfma_queue_isolate [rounds]
// Victims:   V0 production split-precision plain GEMM (gemm_kernel<4,1,4,..,BF16=2>, body qkv shape)
//            V1 the same kernel built on rounded-bf16 operands (BF16=1)
//            V2 bare v_mfma_f32_32x32x16_bf16 loop on register operands (no LDS, no global loads in the loop)
//            V3 bare v_mfma_f32_16x16x32_bf16 loop
//            V4 fp32 plain GEMM (v_mfma_f32_32x32x2_f32) - the control that never failed
// Partners:  none, attention (v_mfma_f32_16x16x4_f32 + LDS), VALU spin, LDS spin, streaming copy, fp32-MFMA register loop,
//            bf16-MFMA register loop, a second split GEMM on its own buffers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void diff_count(const uint32_t* a, const uint32_t* b, size_t n, unsigned* cnt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned bad = 0;
    for (; i < n; i += (size_t)gridDim.x * 256) bad += a[i] != b[i];
    if (bad) atomicAdd(cnt, bad);
}

// ---- bare MFMA victims / partners: operands from registers, deterministic output per lane
template <int SHAPE>  // 0: 32x32x16 bf16, 1: 16x16x32 bf16, 2: 32x32x2 f32, 3: 16x16x4 f32
__global__ void __launch_bounds__(256) mfma_loop(float* out, const uint32_t* rnd, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    u32x4 a, b;
    for (int i = 0; i < 4; ++i) a[i] = rnd[(tid * 8 + i) & 16383], b[i] = rnd[(tid * 8 + 4 + i) & 16383];
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        f32x16 acc[2] = {};
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, a), acc[1], 0, 0, 0);
            if ((it & 63) == 63) for (int i = 0; i < 16; ++i) acc[0][i] *= 0.015625f, acc[1][i] *= 0.015625f;
        }
        for (int i = 0; i < 16; ++i) s += acc[0][i] - acc[1][i];
    } else if constexpr (SHAPE == 1) {
        f32x4 acc[4] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (q & 1) ? a : b), __builtin_bit_cast(bf16x8, (q & 2) ? a : b), acc[q], 0, 0, 0);
            if ((it & 63) == 63) for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) acc[q][i] *= 0.015625f;
        }
        for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) s += acc[q][i] * (float)(q + 1);
    } else if constexpr (SHAPE == 2) {
        f32x16 acc[2] = {};
        const float fa = (float)(a[0] & 1023) * (1.f / 1024.f), fb = (float)(b[0] & 1023) * (1.f / 1024.f);
        for (int it = 0; it < iters; ++it) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb, fa, acc[1], 0, 0, 0);
            if ((it & 63) == 63) for (int i = 0; i < 16; ++i) acc[0][i] *= 0.015625f, acc[1][i] *= 0.015625f;
        }
        for (int i = 0; i < 16; ++i) s += acc[0][i] - acc[1][i];
    } else {
        f32x4 acc[4] = {};
        const float fa = (float)(a[0] & 1023) * (1.f / 1024.f), fb = (float)(b[0] & 1023) * (1.f / 1024.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32((q & 1) ? fa : fb, (q & 2) ? fa : fb, acc[q], 0, 0, 0);
            if ((it & 63) == 63) for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) acc[q][i] *= 0.015625f;
        }
        for (int q = 0; q < 4; ++q) for (int i = 0; i < 4; ++i) s += acc[q][i] * (float)(q + 1);
    }
    out[tid] = s;
}

__global__ void __launch_bounds__(256) valu_spin(float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float x = (float)tid * 1e-3f, y = 0.5f;
    for (int it = 0; it < iters; ++it) x = x * 0.999f + y, y = y * 1.0001f - x * 1e-4f;
    out[tid] = x + y;
}

__global__ void __launch_bounds__(256) lds_spin(float* out, int iters) {
    __shared__ float buf[8192];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 256) buf[i] = (float)i;
    __syncthreads();
    float s = 0.f;
    for (int it = 0; it < iters; ++it) {
        s += buf[(tid * 4 + it * 37) & 8191];
        buf[(tid * 4 + it * 41 + 1) & 8191] = s;
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = s;
}

__global__ void __launch_bounds__(256) stream_copy(const f32x4* src, f32x4* dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

struct Bufs {
    float *A, *W, *bias, *out, *ref, *A2, *out2, *qkv, *o, *scratch, *copy_src, *copy_dst, *loop_out, *loop_ref, *pl_out;
    uint8_t *Ws, *Ws2;
    uint32_t* rnd;
    unsigned* cnt;
};

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    // victim rows: 25920 fills every CU with victim workgroups (partners only overlap at its edges); 5184 (the P = 8
    // denoiser of tests/cabi/queue_concurrency.c: 369 workgroups) leaves room for the partner's waves on the SAME CUs
    const int64_t M = argc > 2 ? atoll(argv[2]) : 25920;
    const int64_t MP = 25920;   // partner kernels keep their size
    const int N = 1152, K = 384;
    Bufs b{};
    CK(hipMalloc(&b.A, MP * K * 4)); CK(hipMalloc(&b.A2, MP * K * 4)); CK(hipMalloc(&b.W, (size_t)N * K * 4));
    CK(hipMalloc(&b.bias, N * 4)); CK(hipMalloc(&b.out, MP * N * 4)); CK(hipMalloc(&b.ref, MP * N * 4)); CK(hipMalloc(&b.out2, MP * N * 4));
    CK(hipMalloc(&b.Ws, (size_t)N * K * 6)); CK(hipMalloc(&b.Ws2, (size_t)N * K * 6));
    CK(hipMalloc(&b.qkv, MP * N * 4)); CK(hipMalloc(&b.o, MP * K * 4));
    CK(hipMalloc(&b.scratch, 1 << 22)); CK(hipMalloc(&b.copy_src, 64 << 20)); CK(hipMalloc(&b.copy_dst, 64 << 20));
    CK(hipMalloc(&b.loop_out, 1 << 22)); CK(hipMalloc(&b.loop_ref, 1 << 22)); CK(hipMalloc(&b.pl_out, 1 << 22));
    CK(hipMalloc(&b.rnd, 16384 * 4)); CK(hipMalloc(&b.cnt, 4));
    {
        std::vector<float> h((size_t)MP * N);
        srand(1);
        for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
        CK(hipMemcpy(b.A, h.data(), MP * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.A2, h.data() + 12345, MP * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.W, h.data() + 777, (size_t)N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.bias, h.data() + 99, N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(b.qkv, h.data(), MP * N * 4, hipMemcpyHostToDevice));
        std::vector<uint32_t> r(16384);
        auto bf = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u >> 16; };
        for (auto& v : r) v = bf((float)rand() / RAND_MAX * 2.f - 1.f) | (bf((float)rand() / RAND_MAX * 2.f - 1.f) << 16);
        CK(hipMemcpy(b.rnd, r.data(), 16384 * 4, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, b.W, b.Ws, N, K);
    hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, b.W, b.Ws2, N, K);
    CK(hipDeviceSynchronize());
    hipStream_t sA, sB, sC;
    CK(hipStreamCreateWithFlags(&sA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sB, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));

    GemmParams g{};
    g.A = b.A, g.W = b.W, g.bias = b.bias, g.out = b.out, g.M = M, g.N = N, g.K = K, g.Wsplit = b.Ws;
    GemmParams g2 = g;
    g2.A = b.A2, g2.out = b.out2, g2.Wsplit = b.Ws2, g2.M = MP;
    AttnParams at{};  // body spatial attention on its own qkv buffer
    at.qkv = b.qkv, at.o = b.o, at.nseq = MP / 24, at.L = 24, at.C = 384, at.heads = 8, at.d = 48;
    at.group = 1, at.group_stride = 24, at.seq_stride = 0, at.tok_stride = 1, at.scale = 0.144f;

    using T44 = GemmTile<4, 1, 4>;
    using T42 = GemmTile<4, 1, 2>;
    const int LOOP_BLOCKS = M >= 25920 ? 2048 : 384;
    struct Victim { const char* name; int id; };
    const Victim victims[] = {{"V0 split GEMM gemm_kernel<4,1,4,BF16=2>", 0}, {"V1 rounded-bf16 GEMM gemm_kernel<4,1,2,BF16=1>", 1},
                              {"V2 bare 32x32x16 bf16 loop", 2}, {"V3 bare 16x16x32 bf16 loop", 3}, {"V4 fp32 GEMM gemm_kernel<4,1,2>", 4}};
    auto launch_victim = [&](int id, hipStream_t s, bool to_ref) {
        GemmParams p = g;
        if (to_ref) p.out = b.ref;
        switch (id) {
            case 0: p.bf16 = 2; hipLaunchKernelGGL((gemm_kernel<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>), dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256), T44::STAGE_FLOATS_SPLIT * 4, s, p); break;
            case 1: p.bf16 = 1; hipLaunchKernelGGL((gemm_kernel<4, 1, 2, EPI_BIAS, 1, 5, 0, 1>), dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), T42::STAGE_FLOATS * 4, s, p); break;
            case 2: hipLaunchKernelGGL(mfma_loop<0>, dim3(LOOP_BLOCKS), dim3(256), 0, s, to_ref ? b.loop_ref : b.loop_out, b.rnd, 6000); break;
            case 3: hipLaunchKernelGGL(mfma_loop<1>, dim3(LOOP_BLOCKS), dim3(256), 0, s, to_ref ? b.loop_ref : b.loop_out, b.rnd, 12000); break;
            case 4: p.bf16 = 0; hipLaunchKernelGGL((gemm_kernel<4, 1, 2, EPI_BIAS, 1, 5>), dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), T42::STAGE_FLOATS * 4, s, p); break;
        }
    };
    struct Partner { const char* name; int id; };
    const Partner partners[] = {{"none", 0}, {"attention attn_kernel<32,48,4> (f32 16x16x4 MFMA)", 1}, {"VALU spin", 2}, {"LDS spin", 3},
                                {"streaming copy", 4}, {"bare f32 32x32x2 MFMA loop", 5}, {"bare f32 16x16x4 MFMA loop", 6},
                                {"bare bf16 32x32x16 MFMA loop", 7}, {"second split GEMM (own buffers)", 8},
                                // three hardware queues, the mix of one denoiser chain: attention on B, another kernel on C
                                {"3 queues: attention + second split GEMM", 9}, {"3 queues: attention + VALU spin", 10},
                                {"3 queues: attention + streaming copy", 11}};
    auto launch_partner = [&](int id, hipStream_t s) {
        switch (id) {
            case 1: {
                constexpr int ITEMS = 4 / 2;
                const int64_t nitems = at.nseq * at.heads;
                hipLaunchKernelGGL((attn_kernel<32, 48, 4>), dim3((unsigned)((nitems + ITEMS - 1) / ITEMS)), dim3(256), (size_t)2 * ITEMS * 32 * 52 * 4, s, at);
                break;
            }
            case 2: hipLaunchKernelGGL(valu_spin, dim3(4096), dim3(256), 0, s, b.scratch, 4000); break;
            case 3: hipLaunchKernelGGL(lds_spin, dim3(2048), dim3(256), 0, s, b.scratch, 600); break;
            case 4: hipLaunchKernelGGL(stream_copy, dim3(2048), dim3(256), 0, s, (const f32x4*)b.copy_src, (f32x4*)b.copy_dst, (size_t)(64 << 20) / 16); break;
            case 5: hipLaunchKernelGGL(mfma_loop<2>, dim3(LOOP_BLOCKS), dim3(256), 0, s, b.pl_out, b.rnd, 1500); break;
            case 6: hipLaunchKernelGGL(mfma_loop<3>, dim3(LOOP_BLOCKS), dim3(256), 0, s, b.pl_out, b.rnd, 1500); break;
            case 7: hipLaunchKernelGGL(mfma_loop<0>, dim3(LOOP_BLOCKS), dim3(256), 0, s, b.pl_out, b.rnd, 6000); break;
            case 8: { GemmParams p = g2; p.bf16 = 2; hipLaunchKernelGGL((gemm_kernel<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>), dim3((unsigned)((MP + 127) / 128 * (N / 128))), dim3(256), T44::STAGE_FLOATS_SPLIT * 4, s, p); break; }
            default: break;
        }
    };
    printf("victim rows M = %ld; ", (long)M);
    printf("rounds per cell: %d   (a cell = launches of the victim whose output differs from its solo result / words that differ)\n", rounds);
    for (const Victim& v : victims) {
        launch_victim(v.id, sA, true);
        CK(hipStreamSynchronize(sA));
        const bool gemm = v.id == 0 || v.id == 1 || v.id == 4;
        const uint32_t* ref = (const uint32_t*)(gemm ? b.ref : b.loop_ref);
        const uint32_t* out = (const uint32_t*)(gemm ? b.out : b.loop_out);
        const size_t words = gemm ? (size_t)M * N : (size_t)LOOP_BLOCKS * 256;
        for (const Partner& pt : partners) {
            int bad_launches = 0;
            unsigned long long bad_words = 0;
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(b.cnt, 0, 4, sA));
                CK(hipStreamSynchronize(sA));
                // partner first (two launches keep queue B busy across the victim's lifetime), victim in the middle
                const int idB = pt.id >= 9 ? 1 : pt.id, idC = pt.id == 9 ? 8 : (pt.id == 10 ? 2 : (pt.id == 11 ? 4 : 0));
                launch_partner(idB, sB);
                launch_partner(idC, sC);
                launch_victim(v.id, sA, false);
                launch_partner(idB, sB);
                launch_partner(idC, sC);
                launch_partner(idB, sB);
                CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB)); CK(hipStreamSynchronize(sC));
                hipLaunchKernelGGL(diff_count, dim3(1024), dim3(256), 0, sA, out, ref, words, b.cnt);
                unsigned c = 0;
                CK(hipMemcpyAsync(&c, b.cnt, 4, hipMemcpyDeviceToHost, sA));
                CK(hipStreamSynchronize(sA));
                if (c) ++bad_launches, bad_words += c;
            }
            printf("%-48s | partner %-52s : %4d of %d launches differ (%llu words)\n", v.name, pt.name, bad_launches, rounds, bad_words);
            fflush(stdout);
        }
    }
    return 0;
}
