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

// a VALU-only victim with ~60 live registers per lane (the register footprint of embed_kernel): every lane runs a fixed
// chain of fp32 multiply-adds over all of them, so a single corrupted VGPR of a single lane shows in the output
template <int NR>
__global__ void __launch_bounds__(256) valu_victim(float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    float r[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) r[k] = (float)((tid * 31 + k * 17) & 1023) * (1.0f / 1024.0f) - 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < NR; ++k) r[k] = r[k] * 0.9990234375f + r[(k + 1) % NR] * 0.0009765625f;
    }
#pragma unroll
    for (int k = 0; k < NR; ++k) out[(size_t)tid * NR + k] = r[k];
}

#ifndef NO_PK_VICTIMS
// the same victim on PACKED fp32 VALU instructions (what hipcc's SLP vectoriser makes of adjacent scalar float operations
// at -O3): v_pk_fma_f32 on register pairs, nothing else in the loop
__global__ void __launch_bounds__(256) pk_victim(float* out, int iters) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x2 r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = f32x2{(float)((tid * 31 + k * 17) & 1023) * (1.0f / 1024.0f) - 0.5f, (float)((tid * 13 + k * 29) & 1023) * (1.0f / 1024.0f) - 0.5f};
    const f32x2 ca = {0.9990234375f, 0.998046875f}, cb = {0.0009765625f, 0.001953125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x2 t;
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(r[(k + 1) & 15]), "v"(cb));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[(size_t)tid * 32 + 2 * k] = r[k][0], out[(size_t)tid * 32 + 2 * k + 1] = r[k][1];
}

// V9.x: one packed-fp32 form each, to find which encoding is hit (embed_kernel's ISA holds all of them)
//   0 v_pk_fma_f32 plain            1 v_pk_fma_f32 op_sel_hi:[0,1,1] (src0.lo to both halves)
//   2 v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1] (src0.hi to both)      3 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)
//   4 v_pk_add_f32 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] (minus src1.lo)   5 v_pk_mul_f32 plain   6 v_pk_add_f32 plain
//   7 plain v_pk_fma_f32 on operands loaded from memory inside the loop (the embed kernel's shape: load, few packed ops, store)
//   8 v_pk_add_f32 src0 halves swapped   9 v_pk_mul_f32 src1 swapped   10 v_pk_fma_f32 src1 swapped   11 v_pk_fma_f32 src2 swapped
//   12 v_pk_add_f32 src1.hi to both halves   13 v_pk_add_f32 src1.lo to both halves
template <int V>
__global__ void __launch_bounds__(256) pk_form_victim(float* out, const float* src, int iters) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x2 r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = f32x2{(float)((tid * 31 + k * 17) & 1023) * (1.0f / 1024.0f) - 0.5f, (float)((tid * 13 + k * 29) & 1023) * (1.0f / 1024.0f) - 0.5f};
    const f32x2 ca = {0.9990234375f, 0.998046875f}, cb = {0.0009765625f, 0.001953125f}, half = {0.5f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f32x2 t, u;
            if (V == 7) {
                const f32x2 m = *reinterpret_cast<const f32x2*>(src + ((size_t)((tid + it * 64 + k) & 65535)) * 2);
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(m));
                continue;
            }
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(r[(k + 1) & 15]), "v"(cb));
            if (V == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
            if (V == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
            if (V == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
            if (V == 3) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r[k]) : "v"(u), "v"(t)); }
            if (V == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r[k]) : "v"(u), "v"(t)); }
            if (V == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r[k]) : "v"(r[k]), "v"(ca)); r[k][0] += t[0]; r[k][1] += t[1]; }
            if (V == 6) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r[k]) : "v"(u), "v"(t)); }
            if (V == 8) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(r[k]) : "v"(u), "v"(t)); }
            if (V == 9) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r[k]) : "v"(r[k]), "v"(ca)); r[k][0] += t[0]; r[k][1] += t[1]; }
            if (V == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
            if (V == 11) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r[k]) : "v"(r[k]), "v"(ca), "v"(t));
            if (V == 12) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r[k]) : "v"(u), "v"(t)); }
            if (V == 13) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(u) : "v"(r[k]), "v"(half)); asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r[k]) : "v"(u), "v"(t)); }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) out[(size_t)tid * 32 + 2 * k] = r[k][0], out[(size_t)tid * 32 + 2 * k + 1] = r[k][1];
}

#endif  // NO_PK_VICTIMS

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
    float *A, *W, *bias, *out, *ref, *A2, *out2, *qkv, *o, *scratch, *copy_src, *copy_dst, *loop_out, *loop_ref, *pl_out, *vv_out, *vv_ref;
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
    const int VV_BLOCKS = 1024, VV_NR = 56;
    CK(hipMalloc(&b.vv_out, (size_t)VV_BLOCKS * 256 * VV_NR * 4)); CK(hipMalloc(&b.vv_ref, (size_t)VV_BLOCKS * 256 * VV_NR * 4));
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
                              {"V2 bare 32x32x16 bf16 loop", 2}, {"V3 bare 16x16x32 bf16 loop", 3}, {"V4 fp32 GEMM gemm_kernel<4,1,2>", 4},
                              {"V5 VALU-only kernel, 56 live registers per lane", 5}, {"V6 attention attn_kernel<32,48,4>", 6},
                              {"V7 packed-fp32 VALU kernel (v_pk_mul_f32 / v_pk_fma_f32)", 7},
                              {"V8 embed_kernel (VALU only; SLP-packed fp32 unless built -fno-slp-vectorize)", 8},
                              {"V9.0 v_pk_fma_f32 plain", 90}, {"V9.1 v_pk_fma_f32 op_sel_hi:[0,1,1]", 91}, {"V9.2 v_pk_fma_f32 op_sel:[1,0,0] op_sel_hi:[0,1,1]", 92},
                              {"V9.3 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", 93}, {"V9.4 v_pk_add_f32 op_sel_hi:[1,0] neg_lo neg_hi", 94},
                              {"V9.5 v_pk_mul_f32 plain + scalar adds", 95}, {"V9.6 v_pk_add_f32 plain", 96}, {"V9.7 v_pk_fma_f32 on operands loaded in the loop", 97},
                              {"V9.8 v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 swapped)", 98}, {"V9.9 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 swapped)", 99},
                              {"V9.10 v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1] (src1 swapped)", 100}, {"V9.11 v_pk_fma_f32 op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 swapped)", 101},
                              {"V9.12 v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1] (src1.hi twice)", 102}, {"V9.13 v_pk_add_f32 op_sel:[0,0] op_sel_hi:[1,0] (src1.lo twice)", 103}};
    const int only = argc > 3 ? atoi(argv[3]) : -1;   // run one victim only
    EmbedParams em{};  // the first kernel of a denoiser pass: patch embedding + positions + time embedding + LayerNorm of block 0
    em.x3d = b.A, em.x2d = b.A2, em.pw = b.W, em.pb = b.bias, em.pos = b.W + 4000, em.temb = b.W + 20000, em.n_w = b.bias + 384, em.n_b = b.bias + 768;
    em.n_eps = 1e-6f, em.xn = b.out2, em.B = 1, em.P = (int)(M / (27 * 24)), em.F = 27, em.J = 24, em.J3 = 24, em.C = 384, em.nflip = 1;
    em.do_clamp = 1, em.scale = 1.0f, em.lim = 1.1f, em.row0 = 0, em.nrows = (int64_t)em.P * 27 * 24;
    auto launch_victim = [&](int id, hipStream_t s, bool to_ref) {
        GemmParams p = g;
        if (to_ref) p.out = b.ref;
        switch (id) {
            case 0: p.bf16 = 2; hipLaunchKernelGGL((gemm_kernel<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>), dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256), T44::STAGE_FLOATS_SPLIT * 4, s, p); break;
            case 1: p.bf16 = 1; hipLaunchKernelGGL((gemm_kernel<4, 1, 2, EPI_BIAS, 1, 5, 0, 1>), dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), T42::STAGE_FLOATS * 4, s, p); break;
            case 2: hipLaunchKernelGGL(mfma_loop<0>, dim3(LOOP_BLOCKS), dim3(256), 0, s, to_ref ? b.loop_ref : b.loop_out, b.rnd, 6000); break;
            case 3: hipLaunchKernelGGL(mfma_loop<1>, dim3(LOOP_BLOCKS), dim3(256), 0, s, to_ref ? b.loop_ref : b.loop_out, b.rnd, 12000); break;
            case 4: p.bf16 = 0; hipLaunchKernelGGL((gemm_kernel<4, 1, 2, EPI_BIAS, 1, 5>), dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), T42::STAGE_FLOATS * 4, s, p); break;
            case 5: hipLaunchKernelGGL(valu_victim<56>, dim3(VV_BLOCKS), dim3(256), 0, s, to_ref ? b.vv_ref : b.vv_out, 400); break;
#ifndef NO_PK_VICTIMS
            case 7: hipLaunchKernelGGL(pk_victim, dim3(VV_BLOCKS), dim3(256), 0, s, to_ref ? b.vv_ref : b.vv_out, 2000); break;
#define PKF(n) case 90 + n: hipLaunchKernelGGL(pk_form_victim<n>, dim3(VV_BLOCKS), dim3(256), 0, s, to_ref ? b.vv_ref : b.vv_out, b.A, n == 7 ? 400 : 1000); break;
            PKF(0) PKF(1) PKF(2) PKF(3) PKF(4) PKF(5) PKF(6) PKF(7) PKF(8) PKF(9) PKF(10) PKF(11) PKF(12) PKF(13)
#undef PKF
#endif
            case 8: {
                EmbedParams e2 = em;
                e2.x = to_ref ? b.ref : b.out;
                hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((e2.nrows + EMBED_ROWS_PER_BLOCK - 1) / EMBED_ROWS_PER_BLOCK)), dim3(256), 0, s, e2);
                break;
            }
            case 6: {
                AttnParams a2 = at;
                a2.o = to_ref ? b.ref : b.out;
                constexpr int ITEMS = 4 / 2;
                const int64_t nitems = a2.nseq * a2.heads;
                hipLaunchKernelGGL((attn_kernel<32, 48, 4>), dim3((unsigned)((nitems + ITEMS - 1) / ITEMS)), dim3(256), (size_t)2 * ITEMS * 32 * 52 * 4, s, a2);
                break;
            }
        }
    };
    struct Partner { const char* name; int id; };
    const Partner partners[] = {{"none", 0}, {"attention attn_kernel<32,48,4> (f32 16x16x4 MFMA)", 1}, {"VALU spin", 2}, {"LDS spin", 3},
                                {"streaming copy", 4}, {"bare f32 32x32x2 MFMA loop", 5}, {"bare f32 16x16x4 MFMA loop", 6},
                                {"bare bf16 32x32x16 MFMA loop", 7}, {"second split GEMM (own buffers)", 8},
                                // three hardware queues, the mix of one denoiser chain: attention on B, another kernel on C
                                {"3 queues: attention + second split GEMM", 9}, {"3 queues: attention + VALU spin", 10},
                                {"3 queues: attention + streaming copy", 11},
                                // MANY kernel boundaries on the other queue(s) while the victim runs: every kernel start is an
                                // acquire (cache invalidate) issued by the command processor
                                {"burst of 60 tiny kernels on queue B", 12}, {"bursts of 60 tiny kernels on queues B and C", 13},
                                {"bare bf16 16x16x32 MFMA loop", 14}};
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
            case 14: hipLaunchKernelGGL(mfma_loop<1>, dim3(LOOP_BLOCKS), dim3(256), 0, s, b.pl_out, b.rnd, 12000); break;
            case 8: { GemmParams p = g2; p.bf16 = 2; hipLaunchKernelGGL((gemm_kernel<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>), dim3((unsigned)((MP + 127) / 128 * (N / 128))), dim3(256), T44::STAGE_FLOATS_SPLIT * 4, s, p); break; }
            default: break;
        }
    };
    printf("victim rows M = %ld; ", (long)M);
    printf("rounds per cell: %d   (a cell = launches of the victim whose output differs from its solo result / words that differ)\n", rounds);
    for (const Victim& v : victims) {
        if (only >= 0 && v.id != only && !(only == 9 && v.id >= 90)) continue;
        launch_victim(v.id, sA, true);
        CK(hipStreamSynchronize(sA));
        const bool gemm = v.id == 0 || v.id == 1 || v.id == 4;
        const uint32_t* ref = (const uint32_t*)(gemm || v.id == 6 || v.id == 8 ? b.ref : (v.id == 5 || v.id == 7 || v.id >= 90 ? b.vv_ref : b.loop_ref));
        const uint32_t* out = (const uint32_t*)(gemm || v.id == 6 || v.id == 8 ? b.out : (v.id == 5 || v.id == 7 || v.id >= 90 ? b.vv_out : b.loop_out));
        const size_t words = gemm ? (size_t)M * N : v.id == 8 ? (size_t)em.nrows * 384 : (v.id == 7 || v.id >= 90 ? (size_t)VV_BLOCKS * 256 * 32 : v.id == 5 ? (size_t)VV_BLOCKS * 256 * VV_NR : (v.id == 6 ? (size_t)MP * 384 : (size_t)LOOP_BLOCKS * 256));
        for (const Partner& pt : partners) {
            int bad_launches = 0;
            unsigned long long bad_words = 0;
            for (int r = 0; r < rounds; ++r) {
                CK(hipMemsetAsync(b.cnt, 0, 4, sA));
                CK(hipStreamSynchronize(sA));
                // partner first (two launches keep queue B busy across the victim's lifetime), victim in the middle
                if (pt.id >= 12) {
                    for (int q = 0; q < 20; ++q) {
                        hipLaunchKernelGGL(valu_spin, dim3(64), dim3(256), 0, sB, b.scratch, 200);
                        if (pt.id == 13) hipLaunchKernelGGL(valu_spin, dim3(64), dim3(256), 0, sC, b.scratch + 65536, 200);
                    }
                    launch_victim(v.id, sA, false);
                    for (int q = 0; q < 40; ++q) {
                        hipLaunchKernelGGL(valu_spin, dim3(64), dim3(256), 0, sB, b.scratch, 200);
                        if (pt.id == 13) hipLaunchKernelGGL(valu_spin, dim3(64), dim3(256), 0, sC, b.scratch + 65536, 200);
                    }
                    CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB)); CK(hipStreamSynchronize(sC));
                    goto check;
                }
                {
                const int idB = pt.id == 14 ? 14 : (pt.id >= 9 ? 1 : pt.id), idC = pt.id == 9 ? 8 : (pt.id == 10 ? 2 : (pt.id == 11 ? 4 : 0));
                launch_partner(idB, sB);
                launch_partner(idC, sC);
                launch_victim(v.id, sA, false);
                launch_partner(idB, sB);
                launch_partner(idC, sC);
                launch_partner(idB, sB);
                }
                CK(hipStreamSynchronize(sA)); CK(hipStreamSynchronize(sB)); CK(hipStreamSynchronize(sC));
            check:
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
