// queue_chain_isolate.hip - what goes wrong when the bf16-MFMA chains run on several hardware queues: does a kernel
// COMPUTE a wrong tile, or does the next kernel of its chain READ data the previous one has not made visible?
// (tests/cabi/queue_concurrency.c reproduces the failure with the whole denoiser, tools/mfma_queue_isolate.hip
// shows that independent kernels never disturb each other.)
//
// NQ independent chains (own buffers), one per HIP stream, each round:
//     hipMemsetAsync(Y, 0xff)  ->  K1: Y = A W1'^T + b   ->  K2: Z = Y W2'^T + b        (plain linear layers, 128 x 128 tiles)
// Y is poisoned with NaN bit patterns before K1 rewrites it, and the chain's input alternates between two tensors from round
// to round, so a K2 that reads a stale line of Y sees either the poison (NaN in Z) or the PREVIOUS round's Y (another input:
// finite but wrong); a K2 (or K1) that computes wrongly also leaves finite wrong numbers - told apart through Y itself.  After the round, Y and Z are compared bit for bit with the
// single-stream results:   Y wrong -> K1 produced / published a wrong tile;   Y right, Z wrong with NaN -> K2 read stale
// Y (a visibility problem at the kernel boundary);   Y right, Z wrong and finite -> K2 computed wrongly from right inputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/queue_chain_isolate.hip -o tools/bin/queue_chain_isolate
//   ./tools/bin/queue_chain_isolate [rounds] [queues] [rows]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../pafuse_amd/csrc/kernels.hpp"
using namespace pafuse;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Chain {
    float *A, *A2, *W1, *W2, *b, *Y, *Z, *Yref, *Zref, *Yref2, *Zref2;
    uint8_t *W1s, *W2s;
    hipStream_t s;
};

template <int MODE>   // 2: split-precision products (v_mfma_f32_32x32x16_bf16), 0: fp32 matrix cores
static void launch(const float* A, const float* W, const uint8_t* Ws, const float* b, float* out, int64_t M, int N, int K, hipStream_t s) {
    GemmParams p{};
    p.A = A, p.W = W, p.Wsplit = Ws, p.bias = b, p.out = out, p.M = M, p.N = N, p.K = K, p.bf16 = MODE;
    if (MODE == 2) {
        using T = GemmTile<4, 1, 4>;
        hipLaunchKernelGGL((gemm_kernel<4, 1, 4, EPI_BIAS, 1, 2, 0, 2>), dim3((unsigned)((M + 127) / 128 * (N / 128))), dim3(256),
                           T::STAGE_FLOATS_SPLIT * 4, s, p);
    } else {
        using T = GemmTile<4, 1, 2>;
        hipLaunchKernelGGL((gemm_kernel<4, 1, 2, EPI_BIAS, 1, 5>), dim3((unsigned)((M + 127) / 128 * (N / 64))), dim3(256), T::STAGE_FLOATS * 4, s, p);
    }
}

template <int M1, int M2>
static void run(const char* tag, std::vector<Chain>& ch, int64_t M, int rounds, int nq) {
    const int N = 384, K = 384;
    const size_t n = (size_t)M * N;
    std::vector<float> hy(n), hz(n), ry(n), rz(n);
    // single-stream references
    for (auto& c : ch) {
        launch<M1>(c.A, c.W1, c.W1s, c.b, c.Yref, M, N, K, ch[0].s);
        launch<M2>(c.Yref, c.W2, c.W2s, c.b, c.Zref, M, N, K, ch[0].s);
        launch<M1>(c.A2, c.W1, c.W1s, c.b, c.Yref2, M, N, K, ch[0].s);
        launch<M2>(c.Yref2, c.W2, c.W2s, c.b, c.Zref2, M, N, K, ch[0].s);
        CK(hipStreamSynchronize(ch[0].s));
    }
    long y_bad = 0, z_stale = 0, z_finite_wrong = 0, rounds_bad = 0, total = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < (int)ch.size(); ++i) {
            Chain& c = ch[i];
            hipStream_t s = ch[i % nq].s;
            CK(hipMemsetAsync(c.Y, 0xff, n * 4, s));
            launch<M1>((r & 1) ? c.A2 : c.A, c.W1, c.W1s, c.b, c.Y, M, N, K, s);
            launch<M2>(c.Y, c.W2, c.W2s, c.b, c.Z, M, N, K, s);
        }
        CK(hipDeviceSynchronize());
        for (auto& c : ch) {
            CK(hipMemcpy(hy.data(), c.Y, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ry.data(), (r & 1) ? c.Yref2 : c.Yref, n * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hz.data(), c.Z, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rz.data(), (r & 1) ? c.Zref2 : c.Zref, n * 4, hipMemcpyDeviceToHost));
            const bool yb = memcmp(hy.data(), ry.data(), n * 4) != 0, zb = memcmp(hz.data(), rz.data(), n * 4) != 0;
            ++total;
            if (yb) ++y_bad;
            if (zb && !yb) {
                bool nan = false;
                for (size_t k = 0; k < n && !nan; ++k) nan = hz[k] != rz[k] && !std::isfinite(hz[k]);
                if (nan) ++z_stale; else ++z_finite_wrong;
            }
            if (yb || zb) ++rounds_bad;
        }
    }
    printf("%-34s %d queue(s), rows %ld: %ld of %ld chains wrong | Y (K1 output) wrong %ld | Y right, Z has NaN (K2 read the poison) %ld | "
           "Y right, Z finite but wrong (K2 read last round's Y, or computed wrongly) %ld\n", tag, nq, (long)M, rounds_bad, total, y_bad, z_stale, z_finite_wrong);
    fflush(stdout);
}

// ---- chain B: K1 = qkv GEMM (N = 1152) -> K2 = attention (body spatial, 8 heads x 48) -> O [M, 384]
struct ChainB {
    float *A, *A2, *W, *b, *Y, *O, *Yref, *Oref, *Yref2, *Oref2;
    uint8_t* Ws;
};
static void launch_attn(const float* qkv, float* o, int64_t M, hipStream_t s) {
    AttnParams at{};
    at.qkv = qkv, at.o = o, at.nseq = M / 24, at.L = 24, at.C = 384, at.heads = 8, at.d = 48;
    at.group = 1, at.group_stride = 24, at.seq_stride = 0, at.tok_stride = 1, at.scale = 0.1443375673f;
    constexpr int ITEMS = 4 / 2;
    const int64_t nitems = at.nseq * at.heads;
    hipLaunchKernelGGL((attn_kernel<32, 48, 4>), dim3((unsigned)((nitems + ITEMS - 1) / ITEMS)), dim3(256), (size_t)2 * ITEMS * 32 * 52 * 4, s, at);
}
template <int M1>
static void runB(const char* tag, std::vector<ChainB>& ch, std::vector<hipStream_t>& st, int64_t M, int rounds, int nq) {
    const int N = 1152, K = 384;
    const size_t ny = (size_t)M * N, no = (size_t)M * 384;
    std::vector<float> hy(ny), ry(ny), ho(no), ro(no);
    for (auto& c : ch) {
        launch<M1>(c.A, c.W, c.Ws, c.b, c.Yref, M, N, K, st[0]); launch_attn(c.Yref, c.Oref, M, st[0]);
        launch<M1>(c.A2, c.W, c.Ws, c.b, c.Yref2, M, N, K, st[0]); launch_attn(c.Yref2, c.Oref2, M, st[0]);
        CK(hipStreamSynchronize(st[0]));
    }
    long y_bad = 0, o_nan = 0, o_wrong = 0, bad = 0, total = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < (int)ch.size(); ++i) {
            ChainB& c = ch[i];
            hipStream_t s = st[i % nq];
            CK(hipMemsetAsync(c.Y, 0xff, ny * 4, s));
            launch<M1>((r & 1) ? c.A2 : c.A, c.W, c.Ws, c.b, c.Y, M, N, K, s);
            launch_attn(c.Y, c.O, M, s);
        }
        CK(hipDeviceSynchronize());
        for (auto& c : ch) {
            CK(hipMemcpy(hy.data(), c.Y, ny * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ry.data(), (r & 1) ? c.Yref2 : c.Yref, ny * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ho.data(), c.O, no * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ro.data(), (r & 1) ? c.Oref2 : c.Oref, no * 4, hipMemcpyDeviceToHost));
            const bool yb = memcmp(hy.data(), ry.data(), ny * 4) != 0, ob = memcmp(ho.data(), ro.data(), no * 4) != 0;
            ++total;
            if (yb) ++y_bad;
            if (ob && !yb) {
                bool nan = false;
                for (size_t k = 0; k < no && !nan; ++k) nan = ho[k] != ro[k] && !std::isfinite(ho[k]);
                if (nan) ++o_nan; else ++o_wrong;
            }
            if (yb || ob) ++bad;
        }
    }
    printf("%-34s %d queue(s), rows %ld: %ld of %ld chains wrong | qkv (K1 output) wrong %ld | qkv right, o has NaN (attention read the poison) %ld | "
           "qkv right, o finite but wrong %ld\n", tag, nq, (long)M, bad, total, y_bad, o_nan, o_wrong);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100;
    const int nq = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t M = argc > 3 ? atoll(argv[3]) : 5184;
    const int N = 384, K = 384;
    std::vector<Chain> ch(3);
    std::vector<float> h((size_t)M * K + (size_t)N * K * 2 + N + 8192);
    srand(7);
    for (auto& c : ch) {
        for (auto& v : h) v = ((float)rand() / (float)RAND_MAX - 0.5f);
        CK(hipMalloc(&c.A, M * K * 4)); CK(hipMalloc(&c.A2, M * K * 4)); CK(hipMalloc(&c.Yref2, M * N * 4)); CK(hipMalloc(&c.Zref2, M * N * 4)); CK(hipMalloc(&c.W1, (size_t)N * K * 4)); CK(hipMalloc(&c.W2, (size_t)N * K * 4)); CK(hipMalloc(&c.b, N * 4));
        CK(hipMalloc(&c.Y, M * N * 4)); CK(hipMalloc(&c.Z, M * N * 4)); CK(hipMalloc(&c.Yref, M * N * 4)); CK(hipMalloc(&c.Zref, M * N * 4));
        CK(hipMalloc(&c.W1s, (size_t)N * K * 6)); CK(hipMalloc(&c.W2s, (size_t)N * K * 6));
        CK(hipMemcpy(c.A, h.data(), M * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(c.A2, h.data() + 4321, M * K * 4, hipMemcpyHostToDevice));
        for (size_t k = 0; k < (size_t)N * K * 2; ++k) h[M * K + k] *= 0.1f;    // keep Z of order one
        CK(hipMemcpy(c.W1, h.data() + M * K, (size_t)N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(c.W2, h.data() + M * K + (size_t)N * K, (size_t)N * K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(c.b, h.data() + M * K + (size_t)N * K * 2, N * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, c.W1, c.W1s, N, K);
        hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)N * (K / 8) + 255) / 256)), dim3(256), 0, 0, c.W2, c.W2s, N, K);
        CK(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
    }
    CK(hipDeviceSynchronize());
    run<2, 2>("K1 bf16x3 -> K2 bf16x3", ch, M, rounds, nq);
    run<2, 2>("K1 bf16x3 -> K2 bf16x3 (one queue)", ch, M, rounds, 1);
    run<0, 0>("K1 f32    -> K2 f32", ch, M, rounds, nq);
    run<2, 0>("K1 bf16x3 -> K2 f32", ch, M, rounds, nq);
    run<0, 2>("K1 f32    -> K2 bf16x3", ch, M, rounds, nq);
    {
        std::vector<ChainB> cb(3);
        std::vector<hipStream_t> st;
        for (auto& c : ch) st.push_back(c.s);
        const int NB = 1152;
        std::vector<float> hb((size_t)M * K + (size_t)NB * K + NB + 8192);
        for (auto& c : cb) {
            for (auto& v : hb) v = ((float)rand() / (float)RAND_MAX - 0.5f);
            CK(hipMalloc(&c.A, M * K * 4)); CK(hipMalloc(&c.A2, M * K * 4)); CK(hipMalloc(&c.W, (size_t)NB * K * 4)); CK(hipMalloc(&c.b, NB * 4));
            CK(hipMalloc(&c.Y, M * NB * 4)); CK(hipMalloc(&c.Yref, M * NB * 4)); CK(hipMalloc(&c.Yref2, M * NB * 4));
            CK(hipMalloc(&c.O, M * 384 * 4)); CK(hipMalloc(&c.Oref, M * 384 * 4)); CK(hipMalloc(&c.Oref2, M * 384 * 4));
            CK(hipMalloc(&c.Ws, (size_t)NB * K * 6));
            CK(hipMemcpy(c.A, hb.data(), M * K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(c.A2, hb.data() + 4321, M * K * 4, hipMemcpyHostToDevice));
            for (size_t k = 0; k < (size_t)NB * K; ++k) hb[M * K + k] *= 0.2f;
            CK(hipMemcpy(c.W, hb.data() + M * K, (size_t)NB * K * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(c.b, hb.data() + M * K + (size_t)NB * K, NB * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(split_weights_kernel<32>, dim3((unsigned)(((int64_t)NB * (K / 8) + 255) / 256)), dim3(256), 0, 0, c.W, c.Ws, NB, K);
        }
        CK(hipDeviceSynchronize());
        runB<2>("qkv bf16x3 -> attention", cb, st, M, rounds, nq);
        runB<2>("qkv bf16x3 -> attention (one queue)", cb, st, M, rounds, 1);
        runB<0>("qkv f32    -> attention", cb, st, M, rounds, nq);
    }
    return 0;
}
