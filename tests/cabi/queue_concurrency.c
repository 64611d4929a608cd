/* queue_concurrency.c - torch-free reproducer of the multi-queue hazard of the bf16-MFMA modes (VERDICT r2 item 5,
 * profiles/r02_bf16_mfma_concurrency.md): three independent copies of ONE body-part denoiser (own weights, own split
 * images, own workspaces, own outputs) run pafuse_mixste2_forward at the same time on three HIP streams; every output is
 * compared bit for bit with the same copy's single-stream result.  Plain C + the HIP runtime + the C ABI only.
 *   gcc tests/cabi/queue_concurrency.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lpafuse_amd -lpafuse_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -o tools/bin/queue_concurrency
 *   LD_LIBRARY_PATH=pafuse_amd ./tools/bin/queue_concurrency <mode 0|1|2> [depth] [rounds] [streams]
 * Prints "<bad> of <n> concurrent outputs differ".  Exit code 0 always (a measurement, not a test). */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pafuse_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
#define PK(x) do { int rc_ = (x); if (rc_ < 0) { printf("pafuse error %d line %d: %s\n", rc_, __LINE__, pafuse_last_error()); exit(3); } } while (0)

static uint64_t g_seed = 88172645463325252ull;
static float rnd(void) {   /* xorshift, uniform in [-1, 1) */
    g_seed ^= g_seed << 13, g_seed ^= g_seed >> 7, g_seed ^= g_seed << 17;
    return (float)((g_seed >> 11) & 0xffffff) / 8388608.0f - 1.0f;
}
static float* dev_rand(size_t n, float scale, float offset) {
    float* h = malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) h[i] = offset + scale * rnd();
    float* d;
    CK(hipMalloc((void**)&d, n * sizeof(float)));
    CK(hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice));
    free(h);
    return d;
}
static void* split_of(const float* W, int N, int K, int whole_row) {
    void* img;
    CK(hipMalloc(&img, pafuse_split_weights_bytes(N, K)));
    PK(pafuse_split_weights(W, N, K, whole_row, img, NULL));
    return img;
}

enum { F = 27, J = 24, C = 384, HEADS = 8, P = 8, B = 1 };

static void make_block(pafuse_block_weights* b, int mode) {
    b->norm1_w = dev_rand(C, 0.1f, 1.0f), b->norm1_b = dev_rand(C, 0.1f, 0.0f);
    b->qkv_w = dev_rand((size_t)3 * C * C, 0.05f, 0.0f), b->qkv_b = dev_rand(3 * C, 0.02f, 0.0f);
    b->proj_w = dev_rand((size_t)C * C, 0.05f, 0.0f), b->proj_b = dev_rand(C, 0.02f, 0.0f);
    b->norm2_w = dev_rand(C, 0.1f, 1.0f), b->norm2_b = dev_rand(C, 0.1f, 0.0f);
    b->fc1_w = dev_rand((size_t)2 * C * C, 0.05f, 0.0f), b->fc1_b = dev_rand(2 * C, 0.02f, 0.0f);
    b->fc2_w = dev_rand((size_t)2 * C * C, 0.05f, 0.0f), b->fc2_b = dev_rand(C, 0.02f, 0.0f);
    if (mode == 2) {
        b->qkv_ws = split_of(b->qkv_w, 3 * C, C, 0), b->proj_ws = split_of(b->proj_w, C, C, 1);
        b->fc1_ws = split_of(b->fc1_w, 2 * C, C, 0), b->fc2_ws = split_of(b->fc2_w, C, 2 * C, 1);
    }
}

static void make_model(pafuse_mixste2_weights* w, int mode, int depth) {
    memset(w, 0, sizeof *w);
    w->frames = F, w->joints = J, w->channels = C, w->depth = depth, w->heads = HEADS, w->in_chans = 5;
    w->operand_bf16 = mode, w->mlp_hidden = 0, w->qk_scale = 0.0f;
    w->patch_w = dev_rand(C * 5, 0.3f, 0.0f), w->patch_b = dev_rand(C, 0.02f, 0.0f);
    w->pos_spatial = dev_rand((size_t)J * C, 0.02f, 0.0f), w->pos_temporal = dev_rand((size_t)F * C, 0.02f, 0.0f);
    w->tm1_w = dev_rand((size_t)2 * C * C, 0.05f, 0.0f), w->tm1_b = dev_rand(2 * C, 0.02f, 0.0f);
    w->tm3_w = dev_rand((size_t)2 * C * C, 0.05f, 0.0f), w->tm3_b = dev_rand(C, 0.02f, 0.0f);
    {   /* sinusoid frequencies exp(-k ln(1e4) / (C/2 - 1)) (common/mixste.py:135-136) */
        float h[C / 2];
        for (int k = 0; k < C / 2; ++k) h[k] = expf(-(float)k * logf(10000.0f) / (float)(C / 2 - 1));
        float* d;
        CK(hipMalloc((void**)&d, sizeof h));
        CK(hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice));
        w->freqs = d;
    }
    w->snorm_w = dev_rand(C, 0.1f, 1.0f), w->snorm_b = dev_rand(C, 0.1f, 0.0f);
    w->tnorm_w = dev_rand(C, 0.1f, 1.0f), w->tnorm_b = dev_rand(C, 0.1f, 0.0f);
    w->hnorm_w = dev_rand(C, 0.1f, 1.0f), w->hnorm_b = dev_rand(C, 0.1f, 0.0f);
    w->head_w = dev_rand(3 * C, 0.05f, 0.0f), w->head_b = dev_rand(3, 0.02f, 0.0f);
    for (int i = 0; i < depth; ++i) make_block(&w->ste[i], mode), make_block(&w->tte[i], mode);
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 2, depth = argc > 2 ? atoi(argv[2]) : 1;
    const int rounds = argc > 3 ? atoi(argv[3]) : 80;
    int ns = argc > 4 ? atoi(argv[4]) : 3;
    if (ns < 1 || ns > 3) ns = 3;
    pafuse_mixste2_weights w[3];
    float *x2d[3], *x3d[3], *out[3], *ref[3];
    void* ws[3];
    int64_t* t;
    const size_t n_out = (size_t)B * P * F * J * 3;
    {
        int64_t ht[B] = {499};
        CK(hipMalloc((void**)&t, sizeof ht));
        CK(hipMemcpy(t, ht, sizeof ht, hipMemcpyHostToDevice));
    }
    size_t nbytes = 0;
    for (int i = 0; i < 3; ++i) {
        make_model(&w[i], mode, depth);
        x2d[i] = dev_rand((size_t)B * F * J * 2, 1.0f, 0.0f), x3d[i] = dev_rand(n_out, 1.0f, 0.0f);
        CK(hipMalloc((void**)&out[i], n_out * 4)); CK(hipMalloc((void**)&ref[i], n_out * 4));
        nbytes = pafuse_mixste2_workspace_bytes(&w[i], B, P);
        CK(hipMalloc(&ws[i], nbytes));
        CK(hipMemset(ws[i], 0xff, nbytes));     /* NaN-poisoned workspace */
    }
    CK(hipDeviceSynchronize());
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking));
    /* single-stream references (twice: they must agree with themselves) */
    float* h0 = malloc(n_out * 4), *h1 = malloc(n_out * 4);
    for (int i = 0; i < 3; ++i) {
        PK(pafuse_mixste2_forward(&w[i], x2d[i], x3d[i], t, B, P, ref[i], ws[i], nbytes, s[0]));
        CK(hipStreamSynchronize(s[0]));
        PK(pafuse_mixste2_forward(&w[i], x2d[i], x3d[i], t, B, P, out[i], ws[i], nbytes, s[0]));
        CK(hipStreamSynchronize(s[0]));
        CK(hipMemcpy(h0, ref[i], n_out * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, out[i], n_out * 4, hipMemcpyDeviceToHost));
        if (memcmp(h0, h1, n_out * 4)) printf("copy %d: single-stream run is not reproducible!\n", i);
        int finite = 1;
        for (size_t k = 0; k < n_out; ++k) finite &= isfinite(h0[k]) != 0;
        if (!finite) printf("copy %d: non-finite reference output\n", i);
    }
    int bad = 0, total = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 3; ++i)   /* copy i on stream i % ns: ns = 1 runs the three copies back to back on one queue */
            PK(pafuse_mixste2_forward(&w[i], x2d[i], x3d[i], t, B, P, out[i], ws[i], nbytes, s[i % ns]));
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 3; ++i) {
            CK(hipMemcpy(h0, ref[i], n_out * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, out[i], n_out * 4, hipMemcpyDeviceToHost));
            bad += memcmp(h0, h1, n_out * 4) != 0, ++total;
        }
    }
    printf("%s, mode %d, depth %d, %d stream(s): %d of %d concurrent outputs differ from the single-stream result\n",
           pafuse_version(), mode, depth, ns, bad, total);
    return 0;
}
