/* queue_concurrency.c - torch-free reproducer of the multi-queue hazard of the bf16-MFMA modes (VERDICT r2 item 5,
 * profiles/r03_bf16_mfma_concurrency.md has the root cause): three independent copies of ONE body-part denoiser (own weights, own split
 * images, own workspaces, own outputs) run pafuse_mixste2_forward at the same time on three HIP streams; every output is
 * compared bit for bit with the same copy's single-stream result.  Plain C + the HIP runtime + the C ABI only.
 *   gcc tests/cabi/queue_concurrency.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lpafuse_amd -lpafuse_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -o tools/bin/queue_concurrency
 *   LD_LIBRARY_PATH=pafuse_amd ./tools/bin/queue_concurrency <mode 0|1|2> [depth] [rounds] [streams]
 * Prints "<bad> of <n> concurrent outputs differ".  Exit code 0 always (a measurement, not a test). */
#define _GNU_SOURCE
#include <dlfcn.h>
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
static void* split_of(const float* W, int N, int K, int layout) {   /* pafuse_split_weights: 0 fc1, 1 proj / fc2, 2 qkv */
    void* img;
    CK(hipMalloc(&img, pafuse_split_weights_bytes(N, K)));
    PK(pafuse_split_weights(W, N, K, layout, img, NULL));
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
        b->qkv_ws = split_of(b->qkv_w, 3 * C, C, 2), b->proj_ws = split_of(b->proj_w, C, C, 1);
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
    }
    /* the three workspaces in ONE arena, each between two guard zones of a known pattern: a kernel that writes outside its
     * own workspace either lands in a neighbour's buffers (wrong results only while that neighbour is in flight, i.e. only
     * under concurrency) or in a guard, which is checked at the end */
    const size_t GUARD = (size_t)8 << 20, slot = (nbytes + 255) / 256 * 256;
    char* arena;
    CK(hipMalloc((void**)&arena, 4 * GUARD + 3 * slot));
    CK(hipMemset(arena, 0xAB, 4 * GUARD + 3 * slot));
    for (int i = 0; i < 3; ++i) {
        ws[i] = arena + GUARD + i * (slot + GUARD);
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
    /* with a -DPAFUSE_DIAG build of the library on LD_LIBRARY_PATH: hash every intermediate tensor of every pass and report,
     * for each wrong output, the first kernel output that differs from the single-stream pass (its inputs were still right) */
    typedef int (*trace_fn)(unsigned long long*, int32_t);
    trace_fn diag_trace = (trace_fn)dlsym(RTLD_DEFAULT, "pafuse_diag_trace");
    typedef void (*snap_fn)(float*);
    snap_fn diag_snap = (snap_fn)dlsym(RTLD_DEFAULT, "pafuse_diag_snapshot");
    const size_t n_x = (size_t)B * P * F * J * C + (size_t)B * C;   /* x, then temb as embed_kernel found it */
    float *snap_dev[3] = {0}, *snap_ref[3] = {0}, *snap_now = NULL;
    int reported = 0;
    enum { SLOTS = 512 };
    unsigned long long *tr_dev[3] = {0}, *tr_ref[3] = {0}, *tr_now = NULL;
    int n_slots = 0;
    static int first_bad_hist[SLOTS];
    if (diag_trace) {
        tr_now = malloc(SLOTS * 8);
        for (int i = 0; i < 3; ++i) {
            CK(hipMalloc((void**)&tr_dev[i], SLOTS * 8));
            CK(hipMemset(tr_dev[i], 0, SLOTS * 8));
            tr_ref[i] = malloc(SLOTS * 8);
            if (diag_snap) {
                CK(hipMalloc((void**)&snap_dev[i], n_x * 4));
                snap_ref[i] = malloc(n_x * 4), snap_now = snap_now ? snap_now : malloc(n_x * 4);
                diag_snap(snap_dev[i]);
            }
            diag_trace(tr_dev[i], SLOTS);
            PK(pafuse_mixste2_forward(&w[i], x2d[i], x3d[i], t, B, P, out[i], ws[i], nbytes, s[0]));
            CK(hipStreamSynchronize(s[0]));
            n_slots = diag_trace(NULL, 0);
            CK(hipMemcpy(tr_ref[i], tr_dev[i], SLOTS * 8, hipMemcpyDeviceToHost));
            if (diag_snap) CK(hipMemcpy(snap_ref[i], snap_dev[i], n_x * 4, hipMemcpyDeviceToHost));
        }
        printf("diagnostic library: %d traced tensors per pass (temb, x, xn, then per block: qkv, o, x, xn, hidden, x, xn)\n", n_slots);
    }
    int bad = 0, total = 0;
    for (int r = 0; r < rounds; ++r) {
        if (diag_trace) {
            for (int i = 0; i < 3; ++i) CK(hipMemset(tr_dev[i], 0, SLOTS * 8));
            CK(hipDeviceSynchronize());
            for (int i = 0; i < 3; ++i) {
                if (diag_snap) diag_snap(snap_dev[i]);
                diag_trace(tr_dev[i], SLOTS);
                PK(pafuse_mixste2_forward(&w[i], x2d[i], x3d[i], t, B, P, out[i], ws[i], nbytes, s[i % ns]));
            }
            diag_trace(NULL, 0);
            CK(hipDeviceSynchronize());
            for (int i = 0; i < 3 && diag_snap && reported < 6; ++i) {   /* what a wrong embedding looks like */
                CK(hipMemcpy(snap_now, snap_dev[i], n_x * 4, hipMemcpyDeviceToHost));
                if (!memcmp(snap_now, snap_ref[i], n_x * 4)) continue;
                size_t nd = 0, first = 0, last = 0;
                long rows_bad = 0;
                {
                    const size_t t0 = (size_t)B * P * F * J * C;
                    int tb = 0;
                    for (int c = 0; c < C; ++c) tb += memcmp(&snap_now[t0 + c], &snap_ref[i][t0 + c], 4) != 0;
                    printf("round %d copy %d: temb as read between time_embed and embed: %d of %d elements differ", r, i, tb, C);
                    if (tb) {
                        printf(" (channels:");
                        for (int c = 0; c < C; ++c) if (memcmp(&snap_now[t0 + c], &snap_ref[i][t0 + c], 4)) printf(" %d", c);
                        printf(")");
                    }
                    printf("\n");
                }
                for (size_t m = 0; m < (size_t)B * P * F * J; ++m) {
                    int rb = 0;
                    for (int c = 0; c < C; ++c)
                        if (memcmp(&snap_now[m * C + c], &snap_ref[i][m * C + c], 4)) { if (!nd) first = m * C + c; last = m * C + c; ++nd, rb = 1; }
                    rows_bad += rb;
                }
                printf("round %d copy %d: x behind embed_kernel differs in %zu of %zu elements, %ld of %zu rows; first (row %zu, ch %zu) "
                       "last (row %zu, ch %zu); first: got %.9g want %.9g", r, i, nd, n_x, rows_bad, (size_t)B * P * F * J, first / C, first % C, last / C,
                       last % C, snap_now[first], snap_ref[i][first]);
                /* is the wrong row another row's right value? (a wrong temb / pos / input index) */
                {
                    const size_t m = first / C;
                    double dsum = 0;
                    for (int c = 0; c < C; ++c) dsum += fabs((double)snap_now[m * C + c] - snap_ref[i][m * C + c]);
                    printf("; mean |d| over that row %.3g; wrong channels of that row:", dsum / C);
                    for (int c = 0; c < C; ++c) if (memcmp(&snap_now[m * C + c], &snap_ref[i][m * C + c], 4)) printf(" %d", c);
                    printf("\n");
                }
                ++reported;
            }
            for (int i = 0; i < 3; ++i) {
                CK(hipMemcpy(tr_now, tr_dev[i], SLOTS * 8, hipMemcpyDeviceToHost));
                for (int k = 0; k < n_slots; ++k)
                    if (tr_now[k] != tr_ref[i][k]) { ++first_bad_hist[k]; break; }
            }
            for (int i = 0; i < 3; ++i) {
                CK(hipMemcpy(h0, ref[i], n_out * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1, out[i], n_out * 4, hipMemcpyDeviceToHost));
                bad += memcmp(h0, h1, n_out * 4) != 0, ++total;
            }
            continue;
        }
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
    {   /* guard zones */
        unsigned char* hg = malloc(GUARD);
        for (int gz = 0; gz < 4; ++gz) {
            const char* gp = arena + gz * (slot + GUARD);
            CK(hipMemcpy(hg, gp, GUARD, hipMemcpyDeviceToHost));
            size_t nbad = 0, first = 0, last = 0;
            for (size_t k = 0; k < GUARD; ++k)
                if (hg[k] != 0xAB) { if (!nbad) first = k; last = k; ++nbad; }
            printf("guard zone %d (%s workspace %d): %zu bytes modified", gz, gz ? "behind" : "in front of", gz ? gz - 1 : 0, nbad);
            if (nbad) printf(" (offsets %zu .. %zu of %zu)", first, last, GUARD);
            printf("\n");
        }
        free(hg);
    }
    if (diag_trace) {
        printf("first differing traced tensor (slot: count):");
        for (int k = 0; k < n_slots; ++k)
            if (first_bad_hist[k]) printf(" %d:%d", k, first_bad_hist[k]);
        printf("\n");
    }
    return 0;
}
