/* linear_smoke.c - the C ABI used from plain C with the HIP runtime only (no torch, no Python):
 * out = A W^T + b on the device, checked against a host loop.  Built and run by tests/test_hip_parity.py on the GPU box:
 *   gcc tests/cabi/linear_smoke.c -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Lpafuse_amd -lpafuse_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -o build/linear_smoke                                                  */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "pafuse_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

int main(void) {
    const int M = 77, N = 96, K = 64;
    float *hA = malloc(sizeof(float) * M * K), *hW = malloc(sizeof(float) * N * K), *hb = malloc(sizeof(float) * N),
          *ho = malloc(sizeof(float) * M * N);
    for (int i = 0; i < M * K; ++i) hA[i] = (float)((i * 37) % 17 - 8);      /* small integers: the result is exact */
    for (int i = 0; i < N * K; ++i) hW[i] = (float)((i * 53) % 13 - 6);
    for (int i = 0; i < N; ++i) hb[i] = (float)(i % 5);
    float *dA, *dW, *db, *dO;
    CK(hipMalloc((void**)&dA, sizeof(float) * M * K));
    CK(hipMalloc((void**)&dW, sizeof(float) * N * K));
    CK(hipMalloc((void**)&db, sizeof(float) * N));
    CK(hipMalloc((void**)&dO, sizeof(float) * M * N));
    CK(hipMemcpy(dA, hA, sizeof(float) * M * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW, sizeof(float) * N * K, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb, sizeof(float) * N, hipMemcpyHostToDevice));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int rc = pafuse_linear(dA, dW, db, dO, M, N, K, 0, (void*)s);
    if (rc != PAFUSE_OK) { printf("pafuse_linear failed: %d %s\n", rc, pafuse_last_error()); return 3; }
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(ho, dO, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            float acc = hb[n];
            for (int k = 0; k < K; ++k) acc += hA[m * K + k] * hW[n * K + k];
            if (acc != ho[m * N + n]) ++bad;
        }
    rc = pafuse_linear(dA, dW, db, dO, M, N, 33, 0, (void*)s);               /* K not a multiple of 32: refused */
    printf("%s: %d mismatches of %d, bad-shape rc=%d (%s)\n", pafuse_version(), bad, M * N, rc, pafuse_last_error());
    return (bad == 0 && rc == PAFUSE_E_SHAPE) ? 0 : 1;
}
