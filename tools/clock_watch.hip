// clock_watch.hip - what shader clock does the chip hold while the production GEMM runs?
// A one-wave monitor kernel on a second stream samples (s_memtime = shader-clock counter, s_memrealtime = 100 MHz
// counter) once a millisecond while pafuse_linear launches run back to back on the main stream; the ratio of the
// two deltas is the clock.  Prints the clock idle, under the GEMM stream, and the GEMM rate in the same window.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/clock_watch.hip -o build/clock_watch -ldl
//   gpurun -- './build/clock_watch pafuse_amd/libpafuse_hip.so'
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void monitor(unsigned long long* samples, int nsamples, unsigned long long period_rt) {
    if (threadIdx.x != 0) return;
    unsigned long long rt0 = __builtin_readcyclecounter();
    (void)rt0;
    for (int i = 0; i < nsamples; ++i) {
        unsigned long long t, rt;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(rt)::"memory");
        samples[2 * i] = t;
        samples[2 * i + 1] = rt;
        unsigned long long now = rt;
        while (now - rt < period_rt) {
            __builtin_amdgcn_s_sleep(100);
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now)::"memory");
        }
    }
}

typedef int (*linear_fn)(const float*, const float*, const float*, float*, long long, int, int, int, void*);

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "pafuse_amd/libpafuse_hip.so";
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    linear_fn linear = (linear_fn)dlsym(lib, "pafuse_linear");
    const long long M = 25920;
    const int N = 1152, K = 384;
    float *A, *W, *b, *out;
    hipMalloc(&A, M * K * 4), hipMalloc(&W, (size_t)N * K * 4), hipMalloc(&b, N * 4), hipMalloc(&out, M * N * 4);
    std::vector<float> h(M * K);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(A, h.data(), M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(W, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    hipMemset(b, 0, N * 4);
    hipStream_t s_main, s_mon;
    hipStreamCreate(&s_main), hipStreamCreate(&s_mon);
    const int NS = 120;
    unsigned long long* samples;
    hipMalloc(&samples, NS * 16);
    std::vector<unsigned long long> hs(2 * NS);
    auto report = [&](const char* tag) {
        hipMemcpy(hs.data(), samples, NS * 16, hipMemcpyDeviceToHost);
        double lo = 1e9, hi = 0, sum = 0;
        int n = 0;
        for (int i = 10; i + 1 < NS - 10; ++i) {
            double mhz = (double)(hs[2 * i + 2] - hs[2 * i]) / (double)(hs[2 * i + 3] - hs[2 * i + 1]) * 100.0;
            lo = mhz < lo ? mhz : lo, hi = mhz > hi ? mhz : hi, sum += mhz, ++n;
        }
        printf("%-18s shader clock: mean %.0f MHz  (min %.0f, max %.0f over %d 1-ms windows)\n", tag, sum / n, lo, hi, n);
    };
    // idle
    hipLaunchKernelGGL(monitor, dim3(1), dim3(64), 0, s_mon, samples, NS, 100000ull);
    hipStreamSynchronize(s_mon);
    report("idle:");
    // under the GEMM stream
    for (int i = 0; i < 20; ++i) linear(A, W, b, out, M, N, K, 0, s_main);
    hipStreamSynchronize(s_main);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    hipLaunchKernelGGL(monitor, dim3(1), dim3(64), 0, s_mon, samples, NS, 100000ull);
    const int reps = 700;
    hipEventRecord(e0, s_main);
    for (int i = 0; i < reps; ++i) linear(A, W, b, out, M, N, K, 0, s_main);
    hipEventRecord(e1, s_main);
    hipStreamSynchronize(s_main);
    hipStreamSynchronize(s_mon);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    report("under qkv GEMMs:");
    const double tf = 2.0 * M * N * K * reps / (ms * 1e-3) / 1e12;
    printf("qkv body GEMM: %.1f us per launch, %.1f TFLOP/s\n", ms * 1e3 / reps, tf);
    return 0;
}
