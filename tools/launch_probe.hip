// Micro-benchmark (not part of the product): cost of back-to-back tiny kernel launches on one stream — the unit
// cost of the deterministic mode's one-launch-per-level schedule.  hipcc --offload-arch=gfx950 -O3 tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(float *p, int n) { if (threadIdx.x < (unsigned)n) p[threadIdx.x] += 1.f; }
int main() {
    float *p; hipMalloc(&p, 4096); hipMemset(p, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 2; ++rep) {
        const int N = 50000;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, p, 64);
        auto t1 = std::chrono::steady_clock::now();
        hipStreamSynchronize(s);
        auto t2 = std::chrono::steady_clock::now();
        printf("%d launches: host enqueue %.2f us each, enqueue+drain %.2f us each\n", N,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / N,
               std::chrono::duration<double, std::micro>(t2 - t0).count() / N);
    }
    // the same through a captured graph
    hipGraph_t g; hipGraphExec_t ge;
    const int M = 10000;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < M; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(256), 0, s, p, 64);
    hipStreamEndCapture(s, &g);
    auto t0 = std::chrono::steady_clock::now();
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    auto t1 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t2 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    auto t3 = std::chrono::steady_clock::now();
    printf("graph of %d nodes: instantiate %.2f us/node, first launch %.2f us/node, second launch %.2f us/node\n", M,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / M, std::chrono::duration<double, std::micro>(t2 - t1).count() / M,
           std::chrono::duration<double, std::micro>(t3 - t2).count() / M);
    return 0;
}
