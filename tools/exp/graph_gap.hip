// How much of the gap between dependent kernel launches a hipGraph removes on this box (probe: not part of the library).
//   hipcc --offload-arch=gfx950 -O2 -o tools/exp/graph_gap tools/exp/graph_gap.hip && tools/exp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(long long cycles, int* sink)
{
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) { }
    if (sink && threadIdx.x == 999) *sink = 1;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int* sink;
    CK(hipMalloc(&sink, 4));
    const long long cyc[3] = { 9000, 6800, 300 }; // ~ 90 + 68 + 3 us at 100 MHz of the constant-rate counter
    const int steps = 300;
    auto run_plain = [&]() {
        for (int i = 0; i < steps; ++i)
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, s, cyc[k], sink);
    };
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        run_plain();
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("plain launches: %.2f us per step of three kernels\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / steps);
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(spin, dim3(1024), dim3(64), 0, s, cyc[k], sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        auto t1 = std::chrono::steady_clock::now();
        printf("graph of three: %.2f us per step\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / steps);
    }
    return 0;
}
