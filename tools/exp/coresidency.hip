// Experiment: which resource stops two 5-wave workgroups from sharing a CU on gfx950?
// A latency-bound chain kernel (dependent global loads) with configurable workgroup size, dynamic LDS and scratch.
// time(grid = 2 x CUs) ~= time(grid = CUs)  <=>  two workgroups are co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int T, int MINW, int SCRATCH, int REGS = 0>
__global__ __launch_bounds__(T, MINW) void chain_kernel(const int* next, int* out, int steps)
{
    extern __shared__ int lds[];
    int scratch[SCRATCH > 0 ? SCRATCH : 1];
    double live[REGS > 0 ? REGS : 1];
#pragma unroll
    for (int i = 0; i < (REGS > 0 ? REGS : 1); ++i) live[i] = threadIdx.x * 0.5 + i;
    int p = (blockIdx.x * 977) & 0xffff; // workgroup-uniform pointer chase: one cache line per wave and step
    if (SCRATCH > 0)
        for (int i = 0; i < SCRATCH; ++i) scratch[i] = p + i;
    for (int s = 0; s < steps; ++s) {
        p = next[p];
        if (SCRATCH > 0 && (s & 255) == 0) scratch[(p + threadIdx.x) % SCRATCH] += p; // dynamic index -> scratch memory, touched rarely
        if (REGS > 0) {
#pragma unroll
            for (int i = 0; i < REGS; ++i) asm volatile("" : "+v"(live[i])); // keeps the registers live, no instructions
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) lds[0] = p;
    int acc = p;
    if (REGS > 0) {
        double t = 0;
#pragma unroll
        for (int i = 0; i < REGS; ++i) t += live[i];
        acc += (int)t;
    }
    if (SCRATCH > 0)
        for (int i = 0; i < SCRATCH; ++i) acc += scratch[i];
    out[blockIdx.x * T + threadIdx.x] = acc + lds[0];
}

template <int T, int MINW, int SCRATCH, int REGS = 0>
static void run(const char* name, const int* dnext, int* dout, int cus, size_t ldsBytes)
{
    auto k = chain_kernel<T, MINW, SCRATCH, REGS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, T, ldsBytes);
    float ms[3] = { 0, 0, 0 };
    for (int mult = 1; mult <= 3; ++mult) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(cus * mult), dim3(T), ldsBytes, 0, dnext, dout, 3000);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mult - 1], e0, e1);
        }
    }
    printf("%-44s occupancy API %d/CU   grid x1 %.2f ms  x2 %.2f ms  x3 %.2f ms\n", name, occ, ms[0], ms[1], ms[2]);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    std::vector<int> next(65536);
    unsigned s = 1;
    for (auto& v : next) {
        s = s * 1664525u + 1013904223u;
        v = (s >> 8) & 0xffff;
    }
    int *dnext, *dout;
    hipMalloc(&dnext, next.size() * 4);
    hipMalloc(&dout, (size_t)cus * 3 * 512 * 4);
    hipMemcpy(dnext, next.data(), next.size() * 4, hipMemcpyHostToDevice);
    printf("%d CUs\n", cus);
    run<256, 1, 0>("T=256 LDS 27K", dnext, dout, cus, 27 * 1024);
    run<320, 1, 0>("T=320 LDS 27K", dnext, dout, cus, 27 * 1024);
    run<320, 3, 0>("T=320 minwaves3 LDS 27K", dnext, dout, cus, 27 * 1024);
    run<320, 3, 0>("T=320 minwaves3 LDS 51K", dnext, dout, cus, 51 * 1024);
    run<320, 3, 0>("T=320 minwaves3 LDS 78K", dnext, dout, cus, 78 * 1024);
    run<320, 3, 200>("T=320 minwaves3 LDS 51K scratch 800B", dnext, dout, cus, 51 * 1024);
    run<320, 3, 200>("T=320 minwaves3 LDS 78K scratch 800B", dnext, dout, cus, 78 * 1024);
    run<256, 2, 200>("T=256 minwaves2 LDS 27K scratch 800B", dnext, dout, cus, 27 * 1024);
    run<320, 3, 0, 70>("T=320 minwaves3 LDS 51K ~160 VGPRs", dnext, dout, cus, 51 * 1024);
    run<320, 3, 200, 70>("T=320 minwaves3 LDS 51K ~160 VGPRs scratch", dnext, dout, cus, 51 * 1024);
    run<320, 2, 0, 110>("T=320 minwaves2 LDS 51K ~240 VGPRs", dnext, dout, cus, 51 * 1024);
    return 0;
}
