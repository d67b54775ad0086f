// How many one-wave workgroups with L bytes of dynamic LDS share a CU on gfx950?  (Finds the LDS allocation granule: the
// occupancy API answers with L rounded to 512 B; the hardware may round further.)  time(grid = m x CUs) = T  <=>  m are co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void spin(long long ticks, int* out)
{
    extern __shared__ int lds[];
    const long long t0 = (long long)__builtin_readcyclecounter();
    while ((long long)__builtin_readcyclecounter() - t0 < ticks) { }
    if (threadIdx.x == 0) lds[0] = 1;
    out[blockIdx.x] = lds[0];
}
int main()
{
    int* out;
    hipMalloc(&out, 1 << 20);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    hipFuncSetAttribute(reinterpret_cast<const void*>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int sizes[] = { 12800, 13312, 14080, 14336, 14848, 15360, 15361, 15872, 16384, 16385, 16640, 17744, 17920, 18204, 18432, 20400, 20480 };
    for (int L : sizes) {
        int api = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, spin, 64, L);
        printf("LDS %6d B  occupancy API %2d  co-resident:", L, api);
        float t1 = 0;
        for (int m = 7; m <= 13; ++m) {
            hipLaunchKernelGGL(spin, dim3(cus * m), dim3(64), L, 0, 100000, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(cus * m), dim3(64), L, 0, 100000, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (m == 7) t1 = ms;
            printf(" %d:%s", m, ms < 1.5f * t1 ? "yes" : "no");
        }
        printf("\n");
    }
    return 0;
}
