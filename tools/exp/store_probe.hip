// What 28 flat 1-KB stores of one wave cost (the results phase of lmpc_axis.hpp), alone and with every SIMD busy, cold and with the pages'
// translations warmed by a touch beforehand.   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/exp/store_probe.hip && /tmp/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double pair2 __attribute__((ext_vector_type(2)));
__device__ inline long long now() { return (long long)__builtin_readcyclecounter(); }

template <int MODE> // 0: stores cold | 1: touch (one load per 4 KB page) first, wait, then stores | 2: the same region a second time (warm)
__global__ __launch_bounds__(64, 1) void probe(double* out, long long* t, int rows, int spin) {
    const int lane = threadIdx.x;
    double* const dst = out + (size_t)blockIdx.x * rows * 128;
    // something to do first so that the waves of a launch drift apart a little, like the solver's
    double acc = lane;
    for (int i = 0; i < spin + (int)(blockIdx.x % 7) * 16; ++i) acc = acc * 1.0000001 + 0.5;
    long long t_touch = 0;
    if (MODE == 1) {
        const long long a = now();
        double s = 0;
        const int pages = (rows * 1024 + 4095) / 4096 + 1;
        if (lane < pages) s = __builtin_nontemporal_load(dst + (size_t)lane * 512 < out + (size_t)gridDim.x * rows * 128 ? dst + (size_t)lane * 512 : dst);
        acc += s * 1e-300;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t_touch = now() - a;
    }
    const int reps = MODE == 2 ? 2 : 1;
    long long issue = 0, done = 0;
    for (int r = 0; r < reps; ++r) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long a = now();
        for (int j = 0; j < rows; ++j) {
            pair2 v;
            v.x = acc + j;
            v.y = acc - j;
            *(pair2*)(dst + j * 128 + 2 * lane) = v;
        }
        const long long b = now();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long c = now();
        issue = b - a;
        done = c - a;
    }
    if (lane == 0) {
        t[3 * blockIdx.x] = issue;
        t[3 * blockIdx.x + 1] = done;
        t[3 * blockIdx.x + 2] = t_touch;
    }
}

template <int MODE> static void run(const char* name, int grid, int rows, double* out, long long* t) {
    std::vector<long long> h(3 * grid);
    hipMemset(out, 0, (size_t)grid * rows * 1024);
    hipDeviceSynchronize();
    probe<MODE><<<grid, 64>>>(out, t, rows, 2000);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, sizeof(long long) * 3 * grid, hipMemcpyDeviceToHost);
    double si = 0, sd = 0, st = 0;
    long long mi = 1LL << 60, md = 1LL << 60;
    for (int g = 0; g < grid; ++g) {
        si += h[3 * g]; sd += h[3 * g + 1]; st += h[3 * g + 2];
        mi = std::min(mi, h[3 * g]); md = std::min(md, h[3 * g + 1]);
    }
    printf("%-34s grid %5d rows %2d: issue mean %7.0f min %6lld | acked mean %7.0f min %6lld | touch %6.0f  (ticks of s_memtime)\n", name, grid, rows, si / grid,
           mi, sd / grid, md, st / grid);
}

int main() {
    const int maxgrid = 3072, rows = 28;
    double* out;
    long long* t;
    hipMalloc(&out, (size_t)maxgrid * rows * 1024 * 2);
    hipMalloc(&t, sizeof(long long) * 3 * maxgrid);
    for (int grid : {1, 256, 1024, 3072}) {
        run<0>("cold", grid, rows, out, t);
        run<1>("touched first", grid, rows, out, t);
        run<2>("second time over the same bytes", grid, rows, out, t);
    }
    for (int r : {4, 8, 16}) run<0>("cold", 1024, r, out, t);
    return 0;
}
