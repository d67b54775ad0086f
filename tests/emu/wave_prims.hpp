// tests/emu/wave_prims.hpp -- CPU stand-in for copra_amd/csrc/wave_prims.hpp (TEST INFRASTRUCTURE ONLY).
//
// The kernel bodies (lmpc_fused.hpp, qp_dense.hpp, gi_core.hpp) are written against a handful of wave primitives.
// This header re-implements them with cooperative fibers (ucontext), one per thread of the workgroup (64 for the
// wave-per-instance kernels, up to 1024 for the workgroup-per-instance ones), that run freely BETWEEN barriers: a
// thread runs until its next wave_sync()/bt_sync()/shuffle and waits there until its wave / workgroup has arrived
// (a barrier some threads never reach is reported as a deadlock).  Consequences:
//   * a missing wave_sync() between an LDS write and a cross-lane read shows up as a wrong result here (the fibers
//     do not execute instruction-by-instruction in lock-step like the hardware does, so this is stricter);
//   * the same source can be built with -fsanitize=undefined on the CPU (GPU sanitizers are not available);
//   * `pytest -m "not gpu"` can check the kernel logic against the oracle without a GPU.
// It is never linked into libcopra_hip.so and never used by the product path.
#ifndef COPRA_WAVE_PRIMS_HPP
#define COPRA_WAVE_PRIMS_HPP
#define COPRA_BLOCK_PRIMS_HPP // this header also stands in for copra_amd/csrc/block_prims.hpp
#include <cmath>
#include <cstddef>

#define COPRA_DEV inline

namespace copra_hip {

namespace emu {
    constexpr int kMaxThreads = 1024;
    struct WaveState {
        int lane; // current fiber == thread id in the workgroup
        int nthreads; // workgroup size (multiple of 64)
        int inst;
        int ninst;
        double* lds;
        double xf[kMaxThreads]; // shuffle staging
        int xi[kMaxThreads];
    };
    extern WaveState g_wave;
    // implemented in emu_harness.cpp
    void barrier_wave(); // all 64 threads of the caller's wave
    void barrier_block(); // all threads of the workgroup
} // namespace emu

COPRA_DEV int lane_id() { return emu::g_wave.lane; } // wave-per-instance kernels: workgroup == one wave
COPRA_DEV int instance_id() { return emu::g_wave.inst; }
COPRA_DEV int instance_stride() { return emu::g_wave.ninst; }
COPRA_DEV void wave_sync() { emu::barrier_block(); }
COPRA_DEV void wave_sync_full() { emu::barrier_block(); }
// workgroup-per-instance kernels (block_prims.hpp)
COPRA_DEV int bt_tid() { return emu::g_wave.lane; }
COPRA_DEV int bt_size() { return emu::g_wave.nthreads; }
COPRA_DEV int bt_lane() { return emu::g_wave.lane & 63; }
COPRA_DEV int bt_wave() { return emu::g_wave.lane >> 6; }
COPRA_DEV int bt_nwaves() { return emu::g_wave.nthreads >> 6; }
COPRA_DEV void bt_sync() { emu::barrier_block(); }
COPRA_DEV double* lds_base() { return emu::g_wave.lds; }
COPRA_DEV long long cycle_counter() { return 0; }
COPRA_DEV int atomic_append(int* counter) { return (*counter)++; }
COPRA_DEV int atomic_add_i32(int* counter, int v)
{
    const int o = *counter;
    *counter += v;
    return o;
}
COPRA_DEV double uniform_load(const double* p, int idx) { return p[idx]; }

// `src` is a lane of the caller's own wave
COPRA_DEV double emu_xchg_f64(double v, int src)
{
    const int me = emu::g_wave.lane, base = me & ~63;
    emu::g_wave.xf[me] = v;
    emu::barrier_wave();
    const double r = (src >= 0 && src < 64) ? emu::g_wave.xf[base + src] : emu::g_wave.xf[me];
    emu::barrier_wave();
    return r;
}
COPRA_DEV int emu_xchg_i32(int v, int src)
{
    const int me = emu::g_wave.lane, base = me & ~63;
    emu::g_wave.xi[me] = v;
    emu::barrier_wave();
    const int r = (src >= 0 && src < 64) ? emu::g_wave.xi[base + src] : emu::g_wave.xi[me];
    emu::barrier_wave();
    return r;
}

COPRA_DEV double shfl_xor_f64(double v, int mask) { return emu_xchg_f64(v, bt_lane() ^ mask); }
COPRA_DEV int shfl_xor_i32(int v, int mask) { return emu_xchg_i32(v, bt_lane() ^ mask); }
COPRA_DEV double shfl_f64(double v, int src) { return emu_xchg_f64(v, src); }
COPRA_DEV int shfl_i32(int v, int src) { return emu_xchg_i32(v, src); }
COPRA_DEV double shfl_down0_f64(double v, int delta)
{
    const int src = bt_lane() + delta;
    const double t = emu_xchg_f64(v, src);
    return (src < 64) ? t : 0.0;
}
COPRA_DEV double shfl_up0_f64(double v, int delta)
{
    const int src = bt_lane() - delta;
    const double t = emu_xchg_f64(v, src);
    return (src >= 0) ? t : 0.0;
}
COPRA_DEV double bcast_f64(double v, int src) { return emu_xchg_f64(v, src); }
COPRA_DEV int bcast_i32(int v, int src) { return (int)emu_xchg_f64((double)v, src); }
COPRA_DEV int wave_prefix_count(bool flag, int& total)
{
    const int me = emu::g_wave.lane, base = me & ~63;
    emu::g_wave.xi[me] = flag ? 1 : 0;
    emu::barrier_wave();
    int before = 0;
    total = 0;
    for (int l = 0; l < 64; ++l) {
        total += emu::g_wave.xi[base + l];
        if (l < (me & 63)) before += emu::g_wave.xi[base + l];
    }
    emu::barrier_wave();
    return before;
}
COPRA_DEV bool wave_any(bool flag)
{
    int total = 0;
    (void)wave_prefix_count(flag, total);
    return total > 0;
}
COPRA_DEV void sched_fence() { }
COPRA_DEV void mfma_settle() { } // (hardware wait states: nothing to emulate)
COPRA_DEV double fast_rsqrt(double x) { return 1.0 / std::sqrt(x); }

COPRA_DEV int uniform_i32(int v) { return v; }

// v_mfma_f64_16x16x4_f64 stand-in with the same operand / result layout as the hardware instruction
struct mfma_acc {
    double v[4];
};
namespace emu {
    extern double g_mfma_a[kMaxThreads], g_mfma_b[kMaxThreads];
}
COPRA_DEV void mfma_f64_16x16x4(double a, double b, mfma_acc& c)
{
    const int l = emu::g_wave.lane & 63, base = emu::g_wave.lane & ~63;
    emu::g_mfma_a[base + l] = a; // A[i = l & 15][k = l >> 4]
    emu::g_mfma_b[base + l] = b; // B[k = l >> 4][j = l & 15]
    emu::barrier_wave();
    const int j = l & 15;
    for (int reg = 0; reg < 4; ++reg) {
        const int i = (l >> 4) + 4 * reg;
        double acc = c.v[reg];
        for (int k = 0; k < 4; ++k) acc += emu::g_mfma_a[base + i + 16 * k] * emu::g_mfma_b[base + j + 16 * k];
        c.v[reg] = acc;
    }
    emu::barrier_wave();
}
// v_mfma_f64_4x4x4_4b_f64 stand-in: lane = 16 q + 4 b + r;  A_b[i = r][k = q], B_b[k = q][j = r], C_b / D_b[i = q][j = r]
COPRA_DEV double mfma_f64_4x4x4(double a, double b, double c)
{
    const int l = emu::g_wave.lane & 63, base = emu::g_wave.lane & ~63;
    emu::g_mfma_a[base + l] = a;
    emu::g_mfma_b[base + l] = b;
    emu::barrier_wave();
    const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
    double acc = c;
    for (int k = 0; k < 4; ++k) acc += emu::g_mfma_a[base + 16 * k + 4 * blk + i] * emu::g_mfma_b[base + 16 * k + 4 * blk + j];
    emu::barrier_wave();
    return acc;
}
template <int N>
COPRA_DEV double row_bcast_f64(double v) { return shfl_f64(v, (lane_id() & ~15) + N); }
COPRA_DEV void wave_argmin(double& key, int& idx, double& payload)
{
    for (int m = 32; m >= 1; m >>= 1) {
        const double ok = shfl_xor_f64(key, m);
        const double op = shfl_xor_f64(payload, m);
        const int oi = shfl_xor_i32(idx, m);
        const bool take = (oi >= 0) && (idx < 0 || ok < key || (ok == key && oi < idx));
        if (take) {
            key = ok;
            payload = op;
            idx = oi;
        }
    }
}

COPRA_DEV double quad_sum(double v)
{
    v += shfl_xor_f64(v, 1);
    v += shfl_xor_f64(v, 2);
    return v;
}
COPRA_DEV double wave_sum(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v += shfl_xor_f64(v, m);
    return v;
}
COPRA_DEV double wave_max(double v)
{
    for (int m = 32; m >= 1; m >>= 1) v = fmax(v, shfl_xor_f64(v, m));
    return v;
}

} // namespace copra_hip
#endif // COPRA_WAVE_PRIMS_HPP
