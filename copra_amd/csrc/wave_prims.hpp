// wave_prims.hpp -- the few wavefront-level primitives the kernels are written against (gfx950, wave64).
// One MPC instance is solved by ONE 64-lane wavefront (a 64-thread workgroup), so every cross-lane exchange is
// either an LDS round trip fenced by wave_sync() or a wave shuffle.
// (tests/emu/ provides a CPU stand-in for this header so that the kernel bodies can run under sanitizers.)
#ifndef COPRA_WAVE_PRIMS_HPP
#define COPRA_WAVE_PRIMS_HPP
#include <hip/hip_runtime.h>

#define COPRA_DEV __device__ __forceinline__

namespace copra_hip {

COPRA_DEV int lane_id() { return (int)threadIdx.x; }
COPRA_DEV int instance_id() { return (int)blockIdx.x; }
COPRA_DEV int instance_stride() { return (int)gridDim.x; }

// workgroup == one wave.  wave_sync() orders the LDS traffic between the lanes of that wave: a wave's LDS instructions
// execute in issue order, so a later ds_read sees an earlier ds_write of ANY lane of the same wave without waiting for
// anything -- only the compiler must not move accesses across the point (fences at wavefront scope + a scheduling
// barrier: no instruction is emitted).  __syncthreads() here used to cost an `s_waitcnt vmcnt(0) lgkmcnt(0)` per call --
// every outstanding load, store and LDS access drained -- although the s_barrier itself is dropped for one-wave groups.
// wave_sync_full() is that heavier form: it also makes the wave's global-memory stores visible to its other lanes'
// later loads (cross-lane traffic through HBM workspaces).
#ifndef COPRA_HEAVY_SYNC
COPRA_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#else
COPRA_DEV void wave_sync() { __syncthreads(); }
#endif
COPRA_DEV void wave_sync_full() { __syncthreads(); }

COPRA_DEV int atomic_append(int* counter) { return atomicAdd(counter, 1); }
COPRA_DEV long long cycle_counter() { return (long long)__builtin_readcyclecounter(); } // s_memtime
COPRA_DEV int atomic_add_i32(int* counter, int v) { return atomicAdd(counter, v); }
// the flag of any lane of the wave (wave-uniform)
COPRA_DEV bool wave_any(bool flag) { return __ballot(flag) != 0ull; }
// number of lanes below this one whose flag is set (and the wave's total): one ballot, no LDS
COPRA_DEV int wave_prefix_count(bool flag, int& total)
{
    const unsigned long long m = __ballot(flag);
    total = __popcll(m);
    return __popcll(m & ((1ull << (threadIdx.x & 63u)) - 1ull));
}
// a table entry at a wave-uniform index: read through the constant address space, so that it is a scalar load (s_load, SGPR operand
// of the multiply-add that uses it) whatever the kernel stores elsewhere
COPRA_DEV double uniform_load(const double* p, int idx)
{
    typedef const double __attribute__((address_space(4))) * cptr_t;
    return ((cptr_t)(unsigned long long)p)[idx];
}

COPRA_DEV double* lds_base()
{
    extern __shared__ __attribute__((aligned(16))) double copra_lds[];
    return copra_lds;
}

COPRA_DEV double shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
COPRA_DEV int shfl_xor_i32(int v, int mask) { return __shfl_xor(v, mask, 64); }
COPRA_DEV double shfl_f64(double v, int src) { return __shfl(v, src, 64); }
COPRA_DEV int shfl_i32(int v, int src) { return __shfl(v, src, 64); }
// value of lane (lane+delta), 0 beyond the end of the wave
COPRA_DEV double shfl_down0_f64(double v, int delta)
{
    const double t = __shfl_down(v, (unsigned)delta, 64);
    return ((int)(threadIdx.x & 63u) + delta < 64) ? t : 0.0;
}
// value of lane (lane-delta), 0 before the start of the wave
COPRA_DEV double shfl_up0_f64(double v, int delta)
{
    const double t = __shfl_up(v, (unsigned)delta, 64);
    return ((int)(threadIdx.x & 63u) - delta >= 0) ? t : 0.0;
}

// value of lane `src` (src wave-uniform): v_readlane_b32 x2, no LDS round trip
COPRA_DEV double bcast_f64(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
COPRA_DEV int bcast_i32(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
// the instruction scheduler moves nothing across this point (bounds how far ahead it hoists the loads of unrolled code)
COPRA_DEV void sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// Ten wait states, unconditionally.  To be placed where a wave-uniform branch MERGES inside a chain of matrix instructions, in front of
// the first reader of a v_mfma_f64_4x4x4 result: hipcc of ROCm 7.2 counts the wait states such a reader needs (6 as a matrix or vector
// operand, 9 as store data) along the fall-through side of the branch only -- on the taken side the hardware, which does not interlock
// here, reads the register's previous contents (lmpc_fused_ric.hpp, the sweep; tools/mfma_hazard_lint.py finds the pattern in the ISA).
COPRA_DEV void mfma_settle() { __asm__ volatile("s_nop 7\n\ts_nop 1" ::: "memory"); }
// 1/sqrt(x): v_rsq_f64 seed + two Newton steps (full double precision to ~1 ulp, no divide)
COPRA_DEV double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
}

// D = A(16x4) * B(4x16) + C on the matrix cores, FP64 (v_mfma_f64_16x16x4_f64).  Operand layout (one double per
// lane): A[i = lane & 15][k = lane >> 4],  B[k = lane >> 4][j = lane & 15];  C/D: four doubles per lane,
// element `reg` is D[row = (lane >> 4) + 4 * reg][col = lane & 15]  (the f64 map, NOT the f32 one).
struct mfma_acc {
    double v[4];
};
COPRA_DEV void mfma_f64_16x16x4(double a, double b, mfma_acc& c)
{
    typedef double v4f64 __attribute__((ext_vector_type(4)));
    v4f64 cc = { c.v[0], c.v[1], c.v[2], c.v[3] };
    cc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, cc, 0, 0, 0);
    c.v[0] = cc[0];
    c.v[1] = cc[1];
    c.v[2] = cc[2];
    c.v[3] = cc[3];
}

// v_mfma_f64_4x4x4_4b_f64: FOUR independent products D_b = A_b (4 x 4) B_b (4 x 4) + C_b, one double per lane everywhere.
// Layout (found by one-hot probing, tools/exp/mfma444_probe.hip; profiles/r02/mfma_f64_4x4x4_probe.txt), lane = 16 q + 4 b + r:
//     A_b[i = r][k = q],   B_b[k = q][j = r],   C_b / D_b[i = q][j = r]
// so that, as with the 16 x 16 x 4 instruction, a result is laid out like the B operand of the next product.  16 cycles of
// the matrix pipe instead of 64.
COPRA_DEV double mfma_f64_4x4x4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
// lane N of every row of 16 lanes, to all lanes of that row (DPP row_newbcast)
template <int N>
COPRA_DEV double row_bcast_f64(double v)
{
    // (every lane is written: no `old` operand to initialise -- mov_dpp instead of update_dpp saves two v_mov per call)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + N, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + N, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// make a wave-uniform value provably uniform (SGPR) so that table look-ups become scalar loads
COPRA_DEV int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- DPP cross-lane moves (no LDS round trip).  ctrl: quad_perm 0x00-0xFF, row_half_mirror 0x141, row_mirror 0x140
template <int CTRL>
COPRA_DEV double dpp_f64(double v)
{
    // (the controls used here are permutations inside a row with every lane written: mov_dpp, no `old` operand to initialise)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
COPRA_DEV int dpp_i32(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, false);
}
// Butterfly inside each 16-lane row: xor 1, xor 2 (quad_perm), then mirror inside 8 and inside 16 (valid for
// commutative reductions once the smaller groups already agree).  Afterwards every lane of a row holds the row
// result; the four row results are combined through v_readlane.
#define COPRA_DPP_XOR1 0xB1 /* quad_perm [1,0,3,2] */
#define COPRA_DPP_XOR2 0x4E /* quad_perm [2,3,0,1] */
#define COPRA_DPP_HMIRROR 0x141
#define COPRA_DPP_MIRROR 0x140

// sum over the four lanes of a quad (two DPP quad_perm steps), in every lane of the quad
COPRA_DEV double quad_sum(double v)
{
    v += dpp_f64<COPRA_DPP_XOR1>(v);
    v += dpp_f64<COPRA_DPP_XOR2>(v);
    return v;
}
COPRA_DEV double wave_sum(double v)
{
    v += dpp_f64<COPRA_DPP_XOR1>(v);
    v += dpp_f64<COPRA_DPP_XOR2>(v);
    v += dpp_f64<COPRA_DPP_HMIRROR>(v);
    v += dpp_f64<COPRA_DPP_MIRROR>(v);
    return (bcast_f64(v, 0) + bcast_f64(v, 16)) + (bcast_f64(v, 32) + bcast_f64(v, 48));
}
COPRA_DEV double wave_max(double v)
{
    v = fmax(v, dpp_f64<COPRA_DPP_XOR1>(v));
    v = fmax(v, dpp_f64<COPRA_DPP_XOR2>(v));
    v = fmax(v, dpp_f64<COPRA_DPP_HMIRROR>(v));
    v = fmax(v, dpp_f64<COPRA_DPP_MIRROR>(v));
    return fmax(fmax(bcast_f64(v, 0), bcast_f64(v, 16)), fmax(bcast_f64(v, 32), bcast_f64(v, 48)));
}
// argmin of (key, idx) over the wave: smallest key, ties -> smallest idx; lanes without a candidate pass idx < 0.
// `payload` travels with the winner.  Result is wave-uniform.
COPRA_DEV void wave_argmin(double& key, int& idx, double& payload)
{
#define COPRA_ARGMIN_STEP(CTRL)                                                                                       \
    {                                                                                                                 \
        const double ok = dpp_f64<CTRL>(key);                                                                         \
        const double op = dpp_f64<CTRL>(payload);                                                                     \
        const int oi = dpp_i32<CTRL>(idx);                                                                            \
        const bool take = (oi >= 0) && (idx < 0 || ok < key || (ok == key && oi < idx));                              \
        key = take ? ok : key;                                                                                        \
        payload = take ? op : payload;                                                                                \
        idx = take ? oi : idx;                                                                                        \
    }
    COPRA_ARGMIN_STEP(COPRA_DPP_XOR1)
    COPRA_ARGMIN_STEP(COPRA_DPP_XOR2)
    COPRA_ARGMIN_STEP(COPRA_DPP_HMIRROR)
    COPRA_ARGMIN_STEP(COPRA_DPP_MIRROR)
#undef COPRA_ARGMIN_STEP
    // combine the four rows (lanes 0, 16, 32, 48 hold the row winners)
    double bk = bcast_f64(key, 0), bp = bcast_f64(payload, 0);
    int bi = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) {
        const double ok = bcast_f64(key, r), op = bcast_f64(payload, r);
        const int oi = __builtin_amdgcn_readlane(idx, r);
        const bool take = (oi >= 0) && (bi < 0 || ok < bk || (ok == bk && oi < bi));
        bk = take ? ok : bk;
        bp = take ? op : bp;
        bi = take ? oi : bi;
    }
    key = bk;
    payload = bp;
    idx = bi;
}

} // namespace copra_hip
#endif // COPRA_WAVE_PRIMS_HPP
