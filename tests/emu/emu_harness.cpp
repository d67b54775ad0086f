// tests/emu/emu_harness.cpp -- runs the HIP kernel BODIES on the CPU, one fiber per lane (TEST INFRASTRUCTURE ONLY).
// See wave_prims.hpp in this directory.  Exposes a tiny C interface for ctypes.
#include "wave_prims.hpp" // must come first: shadows copra_amd/csrc/wave_prims.hpp (same include guard name)

#include "../../copra_amd/csrc/islmpc_fused.hpp"
#include "../../copra_amd/csrc/lmpc_fused.hpp"
#include "../../copra_amd/csrc/lmpc_fused_ric.hpp"
#include "../../copra_amd/csrc/lmpc_lane.hpp"
#include "../../copra_amd/csrc/lmpc_axis.hpp"
#include "../../copra_amd/csrc/lmpc_large.hpp"
#include "../../copra_amd/csrc/lmpc_riccati.hpp"
#include "../../copra_amd/csrc/lmpc_riccati_mfma.hpp"
#include "../../copra_amd/csrc/lmpc_shared.hpp"
#include "../../copra_amd/csrc/plan_builder.hpp"
#include "../../copra_amd/csrc/qp_dense.hpp"
#include "../../copra_amd/csrc/qp_dense_large.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <ucontext.h>
#include <csetjmp>
#include <algorithm>
#include <vector>

namespace copra_hip {
namespace emu {
    WaveState g_wave;
    double g_mfma_a[kMaxThreads], g_mfma_b[kMaxThreads];
    static ucontext_t g_sched;
    static ucontext_t g_fiber[kMaxThreads];
    static bool g_done[kMaxThreads];
    static std::function<void()>* g_body;
    // barriers: a waiting fiber records which generation it waits for; the scheduler skips it until that generation
    // has passed (so a blocked thread costs a comparison, not a context switch)
    static int g_blk_count;
    static unsigned g_blk_gen;
    static int g_wv_count[kMaxThreads / 64];
    static unsigned g_wv_gen[kMaxThreads / 64];
    static int g_wait_kind[kMaxThreads]; // 0 runnable, 1 block barrier, 2 wave barrier
    static unsigned g_wait_gen[kMaxThreads];
    static unsigned long g_progress;

    // Switching: swapcontext saves and restores the signal mask -- a system call per switch, two thirds of the emulator's run time.  A fiber
    // is ENTERED once through its ucontext (its own stack); every later switch is _setjmp / _longjmp, which touch registers only.
    static jmp_buf g_sched_jb;
    static jmp_buf g_fiber_jb[kMaxThreads];
    static bool g_started[kMaxThreads];
    static void yield()
    {
        if (!_setjmp(g_fiber_jb[g_wave.lane])) _longjmp(g_sched_jb, 1);
    }
    static void resume(int l) // scheduler -> fiber l, back at its next yield
    {
        if (_setjmp(g_sched_jb)) return;
        if (!g_started[l]) {
            g_started[l] = true;
            setcontext(&g_fiber[l]);
        }
        _longjmp(g_fiber_jb[l], 1);
    }

    void barrier_block()
    {
        const int me = g_wave.lane;
        const unsigned g = g_blk_gen;
        if (++g_blk_count == g_wave.nthreads) {
            g_blk_count = 0;
            ++g_blk_gen;
            ++g_progress;
            return;
        }
        g_wait_kind[me] = 1;
        g_wait_gen[me] = g;
        yield();
    }
    void barrier_wave()
    {
        const int me = g_wave.lane, w = me >> 6;
        const unsigned g = g_wv_gen[w];
        if (++g_wv_count[w] == 64) {
            g_wv_count[w] = 0;
            ++g_wv_gen[w];
            ++g_progress;
            return;
        }
        g_wait_kind[me] = 2;
        g_wait_gen[me] = g;
        yield();
    }

    static void fiber_main()
    {
        (*g_body)();
        g_done[g_wave.lane] = true;
        ++g_progress;
        _longjmp(g_sched_jb, 1);
    }

    // run one workgroup of `nthreads` threads to completion
    static int run_block(std::function<void()> body, size_t lds_bytes, int inst, int ninst, int nthreads)
    {
        static std::vector<char> stacks;
        const size_t stack_sz = 1024 * 1024; // (the unrolled (instance, axis)-per-lane bodies have frames of several hundred KB at -O1)
        if (nthreads % 64 != 0 || nthreads > kMaxThreads) return -1;
        if (stacks.size() < (size_t)nthreads * stack_sz) stacks.resize((size_t)nthreads * stack_sz);
        std::vector<double> lds(lds_bytes / sizeof(double) + 2, __builtin_nan(""));
        g_wave.lds = lds.data();
        g_wave.inst = inst;
        g_wave.ninst = ninst;
        g_wave.nthreads = nthreads;
        g_body = &body;
        g_blk_count = 0;
        for (int w = 0; w < nthreads / 64; ++w) g_wv_count[w] = 0;
        for (int l = 0; l < nthreads; ++l) {
            g_done[l] = false;
            g_started[l] = false;
            g_wait_kind[l] = 0;
            getcontext(&g_fiber[l]);
            g_fiber[l].uc_stack.ss_sp = stacks.data() + (size_t)l * stack_sz;
            g_fiber[l].uc_stack.ss_size = stack_sz;
            g_fiber[l].uc_link = &g_sched;
            makecontext(&g_fiber[l], fiber_main, 0);
        }
        for (;;) {
            const unsigned long before = g_progress;
            int ndone = 0;
            // (COPRA_EMU_REVERSE: the lanes take their turns in descending order -- a result that depends on the order is a missing wave_sync:
            //  between two syncs the hardware runs the lanes in lockstep, the emulator one after the other)
            static const bool reverse = std::getenv("COPRA_EMU_REVERSE") != nullptr;
            for (int li = 0; li < nthreads; ++li) {
                const int l = reverse ? nthreads - 1 - li : li;
                if (g_done[l]) {
                    ++ndone;
                    continue;
                }
                if (g_wait_kind[l] == 1 && g_wait_gen[l] == g_blk_gen) continue;
                if (g_wait_kind[l] == 2 && g_wait_gen[l] == g_wv_gen[l >> 6]) continue;
                g_wait_kind[l] = 0;
                g_wave.lane = l;
                resume(l);
            }
            if (ndone == nthreads) return 0;
            if (g_progress == before) {
                int waiting = 0;
                for (int l = 0; l < nthreads; ++l) waiting += g_done[l] ? 0 : 1;
                fprintf(stderr, "emu: deadlock -- %d threads wait at a barrier the rest of the workgroup never reaches\n",
                    waiting);
                return -1;
            }
        }
    }
    static int run_wave(std::function<void()> body, size_t lds_bytes, int inst, int ninst)
    {
        return run_block(body, lds_bytes, inst, ninst, 64);
    }
} // namespace emu
} // namespace copra_hip

using namespace copra_hip;

static const double* g_cost_p[copra_hip::kMaxCosts]; // per-instance cost references for the next emu_lmpc_solve

extern "C" {

static int g_lane_hist[copra_hip::kLaneHistBins]; // violated-row histogram of the last lane pass (FusedPlan::lane_hist)
void emu_last_lane_hist(int* out) { std::copy(g_lane_hist, g_lane_hist + copra_hip::kLaneHistBins, out); }
void emu_set_cost_reference(int cost_index, const double* p) { g_cost_p[cost_index] = p; }
// the engine options of the next calls (copra_options_t; what copra_batch_create_with_options takes): every HostPlan built here starts
// from them, and the launch decisions this harness restates from copra_batch_solve consult them
void emu_set_options(const copra_options_t* opts)
{
    copra_options_t builtin {};
    builtin.struct_size = (int)sizeof(copra_options_t);
    default_options() = builtin;
    if (opts) default_options() = resolve_options(opts);
}
// per-instance right-hand sides in STACKED row order [batch][mgen] and control bounds [batch][n] for the next solve
static const double *g_row_f_inst, *g_lb_inst, *g_ub_inst;
void emu_set_instance_rows(const double* row_f, const double* lb, const double* ub)
{
    g_row_f_inst = row_f;
    g_lb_inst = lb;
    g_ub_inst = ub;
}

// Build the plan exactly as copra_batch_create does and run the fused kernel body for every instance.
int emu_lmpc_solve(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const double* A, const double* B, const double* d, const double* x0,
    double* control, double* trajectory, int* status, int* iter, int dump_instance, double* dumpQ, double* dumpc,
    double* dumpA, double* dumpb, int* sizes /* nvar, neq, nineq, lds_bytes, overflowed, rcap */, int use_specialised,
    const copra_initial_state_desc_t* is, const double* x0lb, const double* x0ub, double* x0_opt)
{
    HostPlan hp;
    copra_status_t rc = build_plan(hp, *dims, n_costs, costs, n_cstrs, cstrs, is);
    if (rc != COPRA_OK) {
        fprintf(stderr, "emu: %s\n", hp.error.c_str());
        return (int)rc;
    }
    if (std::getenv("COPRA_EMU_WANT_RIC")) (void)take_ric_layout(hp); // (what copra_batch_specialise does once the shape's kernel is compiled)
    if (const char* steps = std::getenv("COPRA_EMU_LADDER_STEPS")) { // (what adapt_layout does after solves that overflowed: steps down the tier's ladder)
        for (int q = 0; q < std::atoi(steps); ++q) {
            LdsLayout roomier {};
            if (!next_tri_layout(hp.plan, hp.plan.lds, roomier)) break;
            hp.plan.lds = roomier;
            hp.lds_bytes = (size_t)roomier.total * sizeof(double);
        }
    }
    point_plan_to_host(hp);
    FusedPlan& P = hp.plan;
    P.A = A;
    P.B = B;
    P.d = d;
    P.x0 = x0;
    P.control = control;
    P.trajectory = trajectory;
    P.status = status;
    P.iter = iter;
    P.dump_instance = dump_instance;
    P.dumpQ = dumpQ;
    P.dumpc = dumpc;
    P.dumpA = dumpA;
    P.dumpb = dumpb;
    P.x0lb = x0lb;
    P.x0ub = x0ub;
    P.x0_opt = x0_opt;
    for (int k = 0; k < kMaxCosts; ++k) P.cost_p[k] = g_cost_p[k];
    P.row_f_inst = g_row_f_inst;
    P.lb_inst = g_lb_inst;
    P.ub_inst = g_ub_inst;
    if (sizes) {
        sizes[0] = P.initial_state ? P.nx + P.n : P.n;
        sizes[1] = P.meq;
        sizes[2] = P.mineq;
        sizes[3] = (int)hp.lds_bytes;
        sizes[4] = 0;
        sizes[5] = P.lds.rcap;
        sizes[6] = P.lds.tri;
        sizes[7] = P.lds.ric;
        sizes[8] = -1; // instances finished by the one-instance-per-lane pass (-1: it did not run)
    }
    if (!A) return 0; // size query only
    if (P.use_large) { // workgroup-per-instance kernel: one resident workgroup walks the batch (persistent grid)
        std::vector<double> ws((size_t)P.large.ws_total + 8, __builtin_nan(""));
        P.ws = ws.data();
        if (dump_instance >= 0) {
            P.inst_offset = dump_instance; // as copra_batch_dump_qp launches it
            P.dump_only = 1;
        }
        int r = emu::run_block([&]() { lmpc_large_body(P); }, hp.lds_bytes, 0, 1, P.large.threads);
        if (r == 0 && dump_instance >= 0) { // ... followed by the ordinary solve
            P.inst_offset = 0;
            P.dump_only = 0;
            P.dump_instance = -1;
            r = emu::run_block([&]() { lmpc_large_body(P); }, hp.lds_bytes, 0, 1, P.large.threads);
        }
        return r != 0 ? -100 : 0;
    }
    if (P.initial_state) {
        for (int b = 0; b < dims->batch; ++b) {
            int r = emu::run_wave([&]() { islmpc_fused_body(P, b); }, hp.lds_bytes, b, dims->batch);
            if (r != 0) return -100;
        }
        return 0;
    }
    // same dispatch as the HIP launcher (select_fused_kernel): compile-time shapes for the BASELINE configs
    const int rp = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    const bool s6 = use_specialised && P.nx == 6 && rp == 6;
    const bool s2 = use_specialised && P.nx == 2 && rp == 2;
    // two-tier execution exactly as copra_batch_solve does it: compact layout first, overflow queue, full layout
    int ovf_count = 0;
    std::vector<int> ovf_list((size_t)(dims->batch > 0 ? dims->batch : 1));
    P.ovf_count = &ovf_count;
    P.ovf_list = ovf_list.data();
    P.from_list = 0;
    if (dump_instance >= 0) P.lds = hp.lds_full;
    const bool sfull = use_specialised && P.rfull > 0 && P.nx == 6 && P.nu == 3 && P.N == 20; // (headline shape, full-size costs)
    if (!s6 && !sfull && !(use_specialised && P.lds.ric) && P.lds.q1regs > 0) { // the run-time-shape body keeps Q1 in LDS (it never meets a register-Q1 layout in the library)
        LdsLayout lq {};
        if (tri_layout_with_lds_q1(P, P.lds, lq)) P.lds = lq;
    }
    bool lane_failed = false; // (the instance comes from the one-instance-per-lane pass with a failed factorisation)
    auto body = [&](const FusedPlan& PP, int b) {
        // the library's builds with a RUN-TIME horizon (copra_hip_ric.hip; select_fused_kernel) for the shapes of ric_aot_shape -- unless the
        // test asks for what copra_batch_specialise compiles (COPRA_EMU_WANT_RIC: the compile-time instantiations below)
        const bool rt = PP.lds.tri && PP.lds.ric && ric_aot_shape(PP.nx, PP.nu) && !ric_aot_exact(PP.nx, PP.nu, PP.N) && !std::getenv("COPRA_EMU_WANT_RIC");
#define EMU_RIC_RT(NX, NU)                                                                                             \
    (PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<NX, NU, 0, 6, kFusedQ1Regs, true>(PP, b, lane_failed)        \
                                    : lmpc_fused_ric_body<NX, NU, 0, 6, kFusedQ1Regs>(PP, b, lane_failed))             \
                   : (PP.stage_refs ? lmpc_fused_ric_body<NX, NU, 0, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<NX, NU, 0, 6, 0>(PP, b, lane_failed)))
        if (rt && PP.nx == 6)
            EMU_RIC_RT(6, 3);
        else if (rt && PP.nx == 4)
            EMU_RIC_RT(4, 2);
        else if (rt)
            EMU_RIC_RT(2, 1);
#undef EMU_RIC_RT
        // (shapes beyond the library's instantiations: what copra_batch_specialise compiles at run time)
        else if (PP.lds.tri && PP.lds.ric && PP.nx == 6 && PP.nu == 3 && PP.N == 12)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 12, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 12, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 12, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 12, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.nx == 4 && PP.nu == 2 && PP.N == 16)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<4, 2, 16, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<4, 2, 16, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<4, 2, 16, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<4, 2, 16, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.nx == 5 && PP.nu == 3 && PP.N == 12)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<5, 3, 12, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<5, 3, 12, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<5, 3, 12, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<5, 3, 12, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.nx == 2 && PP.nu == 1 && PP.N == 10)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<2, 1, 10, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<2, 1, 10, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<2, 1, 10, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<2, 1, 10, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.nx == 2 && PP.nu == 1 && PP.N == 40)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<2, 1, 40, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<2, 1, 40, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<2, 1, 40, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<2, 1, 40, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.N == 10) // (select_fused_kernel: the factor in Riccati form)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 10, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 10, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 10, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 10, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.N == 15)
            PP.lds.q1regs ? (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 15, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 15, 6, kFusedQ1Regs>(PP, b, lane_failed)) : (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 15, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 15, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric && PP.lds.q1regs)
            (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 20, 6, kFusedQ1Regs, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 20, 6, kFusedQ1Regs>(PP, b, lane_failed));
        else if (PP.lds.tri && PP.lds.ric)
            (PP.stage_refs ? lmpc_fused_ric_body<6, 3, 20, 6, 0, true>(PP, b, lane_failed) : lmpc_fused_ric_body<6, 3, 20, 6, 0>(PP, b, lane_failed));
        else if (PP.lds.tri && s6 && PP.lds.q1regs == kFusedQ1Regs) // (select_fused_kernel: the factor-only first tier, Q1 in registers)
            lmpc_fused_body<6, 3, 20, 6, true, kFusedQ1Regs>(PP, b);
        else if (PP.lds.tri && s6)
            lmpc_fused_body<6, 3, 20, 6, true>(PP, b);
        else if (PP.lds.tri && sfull && PP.lds.q1regs == kFusedQ1Regs)
            lmpc_fused_body<6, 3, 20, 0, true, kFusedQ1Regs>(PP, b);
        else if (PP.lds.tri && sfull)
            lmpc_fused_body<6, 3, 20, 0, true>(PP, b);
        else if (PP.lds.tri)
            lmpc_fused_body<0, 0, 0, 0, true>(PP, b);
        else if (s6)
            lmpc_fused_body<6, 3, 20, 6>(PP, b);
        else if (s2)
            lmpc_fused_body<2, 1, 10, 2>(PP, b);
        else if (use_specialised && PP.rfull > 0 && PP.nx == 6 && PP.nu == 3 && PP.N == 20) // headline shape, full-size costs
            lmpc_fused_body<6, 3, 20, 0>(PP, b);
        else
            lmpc_fused_body<0, 0, 0, 0>(PP, b);
    };
    const size_t bytes1 = (size_t)P.lds.total * sizeof(double);
    // the one-instance-per-lane pass in front of the Riccati-factor tier (lmpc_lane.hpp), as copra_batch_solve runs it
    // (copra_hip.hip: lane_pass_wanted): the instances it does not finish go through the first tier
    bool lane_pass = P.lane_tab >= 0 && !hp.large && !P.initial_state && dump_instance < 0 && (P.lds.ric || std::getenv("COPRA_EMU_LANE_FILTER")) && !default_options().no_lane_pass
        && ((P.nx == 6 && P.nu == 3) || (P.nx == 4 && P.nu == 2) || (P.nx == 5 && P.nu == 3) || (P.nx == 2 && P.nu == 1));
    for (int k = 0; k < kMaxCosts; ++k) lane_pass = lane_pass && (!P.cost_p[k] || P.lane_cref >= 0);
    std::vector<int> lane_list((size_t)dims->batch + 64, -1);
    std::vector<double> lane_ws, lane_ws2;
    int lane_cnt[4] = { 0, 0, 0, 0 }; // (as the device's: [left over | the next solve's] [+ 2: ended by the pass's own steps])
    int &lane_count = lane_cnt[0], &lane_other = lane_cnt[1];
    // ... or, where the controller's axes are decoupled, the one-(instance, axis)-per-lane solver (lmpc_axis.hpp; copra_hip.hip: axis_solver_wanted)
    // (the order of the systems' states, from the first one: copra_hip.hip, see_axis_order)
    if (dims->batch > 0 && P.A && P.B && !hp.large && !P.initial_state && axis_order_of(P.A, P.B, P.nx, P.nu) == 1 && hp.axis1_tab >= 0) {
        P.axis_order = 1;
        P.axis_tab = hp.axis1_tab;
        P.axis_cref = hp.axis1_cref;
        P.axis_rpa = hp.axis1_rpa;
        P.axis_const = hp.axis1_const;
    }
    // (independent of the pass: chains of three states per control have no build of it)
    bool axis_pass = P.axis_tab >= 0 && !hp.large && !P.initial_state && dump_instance < 0 && !default_options().no_lane_pass && (lane_pass || P.nx == 3 * P.nu || P.nx == P.nu)
        && !default_options().no_axis_solver && axis_solver_nmax(P.nx, P.nu, P.N) > 0
        && (!(P.row_f_inst || P.lb_inst || P.ub_inst) || (P.axis_const && (P.lb_inst == nullptr) == (P.ub_inst == nullptr)));
    for (int k = 0; k < kMaxCosts; ++k) axis_pass = axis_pass && (!P.cost_p[k] || (P.axis_cref >= 0 && k < P.ncost));
    if (axis_pass && P.stage_refs) { // (reference trajectories: copra_hip.hip, axis_solver_wanted)
        int oB = 0, oR = 0, rcs = 0;
        (void)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmax, oB, oR, rcs);
        axis_pass = P.axis_cref >= 0 && P.N * (P.nx / P.nu + 1) <= rcs;
    }
    if (axis_pass) {
        int on_spare = 0;
        const int groups = axis_grid(P.nu, dims->batch, on_spare);
        std::vector<int> axis_acc((size_t)groups / (size_t)P.nu + 2, 0);
        P.axis_waves = groups;
        P.axis_pf = groups > 2 ? 2 : 0; // (the touches of a later wave's systems: exercised, without effect here)
        P.axis_acc = axis_acc.data();
        P.lane_list = lane_list.data();
        P.lane_count = &lane_count;
        P.lane_zero = &lane_other;
        std::fill(g_lane_hist, g_lane_hist + kLaneHistBins, 0);
        P.lane_hist = g_lane_hist;
        int oB = 0, oR = 0, rcs = 0;
        const size_t abytes = (size_t)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmax, oB, oR, rcs) * sizeof(double); // (sized for the library's builds; the two-slot test build needs less)
        const bool small_q = std::getenv("COPRA_EMU_AXIS_QMAX2") != nullptr; // (tests: an active set that outgrows the lane -- the hand-over to the tier)
        for (int g = 0; g < groups; ++g) {
            int r = emu::run_wave([&]() {
#define COPRA_EMU_AXIS_B(NU, NMAX, Q, EXACT)                                                                                     \
    (P.axis_const ? (P.axis_rpa <= 1 ? lmpc_axis_body<2, NU, NMAX, Q, EXACT, true, 1>(P, g) : lmpc_axis_body<2, NU, NMAX, Q, EXACT, true, 2>(P, g)) \
                  : lmpc_axis_body<2, NU, NMAX, Q, false, false, 2>(P, g))
#define COPRA_EMU_AXIS(NU)                                                                                                       \
    (small_q ? COPRA_EMU_AXIS_B(NU, 20, 2, false)                                                                                \
             : P.N == 20 && NU == 3 && !P.stage_refs ? COPRA_EMU_AXIS_B(NU, 20, kAxisQmax, true)                                  \
             : P.N <= 20 ? COPRA_EMU_AXIS_B(NU, 20, kAxisQmax, false) : COPRA_EMU_AXIS_B(NU, 31, kAxisQmax, false))
                if (P.nx == P.nu && P.N > 20) { // one state per control in the plane, horizons up to 31
                    const bool ct = P.axis_const && P.axis_rpa <= 1;
                    (small_q ? lmpc_axis_body<1, 2, 31, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<1, 2, 31, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<1, 2, 31, kAxisQmax, false, false, 2>(P, g));
                } else if (P.nx == P.nu) { // one state per control (copra_hip_axis3.hip)
                    const bool ct = P.axis_const && P.axis_rpa <= 1;
                    if (P.nu == 3) (small_q ? lmpc_axis_body<1, 3, 20, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<1, 3, 20, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<1, 3, 20, kAxisQmax, false, false, 2>(P, g));
                    else (small_q ? lmpc_axis_body<1, 2, 20, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<1, 2, 20, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<1, 2, 20, kAxisQmax, false, false, 2>(P, g));
                } else if (P.nx == 3 * P.nu) { // chains of three states per control (copra_hip_axis3.hip)
                    const bool ct = P.axis_const && P.axis_rpa <= 1;
                    if (P.nu == 3) (small_q ? lmpc_axis_body<3, 3, 20, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<3, 3, 20, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<3, 3, 20, kAxisQmax, false, false, 2>(P, g));
                    else (small_q ? lmpc_axis_body<3, 2, 20, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<3, 2, 20, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<3, 2, 20, kAxisQmax, false, false, 2>(P, g));
                } else if (P.nu == 3 && P.N == 21) {
                    const bool ct = P.axis_const && P.axis_rpa <= 1;
                    (small_q ? lmpc_axis_body<2, 3, 21, 2, false, false, 2>(P, g) : ct ? lmpc_axis_body<2, 3, 21, kAxisQmax, false, true, 1>(P, g) : lmpc_axis_body<2, 3, 21, kAxisQmax, false, false, 2>(P, g));
                } else if (P.nu == 3) COPRA_EMU_AXIS(3);
                else COPRA_EMU_AXIS(2);
#undef COPRA_EMU_AXIS
            }, abytes, g, groups);
            if (r != 0) return -100;
        }
        P.lane_hist = nullptr;
        // the second chance of what it listed (copra_lmpc_axis_list_kernel): room for kAxisQmaxBig active constraints per lane, instances from the list
        std::vector<int> list2((size_t)dims->batch + 64, -1);
        int cnt2[4] = { 0, 0, 0, 0 }; // (as the first launch's: [left over | - | ended by its steps | -])
        int& count2 = cnt2[0];
        if (!std::getenv("COPRA_EMU_AXIS_NO_SECOND_CHANCE")) {
            FusedPlan Pl = P;
            Pl.axis_list_in = lane_list.data();
            Pl.axis_list_count = &lane_count;
            Pl.lane_list = list2.data();
            Pl.lane_count = &count2;
            Pl.lane_zero = nullptr;
            Pl.axis_pf = 0;
            const size_t lbytes = (size_t)axis_lds_doubles(P.nx, P.nu, P.N, P.axis_rpa, kAxisQmaxBig, oB, oR, rcs) * sizeof(double);
            const int ipw = 64 / P.nu;
            for (int g = 0; g * ipw < lane_count; ++g) {
                int r = emu::run_wave([&]() {
#define COPRA_EMU_AXIS_L(NU, NMAX)                                                                                       \
    (Pl.axis_const ? lmpc_axis_body<2, NU, NMAX, kAxisQmaxBig, false, true, 2, true>(Pl, g) : lmpc_axis_body<2, NU, NMAX, kAxisQmaxBig, false, false, 2, true>(Pl, g))
                    if (P.nx == P.nu && P.N > 20) lmpc_axis_body<1, 2, 31, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                    else if (P.nx == P.nu) {
                        if (P.nu == 3) lmpc_axis_body<1, 3, 20, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                        else lmpc_axis_body<1, 2, 20, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                    } else if (P.nx == 3 * P.nu) {
                        if (P.nu == 3) lmpc_axis_body<3, 3, 20, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                        else lmpc_axis_body<3, 2, 20, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                    } else if (P.nu == 3 && P.N == 21) lmpc_axis_body<2, 3, 21, kAxisQmaxBig, false, false, 2, true>(Pl, g);
                    else if (P.nu == 3) COPRA_EMU_AXIS_L(3, 20);
                    else if (P.N <= 20) COPRA_EMU_AXIS_L(2, 20);
                    else COPRA_EMU_AXIS_L(2, 31);
#undef COPRA_EMU_AXIS_L
                }, lbytes, g, 1);
                if (r != 0) return -100;
            }
            if (std::getenv("COPRA_EMU_AXIS_REPORT")) { // (tools: who is listed, and why)
                int bad1 = 0, bad2 = 0;
                for (int k = 0; k < lane_count; ++k) bad1 += lane_list[(size_t)k] < 0 ? 1 : 0;
                for (int k = 0; k < count2; ++k) bad2 += list2[(size_t)k] < 0 ? 1 : 0;
                std::fprintf(stderr, "emu axis solver: %d of %d instances listed by the first launch (%d with a failed factorisation), %d by the second chance (%d failed):",
                    lane_count, dims->batch, bad1, count2, bad2);
                for (int k = 0; k < count2 && k < 24; ++k) std::fprintf(stderr, " %d%s", list2[(size_t)k] & 0x7fffffff, list2[(size_t)k] < 0 ? "!" : "");
                std::fprintf(stderr, "\n");
            }
            lane_list = list2;
            lane_count = count2;
        }
        P.lane_from_list = 1;
        P.lane_spec = P.lds.ricC ? 1 : 0; // (nothing is handed over: the tier sweeps for itself)
        P.lane_handover = 0;
        for (int k = 0; k < lane_count; ++k) {
            const int raw = lane_list[(size_t)k], b = raw & 0x7fffffff;
            lane_failed = raw < 0;
            int r = emu::run_wave([&]() { body(P, b); }, bytes1, b, dims->batch);
            if (r != 0) return -100;
        }
        lane_failed = false;
        P.lane_from_list = 0;
        if (sizes) sizes[8] = dims->batch - lane_count;
    } else if (lane_pass) {
        const int groups = (dims->batch + 63) / 64;
        P.lane_bp = (dims->batch + 63) / 64 * 64 + 64;
        lane_ws.assign((size_t)P.N * lane_ws_rows(P.nx, P.nu) * P.lane_bp, 0.0);
        P.lane_ws = lane_ws.data();
        lane_ws2.assign((size_t)P.lane_bp * lane_ws2_doubles(P.nx, P.nu, P.N), 0.0);
        P.lane_ws2 = lane_ws2.data();
        P.lane_list = lane_list.data();
        P.lane_count = &lane_count;
        P.lane_zero = &lane_other;
        P.lane_spec = (P.lds.ricC && !default_options().no_lane_spec) ? 1 : 0; // (as copra_batch_solve: the two forms of the pass)
        P.lane_handover = (P.lds.ricC && !default_options().no_lane_handover && !P.lane_spec) ? 1 : 0;
        std::fill(g_lane_hist, g_lane_hist + kLaneHistBins, 0);
        P.lane_hist = g_lane_hist; // (what the first solve of a controller asks of the pass: copra_batch_solve picks the tier's layout from it)
        for (int g = 0; g < groups; ++g) {
            int oHl = 0;
            const size_t lbytes = (size_t)(lane_lds_doubles(P.nx, P.nu, oHl) + P.lane_tlds) * sizeof(double);
            int r = emu::run_wave([&]() {
#define COPRA_EMU_LANE(NX, NU)                                                                                                    \
    (P.lane_spec ? (P.stage_refs ? lmpc_lane_body<NX, NU, true, true>(P, g) : lmpc_lane_body<NX, NU, false, true>(P, g))            \
                 : (P.stage_refs ? lmpc_lane_body<NX, NU, true, false>(P, g) : lmpc_lane_body<NX, NU, false, false>(P, g)))
                if (P.nx == 6) COPRA_EMU_LANE(6, 3);
                else if (P.nx == 4) COPRA_EMU_LANE(4, 2);
                else if (P.nx == 5) COPRA_EMU_LANE(5, 3);
                else COPRA_EMU_LANE(2, 1);
#undef COPRA_EMU_LANE
            }, lbytes, g, groups);
            if (r != 0) return -100;
        }
        P.lane_hist = nullptr;
        P.lane_from_list = 1;
        for (int k = 0; k < lane_count; ++k) {
            const int raw = lane_list[(size_t)k], b = raw & 0x7fffffff;
            lane_failed = raw < 0;
            int r = emu::run_wave([&]() { body(P, b); }, bytes1, b, dims->batch);
            if (r != 0) return -100;
        }
        lane_failed = false;
        P.lane_from_list = 0;
        if (sizes) sizes[8] = dims->batch - lane_count;
    } else {
    for (int b = 0; b < dims->batch; ++b) {
        int r = emu::run_wave([&]() { body(P, b); }, bytes1, b, dims->batch);
        if (r != 0) return -100;
    }
    }
    if (sizes) sizes[4] = ovf_count;
    if (ovf_count > 0) {
        FusedPlan P2 = P;
        P2.lds = hp.lds_full;
        for (int k = 0; k < ovf_count; ++k) {
            const int b = ovf_list[(size_t)k];
            int r = emu::run_wave([&]() { body(P2, b); }, hp.lds_full_bytes, b, dims->batch);
            if (r != 0) return -100;
        }
    }
    return 0;
}

// The stage-wise Riccati interior-point body (lmpc_riccati.hpp) for every instance; returns -200 when the controller is
// not stage-wise (stage_plan.hpp), else 0.  not_converged[0] = number of instances it queued for the Goldfarb-Idnani path.
int emu_lmpc_solve_riccati(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const double* A, const double* B, const double* d, const double* x0, double* control,
    double* trajectory, int* status, int* iter, const copra_initial_state_desc_t* is, const double* x0lb, const double* x0ub,
    double* x0_opt, int* not_converged)
{
    HostPlan hp;
    copra_status_t rc = build_plan(hp, *dims, n_costs, costs, n_cstrs, cstrs, is);
    if (rc != COPRA_OK) {
        fprintf(stderr, "emu: %s\n", hp.error.c_str());
        return (int)rc;
    }
    point_plan_to_host(hp);
    FusedPlan& P = hp.plan;
    HostStagePlan hs;
    build_stage_plan(hp, hs, g_lb_inst != nullptr);
    if (!hs.eligible) {
        fprintf(stderr, "emu: not stage-wise: %s\n", hs.why.c_str());
        return -200;
    }
    point_stage_plan_to_host(hs);
    P.A = A, P.B = B, P.d = d, P.x0 = x0;
    P.control = control, P.trajectory = trajectory, P.status = status, P.iter = iter;
    P.x0lb = x0lb, P.x0ub = x0ub, P.x0_opt = x0_opt;
    for (int k = 0; k < kMaxCosts; ++k) P.cost_p[k] = g_cost_p[k];
    P.row_f_inst = g_row_f_inst;
    P.lb_inst = g_lb_inst;
    P.ub_inst = g_ub_inst;
    int ovf_count = 0;
    std::vector<int> ovf_list((size_t)(dims->batch > 0 ? dims->batch : 1));
    P.ovf_count = &ovf_count;
    P.ovf_list = ovf_list.data();
    std::vector<double> ws((size_t)hs.sp.ws_total + 8, __builtin_nan(""));
    hs.sp.ws = ws.data();
    int next_instance = 0; // the work queue of the kernel
    hs.sp.next_instance = &next_instance;
    std::vector<double> rec_ws((size_t)hs.sp.N * kRfKStride + 8, __builtin_nan("")); // (the LDS-resident kernel's stage records: one wave here)
    hs.sp.rec_ws = rec_ws.data();
    const StagePlan& S = hs.sp;
    // same dispatch as the HIP launcher: the LDS-resident kernel (lmpc_riccati_mfma.hpp) where the plan fits it ...
    if (not_converged) not_converged[1] = 0;
    bool refs = false;
    for (int k = 0; k < kMaxCosts; ++k) refs = refs || g_cost_p[k] != nullptr;
    if (S.fast_ok && !refs && !default_options().no_ric_fast) {
        int rf = emu::run_wave([&]() { lmpc_riccati_mfma_body(P, S); }, (size_t)S.fast_lds_doubles * sizeof(double), 0, 1);
        if (not_converged) not_converged[0] = ovf_count, not_converged[1] = 1;
        return rf != 0 ? -100 : 0;
    }
    if (std::getenv("COPRA_EMU_DEBUG")) fprintf(stderr, "emu: streaming Riccati kernel (%s)\n", hs.fast_why.c_str());
    // ... else the streaming one (select_riccati_kernel)
    int r = emu::run_wave(
        [&]() {
            if (S.nx == 12 && S.nu == 6)
                lmpc_riccati_body<12, 6>(P, S);
            else if (S.nx == 2 && S.nu == 1 && !std::getenv("COPRA_EMU_GENERIC"))
                lmpc_riccati_body<2, 1>(P, S);
            else
                lmpc_riccati_body<0, 0>(P, S);
        },
        (size_t)S.lds_doubles * sizeof(double), 0, 1);
    if (not_converged) not_converged[0] = ovf_count;
    return r != 0 ? -100 : 0;
}

// Shared-model fast path exactly as copra_batch_set_shared_system + copra_batch_solve run it: nx + 1 probe instances of
// the fused body give c(x0) = c0 + C1 x0, one "prepare" run stores the factorisation, then lmpc_shared_body per
// instance (compact layout first, overflow queue, full layout).  A, B, d: ONE system; x0: [batch][nx].
int emu_lmpc_solve_shared(const copra_dims_t* dims, int n_costs, const copra_cost_desc_t* costs, int n_cstrs,
    const copra_cstr_desc_t* cstrs, const double* A, const double* B, const double* d, const double* x0,
    double* control, double* trajectory, int* status, int* iter, int* sizes /* overflowed */,
    int* warm_set /* [batch][kWarmCap], kept by the caller across ticks; nullptr: cold starts */)
{
    HostPlan hp;
    copra_status_t rc = build_plan(hp, *dims, n_costs, costs, n_cstrs, cstrs, nullptr);
    if (rc != COPRA_OK) {
        fprintf(stderr, "emu: %s\n", hp.error.c_str());
        return (int)rc;
    }
    if (hp.large) return (int)COPRA_ERR_UNSUPPORTED;
    point_plan_to_host(hp);
    FusedPlan P = hp.plan;
    const int nx = P.nx, nu = P.nu, N = P.N, n = P.n, X = P.X, np1 = nx + 1;
    const ModelLayout m = model_layout(nx, nu, N, n, X, hp.lds_full.ldj, P.mgen);
    std::vector<double> model((size_t)m.total, 0.0);
    std::vector<double> Ap((size_t)nx * nx * np1), Bp((size_t)nx * nu * np1), dp((size_t)nx * np1), xp((size_t)nx * np1, 0.0);
    for (int a = 0; a < np1; ++a) {
        memcpy(&Ap[(size_t)a * nx * nx], A, sizeof(double) * nx * nx);
        memcpy(&Bp[(size_t)a * nx * nu], B, sizeof(double) * nx * nu);
        memcpy(&dp[(size_t)a * nx], d, sizeof(double) * nx);
        if (a > 0) xp[(size_t)a * nx + (a - 1)] = 1.0;
    }
    std::vector<double> dQ((size_t)n * n), dC((size_t)n * np1);
    int ovf_count = 0;
    std::vector<int> ovf_list((size_t)(dims->batch > 0 ? dims->batch : 1));
    P.ovf_count = &ovf_count;
    P.ovf_list = ovf_list.data();
    P.from_list = 0;
    P.control = control;
    P.trajectory = trajectory;
    P.status = status;
    P.iter = iter;
    FusedPlan Q = P; // prepare launches
    Q.A = Ap.data(), Q.B = Bp.data(), Q.d = dp.data(), Q.x0 = xp.data();
    Q.batch = np1;
    Q.lds = hp.lds_full;
    auto fused = [&](const FusedPlan& PP, int b) {
        const int rp = specialised_cost_rows(PP.nx, PP.nu, PP.N, PP.rmax, PP.rfull);
        if (PP.nx == 6 && rp == 6)
            lmpc_fused_body<6, 3, 20, 6>(PP, b);
        else if (PP.nx == 2 && rp == 2)
            lmpc_fused_body<2, 1, 10, 2>(PP, b);
        else
            lmpc_fused_body<0, 0, 0, 0>(PP, b);
    };
    for (int a = 0; a < np1; ++a) {
        Q.dump_instance = a;
        Q.dump_only = 1;
        Q.dumpQ = dQ.data();
        Q.dumpc = dC.data() + (size_t)a * n;
        if (emu::run_wave([&]() { fused(Q, a); }, hp.lds_full_bytes, a, np1) != 0) return -100;
    }
    Q.dump_instance = 0;
    Q.dump_only = 0;
    Q.dumpQ = Q.dumpc = nullptr;
    Q.model_out = model.data();
    if (emu::run_wave([&]() { fused(Q, 0); }, hp.lds_full_bytes, 0, np1) != 0) return -100;
    for (int j = 0; j < n; ++j) model[(size_t)m.c0 + j] = dC[(size_t)j];
    for (int a = 0; a < nx; ++a)
        for (int j = 0; j < n; ++j) model[(size_t)m.C1 + (size_t)a * n + j] = dC[(size_t)(a + 1) * n + j] - dC[(size_t)j];
    { // x = -Qinv (c0 + C1 x0) multiplied out once (prepare_shared_model in copra_hip.hip)
        const int ld = hp.lds_full.ldj;
        for (int a = 0; a < np1; ++a)
            for (int i = 0; i < n; ++i) {
                double acc = 0.0;
                const double* col = (a == 0) ? &model[(size_t)m.c0] : &model[(size_t)m.C1 + (size_t)(a - 1) * n];
                for (int j = 0; j < n; ++j) acc += model[(size_t)m.Qinv + (size_t)j * ld + i] * col[j];
                if (a == 0)
                    model[(size_t)m.xu0 + i] = -acc;
                else
                    model[(size_t)m.K1 + (size_t)(a - 1) * n + i] = -acc;
            }
    }
    P.model = model.data();
    P.model_rtot = 0;
    P.x0 = x0;
    P.warm_set = warm_set;
    // per-instance cost references (emu_set_cost_reference): only the records form with the pass in front takes them here (the delta sweep of
    // lmpc_lane_shared_body); the emulated lmpc_shared.hpp path has no reference columns in its model -> refused below
    bool sh_refs = false;
    for (int k = 0; k < kMaxCosts; ++k) sh_refs = sh_refs || g_cost_p[k] != nullptr;
    // Riccati-factor tier in shared-model mode (as copra_batch_solve: cold starts, controller-wide references or the pass's delta sweep): one prepare
    // run of the body leaves the stage records, bkd, G and the row norms; the first tier copies them instead of sweeping
    std::vector<double> ric_model;
    // (as copra_batch_set_shared_system: the library's instantiations of the tier -- the compile-time horizons of (6, 3) and the run-time-horizon
    //  builds of the integrator shapes)
    const bool ric_shared = P.lds.ric && (ric_aot_exact(P.nx, P.nu, P.N) || ric_aot_shape(P.nx, P.nu)) && P.lds.q1regs == kFusedQ1Regs && !warm_set
        && !default_options().no_ric_shared;
    auto ric_tier = [&](const FusedPlan& PP, int b) {
#define EMU_RIC_SH(NX, NU, NH) (PP.stage_refs ? lmpc_fused_ric_body<NX, NU, NH, 6, kFusedQ1Regs, true>(PP, b) : lmpc_fused_ric_body<NX, NU, NH, 6, kFusedQ1Regs>(PP, b))
        if (PP.nx == 6 && PP.N == 20)
            EMU_RIC_SH(6, 3, 20);
        else if (PP.nx == 6 && PP.N == 15)
            EMU_RIC_SH(6, 3, 15);
        else if (PP.nx == 6 && PP.N == 10)
            EMU_RIC_SH(6, 3, 10);
        else if (PP.nx == 6)
            EMU_RIC_SH(6, 3, 0);
        else if (PP.nx == 4)
            EMU_RIC_SH(4, 2, 0);
        else
            EMU_RIC_SH(2, 1, 0);
#undef EMU_RIC_SH
    };
    if (ric_shared) {
        int oBk, oG, oNb;
        ric_model.assign((size_t)ric_model_offsets(nx, nu, N, P.mgen, oBk, oG, oNb), 0.0);
        FusedPlan R = P;
        R.A = A, R.B = B, R.d = d, R.x0 = xp.data(); // (x0 = 0: the records do not depend on it)
        R.batch = 1;
        R.dump_instance = 0;
        R.ric_model_out = ric_model.data();
        if (emu::run_wave([&]() { ric_tier(R, 0); }, hp.lds_bytes, 0, 1) != 0) return -100;
        P.ric_model = ric_model.data();
    } else { // as copra_batch_set_shared_system: the shared-model kernels keep Q1 in LDS
        LdsLayout lq {};
        if (tri_layout_with_lds_q1(P, P.lds, lq)) {
            P.lds = lq;
            hp.lds_bytes = (size_t)lq.total * sizeof(double);
        }
    }
    auto shared = [&](const FusedPlan& PP, int b) {
        if (PP.lds.ric && PP.ric_model)
            ric_tier(PP, b);
        else if (PP.lds.tri && PP.nx == 6 && PP.nu == 3 && PP.N == 20) // (select_shared_kernel: factor-only first tier)
            lmpc_shared_body<6, 3, 20, true>(PP, b);
        else if (PP.lds.tri)
            lmpc_shared_body<0, 0, 0, true>(PP, b);
        else if (PP.nx == 6 && PP.nu == 3 && PP.N == 20)
            lmpc_shared_body<6, 3, 20>(PP, b);
        else if (PP.nx == 2 && PP.nu == 1 && PP.N == 10)
            lmpc_shared_body<2, 1, 10>(PP, b);
        else
            lmpc_shared_body<0, 0, 0>(PP, b);
    };
    // in front of the Riccati-factor tier in shared-model mode: the one-instance-per-lane pass in its shared-model form (as copra_batch_solve)
    std::vector<int> lane_list((size_t)dims->batch + 64, -1);
    int lane_cnt[4] = { 0, 0, 0, 0 };
    int &lane_count = lane_cnt[0], &lane_other = lane_cnt[1];
    int lane_finished = -1;
    const bool lane_sh = ric_shared && P.lane_tab >= 0 && P.lds.ricC && !P.row_f_inst && !default_options().no_lane_pass;
    if (sh_refs && !(lane_sh && P.lane_cref >= 0)) {
        fprintf(stderr, "emu: shared-model references need the records form with the pass: ric_shared %d lane_tab %d ricC %d lane_cref %d\n", (int)ric_shared, P.lane_tab, (int)P.lds.ricC, P.lane_cref);
        return (int)COPRA_ERR_UNSUPPORTED;
    }
    std::vector<double> lane_ws_sh;
    if (lane_sh) { // (select_lane_shared_kernel: every shape of the tier)
        const int groups = (dims->batch + 63) / 64;
        P.lane_bp = groups * 64;
        if (sh_refs) {
            for (int k = 0; k < kMaxCosts; ++k) P.cost_p[k] = g_cost_p[k];
            lane_ws_sh.assign((size_t)P.N * lane_ws_rows(P.nx, P.nu) * (P.lane_bp + 64), 0.0);
            P.lane_ws = lane_ws_sh.data();
        }
        P.lane_list = lane_list.data();
        P.lane_count = &lane_count;
        P.lane_zero = &lane_other;
        P.lane_spec = default_options().no_lane_spec ? 0 : 1; // (copra_batch_solve, solve_shared_model)
        int oHl = 0;
        const size_t lbytes = (size_t)(lane_lds_doubles(P.nx, P.nu, oHl) + P.lane_tlds) * sizeof(double);
        for (int g = 0; g < groups; ++g)
            if (emu::run_wave([&]() {
                    if (P.nx == 6)
                        P.lane_spec ? lmpc_lane_shared_body<6, 3, true>(P, g) : lmpc_lane_shared_body<6, 3, false>(P, g);
                    else if (P.nx == 4)
                        P.lane_spec ? lmpc_lane_shared_body<4, 2, true>(P, g) : lmpc_lane_shared_body<4, 2, false>(P, g);
                    else
                        P.lane_spec ? lmpc_lane_shared_body<2, 1, true>(P, g) : lmpc_lane_shared_body<2, 1, false>(P, g);
                }, lbytes, g, groups) != 0) return -100;
        P.lane_from_list = 1;
        P.lane_handover = 1;
        for (int k = 0; k < lane_count; ++k) {
            const int b = lane_list[(size_t)k];
            if (emu::run_wave([&]() { shared(P, b); }, hp.lds_bytes, b, dims->batch) != 0) return -100;
        }
        P.lane_from_list = 0;
        lane_finished = dims->batch - lane_count;
    } else {
    for (int b = 0; b < dims->batch; ++b)
        if (emu::run_wave([&]() { shared(P, b); }, hp.lds_bytes, b, dims->batch) != 0) return -100;
    }
    if (sizes) sizes[0] = ovf_count;
    if (sizes) sizes[1] = ric_shared ? 1 : 0;
    if (sizes) sizes[2] = lane_finished;
    if (ovf_count > 0 && sh_refs) return (int)COPRA_ERR_UNSUPPORTED; // (the emulated second tier has no reference columns in its model)
    if (ovf_count > 0) {
        FusedPlan P2 = P;
        P2.lds = hp.lds_full;
        P2.ric_model = nullptr;
        for (int k = 0; k < ovf_count; ++k) {
            const int b = ovf_list[(size_t)k];
            if (emu::run_wave([&]() { shared(P2, b); }, hp.lds_full_bytes, b, dims->batch) != 0) return -100;
        }
    }
    return 0;
}

int emu_qp_dense(int batch, int n, int neq, int nineq, const double* Q, const double* c, const double* Aeq,
    const double* beq, const double* Aineq, const double* bineq, const double* XL, const double* XU, double* x,
    int* fail, int* iter)
{
    DensePlan P {};
    P.n = n;
    P.meq = neq;
    P.mineq = nineq;
    P.mgen = neq + nineq;
    P.mtotal = P.mgen + 2 * n;
    P.batch = batch;
    P.Q = Q;
    P.c = c;
    P.Aeq = Aeq;
    P.beq = beq;
    P.Aineq = Aineq;
    P.bineq = bineq;
    P.XL = XL;
    P.XU = XU;
    P.x = x;
    P.fail = fail;
    P.iter = iter;
    P.vsmall = qpgen2_vsmall();
    P.max_iter = 50 * (n + P.mtotal) + 100;
    if (n > kLargeMaxN) return (int)COPRA_ERR_UNSUPPORTED;
    if (n > 64) { // workgroup-per-problem kernel, same launch geometry as copra_qp_solve_dense_batch
        layout_large_solver(P.llds, 0, n, P.mgen, P.meq, P.mtotal);
        const int threads = (n + 63) & ~63;
        const int ld = large_ld(n);
        std::vector<double> ws((size_t)2 * n * ld, __builtin_nan(""));
        P.ws = ws.data();
        // one resident workgroup walking the batch (the persistent-grid loop of the kernel)
        int r = emu::run_block([&]() { qp_dense_large_body(P); }, (size_t)P.llds.total * sizeof(double), 0, 1, threads);
        return r != 0 ? -100 : 0;
    }
    (void)layout_lds(P.lds, 0, 0, 0, n, 0, 1, P.mgen, P.meq, P.mtotal, false);
    const size_t lds_bytes = (size_t)P.lds.total * sizeof(double);
    for (int b = 0; b < batch; ++b) {
        int r = emu::run_wave([&]() { qp_dense_body(P, b); }, lds_bytes, b, batch);
        if (r != 0) return -100;
    }
    return 0;
}
}
