// copra_hip_jit.hip -- run-time specialisation: copra_batch_specialise / copra_qp_dense_specialise compile the kernel bodies for ONE shape with
// `hipcc --genco` from the headers next to libcopra_hip.so and keep the code object in a cache keyed by shape and source hash.
#include "engine.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

#include <cstdio>
#include <cstdlib>
#include <cstring>

// ---- run-time specialisation ------------------------------------------------------------------------------------
// The kernel bodies are templates on (xDim, uDim, nrStep, cost rows); the library ships instantiations for the
// BASELINE shapes and a run-time-shape one that is ~2.5x slower on the same problem (headline shape: 13.4 vs 5.4 M
// solves/s).  copra_batch_specialise compiles the instantiation for THIS controller's shape with hipcc --genco from the
// headers next to the library, keeps the code object in a cache directory and launches it through the module API.
static std::string library_dir()
{
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&copra_abi_version), &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.find_last_of('/');
        return k == std::string::npos ? std::string(".") : p.substr(0, k);
    }
    return ".";
}

// compile `source` (a translation unit that includes headers from the library's directory) into a code object named
// `key` in the cache directory, unless it is already there; returns its path in `obj`
// `check` (may be null) sees every code object BEFORE it is loaded -- cached or freshly compiled; a non-zero answer removes the object
// from the cache and fails the call, so nothing the caller's check turned away ever reaches a handle.
static copra_status_t jit_compile_unchecked(const std::string& key, const std::string& source, const char* cache_dir, std::string& obj);
static copra_status_t jit_compile(const std::string& key, const std::string& source, const char* cache_dir, std::string& obj,
    copra_code_object_check_t check = nullptr, void* user = nullptr)
{
    const copra_status_t rc = jit_compile_unchecked(key, source, cache_dir, obj);
    if (rc != COPRA_OK || !check) return rc;
    if (check(obj.c_str(), user) != 0) {
        (void)unlink(obj.c_str());
        return fail(COPRA_ERR_RUNTIME, "run-time specialisation: the caller's check turned the compiled code object away (" + obj
                + " removed; the controller keeps the library's kernels)");
    }
    return COPRA_OK;
}

static copra_status_t jit_compile_unchecked(const std::string& key, const std::string& source, const char* cache_dir, std::string& obj)
{
    const std::string src_dir = library_dir();
    std::string dir = cache_dir ? cache_dir : "";
    if (dir.empty()) {
        const char* e = std::getenv("COPRA_JIT_CACHE");
        const char* home = std::getenv("HOME");
        dir = e ? e : (std::string(home ? home : "/tmp") + "/.cache/copra_amd");
    }
    (void)mkdir((dir.substr(0, dir.find_last_of('/'))).c_str(), 0755);
    (void)mkdir(dir.c_str(), 0755);
    // the code object depends on the exact sources it was compiled from: the hash of those sources is compiled into this
    // library (Makefile: COPRA_SRC_HASH), so a cache left by another build of the library is never picked up
    const std::string stamp = std::string(copra_source_hash()).substr(0, 12);
    obj = dir + "/" + key + "_" + stamp + ".hsaco";
    if (access(obj.c_str(), R_OK) == 0) return COPRA_OK;
    const std::string src = obj + "." + std::to_string((long)getpid()) + ".hip";
    FILE* f = fopen(src.c_str(), "w");
    if (!f) return fail(COPRA_ERR_RUNTIME, "run-time specialisation: cannot write to the cache directory " + dir);
    fputs(source.c_str(), f);
    fclose(f);
    const char* hipcc_env = std::getenv("HIPCC");
    const std::string hipcc = hipcc_env ? hipcc_env : "/opt/rocm/bin/hipcc";
    const std::string tmp = obj + "." + std::to_string((long)getpid()) + ".tmp";
    const std::string log = src + ".log";
    const std::string inc = "-I" + src_dir;
    // argv, no shell: paths with quotes or spaces cannot break (or inject into) the command
    const char* argv[] = { hipcc.c_str(), "--offload-arch=gfx950", "-O3", "-std=c++17", "--genco", inc.c_str(), "-o", tmp.c_str(),
        src.c_str(), nullptr };
    int rc = -1;
    {
        posix_spawn_file_actions_t fa;
        posix_spawn_file_actions_init(&fa);
        posix_spawn_file_actions_addopen(&fa, 1, log.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        posix_spawn_file_actions_adddup2(&fa, 1, 2);
        pid_t pid = 0;
        if (posix_spawn(&pid, hipcc.c_str(), &fa, nullptr, const_cast<char* const*>(argv), environ) == 0) {
            int st = 0;
            if (waitpid(pid, &st, 0) == pid && WIFEXITED(st)) rc = WEXITSTATUS(st);
        }
        posix_spawn_file_actions_destroy(&fa);
    }
    (void)unlink(src.c_str());
    if (rc != 0 || rename(tmp.c_str(), obj.c_str()) != 0)
        return fail(COPRA_ERR_RUNTIME, "run-time specialisation: hipcc --genco failed (see " + log + ")");
    (void)unlink(log.c_str());
    return COPRA_OK;
}

// dense-QP kernels compiled for a fixed number of variables (copra_qp_dense_specialise): (n, lanes per QP) -> kernel
struct DenseJit {
    int n, lanes;
    hipFunction_t fn;
};

static std::vector<DenseJit> g_dense_jit; // guarded by g_dense_jit_mu

static std::mutex g_dense_jit_mu;


hipFunction_t dense_jit_lookup(int n, int lanes)
{
    std::lock_guard<std::mutex> lock(g_dense_jit_mu);
    for (const DenseJit& d : g_dense_jit)
        if (d.n == n && d.lanes == lanes) return d.fn;
    return nullptr;
}

extern "C" {


copra_status_t copra_qp_dense_specialise(int n, const char* cache_dir)
{
    if (n <= 0 || n > kWave) return COPRA_OK; // (the workgroup-per-problem kernel has no shape parameters)
    std::lock_guard<std::mutex> lock(g_dense_jit_mu);
    for (int lanes : { 64, 32, 16 }) {
        if (lanes < n) continue;
        bool have = false;
        for (const DenseJit& d : g_dense_jit) have = have || (d.n == n && d.lanes == lanes);
        if (have) continue;
        char key[96], source[1024];
        snprintf(key, sizeof key, "copra_jit_dense_%d_l%d", n, lanes);
        if (lanes == 64)
            snprintf(source, sizeof source,
                "#include <hip/hip_runtime.h>\n#include \"qp_dense.hpp\"\nusing namespace copra_hip;\n"
                "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_dense(const DensePlan P)\n"
                "{ qp_dense_body<%d>(P, (int)blockIdx.x); }\n", n);
        else
            snprintf(source, sizeof source,
                "#define COPRA_WAVE_WIDTH %d\n#include \"packed_impl.inc\"\n"
                "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_dense(const DensePlan P)\n"
                "{ const int inst = instance_id(); if (inst < P.batch) qp_dense_body<%d>(P, inst); }\n", lanes, n);
        std::string obj;
        const copra_status_t rc = jit_compile(key, source, cache_dir, obj);
        if (rc != COPRA_OK) return rc;
        hipModule_t mod = nullptr;
        HIP_TRY(hipModuleLoad(&mod, obj.c_str()));
        hipFunction_t fn = nullptr;
        HIP_TRY(hipModuleGetFunction(&fn, mod, "copra_jit_dense"));
        g_dense_jit.push_back(DenseJit { n, lanes, fn }); // (modules stay loaded for the life of the process)
    }
    return COPRA_OK;
}

copra_status_t copra_batch_specialise(copra_batch_t* h, const char* cache_dir)
{
    return copra_batch_specialise_checked(h, cache_dir, nullptr, nullptr);
}

copra_status_t copra_batch_specialise_checked(copra_batch_t* h, const char* cache_dir, copra_code_object_check_t check, void* user)
{
    if (!h) return fail(COPRA_ERR_ARG, "copra_batch_specialise: null handle");
    const FusedPlan& P = h->hp.plan;
    if (h->jit_fused) return COPRA_OK;
    const int rp0 = specialised_cost_rows(P.nx, P.nu, P.N, P.rmax, P.rfull);
    if (h->hp.large || P.initial_state || P.rfull > 0 || rp0 > 0 || P.n > kWave || P.nu > kMaxNu || (P.lds.ric && ric_aot_exact(P.nx, P.nu, P.N)))
        return COPRA_OK; // nothing to gain: the shape already runs on dedicated kernels (or on bodies without shape parameters)
    // ---- the Riccati-factor tier (lmpc_fused_ric.hpp; what the headline runs on) for THIS shape: per-step costs, xDim (xDim + uDim + 1)
    //      <= 64, two or three controls, at most 64 decision variables.  Compiled with Q1 in registers and in LDS (the layout ladder
    //      moves between the two); the controller takes the tier's layout once the kernels exist.
    // Single-control systems with fewer than 48 variables stay on the packed / factor-only kernels: the reference's falling-mass
    // problems hold most of their control bounds active, far beyond this tier's five register columns (measured, M solves/s,
    // this tier vs the others compiled for the shape: N = 5: 94 vs 339, 16: 6.8 vs 55, 32: 7.2 vs 16, 48: 29 vs 22, 64: 75 vs 29).
    const bool ric_pays = P.nu >= 2 || P.n >= 48;
    if (!h->shared && ric_pays && !h->hp.opt.no_ric && !h->hp.opt.no_tri) {
        HostPlan trial = h->hp; // (the layout and the tables are only kept if everything below succeeds)
        if (take_ric_layout(trial)) {
            char keyr[128], srcr[4096];
            const char* const sr = P.stage_refs ? "true" : "false"; // (reference trajectories: the builds with the stage-varying affine term)
            snprintf(keyr, sizeof keyr, "copra_jit_ric_%d_%d_%d%s", P.nx, P.nu, P.N, P.stage_refs ? "_srefs" : "");
            snprintf(srcr, sizeof srcr,
                "#include <hip/hip_runtime.h>\n#include \"lmpc_fused_ric.hpp\"\n#include \"lmpc_lane.hpp\"\nusing namespace copra_hip;\n"
                "extern \"C\" __global__ __launch_bounds__(64, 3) void copra_jit_fused(const FusedPlan P)\n"
                "{ if (P.ovf_zero && blockIdx.x == 0 && threadIdx.x == 0) *P.ovf_zero = 0;\n"
                "  int inst; bool failed; if (!tier_instance(P, (int)blockIdx.x, inst, failed)) return;\n"
                "  lmpc_fused_ric_body<%d, %d, %d, 6, %d, %s>(P, inst, failed); }\n"
                "extern \"C\" __global__ __launch_bounds__(64, 3) void copra_jit_fused_q0(const FusedPlan P)\n"
                "{ if (P.ovf_zero && blockIdx.x == 0 && threadIdx.x == 0) *P.ovf_zero = 0;\n"
                "  int inst; bool failed; if (!tier_instance(P, (int)blockIdx.x, inst, failed)) return;\n"
                "  lmpc_fused_ric_body<%d, %d, %d, 6, 0, %s>(P, inst, failed); }\n"
                "extern \"C\" __global__ __launch_bounds__(64, 1) void copra_jit_lane(const FusedPlan P)\n"
                "{ lmpc_lane_body<%d, %d, %s, true>(P, (int)blockIdx.x); }\n"
                "extern \"C\" __global__ __launch_bounds__(64, 1) void copra_jit_lane_plain(const FusedPlan P)\n"
                "{ lmpc_lane_body<%d, %d, %s, false>(P, (int)blockIdx.x); }\n",
                P.nx, P.nu, P.N, kFusedQ1Regs, sr, P.nx, P.nu, P.N, sr, P.nx, P.nu, sr, P.nx, P.nu, sr);
            std::string objr;
            const copra_status_t rcr = jit_compile(keyr, srcr, cache_dir, objr, check, user);
            if (rcr != COPRA_OK) return rcr;
            hipModule_t modr = nullptr;
            HIP_TRY(hipModuleLoad(&modr, objr.c_str()));
            hipFunction_t fr = nullptr, fq = nullptr, fl = nullptr;
            hipError_t er = hipModuleGetFunction(&fr, modr, "copra_jit_fused");
            if (er == hipSuccess) er = hipModuleGetFunction(&fq, modr, "copra_jit_fused_q0");
            if (er == hipSuccess) er = hipModuleGetFunction(&fl, modr, "copra_jit_lane");
            hipFunction_t flp = nullptr;
            if (er == hipSuccess) er = hipModuleGetFunction(&flp, modr, "copra_jit_lane_plain");
            double* dparams = nullptr;
            if (er == hipSuccess) er = upload(&dparams, trial.params); // (the stage-cost tables were appended)
            if (er != hipSuccess) {
                (void)hipGetLastError();
                (void)hipModuleUnload(modr);
                (void)hipFree(dparams);
                return fail(COPRA_ERR_HIP, std::string("copra_batch_specialise (Riccati-factor tier): ") + hipGetErrorString(er));
            }
            (void)hipFree(h->d_params);
            h->d_params = dparams;
            h->hp = trial;
            h->packed = 0; // (one instance per wavefront on this tier)
            h->lds_attr_set = false;
            h->ad.adapt_left = h->ad.adapt_left > 4 ? h->ad.adapt_left : 4;
            h->ad.lds_top_set = false; // (a new ladder: its first solve chooses the level again)
            h->ad.lane_predict_left = 1;
            h->jit_module = modr;
            h->jit_lanes = 64;
            h->jit_tri = 1;
            h->jit_ric = true;
            h->jit_fused = fr;
            h->jit_fused_q0 = fq;
            h->jit_lane = fl;
            h->jit_lane_plain = flp;
            h->jit_shared = nullptr;
            return COPRA_OK;
        }
    }
    // (always the full register budget: with compile-time trip counts the unrolled bodies spill at 128 VGPRs -- double
    //  integrator N = 32: 9.2 M solves/s at four waves per SIMD, 15.6 M at two, 11.8 M for the run-time-shape kernel)
    char key[128], source[1536];
    snprintf(key, sizeof key, "copra_jit_%d_%d_%d_%d_l%d%s", P.nx, P.nu, P.N, P.rmax, h->packed ? h->packed : 64, P.lds.tri ? "t" : "");
    if (h->packed) // several small instances per wavefront: the same bodies on the group-wide primitives
        snprintf(source, sizeof source,
            "#define COPRA_WAVE_WIDTH %d\n#include \"packed_impl.inc\"\n"
            "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_fused(const FusedPlan P)\n"
            "{ const int inst = P.inst_offset + instance_id(); if (inst < P.batch) lmpc_fused_body<%d, %d, %d, %d>(P, inst); }\n"
            "extern \"C\" __global__ __launch_bounds__(64) void copra_jit_shared(const FusedPlan P)\n"
            "{ const int inst = instance_id(); if (inst < P.batch) lmpc_shared_body<%d, %d, %d>(P, inst); }\n",
            h->packed, P.nx, P.nu, P.N, P.rmax, P.nx, P.nu, P.N);
    else
        snprintf(source, sizeof source,
            "#include <hip/hip_runtime.h>\n#include \"lmpc_fused.hpp\"\n#include \"lmpc_shared.hpp\"\nusing namespace copra_hip;\n"
            "extern \"C\" __global__ __launch_bounds__(64%s) void copra_jit_fused(const FusedPlan P)\n"
            "{ lmpc_fused_body<%d, %d, %d, %d, %s>(P, P.inst_offset + (int)blockIdx.x); }\n"
            "extern \"C\" __global__ __launch_bounds__(64%s) void copra_jit_shared(const FusedPlan P)\n"
            "{ lmpc_shared_body<%d, %d, %d, %s>(P, (int)blockIdx.x); }\n",
            P.lds.tri ? ", 2" : "", P.nx, P.nu, P.N, P.rmax, P.lds.tri ? "true" : "false", P.lds.tri ? ", 2" : "", P.nx, P.nu,
            P.N, P.lds.tri ? "true" : "false");
    std::string obj;
    {
        const copra_status_t rcj = jit_compile(key, source, cache_dir, obj, check, user);
        if (rcj != COPRA_OK) return rcj;
    }
    hipModule_t mod = nullptr;
    HIP_TRY(hipModuleLoad(&mod, obj.c_str()));
    hipFunction_t f1 = nullptr, f2 = nullptr;
    hipError_t e = hipModuleGetFunction(&f1, mod, "copra_jit_fused");
    if (e == hipSuccess) e = hipModuleGetFunction(&f2, mod, "copra_jit_shared");
    const size_t jit_lds = (size_t)(h->packed ? 64 / h->packed : 1) * h->hp.lds_bytes;
    if (e == hipSuccess && jit_lds > 48 * 1024) { // more than the default dynamic-LDS limit
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(f1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jit_lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(f2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jit_lds);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipModuleUnload(mod);
        return fail(COPRA_ERR_HIP, std::string("copra_batch_specialise: ") + hipGetErrorString(e));
    }
    h->jit_module = mod;
    h->jit_lanes = h->packed ? h->packed : 64;
    h->jit_tri = P.lds.tri;
    h->jit_fused = f1;
    h->jit_shared = f2;
    return COPRA_OK;
}

} // extern "C"

