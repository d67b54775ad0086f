#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched condensed-LMPC solves/sec at (nx=6, nu=3, N=20) on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its N ranks itself, spawn_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (copra_batch_solve: condense + Goldfarb-Idnani + results) over one batch of
synthetic CoM preview systems, inputs already resident in HBM.
  N = 1 : BASELINE.json configs[2] -- batch 65536, seed 1 (the configuration the metric is quoted on).
  N > 1 : the batch of independent preview systems is sharded contiguously, one process per GPU, 32768 instances per GPU
          of ONE seed-2 batch of 32768 N instances -- at N = 8 exactly BASELINE.json configs[3] (262144 = 8 x 32768):
          rank g owns [g B / N, (g + 1) B / N).  The only data-path collective is ONE RCCL gather of the
          [U | status | iter | X] slabs to rank 0 per step, inside the timed region; after the timed region rank 0
          VERIFIES what it gathered (per-rank checksums + an oracle sample of every shard).
          `--scaling weak` instead keeps 65536 instances per GPU with per-rank seeds (rank 0 = configs[2]).
Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the launches that are the hot path (copra_lmpc_axis_kernel since round 6: one (instance, axis) per lane -- LQ sweep, roll-out
                  and the whole active-set iteration, gains in registers; copra_lmpc_fused_ric_kernel behind it for the handful of instances
                  it leaves): algorithmic bytes per launch (2016 B/solve: A,B,d,x0 in, U,X out) / their duration from HIP events
                  carried in their dispatch packets on the launch stream, against the 8 TB/s HBM peak; executed-FP64 and VALU-issue figures
                  next to it (roofline_fp64).
  cpu_baseline -- the oracle (C port of the reference's CPU QuadProgDense path) timed on this host's cores on a
                  bounded sample of the same workload, rank 0, N=1 only.
  extra        -- N=1 only, measured OUTSIDE the timed region (SURVEY.md 8d asks for them next to the headline):
                  the other BASELINE configs (2 and 5), the shared-model tick, the tight-workload sensitivity point,
                  the host-inclusive rate, the iteration histogram and the host's CPU model.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

def parity_rel(a, b, floor=1e-3):
    """the parity suite's measure (tests/test_gpu_parity.py::_rel): max_i |a_i - b_i| / max(|b_i|, floor) -- the true relative error
    of every entry above 1e-3 in magnitude, 1e-9 absolute below"""
    import numpy as np
    return float(np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), floor)))


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet, vector = matrix FP64
PMC_SUMMARY = os.path.join("profiles", "r06", "headline_rocprof_summary.json")  # rocprofv3 passes of this command


SPIN_UP_SOLVES = 40  # untimed solves before the warm-up steps (clock ramp; see main)

def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cores():
    """threads actually available to this process: hardware threads, affinity mask and the cgroup CPU quota"""
    hw = os.cpu_count() or 1
    cores = min(hw, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else hw
    note = ""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            lim = max(1, int(round(float(q) / float(per))))
            if lim < cores:
                note = "; cgroup CPU quota %d of %d hardware threads" % (lim, hw)
                cores = lim
    except Exception:
        pass
    return cores, hw, note


from tools.bench_extra import extra_measurements  # noqa: E402  (the side measurements: tools/bench_extra.py)


def spawn_ranks(n_gpus):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as CHILD processes of
    `python -m torch.distributed.run` -- one per GPU, rendezvous on 127.0.0.1 at a free port -- with this command line, relay their
    output (rank 0's JSON line stays the last line of stdout) and return their exit code.  Nothing in this parent touches the GPU
    (no HIP call before the children exist; a process that initialised the GPU must not be replaced or forked on this pool)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = r.stdout.rstrip("\n").splitlines()
    js = [k for k, ln in enumerate(lines) if ln.startswith('{"metric"')]
    if js:  # whatever a child printed after the line (library banners at exit) goes in front of it
        lines.append(lines.pop(js[-1]))
    if lines:
        print("\n".join(lines), flush=True)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)  # (200 x 0.14 ms: the timed region is 28 ms -- at 20 steps a line moved by 1.6 % from run to run)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=0, help="instances per GPU (default: 65536 at N = 1, 32768 at N > 1)")
    ap.add_argument("--scaling", choices=["auto", "weak"], default="auto",
                    help="auto: N > 1 shards ONE seed-2 batch of 32768 N instances (N = 8: BASELINE configs[3]); "
                         "weak: 65536 instances per GPU with per-rank seeds")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the measurements outside the timed region")
    ap.add_argument("--dense-hessian", action="store_true",
                    help="hand the TrajectoryCost over as a full-size entry (126x126 M): the Hessian is then built by "
                         "the dense v_mfma_f64_16x16x4 Psi'WPsi contraction instead of the block-diagonal prefix sums")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: gather synchronously on the compute stream instead of overlapping it with the next solve")
    ap.add_argument("--selftest-rccl", action="store_true",
                    help="(single GPU) run the N > 1 code path for real with a one-rank RCCL process group: "
                         "init_process_group('nccl'), dist.gather of the result slab on the side stream, barrier, "
                         "all_reduce of the timing, verification of the gathered slab")
    ap.add_argument("--payload", choices=["auto", "full", "controls"], default="auto",
                    help="N > 1: what the one gather per step carries: the whole result [U | status | iter | X] (what the north star asks "
                         "for, 1500 B per instance) or only [U | status | iter] (492 B; X is the roll-out of U and rank 0 reproduces it) -- "
                         "for nodes whose xGMI links, not the kernel, bound the step.  auto (default): a probe of a few steps before the timed "
                         "region -- the whole result unless a step with its gather takes more than 1.5 x a step without; the line says which")
    ap.add_argument("--selftest-rccl2", action="store_true",
                    help="(two or more GPUs visible) spawn a world-2 RCCL run of this benchmark -- one process per GPU -- and fail "
                         "unless both ranks took part (dist.get_world_size() == 2), the gathered slabs match their checksums and an "
                         "oracle sample of every shard agrees; the first thing to run on a multi-GPU box")
    ap.add_argument("--selftest-overlap", action="store_true",
                    help="(single GPU) exercise the double-buffer / side-stream plumbing of the N > 1 path with a "
                         "device copy standing in for the RCCL gather")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: libcopra_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    if args.selftest_rccl2:
        # (counting devices does not initialise the GPU: the two ranks are CHILD processes, started before anything here touches it)
        import subprocess
        if torch.cuda.device_count() < 2:
            raise SystemExit("--selftest-rccl2 needs two visible GPUs (found %d)" % torch.cuda.device_count())
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29531", os.path.abspath(__file__), "--gpus", "2", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--payload", args.payload, "--no-extra"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
        line = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else None
        mg = (line or {}).get("multi_gpu_check", {})
        ok = (line is not None and line["n_gpus"] == 2 and mg.get("ranks") == 2 and mg.get("rccl_world_size") == 2
              and mg.get("gathered_slabs_match_per_rank_checksums") and mg.get("status_agree") and mg.get("max_rel_u_err", 1.0) <= 1e-6)
        print(json.dumps({"selftest_rccl2": bool(ok), "line": line, "stderr_tail": r.stderr[-800:] if not ok else ""}), flush=True)
        raise SystemExit(0 if ok else 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))  # plain `python bench.py --gpus N`: start the N ranks ourselves
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d under a launcher that started %d ranks (WORLD_SIZE=%d): they must agree" % (args.gpus, world, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (rank %d of %d): the product path has no CPU fallback" % (rank, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.selftest_rccl
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    from copra_amd import BatchLMPC, workloads
    from copra_amd.batch import to_abi_layout
    from copra_amd.sharding import GatherLoop, alloc_result_slab, head_bytes, rollout_trajectory, shard_range, split_slab

    nx, nu, N = 6, 3, 20
    n, X = nu * N, nx * (N + 1)
    sharded = world > 1 and args.scaling == "auto"
    if sharded:
        # ONE batch for the whole job, contiguous shards: at 8 ranks BASELINE.json configs[3]
        per_gpu = args.batch or 32768
        global_batch = per_gpu * world
        wl_all = workloads.com_preview(global_batch, N=N, seed=2)
        lo, hi = shard_range(global_batch, rank, world)
        wl = dict(wl_all, A=wl_all["A"][lo:hi], B=wl_all["B"][lo:hi], d=wl_all["d"][lo:hi], x0=wl_all["x0"][lo:hi])
        batch = hi - lo
        workload_name = ("CoM preview LMPC nx=6 nu=3 N=20, TrajectoryCost+ControlCost, TrajectoryBound(63 rows)+ControlBound: "
                         "one seed-2 batch of %d sharded contiguously over %d GPUs%s"
                         % (global_batch, world, " (BASELINE configs[3])" if global_batch == 262144 and world == 8 else ""))
    else:
        # every rank owns `batch` instances; seeds differ per rank (rank 0 == BASELINE configs[2], seed 1)
        batch = args.batch or 65536
        global_batch = batch * world
        lo, hi = 0, batch
        wl_all = None
        wl = workloads.com_preview(batch, N=N, seed=1 + rank)
        workload_name = ("CoM preview LMPC nx=6 nu=3 N=20, TrajectoryCost+ControlCost, "
                         "TrajectoryBound(63 rows)+ControlBound (BASELINE configs[2])")
    if args.dense_hessian:
        from copra_amd.autospan import autospan_cost
        c0 = wl["costs"][0]
        wl["costs"] = [autospan_cost(dict(c0, p=np.tile(c0["p"], N + 1))), wl["costs"][1]]
    Ab, Bb, db, xb = to_abi_layout(wl["A"], wl["B"], wl["d"], wl["x0"])
    tA, tB = torch.from_numpy(Ab).to(dev), torch.from_numpy(Bb).to(dev)
    td, tx0 = torch.from_numpy(db).to(dev), torch.from_numpy(xb).to(dev)
    # The engine writes straight into the gather payload.  N > 1: two slabs, so that the RCCL gather of step k (on a
    # side stream, over xGMI) overlaps the solve of step k+1; the timed region ends only when both streams are idle.
    comm_path = use_dist or args.selftest_overlap
    overlap = comm_path and not args.no_overlap
    n_slabs = 2 if overlap else 1
    slabs = [alloc_result_slab(batch, n, X, dev) for _ in range(n_slabs)]
    views0 = slabs[0][1]

    # (--dense-hessian measures the dense contraction, not the per-step form the plan builder would recognise in the block-diagonal entry)
    eng = BatchLMPC(nx, nu, N, batch, wl["costs"], wl["cstrs"], options=dict(no_stage_refs=1) if args.dense_hessian else None)
    eng.set_system(tA, tB, td, tx0)
    stream = torch.cuda.current_stream().cuda_stream

    def solve_into(v, k):
        eng.set_outputs(v["control"], v["trajectory"], v["status"], v["iter"])
        eng.solve(stream)

    payload_probe = None
    payload_choice = args.payload
    if payload_choice == "auto":
        payload_choice = "full"
        if use_dist and world > 1:
            # the solve is 0.1 ms per shard; rank 0 ingests (world - 1) x 48.8 MB per step with the whole result: which of the two bounds the
            # step depends on the node's links.  Probe: a few steps without and with the (full) gather; every rank takes rank 0's verdict.
            probe = GatherLoop(slabs, rank, world, solve_into, dev, use_dist=True, overlap=overlap, payload_bytes=None)
            times = []
            for comm in (False, True):
                for _ in range(3):
                    probe.step(communicate=comm)
                torch.cuda.synchronize()
                dist.barrier()
                t0p = time.perf_counter()
                for _ in range(5):
                    probe.step(communicate=comm)
                torch.cuda.synchronize()
                dist.barrier()
                times.append((time.perf_counter() - t0p) / 5)
            verdict = torch.tensor([1 if times[1] > 1.5 * times[0] else 0], device=dev)
            dist.broadcast(verdict, src=0)
            payload_choice = "controls" if int(verdict.item()) else "full"
            payload_probe = {"ms_per_step_without_gather": times[0] * 1e3, "ms_per_step_with_full_gather": times[1] * 1e3}
    args.payload = payload_choice
    payload_bytes = head_bytes(batch, n, X) if args.payload == "controls" else None
    loop = GatherLoop(slabs, rank, world, solve_into, dev, use_dist=use_dist, overlap=overlap, force_gather=args.selftest_rccl,
                      payload_bytes=payload_bytes)
    # the device at the clock it has when work arrives after an idle period (see below): a few timed steps BEFORE the spin-up
    cold_ms = None
    if world == 1 and not comm_path:
        solve_into(views0, 0)  # (module load, LDS opt-in, layout controller)
        torch.cuda.synchronize()
        time.sleep(0.5)
        t0c = time.perf_counter()
        for _ in range(5):
            solve_into(views0, 0)
        torch.cuda.synchronize()
        cold_ms = (time.perf_counter() - t0c) / 5 * 1e3

    # Clock ramp: the first ~25 launches after an idle period run at a lower shader clock (kernel 751 us -> 690 us over them,
    # profiles/r02/dispatch_timeline.txt).  A fixed number of untimed solves -- about 30 ms -- gets the device to its steady
    # state before the W warm-up steps the contract asks for; reported as `spin_up_solves`.
    # ... and, since one run in a dozen on this pool was still at the idle clock after them (profiles/r04/README.md: a bench line at the
    # cold-clock rate, 0.59 ms, behind four minutes of counter passes), further untimed groups of ten until the time per solve has stopped
    # falling (three groups within 1 %) or one second has passed.  Untimed, like the W warm-up steps; the count is in the line.
    spin_up_done = 0
    for _ in range(SPIN_UP_SOLVES):
        solve_into(views0, 0)
    torch.cuda.synchronize()
    spin_up_done += SPIN_UP_SOLVES
    if world == 1 and not comm_path:
        t_end, last, steady = time.perf_counter() + 1.0, None, 0
        while time.perf_counter() < t_end and steady < 3:
            t1 = time.perf_counter()
            for _ in range(10):
                solve_into(views0, 0)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t1
            steady = steady + 1 if (last is not None and abs(dt - last) <= 0.01 * last) else 0
            last = dt
            spin_up_done += 10
    for _ in range(args.warmup):
        loop.step(communicate=comm_path)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loop.step(communicate=comm_path)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- N > 1: rank 0 verifies what it gathered (after the timed region) ----
    multi = None
    if comm_path:
        ok_sum, sums = loop.verify()
        if rank == 0:
            sent = payload_bytes if payload_bytes is not None else slabs[0][0].numel()
            step_s = elapsed / args.steps
            multi = {"gathered_slabs_match_per_rank_checksums": bool(ok_sum), "ranks": world,
                     "rccl_world_size": dist.get_world_size() if use_dist else 1, "payload": args.payload, "payload_probe": payload_probe,
                     "payload_bytes_per_rank_per_step": int(sent),
                     # each peer reaches rank 0 over its own xGMI link (7 links x ~153 GB/s per GPU): what the measured step asks of one
                     # link if the gather fills it, and what the gather alone would take at the nominal link rate
                     "per_link_GBps_if_gather_filled_the_step": sent / step_s / 1e9,
                     "gather_ms_at_nominal_153_GBps_per_link": sent / 153e9 * 1e3, "measured_ms_per_step": step_s * 1e3}
            if use_dist and not args.dense_hessian:
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                import pyoracle
                worst, agree, checked = 0.0, True, 0
                for g, buf in enumerate(loop.gathered()):
                    glo, ghi = shard_range(global_batch, g, world) if sharded else (0, batch)
                    gw = wl_all if sharded else workloads.com_preview(batch, N=N, seed=1 + g)
                    part = split_slab(buf, ghi - glo, n, X)
                    pick = np.linspace(0, ghi - glo - 1, 16).astype(int)
                    ref = pyoracle.lmpc_solve_batch(gw["A"][glo:ghi][pick], gw["B"][glo:ghi][pick], gw["d"][glo:ghi][pick],
                                                    gw["x0"][glo:ghi][pick], N, gw["costs"], gw["cstrs"])
                    u = part["control"].cpu().numpy()[pick]
                    st = part["status"].cpu().numpy()[pick]
                    if args.payload == "controls":  # X did not travel: rank 0 reproduces it by the roll-out and it must be the oracle's
                        tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                        xr = rollout_trajectory(tt(gw["A"][glo:ghi][pick]), tt(gw["B"][glo:ghi][pick]), tt(gw["d"][glo:ghi][pick]),
                                                tt(gw["x0"][glo:ghi][pick]), part["control"][torch.from_numpy(pick).to(dev)]).cpu().numpy()
                        okx = ref["status"] == 0
                        worst = max(worst, parity_rel(xr[okx], ref["trajectory"][okx]))
                    agree = agree and bool((st == ref["status"]).all())
                    okm = ref["status"] == 0
                    worst = max(worst, parity_rel(u[okm], ref["control"][okm]))
                    checked += len(pick)
                multi.update({"oracle_sample_per_shard": 16, "instances_checked": checked, "max_rel_u_err": worst,
                              "status_agree": agree})

    # device time of the dominant kernel: HIP events recorded by the C ABI around the launch on `stream`
    # (measured in a separate short loop so that the event sync does not perturb the timed region)
    eng.set_outputs(views0["control"], views0["trajectory"], views0["status"], views0["iter"])
    for _ in range(min(args.steps, 10)):
        eng.solve(stream)
        kernel_s.append(eng.last_first_tier_seconds())  # (the dominant kernel alone: what rocprofv3's kernel trace shows)
    torch.cuda.synchronize()
    out_u, out_s, out_i = views0["control"], views0["status"], views0["iter"]
    status = out_s.cpu().numpy()
    iters = out_i.cpu().numpy()
    n_ok = int((status == 0).sum())
    lane_ran, lane_done = eng.lane_pass_info()  # (the one-instance-per-lane pass in front of the first tier, lmpc_lane.hpp)

    line = None
    if rank == 0:
        total = global_batch * args.steps
        value = total / elapsed
        kern = float(np.mean(kernel_s))
        alg_bytes = 8 * (nx * nx + nx * nu + 2 * nx) + 8 * (n + X)  # 528 in + 1488 out = 2016 B / solve
        achieved = alg_bytes * batch / kern / 1e9
        # HBM traffic per launch and the issue figures from the committed rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE + WRITE_SIZE in KB; 8-byte-per-lane accesses calibrate at x1.0 on gfx950) -- not
        # re-measured live
        traffic, traffic_src, issue = None, None, {}
        axis_ran = eng.axis_solver_ran()
        dominant = ("copra_lmpc_axis_kernel" if axis_ran else "copra_lmpc_lane_kernel") + " + copra_lmpc_fused_ric_kernel"  # (the launches the headline controller runs on)
        prof = os.path.join(ROOT, PMC_SUMMARY)
        if not os.path.exists(prof):
            prof = os.path.join(ROOT, "profiles", "r05", "headline_rocprof_summary.json")
        traffic_stale = None
        if batch == 65536 and not args.dense_hessian and os.path.exists(prof):
            try:
                prof_json = json.load(open(prof))
                # the counters are quoted from a committed profile of this command: they describe THIS run only if that profile was
                # taken on the build of the library that is loaded now (tools/pmc_summary.py records its source hash)
                from copra_amd import _capi
                traffic_stale = prof_json.get("library_source_hash") != _capi.library_source_hash()
                ctr = prof_json["counters"]
                # the dominant kernel: the first-tier fused kernel (Riccati-factor tier `copra_lmpc_fused_ric_kernel` since
                # round 2; `copra_lmpc_fused_tri_kernel` before) -- the one with the most wave cycles in the profile
                cands = [k for k in ctr if "copra_lmpc_fused_ric_kernel" in k or "copra_lmpc_fused_tri_kernel" in k]
                kname = max(cands, key=lambda k: ctr[k].get("SQ_WAVE_CYCLES", {}).get("mean_per_launch", 0.0))
                dominant = kname.split("<")[0].replace("void ", "")
                c = {k: v["mean_per_launch"] for k, v in ctr[kname].items()}
                # since round 3 the hot path is a PAIR of launches: the one-instance-per-lane pass (LQ sweep + roll-out for the whole
                # batch, lmpc_lane.hpp) and the first tier for the instances it leaves over -- `kernel_ms` spans both (events in
                # their dispatch packets), so the counters are those of both, per launch of the pair
                lane = [k for k in ctr if ("copra_lmpc_axis_kernel" if axis_ran else "copra_lmpc_lane_kernel") in k]
                if lane and axis_ran and eng.lane_pass_info()[1] == batch:
                    # (round 6: every instance ends in the (instance, axis)-per-lane solver, and a controller whose lists stay empty launches
                    #  no first tier at all -- `kernel_ms` is that ONE kernel, the counters are its own)
                    dominant = "copra_lmpc_axis_kernel"
                    c = {k: v["mean_per_launch"] for k, v in ctr[lane[0]].items()}
                elif lane:
                    dominant = ("copra_lmpc_axis_kernel + " if axis_ran else "copra_lmpc_lane_kernel + ") + dominant
                    for k, v in ctr[lane[0]].items():
                        c[k] = c.get(k, 0.0) + v["mean_per_launch"]
                traffic = 1024.0 * (c["FETCH_SIZE"] + c["WRITE_SIZE"])
                traffic_src = os.path.relpath(prof, ROOT) + " (rocprofv3 --pmc, separate passes)"
                issue = {"valu_instructions_per_solve": c["SQ_INSTS_VALU"] / batch,
                         "lds_instructions_per_solve": c["SQ_INSTS_LDS"] / batch,
                         "valu_issue_share_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                         "fp64_mfma_mops_per_solve": c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) / batch}
                if "SQ_INSTS_VALU_FMA_F64" in c:  # executed FP64 wave-instructions by class (their own --pmc pass)
                    fma, add, mul = c["SQ_INSTS_VALU_FMA_F64"], c.get("SQ_INSTS_VALU_ADD_F64", 0.0), c.get("SQ_INSTS_VALU_MUL_F64", 0.0)
                    issue["fp64_valu_instructions_per_solve"] = (fma + add + mul + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) / batch
                    # 64 lanes x (2 per FMA, 1 per add / mul) + 512 per MFMA "MOP" unit, over the kernel time of THIS run
                    issue["executed_fp64_tflops"] = (64 * (2 * fma + add + mul)
                                                     + 512 * c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)) / kern / 1e12
            except Exception:
                traffic = None
        hist = np.bincount(np.minimum(iters[:, 0], 15), minlength=16)
        line = {
            "metric": "MPC solves/sec (batched) at (nx=6,nu=3,N=20); max |u-u_ref|",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spin_up_solves": spin_up_done,
            "cold_clock_ms_per_step": cold_ms,  # five steps after half a second of idle, before the spin-up: the clock a sporadic caller gets
            "cold_clock_solves_per_s": (batch / (cold_ms * 1e-3)) if cold_ms else None,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload_name,
                       "hessian": "dense MFMA f64 contraction (full-size cost entry)" if args.dense_hessian
                       else "none (Riccati stage form: the condensed Hessian is never built; per-step cost entries)",
                       "structure": ("the CoM model's three axes are decoupled (BASELINE configs[2] as specified): every (instance, axis) is solved by one "
                                     "lane of copra_lmpc_axis_kernel, systems checked per instance; the same workload on coupled systems runs the "
                                     "round-5 pair: extra.headline_with_coupled_axes_batch65536"),
                       "batch_per_gpu": batch, "global_batch": global_batch, "nvar": n, "ineq_rows": 63,
                       "bound_rows": 2 * n,
                       "parallelism": ("contiguous batch shards x%d + 1 RCCL gather/step (%s)"
                                       % (world, "overlapped with the next solve" if overlap else "synchronous"))
                       if world > 1 else "single GPU"},
            "solved_ok": n_ok,
            "first_kernel": {"kernel": "copra_lmpc_axis_kernel" if axis_ran else ("copra_lmpc_lane_kernel" if lane_ran else "copra_lmpc_fused_ric_kernel"),
                             "finished": lane_done, "left_to_the_first_tier": (batch - lane_done) if lane_ran else None,
                             "finished_at_unconstrained_minimiser": int(((iters[:, 0] == 1) & (status == 0)).sum()),
                             "what": ("one (instance, axis) per lane (lmpc_axis.hpp, round 6): backward Riccati sweep of the axis' chain with the gains in "
                                      "registers, roll-out, and the Goldfarb-Idnani iteration in range-space form on the Riccati factor -- the same "
                                      "picks, step lengths and counters as qpgen2; an instance one of whose axes outgrows the lane's six active "
                                      "constraints (or is degenerate, or whose system couples two axes) goes to the first tier, from scratch")
                             if axis_ran else
                             ("LQ sweep + roll-out with one instance per lane in front of the first tier (lmpc_lane.hpp): an instance ends there when "
                              "its unconstrained minimiser violates nothing or when its first picks are bounds on u_0; the others go to the "
                              "active-set kernel, from scratch")},
            "mean_active_set_iters": float(iters[:, 0].mean()),
            "max_active_set_iters": int(iters[:, 0].max()),
            "active_set_iteration_histogram": {("%d" % k if k < 15 else "15+"): int(v) for k, v in enumerate(hist) if v},
            "kernel_ms": kern * 1e3,
            "kernel_solves_per_s": batch / kern,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "traffic_stale": traffic_stale,  # True: the quoted counters were taken on ANOTHER build of the library
                         "library_source_hash": __import__("copra_amd")._capi.library_source_hash(),
                         "algorithmic_bytes_per_launch": alg_bytes * batch,
                         "kernel": dominant, "algorithmic_bytes_per_solve": alg_bytes,
                         "note": ("algorithmic bytes over the time of the launches; copra_lmpc_axis_kernel moves nothing but them (gains, iterates and the "
                                  "active sets stay in registers and LDS) and is bound by FP64 issue at one wave per SIMD -- its lanes hold 100 doubles of "
                                  "stage data each --: DESIGN.md 3.2, `roofline_fp64`"),
                         "issue": issue, "fp64_peak_tflops": FP64_PEAK_TFLOPS},
            # the same pair of launches against the FP64 peak (vector and matrix FP64 share one pipe on gfx950: 78.6 TFLOP/s either way):
            # EXECUTED multiply-adds of the PMC pass over the kernel time of this run -- what DESIGN.md calls the path's real bound
            "roofline_fp64": {"bound": "fp64 issue (VALU + MFMA share the pipe)", "achieved": issue.get("executed_fp64_tflops"),
                              "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": (issue["executed_fp64_tflops"] / FP64_PEAK_TFLOPS) if issue.get("executed_fp64_tflops") else None,
                              "source": traffic_src, "stale": traffic_stale},
        }
        if multi is not None:
            line["multi_gpu_check"] = multi

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        cores, hw, quota_note = host_cores()
        # warm the -O3 -march=native build of the oracle (compiled on first use on a new host) before anything is timed
        pyoracle.lmpc_solve_batch(wl["A"][:8], wl["B"][:8], wl["d"][:8], wl["x0"][:8], N, wl["costs"], wl["cstrs"],
                                  nthreads=1, native=True)
        # whole passes over the batch of this run until about --cpu-seconds of CPU work have been done (at least one pass,
        # whose results are the ones compared with the GPU's)
        passes, cpu_t, ref = 0, 0.0, None
        while passes == 0 or cpu_t < args.cpu_seconds:
            t = time.perf_counter()
            r = pyoracle.lmpc_solve_batch(wl["A"], wl["B"], wl["d"], wl["x0"], N, wl["costs"], wl["cstrs"], nthreads=cores,
                                          native=True)
            cpu_t += time.perf_counter() - t
            passes += 1
            ref = ref or r
        sample = batch
        rate = passes * batch / cpu_t
        # single-thread figure on a smaller sample
        s1 = max(64, min(sample, int(rate / cores * 3.0)))
        t = time.perf_counter()
        pyoracle.lmpc_solve_batch(wl["A"][:s1], wl["B"][:s1], wl["d"][:s1], wl["x0"][:s1], N, wl["costs"],
                                  wl["cstrs"], nthreads=1, native=True)
        cpu1 = s1 / (time.perf_counter() - t)
        u = out_u[:sample].cpu().numpy()
        ok = (ref["status"] == 0) & (status[:sample] == 0)
        err = float(np.nanmax(np.abs(u[ok] - ref["control"][ok])))
        rel = parity_rel(u[ok], ref["control"][ok])
        xdev = views0["trajectory"][:sample].cpu().numpy()
        relx = parity_rel(xdev[ok], ref["trajectory"][ok])
        line["cpu_baseline"] = {"value": rate, "unit": "solves/s", "cores": cores, "kind": "port",
                                "sample": "%d pass(es) over the %d instances of this run (%.1f s of CPU work), one oracle "
                                          "controller per instance, static partition over %d pthreads, gcc -O3 "
                                          "-march=native%s" % (passes, batch, cpu_t, cores, quota_note),
                                "single_thread_solves_per_s": cpu1, "cpu_model": cpu_model(), "hardware_threads": hw}
        line["max_abs_u_err"] = err
        line["max_rel_u_err"] = rel  # entry-wise, floor 1e-3 (parity_rel: the measure of the parity tests), over every instance of the batch
        line["max_rel_x_err"] = relx
        # ... and with the floor at 1e-6: the true relative error of every entry above 1e-6 in magnitude (round-5 verdict: the north star says
        # "1e-6 relative", the 1e-3 floor holds the controls that vanish at the optimum to 1e-9 absolute instead)
        line["max_rel_u_err_floor_1e-6"] = parity_rel(u[ok], ref["control"][ok], floor=1e-6)
        line["max_rel_x_err_floor_1e-6"] = parity_rel(xdev[ok], ref["trajectory"][ok], floor=1e-6)
        line["iterations_agree"] = bool((ref["iter"][ok] == iters[:sample][ok]).all())
        line["status_agree"] = bool((ref["status"] == status[:sample]).all())
    if rank == 0 and world == 1 and not args.no_extra and not comm_path and not args.dense_hessian:
        eng.close()
        try:
            line["extra"] = extra_measurements(np, torch, dev)
        except Exception as e:  # the headline line must not be lost to a side measurement
            line["extra"] = {"error": repr(e)}
    if use_dist:
        # RCCL prints its version banner (NCCL_DEBUG=VERSION) through C stdio, which is block-buffered on a pipe and
        # would otherwise land AFTER the JSON line at exit: flush it now so that the JSON line stays the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        dist.barrier()  # every rank has flushed before rank 0 prints
    if rank == 0:
        print(json.dumps(line), flush=True)
    if use_dist:
        if rank == 0 and multi is not None:
            assert multi["gathered_slabs_match_per_rank_checksums"], "RCCL gather mismatch"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
