#!/usr/bin/env python3
"""bench.py -- headline benchmark: batched condensed-LMPC solves/sec at (nx=6, nu=3, N=20) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (copra_batch_solve: condense + Goldfarb-Idnani + results) over one batch of
synthetic CoM preview systems (BASELINE.json configs[2]: batch=65536 per GPU, inputs already resident in HBM).
N > 1: the batch of independent preview systems is sharded, one process per GPU, weak scaling (65536 per GPU);
the only data-path collective is ONE RCCL gather of the [U | X | status] slabs to rank 0 per step, inside the
timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- dominant kernel (copra_lmpc_fused_kernel): algorithmic bytes per launch (2016 B/solve: A,B,d,x0 in,
                  U,X out) / average launch duration from HIP events recorded by the C ABI on the launch
                  stream, against the 8 TB/s HBM peak.  The path is FP64-latency/LDS bound, not HBM bound (see
                  DESIGN.md), so the fraction is small by construction; fp64 figures are reported next to it.
  cpu_baseline -- the oracle (C port of the reference's CPU QuadProgDense path) timed on this host's cores on a
                  bounded sample of the same workload, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet, vector = matrix FP64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-hessian", action="store_true",
                    help="hand the TrajectoryCost over as a full-size entry (126x126 M): the Hessian is then built by "
                         "the dense v_mfma_f64_16x16x4 Psi'WPsi contraction instead of the block-diagonal prefix sums")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline leg")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: gather synchronously on the compute stream instead of overlapping it with the next solve")
    ap.add_argument("--selftest-rccl", action="store_true",
                    help="(single GPU) run the N > 1 code path for real with a one-rank RCCL process group: "
                         "init_process_group('nccl'), dist.gather of the result slab on the side stream, barrier, "
                         "all_reduce of the timing")
    ap.add_argument("--selftest-overlap", action="store_true",
                    help="(single GPU) exercise the double-buffer / side-stream plumbing of the N > 1 path with a "
                         "device copy standing in for the RCCL gather")
    args = ap.parse_args()

    import numpy as np
    import torch  # first: libcopra_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
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
    from copra_amd.sharding import alloc_gather_buffers, alloc_result_slab, gather_results

    nx, nu, N = 6, 3, 20
    batch = args.batch
    n, X = nu * N, nx * (N + 1)
    # weak scaling: every rank owns `batch` instances; seeds differ per rank (rank 0 == BASELINE config 3, seed 1)
    wl = workloads.com_preview(batch, N=N, seed=1 + rank)
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
    slab, views = slabs[0]
    out_u, out_x, out_s, out_i = views["control"], views["trajectory"], views["status"], views["iter"]

    eng = BatchLMPC(nx, nu, N, batch, wl["costs"], wl["cstrs"])
    eng.set_system(tA, tB, td, tx0)
    eng.set_outputs(out_u, out_x, out_s, out_i)
    cur = torch.cuda.current_stream()
    stream = cur.cuda_stream

    if use_dist:
        gather_bufs = [alloc_gather_buffers(sl[0], rank, world) for sl in slabs]
    else:
        gather_bufs = [[torch.empty_like(sl[0])] for sl in slabs]  # self-test stand-in
    comm_stream = torch.cuda.Stream(device=dev) if overlap else None
    ev_solved = [torch.cuda.Event() for _ in range(n_slabs)]
    ev_sent = [torch.cuda.Event() for _ in range(n_slabs)]
    step_no = [0]

    def send(k):
        if use_dist:
            gather_results(slabs[k][0], rank, world, gather_bufs[k], force=True)
        else:
            gather_bufs[k][0].copy_(slabs[k][0], non_blocking=True)

    def step():
        k = step_no[0] % n_slabs
        step_no[0] += 1
        if overlap:
            v = slabs[k][1]
            if step_no[0] > n_slabs:
                cur.wait_event(ev_sent[k])  # this slab's previous gather must be done before it is overwritten
            eng.set_outputs(v["control"], v["trajectory"], v["status"], v["iter"])
        eng.solve(stream)
        if comm_path:
            if overlap:
                ev_solved[k].record(cur)
                with torch.cuda.stream(comm_stream):
                    comm_stream.wait_event(ev_solved[k])
                    send(k)
                    ev_sent[k].record(comm_stream)
            else:
                send(k)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_s = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # device time of the dominant kernel: HIP events recorded by the C ABI around the launch on `stream`
    # (measured in a separate short loop so that the event sync does not perturb the timed region)
    eng.set_outputs(out_u, out_x, out_s, out_i)
    for _ in range(min(args.steps, 10)):
        eng.solve(stream)
        kernel_s.append(eng.last_solve_seconds())
    torch.cuda.synchronize()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    status = out_s.cpu().numpy()
    iters = out_i.cpu().numpy()
    n_ok = int((status == 0).sum())

    line = None
    if rank == 0:
        total = batch * world * args.steps
        value = total / elapsed
        kern = float(np.mean(kernel_s))
        alg_bytes = 8 * (nx * nx + nx * nu + 2 * nx) + 8 * (n + X)  # 528 in + 1488 out = 2016 B / solve
        achieved = alg_bytes * batch / kern / 1e9
        # dense-as-the-reference flop model: 2 n^2 X build + 2 n^3/3 factor + k (2 m n + 4 n^2)
        m_rows = 63 + 2 * n
        flops = 2 * n * n * X + 2 * (nx ** 3 + nx * nx * nu) * (N - 1) + 2 * n ** 3 / 3.0 \
            + float(iters[:, 0].mean()) * (2 * m_rows * n + 4 * n * n)
        # HBM traffic per launch from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE + WRITE_SIZE,
        # KB units; 8-byte-per-lane accesses calibrate at x1.0 on gfx950) -- not re-measured live
        traffic, traffic_src = None, None
        prof = os.path.join(ROOT, "profiles", "r01", "headline_rocprof_summary_final.json")
        if batch == 65536 and not args.dense_hessian and os.path.exists(prof):
            try:
                ctr = json.load(open(prof))["counters"]
                kname = [k for k in ctr if "copra_lmpc_fused_tri_kernel" in k][0]
                traffic = 1024.0 * (ctr[kname]["FETCH_SIZE"]["mean_per_launch"] + ctr[kname]["WRITE_SIZE"]["mean_per_launch"])
                traffic_src = "profiles/r01/headline_rocprof_summary_final.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
            except Exception:
                traffic = None
        line = {
            "metric": "MPC solves/sec (batched) at (nx=6,nu=3,N=20); max |u-u_ref|",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "CoM preview LMPC nx=6 nu=3 N=20, TrajectoryCost+ControlCost, "
                                   "TrajectoryBound(63 rows)+ControlBound (BASELINE configs[2])",
                       "hessian": "dense MFMA f64 contraction (full-size cost entry)" if args.dense_hessian
                       else "block-diagonal prefix sums (per-step cost entry)",
                       "batch_per_gpu": batch, "global_batch": batch * world, "nvar": n, "ineq_rows": 63,
                       "bound_rows": 2 * n,
                       "parallelism": ("batch-shard x%d + 1 RCCL gather/step (%s)"
                                       % (world, "overlapped with the next solve" if overlap else "synchronous"))
                       if world > 1 else "single GPU"},
            "solved_ok": n_ok,
            "mean_active_set_iters": float(iters[:, 0].mean()),
            "max_active_set_iters": int(iters[:, 0].max()),
            "kernel_ms": kern * 1e3,
            "kernel_solves_per_s": batch / kern,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes * batch,
                         "kernel": "copra_lmpc_fused_tri_kernel", "algorithmic_bytes_per_solve": alg_bytes,
                         "note": "path is FP64-latency/LDS bound at n=60 (SURVEY 8d): HBM fraction is small by "
                                 "construction",
                         "fp64_model_tflops": flops * batch / kern / 1e12, "fp64_peak_tflops": FP64_PEAK_TFLOPS},
        }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pyoracle
        # threads actually available to this process: hardware threads, affinity mask and the cgroup CPU quota
        hw = os.cpu_count() or 1
        cores = min(hw, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else hw
        quota_note = ""
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                lim = max(1, int(round(float(q) / float(per))))
                if lim < cores:
                    quota_note = "; cgroup CPU quota %d of %d hardware threads" % (lim, hw)
                    cores = lim
        except Exception:
            pass
        probe = min(batch, 64 * cores)
        sl = slice(0, probe)
        t = time.perf_counter()
        ref = pyoracle.lmpc_solve_batch(wl["A"][sl], wl["B"][sl], wl["d"][sl], wl["x0"][sl], N, wl["costs"],
                                        wl["cstrs"], nthreads=cores, native=True)
        rate = probe / (time.perf_counter() - t)
        sample = int(min(batch, max(probe, rate * args.cpu_seconds)))
        sl = slice(0, sample)
        t = time.perf_counter()
        ref = pyoracle.lmpc_solve_batch(wl["A"][sl], wl["B"][sl], wl["d"][sl], wl["x0"][sl], N, wl["costs"],
                                        wl["cstrs"], nthreads=cores, native=True)
        cpu_t = time.perf_counter() - t
        # single-thread figure on a smaller sample
        s1 = max(64, min(sample, int(rate / cores * 3.0)))
        t = time.perf_counter()
        pyoracle.lmpc_solve_batch(wl["A"][:s1], wl["B"][:s1], wl["d"][:s1], wl["x0"][:s1], N, wl["costs"],
                                  wl["cstrs"], nthreads=1, native=True)
        cpu1 = s1 / (time.perf_counter() - t)
        u = out_u[:sample].cpu().numpy()
        ok = (ref["status"] == 0) & (status[:sample] == 0)
        err = float(np.nanmax(np.abs(u[ok] - ref["control"][ok])))
        rel = float(np.nanmax(np.abs(u[ok] - ref["control"][ok]) / (1.0 + np.abs(ref["control"][ok]))))
        line["cpu_baseline"] = {"value": sample / cpu_t, "unit": "solves/s", "cores": cores, "kind": "port",
                                "sample": "first %d of the %d instances of this run, one oracle controller per "
                                          "instance, static partition over %d pthreads, gcc -O3 -march=native%s"
                                          % (sample, batch, cores, quota_note),
                                "single_thread_solves_per_s": cpu1}
        line["max_abs_u_err"] = err
        line["max_rel_u_err"] = rel
        line["status_agree"] = bool((ref["status"] == status[:sample]).all())
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
        if rank == 0 and args.selftest_rccl:  # the gathered copy of this rank's slab must equal the slab
            assert all(torch.equal(gather_bufs[k][0], slabs[k][0]) for k in range(n_slabs)), "RCCL gather mismatch"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
