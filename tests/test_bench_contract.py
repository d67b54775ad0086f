"""bench.py's output contract: ONE JSON line (the last line of stdout) with the keys the driver reads, for the
single-GPU path and for the N > 1 code path exercised with a one-rank RCCL group."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
        "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(*extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "4096",
                        *extra], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    out = json.loads(line)
    assert KEYS <= set(out), KEYS - set(out)
    assert out["unit"] == "solves/s" and out["value"] > 0 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["workload"].startswith("CoM preview") and out["solved_ok"] == 4096
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    return out


def test_plain_multi_gpu_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (round-3 verdict: it exited 1 with "launch with
    torch.distributed.run"): the parent starts two ranks through torch.distributed.run before touching the GPU.  On this CPU-only
    container the ranks must get as far as their GPU check -- which names rank and world size -- and the parent returns their
    failure; on a GPU box the same command is covered by test_world_2_... below."""
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is visible: the children would run the benchmark")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr
    # (the launcher ends the sibling as soon as one rank fails: on a loaded machine the slower one may not reach its own check)
    assert "rank 0 of 2" in r.stderr or "rank 1 of 2" in r.stderr, r.stderr[-1500:]


@pytest.mark.gpu
def test_single_gpu_line_with_cpu_baseline():
    out = _run("--cpu-seconds", "1", "--no-extra")
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["cpu_model"]
    assert out["max_rel_u_err"] <= 1e-6 and out["status_agree"]
    assert sum(out["active_set_iteration_histogram"].values()) == 4096


@pytest.mark.gpu
def test_extra_measurements_outside_the_timed_region():
    """configs 2 and 5, the shared-model tick, the tight-workload point and the host-inclusive rate ride
    along in `extra`"""
    out = _run("--no-cpu-baseline")
    ex = out["extra"]
    assert "error" not in ex, ex
    for key in ("config2_double_integrator_batch4096", "config5_initial_state_12_6_50_riccati_ipm",
                "config5_initial_state_12_6_50_quadprog_dense", "shared_model_tick_batch65536",
                "tight_workload_vmax0.25_umax1.2", "host_inclusive_numpy_in_numpy_out", "host_inclusive_pinned_pipelined",
                "dense_hessian_mfma_f64_16x16x4_batch65536", "reference_trajectory_tracking_batch65536",
                "single_problem_latency_cpp_mirror"):
        assert "error" not in ex[key], (key, ex[key])
        assert ex[key]["solves_per_s"] > 0, key
    assert ex["single_problem_latency_cpp_mirror"]["median_us"] > 0 and out["cold_clock_ms_per_step"] > 0
    assert ex["config5_initial_state_12_6_50_riccati_ipm"]["solved_ok"] == 16384


@pytest.mark.gpu
def test_multi_gpu_code_path_with_one_rank_rccl():
    out = _run("--no-cpu-baseline", "--selftest-rccl")
    assert out["n_gpus"] == 1 and out["scaling"] == "weak"
    mg = out["multi_gpu_check"]  # rank 0 verified what it gathered: checksums + an oracle sample of the shard
    assert mg["gathered_slabs_match_per_rank_checksums"] and mg["status_agree"] and mg["max_rel_u_err"] <= 1e-6


@pytest.mark.gpu
def test_multi_gpu_code_path_controls_only_payload():
    """--payload controls: the gather carries [U | status | iter] (492 of 1500 bytes per instance), rank 0 reproduces X by the
    roll-out and the oracle sample checks BOTH; the line says what travelled and what it asks of one xGMI link"""
    out = _run("--no-cpu-baseline", "--selftest-rccl", "--payload", "controls")
    mg = out["multi_gpu_check"]
    assert mg["payload"] == "controls" and mg["payload_bytes_per_rank_per_step"] < 4096 * 500
    assert mg["gathered_slabs_match_per_rank_checksums"] and mg["status_agree"] and mg["max_rel_u_err"] <= 1e-6
    assert mg["rccl_world_size"] == 1 and mg["gather_ms_at_nominal_153_GBps_per_link"] > 0


@pytest.mark.gpu
def test_world_2_rccl_selftest_when_two_gpus_are_visible():
    """the first thing to run on a multi-GPU box: two processes, one per GPU, RCCL gather; fails unless RCCL really saw two
    ranks and rank 0 verified both shards (skipped on the single-GPU boxes this repo is developed on)"""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--selftest-rccl2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["selftest_rccl2"] is True
    # and the plain command the contract documents: no launcher, the parent starts the two ranks
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extra"],
                       capture_output=True, text=True, timeout=1800, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["multi_gpu_check"]["rccl_world_size"] == 2
