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


@pytest.mark.gpu
def test_single_gpu_line_with_cpu_baseline():
    out = _run("--cpu-seconds", "1")
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert out["max_rel_u_err"] <= 1e-6 and out["status_agree"]


@pytest.mark.gpu
def test_multi_gpu_code_path_with_one_rank_rccl():
    out = _run("--no-cpu-baseline", "--selftest-rccl")
    assert out["n_gpus"] == 1 and out["scaling"] == "weak"
