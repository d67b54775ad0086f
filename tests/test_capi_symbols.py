"""The C-ABI library loads and exports every symbol include/copra_hip.h declares (no compute calls: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "copra_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(copra_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared_functions()
    for must in ("copra_batch_create", "copra_batch_set_system", "copra_batch_solve", "copra_batch_get_results",
                 "copra_batch_dump_qp", "copra_qp_solve_dense_batch", "copra_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from copra_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), "libcopra_hip.so does not export %s" % name
    lib.copra_abi_version.restype = ctypes.c_int
    assert lib.copra_abi_version() >= 1


def test_no_cpu_fallback_without_a_device():
    """Without a usable HIP device every compute entry point must fail loudly (COPRA_ERR_HIP), never fall back."""
    import numpy as np
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    from copra_amd import BatchLMPC
    with pytest.raises(RuntimeError):
        BatchLMPC(2, 1, 10, 4, [dict(kind="target", M=np.eye(2), p=[0.0, -1.0])], [])


def test_product_package_never_imports_the_oracle_or_emulator():
    """The oracle / emulator are test infrastructure: nothing under copra_amd/ may reference them."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "copra_amd")):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in src and "copra_oracle" not in src and "pyemu" not in src, f
                if f.endswith((".py",)):
                    assert "tests.emu" not in src and "emu_harness" not in src, f


def test_no_matrix_instruction_hazard_across_a_branch():
    """tools/mfma_hazard_lint.py on the two translation units that hold chains of v_mfma_f64_4x4x4 (the headline's kernels and their
    run-time-horizon builds, the interior-point kernel on the matrix cores, the dense contraction): no matrix-instruction result is read
    fewer wait states later than the hardware needs on ANY path of the compiled control flow.  Round 4 found the compiler counting them
    on the fall-through side of a wave-uniform branch only -- wrong Riccati factors on the taken side, invisible to the emulator."""
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import mfma_hazard_lint as lint
    tmp = tempfile.mkdtemp(prefix="copra_lint_")
    outs, procs = [], []
    for f in ("copra_hip.hip", "copra_hip_ric.hip"):
        out = os.path.join(tmp, f[:-4] + ".s")
        outs.append(out)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-pass-failed",
                                       '-DCOPRA_SRC_HASH="lint"', "--cuda-device-only", "-S", "-o", out, f], cwd=lint.CSRC, stderr=subprocess.DEVNULL))
    assert all(p.wait() == 0 for p in procs)
    hits = [h for o in outs for h in lint.lint(o)]
    assert not hits, hits[:5]
    # ... and the lint does find the pattern where it exists (the sweep as it was compiled before the fix)
    bad = os.path.join(tmp, "synthetic.s")
    with open(bad, "w") as fh:
        fh.write("copra_synthetic_kernel:\n\tv_mfma_f64_4x4x4_4b_f64 v[28:29], v[28:29], v[10:11], v[32:33]\n\ts_cbranch_vccnz .LBB0_2\n"
                 "\ts_nop 4\n\tv_cndmask_b32_e64 v11, v11, v29, s[12:13]\n\tv_cndmask_b32_e64 v10, v10, v28, s[12:13]\n.LBB0_2:\n"
                 "\tv_mfma_f64_4x4x4_4b_f64 v[30:31], v[22:23], v[28:29], v[2:3]\n\ts_endpgm\n")
    found = lint.lint(bad)
    assert len(found) == 1 and found[0][3] == 8 and found[0][5] == 1 and found[0][6] == 6  # (the reader behind the label, one wait state away)
