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
