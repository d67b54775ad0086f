import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "emu")):
    if p not in sys.path:
        sys.path.insert(0, p)

from copra_amd._capi import OPTIONS  # noqa: E402  engine options (copra_options_t): tests pin a tier by switching the others off


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.lib()
    return pyoracle


def pytest_collection_modifyitems(config, items):
    """GPU runs: bring torch's HIP runtime up BEFORE the first test drives the device through libcopra_hip.so.  A test that
    is the first to touch torch.cuda late in a session (after hundreds of launches through the C ABI in the same process)
    has been seen to fail with "No HIP GPUs are available" on the GPU box; with the runtime initialised first it never does."""
    if not any(item.get_closest_marker("gpu") for item in items):
        return
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:  # no torch / no GPU: the tests that need them say so themselves
        pass


@pytest.fixture
def full_size_paths(monkeypatch):
    """the plan builder classifies full-size entries with a per-step structure (block-diagonal costs with repeating blocks: reference
    trajectories; constraint rows inside one step) as per-step entries.  Tests that are ABOUT the full-size machinery -- the dense
    Psi' W Psi contraction on the matrix cores, the full-row slack evaluation -- build their inputs with AutoSpan, which produces exactly
    that structure: they switch the classification off, so that they keep covering what they were written for."""
    monkeypatch.setitem(OPTIONS, "no_stage_refs", 1)
    monkeypatch.setitem(OPTIONS, "no_step_rows", 1)
