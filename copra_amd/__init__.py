"""copra_amd -- MI355X-native batched condensed linear-MPC engine with copra's plug-in surface.

The compute path is hand-written HIP (copra_amd/csrc, libcopra_hip.so, C ABI in include/copra_hip.h); this package
is the thin host layer.  Importing the package does not need a GPU; any compute call does, and raises without one.
"""
from ._capi import CopraDomainError, CopraRuntimeError, CopraUnsupported  # noqa: F401
from .batch import BatchLMPC, qp_dense_specialise, qp_solve_dense_batch  # noqa: F401

__all__ = ["BatchLMPC", "qp_solve_dense_batch", "qp_dense_specialise", "CopraDomainError", "CopraRuntimeError", "CopraUnsupported"]
