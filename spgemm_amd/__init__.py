"""spgemm_amd: MI355X (gfx950) SpGEMM engine behind a C ABI (include/spgemm_amd.h).

capi     - ctypes binding of libspgemm_amd.so (contexts, device CSR handles, multiply, helpers)
hostcsr  - host CSR holder, Matrix Market I/O, synthetic generators (no arithmetic on matrices)
dist     - one-process-per-GPU driver of the row-block distributed multiply (torch.distributed)
"""
from . import capi, hostcsr  # noqa: F401
from .capi import Context, DeviceCsr, SpgemmError  # noqa: F401
from .hostcsr import HostCsr  # noqa: F401

__all__ = ["capi", "hostcsr", "Context", "DeviceCsr", "SpgemmError", "HostCsr"]
