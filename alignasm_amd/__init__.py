"""alignasm_amd -- MI355X-native per-contig path inference (drop-in for alignasm's solve_ctg_read).

Only what the hot path needs lives here: `csrc/` (HIP kernels + C-ABI + host PAF codec) and
`api.py`, the host-side mirror of the reference's operator boundary.  See DESIGN.md.
"""
from .api import (AlignasmError, DeviceBatch, DeviceResult, Paf, device_count, solve_batch)  # noqa: F401
from ._abi import HostBatch  # noqa: F401

__all__ = ["AlignasmError", "DeviceBatch", "DeviceResult", "Paf", "HostBatch", "device_count", "solve_batch"]
