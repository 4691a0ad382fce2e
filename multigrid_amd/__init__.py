"""multigrid_amd — MI355X-native batched MultiGrid step path (drop-in for envs/gym_multigrid).

The compute path is the hand-written HIP library `libmgrid.so` (multigrid_amd/csrc/mgrid.hip,
C ABI in include/mgrid.h).  This package is the thin host side above it: a ctypes binding, a
batched env object, and adapters that present the reference's `reset()/step()` contract.
There is no CPU fallback: importing `multigrid_amd.lib` without the built library raises.
"""
from .registry import make, make_batched, register, registered_ids  # noqa: F401
