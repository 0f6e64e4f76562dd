"""Behaviour switches of the rasterizer (include/gsr.h: GSR_FLAG_*).

The native library keeps no option state: the flags travel with every call.  This module holds the DEFAULT a render
starts with; `_RasterizeGaussians.forward` reads it once, stores the value in the autograd context, and the backward
of that render reuses the stored value -- so changing the default between a forward and its backward (or from another
thread: the web UI renders while a training thread steps) can never pair a view's kernels with different flags.
`override(...)` sets flags for the calling thread only.
"""
from __future__ import annotations

import contextlib
import threading

FLAG_TILE_BOUNDS_ALPHA = 1
FLAG_FAST_EXP = 2
FLAG_ALL = 3             # the behaviour switches a caller may set
FLAG_CLEAR_GRADS = 4     # (internal to the binding: the backward clears its accumulators itself; include/gsr.h)
FLAG_FORWARD_ONLY = 8    # (internal to the binding: a render none of whose inputs requires a gradient)
FLAG_SHARED_SIMDS = 16   # (set by multiview_batch_step for a rank's pipelined views: 2 persistent blend waves per SIMD)
FLAG_ACC_SELF_CLEAN = 32  # (internal to the binding: the backward's accumulator table is kept across calls and left zero by K8+K9)

_default = 0
_local = threading.local()


def current_flags() -> int:
    """Flags a render started now by this thread runs with."""
    f = getattr(_local, "flags", None)
    return _default if f is None else f


def default_flags() -> int:
    """The process-wide default, whatever `override` the calling thread is inside of."""
    return _default


def set_default_flags(flags: int) -> None:
    global _default
    if flags & ~FLAG_ALL:
        raise ValueError(f"unknown flag bits in {flags:#x}")
    _default = int(flags)


@contextlib.contextmanager
def override(flags: int):
    """Run the renders started inside the block, by this thread, with `flags` (behaviour switches, and the scheduling hint
    FLAG_SHARED_SIMDS)."""
    if flags & ~(FLAG_ALL | FLAG_SHARED_SIMDS):
        raise ValueError(f"unknown flag bits in {flags:#x}")
    prev = getattr(_local, "flags", None)
    _local.flags = int(flags)
    try:
        yield
    finally:
        _local.flags = prev
