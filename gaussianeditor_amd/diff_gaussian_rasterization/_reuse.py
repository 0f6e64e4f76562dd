"""View reuse: the SECOND render of a view runs the blend kernel only.

The reference renders every training view and every GUI frame twice with the same camera and the same Gaussians -- once
with the SH colours, once with `override_color` (the semantic mask; threestudio/systems/GassuianEditor.py:166-191,
webui.py:693-713) -- and each `render()` runs the whole rasterizer: preprocessing, duplication, sort, tile ranges, blend
(gaussian_renderer/__init__.py:45-150).  Everything in front of the blend depends on the camera and on positions, scales,
rotations and opacities only, not on the colours.  So the rasterizer remembers the state of its latest full render (per
host thread and device) and serves a following colour-override render of the SAME view from it: `gsr_blend_forward_aux`
on the remembered geometry / lists (K6 alone), the first render's radii and depth.  An unmodified GaussianEditor gets what
`render(..., semantic_color=)` gives a patched one.

"The same view" is PROVEN, not assumed:
  * scalars (image size, fov tangents, scale modifier, flags, P, device, stream) by value;
  * every tensor that shaped the remembered state (means3D, scales, rotations, opacities, cov3D_precomp, the two matrices)
    either IS the tensor of the first render (same object, or the same storage at the same offset) with an unchanged
    version counter -- an optimizer step, a densification or any other in-place write bumps it --, or is compared with it
    bit for bit on the device (`gsr_arrays_equal`): the reference's `pc.get_opacity` / `get_scaling` / `get_rotation` are
    fresh activation tensors on every call, equal in content and nothing else;
  * a remembered tensor whose version moved since the render cannot vouch for anything: miss.
A miss costs a few attribute reads; a hit one compare launch (0.1 ms per 500 MB) plus K6.

The served image is connected to autograd like any render: if somebody differentiates through it -- GaussianEditor never
does, the semantic image is thresholded -- the backward first runs the full forward it had skipped, then the ordinary backward.

`GSR_VIEW_REUSE=0` / `gaussianeditor_amd.set_view_reuse(False)` turns it off.  Cost of leaving it on: the state of the
latest render (about 170 bytes per Gaussian + the lists) stays allocated until the next render replaces it.
"""
from __future__ import annotations

import os
import threading

import torch

from . import _C
from .. import options as _options

_enabled = os.environ.get("GSR_VIEW_REUSE", "1") != "0"
_local = threading.local()
#: counters for tests and `bench.py` (per process; not synchronised -- diagnostics only)
stats = {"hits": 0, "misses": 0, "compares": 0, "remembered": 0}


def set_view_reuse(on: bool) -> None:
    global _enabled
    _enabled = bool(on)
    if not on:
        forget()


def view_reuse() -> bool:
    return _enabled


def forget() -> None:
    """Drops what this thread remembers (all devices)."""
    _local.entries = {}


class _Tracked:
    """A tensor as it was when the remembered render read it."""
    __slots__ = ("t", "version")

    def __init__(self, t: torch.Tensor):
        self.t = t
        self.version = t._version

    def match(self, cur: torch.Tensor):
        """True: `cur` is this tensor, unchanged.  None: only a content comparison can tell.  False: cannot match."""
        old = self.t
        if old._version != self.version:
            return False  # written in place since the render: what the state was built from is gone
        if cur is old:
            return True
        if cur.shape != old.shape or cur.dtype != old.dtype or cur.device != old.device:
            return False
        if old.numel() == 0:
            return True
        if (cur.data_ptr() == old.data_ptr() and cur.stride() == old.stride()
                and cur.untyped_storage().data_ptr() == old.untyped_storage().data_ptr() and cur._version == self.version):
            return True  # another view of the same memory (they share the version counter)
        return None


class _Entry:
    __slots__ = ("key", "tensors", "R", "geom", "binning", "img", "radii", "depth")


_ROLES = ("means3D", "scales", "rotations", "opacities", "cov3D_precomp", "viewmatrix", "projmatrix")
#: flags that do not change the state a render leaves for the blend kernel
_IGNORED_FLAGS = _options.FLAG_FORWARD_ONLY | _options.FLAG_CLEAR_GRADS | _options.FLAG_SHARED_SIMDS


def _key(rs, flags, means3D):
    dev = means3D.device
    return (dev, int(means3D.size(0)), int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
            float(rs.scale_modifier), int(flags) & ~_IGNORED_FLAGS, torch.cuda.current_stream(dev).cuda_stream)


def remember(rs, flags, means3D, scales, rotations, opacities, cov3D_precomp, num_rendered, geom, binning, img, radii, depth):
    """Called by the full render (`_RasterizeGaussians.forward`) with what it read and what it left."""
    if not _enabled or means3D.size(0) == 0 or not means3D.is_cuda:  # (the CPU test suite runs the L1 module on an oracle backend)
        return
    e = _Entry()
    e.key = _key(rs, flags, means3D)
    e.tensors = {r: _Tracked(t) for r, t in zip(_ROLES, (means3D, scales, rotations, opacities, cov3D_precomp,
                                                           rs.viewmatrix, rs.projmatrix))}
    e.R, e.geom, e.binning, e.img, e.radii, e.depth = num_rendered, geom, binning, img, radii, depth
    if not hasattr(_local, "entries"):
        _local.entries = {}
    _local.entries[means3D.device] = e
    stats["remembered"] += 1


def lookup(rs, flags, means3D, scales, rotations, opacities, cov3D_precomp):
    """The remembered state if it is provably the state a full render of these arguments would leave, else None."""
    if not _enabled:
        return None
    e = getattr(_local, "entries", {}).get(means3D.device)
    if e is None:
        return None
    if e.key != _key(rs, flags, means3D):
        stats["misses"] += 1
        return None
    pending = []
    for role, cur in zip(_ROLES, (means3D, scales, rotations, opacities, cov3D_precomp, rs.viewmatrix, rs.projmatrix)):
        m = e.tensors[role].match(cur)
        if m is False:
            stats["misses"] += 1
            return None
        if m is None:
            pending.append((e.tensors[role].t, cur))
    if pending:
        stats["compares"] += 1
        with torch.no_grad():
            pairs = [(a.detach().contiguous(), b.detach().contiguous()) for a, b in pending]
            if not _C.arrays_equal(pairs):
                stats["misses"] += 1
                return None
    stats["hits"] += 1
    return e
