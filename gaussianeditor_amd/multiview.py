"""Multi-view data parallelism for the rasterizer path (SURVEY.md section 8(e) -- new design;
the reference is single-GPU and renders the views of a batch sequentially,
threestudio/systems/GassuianEditor.py:165-207).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The Gaussian
parameters are replicated; a batch of K views is sharded one view per rank; after the local
backward the rasterizer-input gradients of all ranks are summed with ONE all-reduce over a
single flat fp32 bucket, and the per-Gaussian screen radii are combined with one MAX
all-reduce (the reference takes `torch.max` over the views, GassuianEditor.py:175-178).

The bucket is filled without copies: the backward's gradient tensors are *allocated as views
into the bucket* (see `_C.set_grad_allocator`), so the kernels write straight into the
buffer RCCL reduces.
"""
from __future__ import annotations

import contextlib
from typing import Dict, Optional, Tuple

import torch
import torch.distributed as dist

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C

__all__ = ["GradBucket", "render_view_grads", "allreduce_view_grads", "multiview_step"]

#: bucket layout per Gaussian: name -> number of floats (3M for the SH block is filled in at construction)
_SLOTS = ("means3D", "sh", "scales", "rotations", "means2D", "opacities")


class GradBucket:
    """Flat fp32 buffer `[means3D 3P | sh 3MP | scales 3P | rotations 4P | means2D 3P | opacities P]`
    = (14 + 3M) * P floats (248 MB at P = 1M, M = 16).  means2D and opacities, the two gradients the backward
    accumulates with atomics, are adjacent so that one fill clears both."""

    def __init__(self, P: int, M: int, device):
        self.P, self.M = int(P), int(M)
        shapes = {"means3D": (P, 3), "sh": (P, M, 3), "opacities": (P, 1), "scales": (P, 3), "rotations": (P, 4),
                  "means2D": (P, 3)}
        n = sum(int(torch.Size(s).numel()) for s in shapes.values())
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        off = 0
        for name in _SLOTS:
            cnt = int(torch.Size(shapes[name]).numel())
            # every segment starts on a 16-byte boundary as long as P % 4 == 0; otherwise the kernels'
            # dwordx4 stores on (P,4) rows would be misaligned -> fall back to private tensors for those.
            self.views[name] = self.flat[off:off + cnt].view(shapes[name])
            off += cnt

    def allocator(self, name: str, shape: Tuple[int, ...], zero: bool) -> Optional[torch.Tensor]:
        if name == "means2D+opacities":  # one contiguous, zeroed (4P,) block for both accumulators
            m2, op = self.views["means2D"], self.views["opacities"]
            if tuple(shape) != (4 * self.P,) or m2.data_ptr() % 16 != 0 or op.data_ptr() != m2.data_ptr() + 12 * self.P:
                return None
            off = m2.storage_offset()
            block = self.flat[off:off + 4 * self.P]
            if zero:
                block.zero_()
            return block
        v = self.views.get(name)
        if v is None or tuple(v.shape) != tuple(shape) or v.data_ptr() % 16 != 0:
            return None
        if zero:
            v.zero_()
        return v

    @contextlib.contextmanager
    def capture(self):
        """While active, the rasterizer backward writes its gradients into this bucket."""
        _C.set_grad_allocator(self.allocator)
        try:
            yield self
        finally:
            _C.set_grad_allocator(None)

    def grads(self) -> Dict[str, torch.Tensor]:
        return dict(self.views)


def render_view_grads(settings: GaussianRasterizationSettings, means3D, opacities, shs, scales, rotations,
                      dL_dcolor: torch.Tensor, bucket: Optional[GradBucket] = None):
    """Forward + backward of ONE view through the drop-in L1 API with `dL_dcolor` as the
    pixel gradient.  Returns (color, radii, depth, grads) where grads has the six
    rasterizer-input gradients (views of `bucket` when one is given)."""
    leaves = [t.detach().requires_grad_(True) for t in (means3D, shs, opacities, scales, rotations)]
    m3, sh, op, sc, rot = leaves
    m2 = torch.zeros_like(m3, requires_grad=True)
    color, radii, depth = GaussianRasterizer(settings)(m3, m2, op, shs=sh, scales=sc, rotations=rot)
    ctx = bucket.capture() if bucket is not None else contextlib.nullcontext()
    with ctx:
        g = torch.autograd.grad([color], [m3, sh, op, sc, rot, m2], grad_outputs=[dL_dcolor])
    names = ("means3D", "sh", "opacities", "scales", "rotations", "means2D")
    return color.detach(), radii, depth.detach(), dict(zip(names, g))


def _touched_rows(bucket: GradBucket) -> torch.Tensor:
    """(P,) bool: Gaussians with at least one non-zero gradient entry.  A Gaussian that no pixel of this
    rank's view blended has an exactly zero row in every segment (the backward writes zeros there)."""
    P = bucket.P
    t = torch.zeros(P, dtype=torch.bool, device=bucket.flat.device)
    for name, v in bucket.views.items():
        # dL_dsh[k] = basis_k(dir) * dL_dRGB with basis_0 = SH_C0 != 0 (backward.cu:47-48): the whole SH row is zero
        # iff its coefficient-0 triple is, so the 12(M-1) other bytes per Gaussian need not be scanned.
        rows = v[:, 0, :] if name == "sh" and v.shape[1] > 0 else v.reshape(P, -1)
        t |= (rows != 0).any(dim=1)
    return t


def allreduce_view_grads(bucket: GradBucket, radii: Optional[torch.Tensor] = None, group=None, sparse="auto",
                         sparse_threshold: float = 0.5):
    """The exchange step of an iteration: SUM over ranks of the gradient bucket, MAX over ranks of the screen
    radii.  No-op in a single-process run.

    A view only produces gradients for the Gaussians it actually blends (the front layers: ~10 % of the
    benchmark scene per view), so summing the dense (14+3M)*P buffer mostly moves zeros over xGMI -- 248 MB per
    step at 1 M Gaussians, more than the step's compute time at 8 GPUs.  With `sparse` enabled the ranks first
    MAX-all-reduce a one-byte-per-Gaussian "touched" mask (P bytes), and if the union is at most
    `sparse_threshold` of the scene only the union rows are packed, summed with ONE all-reduce and scattered
    back; otherwise the dense bucket is reduced.  Every rank takes the same branch (the mask is reduced), and
    rows outside the union are zero on every rank, so the result equals the dense all-reduce."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return "local"
    if radii is not None:
        dist.all_reduce(radii, op=dist.ReduceOp.MAX, group=group)
    mode = "dense"
    if sparse:
        P = bucket.P
        mask = _touched_rows(bucket).to(torch.uint8)
        dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=group)
        idx = mask.nonzero(as_tuple=False).view(-1)  # one host sync; identical on every rank
        if idx.numel() <= sparse_threshold * P:
            segs = [v.reshape(P, -1) for v in bucket.views.values()]
            packed = torch.cat([sg.index_select(0, idx) for sg in segs], dim=1)  # (U, 14+3M)
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
            off = 0
            for sg in segs:
                w = sg.shape[1]
                sg.index_copy_(0, idx, packed[:, off:off + w])
                off += w
            mode = "sparse"
    if mode == "dense":
        dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=group)
    return mode


def multiview_step(settings: GaussianRasterizationSettings, params: Dict[str, torch.Tensor], dL_dcolor: torch.Tensor,
                   bucket: GradBucket, group=None):
    """One data-parallel iteration for this rank's view: forward, backward, gradient all-reduce.
    `params`: xyz, opacity, features, scaling, rotation (activated, as the rasterizer consumes them).
    After the call `bucket.views[...]` hold the batch-summed gradients on every rank and
    `radii` the batch-max radii."""
    color, radii, depth, grads = render_view_grads(settings, params["xyz"], params["opacity"], params["features"],
                                                   params["scaling"], params["rotation"], dL_dcolor, bucket)
    allreduce_view_grads(bucket, radii, group)
    return color, radii, depth, grads
