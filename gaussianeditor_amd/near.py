"""The Delete path's neighbour query on the GPU: which remaining Gaussians lie within `dist_thresh` of the masked object.

Reference: GaussianModel.get_near_gaussians_by_mask (gaussiansplatting/scene/gaussian_model.py:865-898), called by the
Delete system (threestudio/systems/GassuianEditorDel.py:44) and the web UI (webui.py:1075).  The reference copies both
point sets to the host, builds a scipy KDTree (gaussiansplatting/knn.py) and copies the distances back; here the query
runs where the points live, over the C ABI (gsr_near_points, include/gsr.h), with the same compared value (the float64
Euclidean distance rounded to float32 once).
"""
from __future__ import annotations

import ctypes

import torch

from . import _native


def near_points(ref_xyz: torch.Tensor, query_xyz: torch.Tensor, dist_thresh: float, return_dist: bool = False,
                check_finite: bool = True):
    """near[q] = some row of ref_xyz lies within dist_thresh of query_xyz[q]; (n_ref,3), (n_query,3) float32 on the GPU.

    check_finite (default): a NaN / infinite coordinate raises ValueError, as scipy's KDTree does on the reference's route
    ("data must be finite", "'x' must be finite"); it costs one small reduction and a readback.  Without it such points are
    simply never near anything.

    return_dist: also the 1-NN distances, exact where <= dist_thresh and +inf where nothing lies within the search radius
    (the reference's KDTree returns the true distance everywhere; its caller only thresholds it)."""
    for name, t in (("ref_xyz", ref_xyz), ("query_xyz", query_xyz)):
        if not t.is_cuda:
            raise RuntimeError(f"near_points: {name} must be on the ROCm GPU (device 'cuda'); there is no CPU fallback")
        if t.dtype != torch.float32 or t.ndimension() != 2 or t.size(1) != 3:
            raise RuntimeError(f"near_points: {name} must be a float32 tensor of shape (n, 3)")
    if ref_xyz.device != query_xyz.device:
        raise RuntimeError("near_points: ref_xyz and query_xyz must be on the same device")
    if not dist_thresh >= 0:
        raise ValueError("near_points: dist_thresh must be >= 0")
    if check_finite:
        ok = torch.stack([torch.isfinite(ref_xyz).all(), torch.isfinite(query_xyz).all()]).tolist()
        if not ok[0]:
            raise ValueError("near_points: ref_xyz must be finite, check for nan or inf values")
        if not ok[1]:
            raise ValueError("near_points: query_xyz must be finite, check for nan or inf values")
    dev = query_xyz.device
    n_ref, n_query = int(ref_xyz.size(0)), int(query_xyz.size(0))
    near = torch.zeros((n_query,), dtype=torch.uint8, device=dev)
    dist = torch.full((n_query,), float("inf"), dtype=torch.float32, device=dev) if return_dist else None
    if n_query:
        L = _native.lib()
        ref, qry = ref_xyz.detach().contiguous(), query_xyz.detach().contiguous()
        nbytes = ctypes.c_size_t(0)
        _native.check("gsr_near_workspace_size", L.gsr_near_workspace_size(n_ref, ctypes.byref(nbytes)))
        work = torch.empty(max(int(nbytes.value), 1), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _native.check("gsr_near_points", L.gsr_near_points(
                torch.cuda.current_stream(dev).cuda_stream, n_ref, ref.data_ptr(), n_query, qry.data_ptr(),
                ctypes.c_float(dist_thresh), work.data_ptr(), near.data_ptr(), dist.data_ptr() if return_dist else None))
    return (near.bool(), dist) if return_dist else near.bool()


@torch.no_grad()
def get_near_gaussians_by_mask(xyz: torch.Tensor, mask: torch.Tensor, dist_thresh: float = 0.1) -> torch.Tensor:
    """GaussianModel.get_near_gaussians_by_mask with `self._xyz` passed in: a bool mask over the REMAINING points
    (xyz[~mask]) that are inside the object's 3 %..97 % quantile box widened 1.3x and within dist_thresh of an object point.
    Same statements as the reference up to the neighbour query; that one is gsr_near_points."""
    mask = mask.squeeze()
    object_xyz = xyz[mask]
    remaining_xyz = xyz[~mask]
    q = torch.tensor([0.03, 0.97], dtype=object_xyz.dtype, device=object_xyz.device)
    lo_hi = torch.stack([torch.quantile(object_xyz[:, c], q) for c in range(3)])  # (3, 2): the six quantile calls of :872-874
    scale = (lo_hi[:, 1] - lo_hi[:, 0])
    mid = (lo_hi[:, 1] + lo_hi[:, 0]) / 2
    scale = scale * 1.3
    lo, hi = mid - scale / 2, mid + scale / 2
    in_bbox = ((remaining_xyz >= lo) & (remaining_xyz <= hi)).all(dim=1)
    in_box_remaining_xyz = remaining_xyz[in_bbox]
    valid_mask = near_points(object_xyz, in_box_remaining_xyz, dist_thresh)
    mask_to_update = torch.zeros_like(remaining_xyz[:, 0], dtype=torch.bool)
    true_indices = torch.nonzero(in_bbox)[:, 0]
    mask_to_update[true_indices[valid_mask]] = True
    return mask_to_update


def patch_gaussian_model(cls) -> None:
    """Bind the GPU query as `cls.get_near_gaussians_by_mask` (cls: the reference's GaussianModel); INTEGRATION.md section 6."""

    def _method(self, mask, dist_thresh: float = 0.1):
        return get_near_gaussians_by_mask(self._xyz, mask, dist_thresh)

    cls.get_near_gaussians_by_mask = _method
