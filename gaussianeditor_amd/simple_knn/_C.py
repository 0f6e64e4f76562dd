"""`simple_knn._C`: distCUDA2 over the C ABI (gsr_knn_mean_dist2).  Reference: simple-knn/spatial.cu:15-25."""
from __future__ import annotations

import ctypes

import torch

from .. import _native

_native.lib()


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """Mean squared distance of every point to its 3 nearest neighbours; (P,3) float32 on the GPU -> (P,) float32."""
    if not points.is_cuda:
        raise RuntimeError("simple_knn.distCUDA2: points must be on the ROCm GPU (device 'cuda'); there is no CPU fallback")
    if points.dtype != torch.float32 or points.ndimension() != 2 or points.size(1) != 3:
        raise RuntimeError("simple_knn.distCUDA2: points must be a float32 tensor of shape (P, 3)")
    P = int(points.size(0))
    dev = points.device
    out = torch.zeros((P,), dtype=torch.float32, device=dev)  # torch::full({P}, 0.0), spatial.cu:20
    if P == 0:
        return out
    pts = points.contiguous()
    nbytes = ctypes.c_size_t(0)
    L = _native.lib()
    _native.check("gsr_knn_workspace_size", L.gsr_knn_workspace_size(P, ctypes.byref(nbytes)))
    work = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _native.check("gsr_knn_mean_dist2", L.gsr_knn_mean_dist2(torch.cuda.current_stream(dev).cuda_stream, P, pts.data_ptr(),
                                                                 work.data_ptr(), out.data_ptr()))
    return out
