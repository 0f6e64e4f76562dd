"""Prune = one stable row compaction for all tensors of a Gaussian model.

`GaussianModel.prune_points` / `_prune_optimizer` (gaussiansplatting/scene/gaussian_model.py:568-609) index six
parameters, their twelve Adam moment tensors and five bookkeeping tensors with the same boolean mask, one after the
other; every `tensor[mask]` runs its own nonzero + host sync + gather.  `compact_rows` scans the mask once (one host
readback for the number of survivors) and moves the surviving rows of ALL tensors with one launch of the HIP kernel
behind `gsr_compact_apply` (include/gsr.h).  The survivors keep their order, so each output equals `tensor[mask]` bit
for bit.  `prune_optimizer` is `_prune_optimizer` on top of it.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence

import torch

from . import _native

__all__ = ["compact_rows", "prune_optimizer"]


def compact_rows(tensors: Sequence[torch.Tensor], keep: torch.Tensor) -> List[torch.Tensor]:
    """[t[keep] for t in tensors] for tensors that share their leading dimension P with the (P,) bool mask `keep`."""
    if keep.dim() != 1 or keep.dtype not in (torch.bool, torch.uint8):
        raise RuntimeError("compact_rows: keep must be a 1-D bool / uint8 mask")
    if not keep.is_cuda:
        raise RuntimeError("compact_rows: tensors must live on the ROCm GPU; there is no CPU fallback")
    P = int(keep.numel())
    dev = keep.device
    srcs = []
    for t in tensors:
        if t.device != dev or t.dim() < 1 or int(t.shape[0]) != P:
            raise RuntimeError("compact_rows: every tensor needs the mask's device and leading dimension")
        srcs.append(t.detach().contiguous())
    if P == 0:
        return [s.clone() for s in srcs]
    k8 = keep.contiguous().view(torch.uint8) if keep.dtype == torch.bool else keep.contiguous()
    L = _native.lib()
    nbytes = ctypes.c_size_t(0)
    _native.check("gsr_compact_workspace_size", L.gsr_compact_workspace_size(P, ctypes.byref(nbytes)))
    work = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
    kept = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        s = torch.cuda.current_stream(dev).cuda_stream
        _native.check("gsr_compact_plan", L.gsr_compact_plan(s, P, k8.data_ptr(), work.data_ptr(), ctypes.byref(kept)))
        n = int(kept.value)
        outs = [torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in srcs]
        if n > 0:
            for lo in range(0, len(srcs), 32):
                chunk = list(zip(srcs[lo:lo + 32], outs[lo:lo + 32]))
                arr = (_native.CompactTensor * len(chunk))()
                for i, (src, dst) in enumerate(chunk):
                    row_bytes = src.element_size() * (src.numel() // P)
                    if row_bytes == 0:
                        raise RuntimeError("compact_rows: tensors with empty rows are not supported")
                    arr[i] = _native.CompactTensor(src.data_ptr(), dst.data_ptr(), row_bytes)
                _native.check("gsr_compact_apply", L.gsr_compact_apply(s, P, k8.data_ptr(), work.data_ptr(), len(chunk), arr))
    return outs


def prune_optimizer(optimizer: torch.optim.Optimizer, keep: torch.Tensor) -> Dict[str, torch.nn.Parameter]:
    """`GaussianModel._prune_optimizer(mask)` (gaussian_model.py:568-591) with one compaction for all groups: every
    group's single parameter and its `exp_avg` / `exp_avg_sq` lose the rows where `keep` is False.  Returns
    {group["name"]: new parameter}."""
    items = []  # (group, old param, state or None)
    flat: List[torch.Tensor] = []
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        p = group["params"][0]
        st = optimizer.state.get(p, None)
        items.append((group, p, st))
        flat.append(p)
        if st is not None and "exp_avg" in st:
            flat += [st["exp_avg"], st["exp_avg_sq"]]
    outs = compact_rows(flat, keep)
    result: Dict[str, torch.nn.Parameter] = {}
    i = 0
    for group, p, st in items:
        new_p = torch.nn.Parameter(outs[i].requires_grad_(True))
        i += 1
        if st is not None and "exp_avg" in st:
            st["exp_avg"], st["exp_avg_sq"] = outs[i], outs[i + 1]
            i += 2
        if st is not None:
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        result[group["name"]] = new_p
    return result
