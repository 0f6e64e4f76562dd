"""Densify / prune surgery on the tensors of a Gaussian model.

PRUNE = one stable row compaction for all tensors:

`GaussianModel.prune_points` / `_prune_optimizer` (gaussiansplatting/scene/gaussian_model.py:568-609) index six
parameters, their twelve Adam moment tensors and five bookkeeping tensors with the same boolean mask, one after the
other; every `tensor[mask]` runs its own nonzero + host sync + gather.  `compact_rows` scans the mask once (one host
readback for the number of survivors) and moves the surviving rows of ALL tensors with one launch of the HIP kernel
behind `gsr_compact_apply` (include/gsr.h).  The survivors keep their order, so each output equals `tensor[mask]` bit
for bit.  `prune_optimizer` is `_prune_optimizer` on top of it.

DENSIFY = rows appended to all tensors: `cat_tensors_to_optimizer` (:609-641), which `densification_postfix` (:643-671)
calls at the end of both `densify_and_clone` (:730-766) and `densify_and_split` (:673-728), runs `torch.cat` on each of
the six parameters and on their twelve Adam moments (the moments extended by `torch.zeros_like`): 18 cats + 12 fills.
`append_rows` writes every `[old rows ; new rows | zeros]` with ONE launch of the kernel behind `gsr_append_rows`;
`cat_tensors_to_optimizer` is the reference's method on top of it, `clone_rows` = select by mask (one compaction) +
append, i.e. the tensor side of `densify_and_clone`.  WHICH rows are cloned / how split samples are drawn stays in
torch with the caller, as in the reference.  No CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence

import torch

from . import _native

__all__ = ["compact_rows", "prune_optimizer", "append_rows", "cat_tensors_to_optimizer", "clone_rows"]


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


def append_rows(tensors: Sequence[torch.Tensor], extensions: Sequence[Optional[torch.Tensor]], n: Optional[int] = None
                ) -> List[torch.Tensor]:
    """[torch.cat((t, e)) for t, e in zip(tensors, extensions)], an extension of None standing for `n` zero rows
    (torch.cat((t, zeros)) -- what the reference does to the Adam moments).  All tensors share their leading dimension P,
    all extensions theirs (n); trailing shapes and dtypes of a pair must agree."""
    if len(tensors) != len(extensions):
        raise RuntimeError("append_rows: one extension (or None) per tensor")
    if not tensors:
        return []
    dev = tensors[0].device
    if not tensors[0].is_cuda:
        raise RuntimeError("append_rows: tensors must live on the ROCm GPU; there is no CPU fallback")
    P = int(tensors[0].shape[0])
    for e in extensions:
        if e is not None:
            n = int(e.shape[0]) if n is None else n
            if int(e.shape[0]) != n:
                raise RuntimeError("append_rows: all extensions need the same number of rows")
    if n is None:
        raise RuntimeError("append_rows: give n when every extension is None")
    srcs, exts, outs = [], [], []
    for t, e in zip(tensors, extensions):
        if t.device != dev or t.dim() < 1 or int(t.shape[0]) != P:
            raise RuntimeError("append_rows: every tensor needs the same device and leading dimension")
        if e is not None and (e.device != dev or e.dtype != t.dtype or tuple(e.shape[1:]) != tuple(t.shape[1:])):
            raise RuntimeError("append_rows: an extension must match its tensor's device, dtype and row shape")
        srcs.append(t.detach().contiguous())
        exts.append(None if e is None else e.detach().contiguous())
        outs.append(torch.empty((P + n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev))
    if P + n == 0:
        return outs
    L = _native.lib()
    with torch.cuda.device(dev):
        s = torch.cuda.current_stream(dev).cuda_stream
        for lo in range(0, len(srcs), 32):
            chunk = list(zip(srcs[lo:lo + 32], exts[lo:lo + 32], outs[lo:lo + 32]))
            arr = (_native.AppendTensor * len(chunk))()
            for i, (src, ext, dst) in enumerate(chunk):
                row_bytes = dst.element_size() * (dst.numel() // (P + n))
                if row_bytes == 0:
                    raise RuntimeError("append_rows: tensors with empty rows are not supported")
                arr[i] = _native.AppendTensor(src.data_ptr() if P else None, None if ext is None or n == 0 else ext.data_ptr(),
                                              dst.data_ptr(), row_bytes)
            _native.check("gsr_append_rows", L.gsr_append_rows(s, P, n, len(chunk), arr))
    return outs


def cat_tensors_to_optimizer(optimizer: torch.optim.Optimizer, tensors_dict: Dict[str, torch.Tensor]
                             ) -> Dict[str, torch.nn.Parameter]:
    """`GaussianModel.cat_tensors_to_optimizer(tensors_dict)` (gaussian_model.py:609-641) with one launch for all groups:
    every group's single parameter gets `tensors_dict[group["name"]]` appended, its `exp_avg` / `exp_avg_sq` the same
    number of zero rows.  Returns {group["name"]: new parameter}."""
    items, flat, ext = [], [], []
    n = None
    for group in optimizer.param_groups:
        assert len(group["params"]) == 1
        p = group["params"][0]
        e = tensors_dict[group["name"]]
        n = int(e.shape[0]) if n is None else n
        st = optimizer.state.get(p, None)
        has_moments = st is not None and "exp_avg" in st
        items.append((group, p, st, has_moments))
        flat.append(p)
        ext.append(e)
        if has_moments:
            flat += [st["exp_avg"], st["exp_avg_sq"]]
            ext += [None, None]
    outs = append_rows(flat, ext, n=n)
    result: Dict[str, torch.nn.Parameter] = {}
    i = 0
    for group, p, st, has_moments in items:
        new_p = torch.nn.Parameter(outs[i].requires_grad_(True))
        i += 1
        if has_moments:
            st["exp_avg"], st["exp_avg_sq"] = outs[i], outs[i + 1]
            i += 2
        if st is not None:
            del optimizer.state[p]
            optimizer.state[new_p] = st
        group["params"][0] = new_p
        result[group["name"]] = new_p
    return result


def clone_rows(optimizer: torch.optim.Optimizer, selected: torch.Tensor) -> Dict[str, torch.nn.Parameter]:
    """The tensor side of `densify_and_clone` (gaussian_model.py:730-766): the rows of every group's parameter where
    `selected` is set are appended to it (moments: zero rows) -- `param[selected]` for all groups by one compaction,
    then `cat_tensors_to_optimizer`."""
    names = [g["name"] for g in optimizer.param_groups]
    picked = compact_rows([g["params"][0] for g in optimizer.param_groups], selected)
    return cat_tensors_to_optimizer(optimizer, dict(zip(names, picked)))
