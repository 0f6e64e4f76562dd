"""Fused, row-masked Adam for the parameter groups of a Gaussian model.

`FusedMaskedAdam` is a `torch.optim.Optimizer` with the state layout of `torch.optim.Adam` (`step`, `exp_avg`,
`exp_avg_sq` per parameter), so the reference's densification code, which cuts and concatenates those state tensors
(gaussiansplatting/scene/gaussian_model.py:553-671), works on it unchanged.  `step()` updates ALL groups with one
launch of the HIP kernel behind `gsr_adam_step` (include/gsr.h): every scalar is read and written once, the row mask
of `GaussianModel.apply_grad_mask` (:841-856) is applied inside, and optionally the gradient of the anchor loss
(:152-184) is added on the fly.  The arithmetic is torch's single-tensor Adam in float32; there is no CPU fallback.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _native

__all__ = ["FusedMaskedAdam"]


class FusedMaskedAdam(torch.optim.Optimizer):
    """Adam(params, lr, betas=(0.9, 0.999), eps=1e-8) without weight decay / amsgrad / maximize.

    Extra per-group options:
      masked   (bool, default False)  the group's gradient rows are zeroed where the row mask is 0
    Extra state set through methods:
      set_row_mask(mask)              (P,) bool / uint8 tensor, or None: the mask of `apply_grad_mask`
      set_anchor(param, anchor, scale, row_weight=None)
                                      adds scale * row_weight[row] * (param - anchor) to the gradient of `param`
      set_grad_valid(mask)            (P,) uint8 tensor, or None: gradient rows with mask 0 were never written and count as
                                      zeros without being read (multiview.GradBucket(..., sparse_rows=True).row_valid);
                                      applies to every group, independently of `masked`
    A parameter's leading dimension is the Gaussian index; the row length is numel / shape[0]."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, masked=False))
        self._row_mask: Optional[torch.Tensor] = None
        self._row_weight: Optional[torch.Tensor] = None
        self._anchors: Dict[torch.Tensor, tuple] = {}
        self._grad_valid: Optional[torch.Tensor] = None

    def set_grad_valid(self, mask: Optional[torch.Tensor]) -> None:
        if mask is not None and (mask.dtype != torch.uint8 or not mask.is_contiguous()):
            raise RuntimeError("FusedMaskedAdam.set_grad_valid: a contiguous uint8 tensor with one entry per Gaussian")
        self._grad_valid = mask  # (not copied: the exchange refills it every step)

    def set_row_mask(self, mask: Optional[torch.Tensor]) -> None:
        self._row_mask = None if mask is None else mask.detach().to(torch.uint8).contiguous()

    def set_anchor(self, param: torch.Tensor, anchor: Optional[torch.Tensor], scale: float = 0.0,
                   row_weight: Optional[torch.Tensor] = None) -> None:
        if anchor is None:
            self._anchors.pop(param, None)
        else:
            self._anchors[param] = (anchor.detach().contiguous(), float(scale))
        if row_weight is not None:
            self._row_weight = row_weight.detach().float().contiguous()

    def clear_anchors(self) -> None:
        """Forget every anchor and the row weights (call after a densification / prune replaced the parameters)."""
        self._anchors.clear()
        self._row_weight = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # anchors are keyed by parameter identity: one whose tensor left the optimizer (densify / prune replace the
        # parameters) would silently stop acting -- refuse instead
        live = {id(p) for group in self.param_groups for p in group["params"]}
        stale = [a for a in self._anchors if id(a) not in live]
        if stale:
            raise RuntimeError(f"FusedMaskedAdam: {len(stale)} anchor(s) refer to tensors that are no longer parameters of "
                               "this optimizer; call clear_anchors() / set_anchor() again after densify or prune")
        # tensors that share (betas, eps, step) go into one launch (normally: everything)
        batches: Dict[tuple, list] = {}
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedMaskedAdam: parameters must live on the ROCm GPU; there is no CPU fallback")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("FusedMaskedAdam supports dense float32 parameters and gradients only")
                if not p.is_contiguous():
                    raise RuntimeError("FusedMaskedAdam: parameters must be contiguous")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)  # host tensor, as torch.optim.Adam's default
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                key = (p.device, group["betas"], group["eps"], int(st["step"]))
                batches.setdefault(key, []).append((p, group, st))
        L = _native.lib()
        for (dev, betas, eps, step), items in batches.items():
            for lo in range(0, len(items), 8):
                chunk = items[lo:lo + 8]
                arr = (_native.AdamTensor * len(chunk))()
                keep = []  # keeps contiguous copies alive until the launch is enqueued
                for i, (p, group, st) in enumerate(chunk):
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    m, v = st["exp_avg"], st["exp_avg_sq"]
                    if not (m.is_contiguous() and v.is_contiguous()):
                        raise RuntimeError("FusedMaskedAdam: optimizer state tensors must be contiguous")
                    rows = int(p.shape[0]) if p.dim() > 0 and p.shape[0] > 0 else 1
                    anchor = self._anchors.get(p)
                    masked = bool(group.get("masked", False)) and self._row_mask is not None
                    if masked and self._row_mask.numel() != rows:
                        raise RuntimeError("FusedMaskedAdam: the row mask must have one entry per Gaussian")
                    if self._grad_valid is not None and self._grad_valid.numel() != rows:
                        raise RuntimeError("FusedMaskedAdam: grad_valid must have one entry per Gaussian")
                    if anchor is not None and anchor[0].shape != p.shape:
                        raise RuntimeError("FusedMaskedAdam: anchor and parameter shapes differ")
                    # the kernel indexes row_weight[row] for every row of an anchored tensor: a weight vector of another
                    # length (left over from before a densification / prune) would be read out of bounds
                    if anchor is not None and self._row_weight is not None and self._row_weight.numel() != rows:
                        raise RuntimeError(f"FusedMaskedAdam: row_weight has {self._row_weight.numel()} entries, the anchored "
                                           f"parameter {rows} rows (clear_anchors() after densify / prune)")
                    keep.append(g)
                    arr[i] = _native.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(),
                                                anchor[0].data_ptr() if anchor is not None else None, p.numel(),
                                                max(1, p.numel() // rows), int(masked), float(group["lr"]),
                                                anchor[1] if anchor is not None else 0.0)
                mask_ptr = self._row_mask.data_ptr() if self._row_mask is not None else None
                w_ptr = self._row_weight.data_ptr() if self._row_weight is not None else None
                valid_ptr = self._grad_valid.data_ptr() if self._grad_valid is not None else None
                with torch.cuda.device(dev):
                    _native.check("gsr_adam_step_rows", L.gsr_adam_step_rows(
                        torch.cuda.current_stream(dev).cuda_stream, len(chunk), arr, step, float(betas[0]), float(betas[1]),
                        float(eps), mask_ptr, w_ptr, valid_ptr))
                del keep
        return loss
