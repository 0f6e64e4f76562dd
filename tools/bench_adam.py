"""Optimizer step over the six parameter groups of a 1M-Gaussian model (59 scalars per Gaussian at M = 16): the fused
HIP Adam (one launch) against torch.optim.Adam (foreach, the reference's) and torch's own fused implementation."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd.optim import FusedMaskedAdam  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = "cuda"
shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4)}
lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}
mask = torch.rand(P, device=dev) > 0.5


def make(kind):
    params = {k: torch.randn(s, device=dev).requires_grad_(True) for k, s in shapes.items()}
    groups = [{"params": [p], "lr": lrs[k], "name": k, **({"masked": k != "rotation"} if kind == "hip" else {})} for k, p in params.items()]
    if kind == "hip":
        opt = FusedMaskedAdam(groups, lr=0.0, eps=1e-15)
        opt.set_row_mask(mask)
    else:
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15, fused=(kind == "torch_fused"))
    for p in params.values():
        p.grad = torch.randn_like(p)
    return params, opt


out = {"gaussians": P, "scalars_per_gaussian": 59, "bytes_per_step": 28 * 59 * P}
for kind in ("torch_foreach", "torch_foreach_mask_hooks", "torch_fused", "hip"):
    params, opt = make("torch_foreach" if kind.startswith("torch_foreach") else kind)

    def step():
        if kind == "torch_foreach_mask_hooks":  # what apply_grad_mask's hooks add per step (gaussian_model.py:841-856)
            for k, p in params.items():
                if k != "rotation":
                    p.grad = p.grad * (mask[:, None] if p.grad.ndim == 2 else mask[:, None, None])
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    out[kind] = {"ms_per_step": round(ms, 4), "GB_per_s_at_28B_per_scalar": round(28 * 59 * P / ms / 1e6, 1)}
# gradients of which only the rows some view touched were written (multiview: GradBucket(..., sparse_rows=True)):
# the step with FusedMaskedAdam.set_grad_valid reads no gradient of an invalid row
for frac in (0.14, 0.32):
    params, opt = make("hip")
    opt.set_grad_valid((torch.rand(P, device=dev) < frac).to(torch.uint8))
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        opt.step()
    torch.cuda.synchronize()
    out[f"hip_grad_valid_{frac}"] = {"ms_per_step": round((time.perf_counter() - t0) / 20 * 1e3, 4)}
print(json.dumps(out, indent=1))
