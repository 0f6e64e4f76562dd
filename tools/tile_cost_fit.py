#!/usr/bin/env python3
"""What predicts the time the backward blend spends on a tile?  Per-item cycles from gsr_debug_blend_backward_profile
(items extension) against the forward's per-quadrant counts and the walked list length; least-squares fits of a few
candidate estimates.  Run on the GPU box: python tools/tile_cost_fit.py [s0]"""
import ctypes
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd import _native  # noqa: E402
from gaussianeditor_amd.diff_gaussian_rasterization import _C  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

P, W, H = 1_000_000, 1920, 1080
s0 = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=s0)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(dev)  # noqa: E731
e = torch.empty(0, device=dev)
bg = d(sc["bg"])
R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(
    bg, d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e, d(cam.world_view_transform),
    d(cam.full_proj_transform), tfx, tfy, H, W, d(sc["features"]), 3, d(cam.camera_center), False, False)
L = _native.lib()
s = torch.cuda.current_stream(dev).cuda_stream
G = seed_gradient(H, W, 0).to(dev)
z = torch.zeros(P * 11, device=dev)
ptrs = [z[:3 * P].data_ptr(), z[7 * P:].data_ptr(), z[6 * P:7 * P].data_ptr(), z[3 * P:6 * P].data_ptr()]
n = ctypes.c_int64(0)
L.gsr_debug_blend_backward_profile(s, P, R, W, H, bg.data_ptr(), geom.data_ptr(), binning.data_ptr(), img.data_ptr(),
                                   G.data_ptr(), *ptrs, 1, 0, ctypes.byref(n))
n = int(n.value)
T = ((W + 15) // 16) * ((H + 15) // 16)
rec = torch.zeros((n + T, 8), dtype=torch.int64, device=dev)
for _ in range(2):
    z.zero_()
    _native.check("profile", L.gsr_debug_blend_backward_profile(s, P, R, W, H, bg.data_ptr(), geom.data_ptr(), binning.data_ptr(),
                                                                img.data_ptr(), G.data_ptr(), *ptrs, rec.data_ptr(), n + T,
                                                                ctypes.byref(ctypes.c_int64(0))))
torch.cuda.synchronize()
it = rec[n:].cpu().numpy().view(np.uint64).reshape(T * 2, 4)
live = it[:, 0] > 0
cyc = it[live, 0].astype(np.float64)
q = np.stack([(it[live, 1] >> np.uint64(16 * k)) & np.uint64(0xffff) for k in range(4)], 1).astype(np.float64)
tmax = it[live, 2].astype(np.float64)
code = it[live, 3]
half = (code >> np.uint64(31)) & np.uint64(1)
part = (code >> np.uint64(30)) & np.uint64(1)
qq = q.copy()
for i in range(len(qq)):  # a half item only runs two of the quadrants
    if half[i]:
        qq[i] = [q[i, 2], q[i, 3], 0, 0] if part[i] else [q[i, 0], q[i, 1], 0, 0]
chunks = np.ceil(tmax / 64.0)
print(f"items {live.sum()} (halves {int(half.sum())}); cycles: mean {cyc.mean():.0f} max {cyc.max():.0f}; kernel total item cycles {cyc.sum():.3e}")
cands = {
    "sum of quadrant counts (current)": qq.sum(1, keepdims=True),
    "max quadrant count": qq.max(1, keepdims=True),
    "sum + chunks": np.stack([qq.sum(1), chunks], 1),
    "max + chunks": np.stack([qq.max(1), chunks], 1),
    "sum + max + chunks": np.stack([qq.sum(1), qq.max(1), chunks], 1),
}
for name, X in cands.items():
    A = np.concatenate([X, np.ones((X.shape[0], 1))], 1)
    coef, *_ = np.linalg.lstsq(A, cyc, rcond=None)
    pred = A @ coef
    r = np.corrcoef(pred, cyc)[0, 1]
    rel = np.abs(pred - cyc) / cyc
    print(f"{name:34s} corr {r:.3f}  median |rel err| {np.median(rel):.3f}  p90 {np.quantile(rel, 0.9):.3f}  coef {np.round(coef, 1)}")
