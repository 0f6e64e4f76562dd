#!/usr/bin/env python3
"""Per-ITEM profile of the backward blend kernel (K7): which tiles are its longest items, how deep their lists are walked
and how much work the forward measured there.  Answers "are the heavy items DEEP (cut them by list segments) or DENSE?".
Run on the GPU box: python tools/bwd_items_profile.py [--s0 0.01] [--gaussians 1000000]"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--s0", type=float, default=0.01)
a = ap.parse_args()
P, W, H = a.gaussians, 1920, 1080
T = ((W + 15) // 16) * ((H + 15) // 16)
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=a.s0)
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
z = torch.zeros(P * _native.ACC_ROW, device=dev)  # the blend backward's accumulator table (include/gsr.h: GSR_ACC_*)
ptrs = [z.data_ptr()]
n = ctypes.c_int64(0)
L.gsr_debug_blend_backward_profile(s, P, R, W, H, bg.data_ptr(), geom.data_ptr(), binning.data_ptr(), img.data_ptr(),
                                   G.data_ptr(), *ptrs, 1, 0, ctypes.byref(n))
n = int(n.value)
rec = torch.zeros((n + T, 8), dtype=torch.int64, device=dev)
for _ in range(3):
    z.zero_()
    _native.check("profile", L.gsr_debug_blend_backward_profile(s, P, R, W, H, bg.data_ptr(), geom.data_ptr(), binning.data_ptr(),
                                                                img.data_ptr(), G.data_ptr(), *ptrs, rec.data_ptr(), n + T,
                                                                ctypes.byref(ctypes.c_int64(0))))
torch.cuda.synchronize()
r = rec.cpu().numpy().view(np.uint64)
wg = r[:n]
live = wg[:, 1] > 0
dur = (wg[live, 1] - wg[live, 0]).astype(np.int64)
items = r[n:].reshape(-1, 4)  # (2T, 4): cycles, 4 x 16-bit forward counts, positions walked, item code
cyc = items[:, 0].astype(np.int64)
used = cyc > 0
est = ((items[:, 1] & np.uint64(0xffff)) + ((items[:, 1] >> np.uint64(16)) & np.uint64(0xffff))
       + ((items[:, 1] >> np.uint64(32)) & np.uint64(0xffff)) + (items[:, 1] >> np.uint64(48))).astype(np.int64)
walked = items[:, 2].astype(np.int64)
code = items[:, 3].astype(np.int64)
print(f"P={P} s0={a.s0}: workgroups {int(live.sum())}, kernel length ~ max workgroup {dur.max()} cycles, mean workgroup {dur.mean():.0f}; "
      f"items recorded {int(used.sum())}")
order = np.argsort(-cyc)[:24]
print("longest items: cycles | share of the longest workgroup | positions walked | forward work estimate (4 quadrants) | half item?")
for i in order:
    print(f"  {cyc[i]:8d} | {cyc[i] / dur.max():5.2f} | {walked[i]:5d} | {est[i]:5d} | {'half' if code[i] & 0x80000000 else 'tile'}")
top = np.argsort(-cyc)[:100]
print(f"of the 100 longest items: walked > 512 positions: {int((walked[top] > 512).sum())}, > 256: {int((walked[top] > 256).sum())}; "
      f"median walked {int(np.median(walked[top]))}; median of all items {int(np.median(walked[used]))}")
print(f"cycles per walked position, 100 longest items: median {np.median(cyc[top] / np.maximum(walked[top], 1)):.0f}; all items: "
      f"{np.median(cyc[used] / np.maximum(walked[used], 1)):.0f}")
