#!/usr/bin/env python3
"""Per-wave timeline of the forward blend kernel (K6) on the headline workload.
Prints: kernel span, distribution of wave durations, correlation with list depth, and the
load per XCD / CU / SIMD.  Run on the GPU box: python tools/wave_profile.py [P W H]"""
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
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=0.01)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(dev)  # noqa: E731
e = torch.empty(0, device=dev)
args = (d(sc["bg"]), d(sc["xyz"]), e, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), 1.0, e,
        d(cam.world_view_transform), d(cam.full_proj_transform), tfx, tfy, H, W, d(sc["features"]), 3, d(cam.camera_center),
        False, False)
R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(*args)
L = _native.lib()
n = ctypes.c_int64(0)
L.gsr_debug_blend_forward_profile(0, P, R, W, H, args[0].data_ptr(), geom.data_ptr(), binning.data_ptr(), img.data_ptr(),
                                  color.data_ptr(), depth.data_ptr(), 1, 0, ctypes.byref(n))
n = int(n.value)
rec = torch.zeros((n, 4), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    _native.check("profile", L.gsr_debug_blend_forward_profile(s, P, R, W, H, args[0].data_ptr(), geom.data_ptr(),
                                                               binning.data_ptr(), img.data_ptr(), color.data_ptr(),
                                                               depth.data_ptr(), rec.data_ptr(), n, ctypes.byref(ctypes.c_int64(0))))
torch.cuda.synchronize()
r = rec.cpu().numpy().view(np.uint64)
live = r[:, 1] > 0
t0, t1 = r[live, 0].astype(np.int64), r[live, 1].astype(np.int64)
hw, xcc = (r[live, 2] & np.uint64(0xffffffff)).astype(np.int64), (r[live, 2] >> np.uint64(32)).astype(np.int64) & 0xf
items = (r[live, 3] >> np.uint64(32)).astype(np.int64)
visited = (r[live, 3] & np.uint64(0xffffffff)).astype(np.int64)
dur = t1 - t0
print(f"persistent waves {n}, recorded {live.sum()}; R={R}")
print("wave duration ticks (shader cycles): max %d p99 %.0f p90 %.0f median %.0f min %d mean %.0f" % (
    dur.max(), np.percentile(dur, 99), np.percentile(dur, 90), np.median(dur), dur.min(), dur.mean()))
print("items per wave: min %d mean %.1f max %d; total %d" % (items.min(), items.mean(), items.max(), items.sum()))
print("visited entries per wave: min %d mean %.0f max %d; total %.3e; cycles per visited entry (mean over waves) %.1f" % (
    visited.min(), visited.mean(), visited.max(), visited.sum(), (dur / np.maximum(visited, 1)).mean()))
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"  xcd {x}: waves {m.sum()} span {t1[m].max() - t0[m].min()} start spread {t0[m].max() - t0[m].min()} "
              f"end spread {t1[m].max() - t1[m].min()} visited {visited[m].sum()}")
