#!/usr/bin/env python3
"""Per-workgroup timeline of the backward blend kernel (K7) on the headline workload: is it bound by its longest
tile or by throughput?  Run on the GPU box: python tools/wave_profile_bwd.py"""
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
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=0.01)
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
rec = torch.zeros((n, 8), dtype=torch.int64, device=dev)
for _ in range(3):
    z.zero_()
    _native.check("profile", L.gsr_debug_blend_backward_profile(s, P, R, W, H, bg.data_ptr(), geom.data_ptr(), binning.data_ptr(),
                                                                img.data_ptr(), G.data_ptr(), *ptrs, rec.data_ptr(), n,
                                                                ctypes.byref(ctypes.c_int64(0))))
torch.cuda.synchronize()
r = rec.cpu().numpy().view(np.uint64)
live = r[:, 1] > 0
dur = (r[live, 1] - r[live, 0]).astype(np.int64)
tiles = (r[live, 3] >> np.uint64(32)).astype(np.int64)
west = (r[live, 3] & np.uint64(0xffffffff)).astype(np.int64)
longest, first = r[live, 4].astype(np.int64), r[live, 5].astype(np.int64)
hw = (r[live, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (r[live, 2] >> np.uint64(32)).astype(np.int64) & 0xf
print(f"workgroups {n}, recorded {live.sum()}, R={R}; tiles processed {tiles.sum()}")
print("workgroup duration (cycles): max %d p99 %.0f p90 %.0f median %.0f min %d mean %.0f" % (
    dur.max(), np.percentile(dur, 99), np.percentile(dur, 90), np.median(dur), dur.min(), dur.mean()))
print("longest single tile (cycles): max %d p99 %.0f median %.0f; first tile: max %d median %.0f" % (
    longest.max(), np.percentile(longest, 99), np.median(longest), first.max(), np.median(first)))
print("tiles per workgroup: min %d mean %.1f max %d; forward-work estimate per workgroup: min %d mean %.0f max %d" % (
    tiles.min(), tiles.mean(), tiles.max(), west.min(), west.mean(), west.max()))
print("corr(duration, work estimate) = %.3f; cycles per unit of work estimate (mean) %.1f" % (
    np.corrcoef(dur, west)[0, 1], (dur / np.maximum(west, 1)).mean()))
cu_key = xcc * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 0x1) * 256 + ((hw >> 8) & 0xf) * 16
ids, inv = np.unique(cu_key, return_inverse=True)
w_cu = np.bincount(inv, weights=west)
d_cu = np.zeros(len(ids))
np.maximum.at(d_cu, inv, dur)
print(f"per CU: {len(ids)} units; work estimate min {w_cu.min():.0f} mean {w_cu.mean():.0f} max {w_cu.max():.0f}; "
      f"longest workgroup on CU min {d_cu.min():.0f} mean {d_cu.mean():.0f} max {d_cu.max():.0f}")
order = np.argsort(-dur)[:8]
print("slowest workgroups: dur, tiles, work estimate, longest tile, first tile")
for i in order:
    print(f"  {dur[i]:8d} {tiles[i]:3d} {west[i]:6d} {longest[i]:8d} {first[i]:8d}")
