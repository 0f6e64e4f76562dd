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
rec = torch.zeros((n, 8), dtype=torch.int64, device=dev)
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
# tail analysis: the slowest waves, and how busy the chip is over time (per XCD, relative to that XCD's first start)
order = np.argsort(-dur)[:12]
cyc_stage, cyc_group, chunks, cyc_cull = (r[live, k].astype(np.int64) for k in (4, 5, 6, 7))
print("all waves: sum dur %.3e = staging %.3e (of which gather wait + cull %.3e) + group loops %.3e + rest %.3e; chunks %d" % (
    dur.sum(), cyc_stage.sum(), cyc_cull.sum(), cyc_group.sum(), (dur - cyc_stage - cyc_group).sum(), chunks.sum()))
print("slowest waves: dur, items, visited, cycles/visited, xcd | chunks, staging cyc, (cull part), group cyc, group cyc / visited")
for i in order:
    print(f"  {dur[i]:8d} {items[i]:3d} {visited[i]:5d} {dur[i] / max(visited[i], 1):7.1f} {xcc[i]} | {chunks[i]:4d} {cyc_stage[i]:8d} "
          f"{cyc_cull[i]:8d} {cyc_group[i]:8d} {cyc_group[i] / max(visited[i], 1):6.1f}")
fast = np.argsort(dur)[:5]
for i in fast:
    print(f"  fastest: {dur[i]:8d} {items[i]:3d} {visited[i]:5d} {dur[i] / max(visited[i], 1):7.1f} {xcc[i]}")
rel_end = np.zeros_like(t1)
for x in range(8):
    m = xcc == x
    if m.any():
        rel_end[m] = t1[m] - t0[m].min()
span = rel_end.max()
edges = np.linspace(0, span, 15)
alive = [(rel_end > e).sum() for e in edges]
print("waves still running at t = k/14 of the kernel span:", alive)
print("corr(dur, visited) = %.3f" % np.corrcoef(dur, visited)[0, 1])
# balance across the hardware: visited entries and group-loop cycles per CU and per SIMD
cu_key = xcc * 4096 + ((hw >> 13) & 0x7) * 512 + ((hw >> 12) & 0x1) * 256 + ((hw >> 8) & 0xf) * 16
simd_key = cu_key + ((hw >> 4) & 0x3)
for name, key in (("CU", cu_key), ("SIMD", simd_key)):
    ids, inv = np.unique(key, return_inverse=True)
    v = np.bincount(inv, weights=visited)
    g = np.bincount(inv, weights=cyc_group)
    n = np.bincount(inv)
    last = np.zeros(len(ids))
    np.maximum.at(last, inv, dur)
    print(f"per {name}: {len(ids)} units, waves/unit min {n.min()} max {n.max()}; visited min {v.min():.0f} mean {v.mean():.0f} max {v.max():.0f}; "
          f"group cycles (sum over waves) min {g.min():.3e} mean {g.mean():.3e} max {g.max():.3e}; longest wave on unit: min {last.min():.0f} "
          f"mean {last.mean():.0f} max {last.max():.0f}")
    print(f"   corr(visited on unit, longest wave on unit) = {np.corrcoef(v, last)[0, 1]:.3f}")
# placement of persistent workgroups: blockIdx -> (xcc, se, sh, cu, simd, wave slot)
allr = rec.cpu().numpy().view(np.uint64)
hw_all = (allr[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
xcc_all = (allr[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
print("blockIdx: xcc se sh cu simd slot   (first 24, then every 8th up to 1100)")
idxs = list(range(24)) + list(range(24, 1100, 8))
print(" ".join(f"{b}:{xcc_all[b]}/{(hw_all[b] >> 13) & 7}/{(hw_all[b] >> 12) & 1}/{(hw_all[b] >> 8) & 15}/{(hw_all[b] >> 4) & 3}/{hw_all[b] & 15}" for b in idxs))
# consistency of the guess "blocks b and b + 1024k share a SIMD"
simd_all = xcc_all * 4096 + ((hw_all >> 13) & 7) * 512 + ((hw_all >> 12) & 1) * 256 + ((hw_all >> 8) & 15) * 16 + ((hw_all >> 4) & 3)
for stride in (8, 128, 256, 512, 1024, 2048):
    same = (simd_all[:len(simd_all) - stride] == simd_all[stride:]).mean()
    print(f"  fraction of blocks b with simd(b) == simd(b + {stride}): {same:.3f}")
