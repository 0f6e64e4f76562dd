#!/usr/bin/env python3
"""How many (tile, Gaussian) instances of the benchmark view could never reach alpha >= 1/255 anywhere in their tile?
("opacity-aware tile bounds".)  CPU only: runs the oracle's preprocessing on the benchmark scene and
compares the reference's tile rectangle (square of side 2 ceil(3 sigma_max), auxiliary.h:46-56) with what
gaussianeditor_amd.set_tile_bounds("alpha") bins (K1 in gsr_preprocess.hip): the tiles the axis-aligned bounding box of
the alpha = 1/255 level set can reach, with the conservative margins of gsr_blend.hip: can_touch_quad.  Usage: python tools/tile_bound_study.py [P W H]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402
from oracle import cpu as O  # noqa: E402  (test infrastructure: this tool is a study, not part of the product)

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
sc = synth_scene(P, seed=0, s0=0.01, sh_degree=3)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
f = O.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None, cam.world_view_transform,
              cam.full_proj_transform, cam.camera_center, sc["bg"], W, H, tfx, tfy, 1.0, 3)
vis = f["radii"] > 0
m, r, co = f["means2D"][vis], f["radii"][vis].astype(np.float32), f["conic_opacity"][vis]
gx, gy = (W + 15) // 16, (H + 15) // 16


def rect(hx, hy, upper):
    minx = np.clip(((m[:, 0] - hx) / 16).astype(np.int64), 0, gx)
    miny = np.clip(((m[:, 1] - hy) / 16).astype(np.int64), 0, gy)
    maxx = np.clip(((m[:, 0] + hx + upper) / 16).astype(np.int64), 0, gx)
    maxy = np.clip(((m[:, 1] + hy + upper) / 16).astype(np.int64), 0, gy)
    return minx, miny, maxx, maxy


def area(rc):
    return np.maximum(rc[2] - rc[0], 0) * np.maximum(rc[3] - rc[1], 0)


r0 = rect(r, r, 15)  # the reference: floor((p + r + 15) / 16) stops one pixel short of p + r
ref = area(r0)
assert int(ref.sum()) == int(f["num_rendered"])
A, B, C, o = co[:, 0], co[:, 1], co[:, 2], co[:, 3]
det = A * C - B * B
ok = (o >= 1 / 255) & (det > 0)
tau2 = 2 * (np.log(np.maximum(255 * o, 1e-30)) * 1.001 + 0.01)
hx = np.where(ok, np.sqrt(np.maximum(tau2 * C / np.where(ok, det, 1) * 1.0001, 0)) + 0.01, r)
hy = np.where(ok, np.sqrt(np.maximum(tau2 * A / np.where(ok, det, 1) * 1.0001, 0)) + 0.01, r)
r1 = rect(np.minimum(hx, r), np.minimum(hy, r), 16)  # tile t is reachable iff 16 t <= p + h ...
r1 = (np.maximum(r1[0], r0[0]), np.maximum(r1[1], r0[1]), np.minimum(r1[2], r0[2]), np.minimum(r1[3], r0[3]))  # ... inside the reference's
tight = np.where(o < 1 / 255, 0, area(r1))
print(f"{P} Gaussians, {W}x{H}: visible {int(vis.sum())}; instances {int(ref.sum())} (reference rectangle) -> "
      f"{int(tight.sum())} (alpha >= 1/255 bounding box) = {tight.sum() / ref.sum():.3f}")
