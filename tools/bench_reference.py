#!/usr/bin/env python3
"""Times the REFERENCE's own rasterizer (oracle/_ref: its .cu sources compiled for gfx950 by hipcc, i.e. a
straight recompilation of the CUDA design) on the headline workload, next to the product's numbers from
bench.py.  Baseline measurement only -- nothing here is shipped.  Run on the GPU box:
    python tools/bench_reference.py [--variant fma|nofma] [--steps K]"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402
from oracle import ref  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variant", default="fma")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
P, W, H = a.gaussians, a.width, a.height
sc = synth_scene(P, seed=0, s0=0.01)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
R = ref.Reference(a.variant)
G = seed_gradient(H, W, 0)


def fwd():
    return R.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None,
                     cam.world_view_transform, cam.full_proj_transform, cam.camera_center, sc["bg"], W, H, tfx, tfy, 1.0, 3)


# the wrapper re-uploads inputs and exports intermediates on every call; time the native calls only
import ctypes  # noqa: E402

f = fwd()
i = R.inp
L, h = R.L, R.h
p = ref._p
color = torch.empty((3, H, W), device="cuda")
depth = torch.empty((1, H, W), device="cuda")
radii = torch.empty(P, dtype=torch.int32, device="cuda")
dG = G.cuda()
g = R.backward(G)


def native_fwd():
    return L.gsrref_forward(h, P, 3, 16, p(i["bg"]), W, H, p(i["means3D"]), p(i["shs"]), p(None), p(i["opacities"]),
                            p(i["scales"]), ctypes.c_float(1.0), p(i["rotations"]), p(None), p(i["view"]), p(i["proj"]),
                            p(i["campos"]), ctypes.c_float(tfx), ctypes.c_float(tfy), 0, p(color), p(depth), p(radii))


def native_bwd():
    return L.gsrref_backward(h, 3, 16, p(i["bg"]), p(i["means3D"]), p(i["shs"]), p(None), p(i["scales"]), ctypes.c_float(1.0),
                             p(i["rotations"]), p(None), p(i["view"]), p(i["proj"]), p(i["campos"]), ctypes.c_float(tfx),
                             ctypes.c_float(tfy), p(radii), p(dG), p(g["dL_dmeans2D"]), p(g["dL_dconic"]), p(g["dL_dopacity"]),
                             p(g["dL_dcolors"]), p(g["dL_dmeans3D"]), p(g["dL_dcov3D"]), p(g["dL_dsh"]), p(g["dL_dscales"]),
                             p(g["dL_drotations"]))


for _ in range(3):
    native_fwd(); native_bwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    native_fwd()  # synchronises internally (hipDeviceSynchronize in the driver)
tf = (time.perf_counter() - t0) / a.steps
t0 = time.perf_counter()
for _ in range(a.steps):
    native_fwd(); native_bwd()
tt = (time.perf_counter() - t0) / a.steps
print(json.dumps({"what": f"reference .cu sources, hipcc -O3 ({a.variant}) on MI355X, synth-v1 {P} Gaussians, {W}x{H}, ring view 0", "num_rendered": f["num_rendered"],
                  "forward_ms": 1e3 * tf, "train_iter_ms": 1e3 * tt, "forward_renders_per_s": 1 / tf, "train_iters_per_s": 1 / tt}))
