"""Development tool (GPU): time of the whole forward (K1 ... K6) of a synthetic view from HIP events, under the current
environment (GSR_LIBRARY_PATH selects a build variant): quick same-box A/B of forward changes.

    python tools/forward_probe.py [--s0 0.01] [--gaussians 1000000] [--iters 50] [--width 1920 --height 1080]
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.diff_gaussian_rasterization import _C  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--s0", type=float, default=0.01)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
dev = "cuda:0"
W, H = args.width, args.height
sc = synth_scene(args.gaussians, seed=0, s0=args.s0)
cam = ring_cameras(8, W, H)[0]
d = lambda t: t.to(dev)  # noqa: E731
e = torch.empty(0, device=dev)
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
xyz, op, scl, rot, feat, bg = d(sc["xyz"]), d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), d(sc["features"]), d(sc["bg"])
wv, pj, cc = d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center)


def fwd():
    return _C.rasterize_gaussians(bg, xyz, e, op, scl, rot, 1.0, e, wv, pj, tfx, tfy, H, W, feat, 3, cc, False, False)


for _ in range(5):
    out = fwd()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.iters + 1)]
ev[0].record()
for i in range(args.iters):
    out = fwd()
    ev[i + 1].record()
torch.cuda.synchronize()
ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.iters))
lib = os.path.basename(os.environ.get("GSR_LIBRARY_PATH", "in-tree"))
print(f"{lib}: P={args.gaussians} s0={args.s0} {W}x{H}: R={out[0]}; forward median {1e3 * ts[len(ts) // 2]:.1f} us, "
      f"p10 {1e3 * ts[len(ts) // 10]:.1f} us, checksum {float(out[1].double().sum()):.6f}")
