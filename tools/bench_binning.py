"""Development aid: time gsr_preprocess and gsr_bin alone on a synth-v1 scene (events on the launch stream), and check the
point list against the first iteration's (and optionally the CPU oracle's).  Usage:
    python tools/bench_binning.py [--gaussians N] [--s0 S] [--width W --height H] [--iters K] [--oracle]"""
import argparse
import ctypes
import math
import sys

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd import _native  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--s0", type=float, default=0.01)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--oracle", action="store_true")
a = ap.parse_args()
DEV = "cuda:0"
P, W, H = a.gaussians, a.width, a.height
sc = synth_scene(P, seed=0, s0=a.s0)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(DEV).contiguous()  # noqa: E731
t = dict(xyz=d(sc["xyz"]), sca=d(sc["scaling"]), rot=d(sc["rotation"]), op=d(sc["opacity"]), sh=d(sc["features"]),
         view=d(cam.world_view_transform), proj=d(cam.full_proj_transform), cp=d(cam.camera_center))
p = lambda x: x.data_ptr()  # noqa: E731
L = _native.lib()
st = torch.cuda.current_stream()
s = st.cuda_stream
gb, _, ib = _native.scratch_sizes(P, 0, W, H)
geom = torch.empty(gb, dtype=torch.uint8, device=DEV)
img = torch.empty(ib, dtype=torch.uint8, device=DEV)
radii = torch.empty(P, dtype=torch.int32, device=DEV)
counts = (ctypes.c_int64 * 2)()
ref_pl = None
tp = tb = 0.0
for it in range(a.iters + 3):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(st)
    _native.check("pre", L.gsr_preprocess(s, P, 3, 16, p(t["xyz"]), p(t["sca"]), 1.0, p(t["rot"]), p(t["op"]), p(t["sh"]), None,
                                          None, p(t["view"]), p(t["proj"]), p(t["cp"]), W, H, tfx, tfy, 0, 0, 0, p(radii), p(geom),
                                          counts))
    e[1].record(st)
    R, G = int(counts[0]), int(counts[1])
    _, bb, _ = _native.scratch_sizes(P, R, W, H, G)
    binning = torch.empty(bb, dtype=torch.uint8, device=DEV)
    _native.check("bin", L.gsr_bin(s, P, R, G, W, H, p(geom), p(binning), p(img)))
    e[2].record(st)
    torch.cuda.synchronize()
    pl = binning[:4 * R].view(torch.int32)
    if ref_pl is None:
        ref_pl = pl.clone()
    elif not torch.equal(pl, ref_pl):
        print("POINT LIST CHANGED between iterations")
    if it >= 3:
        tp += e[0].elapsed_time(e[1])
        tb += e[1].elapsed_time(e[2])
print(f"P={P} {W}x{H} s0={a.s0}: R={R} G={G} binning scratch {bb / 1e6:.1f} MB; preprocess {1e3 * tp / a.iters:.1f} us, "
      f"bin {1e3 * tb / a.iters:.1f} us")
if a.oracle:
    from oracle import cpu as O

    O.build()
    f = O.forward(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], None, None, cam.world_view_transform,
                  cam.full_proj_transform, cam.camera_center, torch.zeros(3), W, H, tfx, tfy, 1.0, 3)
    import numpy as np

    print("point list == oracle:", np.array_equal(ref_pl.cpu().numpy().view(np.uint32), f["point_list"]))
