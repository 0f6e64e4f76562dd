#!/usr/bin/env python3
"""Development tool (GPU): cProfile of the launch thread over the pipelined 8-view batch step (multiview_batch_step) -- where
the host's time per view goes (tools/views8_probe.py --timeline shows the pipelined form is bound by it).
    python tools/host_profile.py [--steps 30] [--serial]"""
import argparse
import cProfile
import math
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gaussianeditor_amd.multiview as mv  # noqa: E402
from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--serial", action="store_true")
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
sc = synth_scene(P, seed=0, s0=0.01)
ring = ring_cameras(8, W, H)
p1 = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
rs8 = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), sc["bg"].to(dev), 1.0, c.world_view_transform.to(dev),
                                     c.full_proj_transform.to(dev), 3, c.camera_center.to(dev), False, False) for c in ring]
G = seed_gradient(H, W, 0).to(dev)
mv._VIEW_PIPELINE = not a.serial
b8 = mv.GradBucket(P, 16, dev, sh_exchange="rgb")
for _ in range(10):
    mv.multiview_batch_step(rs8, p1, [G] * 8, b8)
torch.cuda.synchronize(dev)
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    mv.multiview_batch_step(rs8, p1, [G] * 8, b8)
torch.cuda.synchronize(dev)
pr.disable()
views = a.steps * 8
st = pstats.Stats(pr)
print(f"{views} views; times below are totals over them (divide by {views} for per view)")
for key in ("tottime", "cumulative"):
    print(f"==== by {key}")
    st.sort_stats(key)
    st.print_stats(a.top)
