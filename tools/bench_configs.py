#!/usr/bin/env python3
"""Timings for the other BASELINE.json configurations with the substitutes SURVEY.md section 8(d) prescribes
(the real assets -- bicycle.ply, the bear scene, diffusion weights -- are not available offline):

  C2  synth-v1(6,000,000) forward at 1920x1080                      (substitute for the 6 M-Gaussian bicycle.ply)
  C3  edit loop shape at 512x512: 2 forwards (SH colours, then override_color) + 1 backward per step
  C5  semantic tracing: apply_weights over 12 views with a 1-channel mask at 512x512

Not bench lines (bench.py reports the headline metric); run on the GPU box: python tools/bench_configs.py"""
import json
import math
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.gaussian_renderer import camera2rasterizer, render  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

dev = torch.device("cuda:0")
PIPE = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)


class PC:
    def __init__(self, sc, grad=False):
        self.t = {k: v.to(dev).requires_grad_(grad) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
        self.active_sh_degree, self.max_sh_degree = 3, 3

    get_xyz = property(lambda s: s.t["xyz"])
    get_opacity = property(lambda s: s.t["opacity"])
    get_scaling = property(lambda s: s.t["scaling"])
    get_rotation = property(lambda s: s.t["rotation"])
    get_features = property(lambda s: s.t["features"])


def timed(fn, steps=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


out = {}
# ---- C2 ----
sc = synth_scene(6_000_000, seed=0, s0=0.01)
pc = PC(sc)
cam = ring_cameras(8, 1920, 1080)[0].to(dev)
bg = sc["bg"].to(dev)
with torch.no_grad():
    r = render(cam, pc, PIPE, bg)
    t = timed(lambda: render(cam, pc, PIPE, bg), steps=10)
out["C2_synth6M_1080p_forward"] = {"ms": 1e3 * t, "renders_per_s": 1 / t, "mpixels_per_s": 1920 * 1080 / t / 1e6,
                                   "visible": int((r["radii"] > 0).sum())}
del pc, sc, r
torch.cuda.empty_cache()

# ---- C3 ----
sc = synth_scene(1_000_000, seed=0, s0=0.01)
pc = PC(sc, grad=True)
cam = ring_cameras(8, 512, 512)[0].to(dev)
G = seed_gradient(512, 512, 0).to(dev)
mask = (torch.rand(1_000_000, 1, device=dev) > 0.5).float().repeat(1, 3)


def edit_step():
    a = render(cam, pc, PIPE, bg)
    with torch.no_grad():
        render(cam, pc, PIPE, bg, override_color=mask)  # the semantic pass, thresholded by the caller (GassuianEditor.py:183-191)
    (a["render"] * G).sum().backward()
    for v in pc.t.values():
        v.grad = None


t = timed(edit_step)
out["C3_edit_loop_512_1M"] = {"ms_per_step": 1e3 * t, "steps_per_s": 1 / t}


def edit_step_fused():  # the same two images from ONE preprocessing / sort: render(..., semantic_color=mask)
    a = render(cam, pc, PIPE, bg, semantic_color=mask)
    (a["render"] * G).sum().backward()
    for v in pc.t.values():
        v.grad = None


t = timed(edit_step_fused)
out["C3_edit_loop_512_1M_fused_semantic"] = {"ms_per_step": 1e3 * t, "steps_per_s": 1 / t}

# ---- C5 ----
cams = [c.to(dev) for c in ring_cameras(12, 512, 512)]
masks = [(torch.rand(1, 512, 512, device=dev) > 0.5).float() for _ in cams]
zero_bg = torch.zeros(3, device=dev)


def trace_all():
    w = torch.zeros(1_000_000, 1, device=dev)
    cnt = torch.zeros(1_000_000, 1, dtype=torch.int32, device=dev)
    with torch.no_grad():
        for c, m in zip(cams, masks):
            camera2rasterizer(c, zero_bg).apply_weights(pc.get_xyz, None, pc.get_opacity, None, w, pc.get_scaling,
                                                        pc.get_rotation, None, cnt, m)
    return w, cnt


t = timed(trace_all, steps=5, warmup=1)
out["C5_apply_weights_12views_512_1M"] = {"ms_total": 1e3 * t, "ms_per_view": 1e3 * t / len(cams)}
print(json.dumps(out, indent=1))
