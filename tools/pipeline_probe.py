"""Development experiment (GPU): two-stream VIEW PIPELINING on one GPU.  When a rank renders several views
per step (multiview_batch_step), view v + 1's forward -- K1, the depth sort and the binning: a bandwidth kernel and a chain of
small latency-bound kernels -- could run on a second stream underneath view v's backward, whose K7 is bound by VALU issue.

Measured here through the L1 API (forward = GaussianRasterizer, backward = torch.autograd.grad with the seed gradient), the
same work in three schedules, n views each, wall clock between two device synchronisations:
  serial      forward(v), backward(v) one after the other on one stream                      (what the batch step does)
  pipelined   backward(v) enqueued on stream v % 2, THEN forward(v + 1) on the other stream  (the host blocks in the forward's
              one readback while the backward runs)
  pipelined3w the same with the blend kernels at 3 waves per SIMD (GSR_BLEND_WAVES_PER_SIMD=3: their persistent waves then
              leave register space for the other stream's kernels) -- run as a second process by the caller.

    python tools/pipeline_probe.py [--views 40] [--gaussians 1000000]
"""
import argparse
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=40)
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--s0", type=float, default=0.01)
args = ap.parse_args()
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, args.gaussians
sc = synth_scene(P, seed=0, s0=args.s0)
ring = ring_cameras(8, W, H)
params = [sc[k].to(dev) for k in ("xyz", "features", "opacity", "scaling", "rotation")]
G = seed_gradient(H, W, 0).to(dev)
bg = sc["bg"].to(dev)


def settings(v):
    c = ring[v % 8]
    return GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), bg, 1.0, c.world_view_transform.to(dev),
                                         c.full_proj_transform.to(dev), 3, c.camera_center.to(dev), False, False)


RS = [settings(v) for v in range(8)]


def forward(v):
    leaves = [t.detach().requires_grad_(True) for t in params]
    m3, sh, op, scl, rot = leaves
    m2 = torch.empty_like(m3).requires_grad_(True)
    color, radii, depth = GaussianRasterizer(RS[v % 8])(m3, m2, op, shs=sh, scales=scl, rotations=rot)
    return color, leaves + [m2]


def backward(state):
    color, leaves = state
    return torch.autograd.grad([color], leaves, grad_outputs=[G])


def serial(n):
    for v in range(n):
        backward(forward(v))


S = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def pipelined(n):
    with torch.cuda.stream(S[0]):
        st = forward(0)
    for v in range(n):
        with torch.cuda.stream(S[v % 2]):
            g = backward(st)  # enqueued behind forward(v) on the same stream
        if v + 1 < n:
            with torch.cuda.stream(S[(v + 1) % 2]):
                st = forward(v + 1)  # the other stream: runs underneath backward(v)
        del g


def timed(fn, n):
    fn(4)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize(dev)
    return 1e3 * (time.perf_counter() - t0) / n


waves = os.environ.get("GSR_BLEND_WAVES_PER_SIMD", "4 (default)")
a = timed(serial, args.views)
b = timed(pipelined, args.views)
a2 = timed(serial, args.views)
b2 = timed(pipelined, args.views)
print(f"P={P} s0={args.s0} blend waves/SIMD={waves}: serial {a:.4f} / {a2:.4f} ms per view, two-stream pipelined {b:.4f} / {b2:.4f} ms per view")
