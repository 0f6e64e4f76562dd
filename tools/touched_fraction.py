#!/usr/bin/env python3
"""Fraction of the Gaussians each ring view of the benchmark scene produces gradients for (the rows the touched-rows
exchange of gaussianeditor_amd/multiview.py sends), and the time of the exchange's local kernels for 8 views.
Run on the GPU box: python tools/touched_fraction.py"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings, _C  # noqa: E402
from gaussianeditor_amd.multiview import _ROW_SEGS, GradBucket, render_view_grads  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

P, W, H, M = 1_000_000, 1920, 1080, 16
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=0.01, sh_degree=3)
params = [sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")]
G = seed_gradient(H, W, 0).to(dev)
buckets, plans = [], []
cams = ring_cameras(8, W, H)
for v, cam in enumerate(cams):
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc["bg"].to(dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                       cam.camera_center.to(dev), False, False)
    b = GradBucket(P, M, dev, sh_exchange="rgb")
    render_view_grads(rs, *params, G, b)
    buckets.append(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan, count = _C.view_message_plan([b.views[n] for n in _ROW_SEGS], b.rgb)
    t1 = time.perf_counter()
    plans.append((plan, count))
    print(f"view {v}: {count} touched rows = {100.0 * count / P:.1f} % ({72 * count / 1e6:.1f} MB), plan {1e3 * (t1 - t0):.3f} ms")
cap = max(c for _, c in plans)
words = _C.view_message_words(P, cap)
messages = torch.empty((8, words), device=dev)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _C.view_message_pack(plans[0][0], [buckets[0].views[n] for n in _ROW_SEGS], buckets[0].rgb, cams[0].camera_center.to(dev), cap,
                         messages[0])
    torch.cuda.synchronize()
    t1 = time.perf_counter()
print(f"pack one message ({4 * words / 1e6:.1f} MB): {1e3 * (t1 - t0):.3f} ms")
for v in range(1, 8):
    _C.view_message_pack(plans[v][0], [buckets[v].views[n] for n in _ROW_SEGS], buckets[v].rgb, cams[v].camera_center.to(dev), cap,
                         messages[v])
dense = [torch.empty((P, k), device=dev) for k in (3, 3, 4, 3, 1)] + [torch.empty((P, M, 3), device=dev)]
for nv in (2, 4, 8):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _C.view_messages_accumulate(messages[:nv], P, cap, 3, M, params[0], dense)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    print(f"accumulate {nv} views into the dense gradients: {1e3 * (t1 - t0):.3f} ms")
    valid = torch.empty(P, dtype=torch.uint8, device=dense[0].device)
    ref = [t.clone() for t in dense]
    for t in dense:
        t.fill_(float("nan"))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _C.view_messages_accumulate(messages[:nv], P, cap, 3, M, params[0], dense, row_valid=valid)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    vb = valid.bool()
    same = all(torch.equal(t[vb], r[vb]) for t, r in zip(dense, ref)) and all(bool((r[~vb] == 0).all()) and bool(t[~vb].isnan().all())
                                                                               for t, r in zip(dense, ref))
    print(f"   only the rows some view touched ({100.0 * float(vb.float().mean()):.1f} %): {1e3 * (t1 - t0):.3f} ms; "
          f"valid rows identical, invalid rows untouched: {same}")
