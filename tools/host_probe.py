#!/usr/bin/env python3
"""Where the HOST spends an iteration of the headline train step (bench.py's multiview_step through the autograd drop-in):
perf_counter stamps around every C-ABI call of a step (no added synchronisation), then a cProfile of the same loop.
The GPU timeline of the same step is profiles/*_kernel_stats.md; together they say whether the 20-30 us the GPU idles
between the forward's last kernel and the backward's first are host time and whose."""
import argparse
import cProfile
import io
import math
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--s0", type=float, default=0.01)
    a = ap.parse_args()
    from gaussianeditor_amd import _native
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.multiview import GradBucket, multiview_step
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene

    dev = torch.device("cuda:0")
    W, H = 979, 546
    sc = synth_scene(a.gaussians, seed=0, s0=a.s0, sh_degree=3)
    cam = ring_cameras(8, W, H)[0]
    params = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
    G = seed_gradient(H, W, 0).to(dev)
    rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), sc["bg"].to(dev), 1.0,
                                       cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3,
                                       cam.camera_center.to(dev), False, False)
    bucket = GradBucket(a.gaussians, sc["features"].shape[1], dev)
    L = _native.lib()
    log = []
    names = [n for n in _native.SIGNATURES if n.startswith("gsr_")]
    for n in names:
        f = getattr(L, n)

        def wrap(f=f, n=n):
            def g(*args):
                t0 = time.perf_counter_ns()
                r = f(*args)
                log.append((n, t0, time.perf_counter_ns()))
                return r
            return g
        setattr(L, n, wrap())

    def step():
        multiview_step(rs, params, G, bucket)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    rows = {}
    t_begin = time.perf_counter_ns()
    for _ in range(a.steps):
        log.clear()
        t0 = time.perf_counter_ns()
        step()
        t1 = time.perf_counter_ns()
        for n, s, e in log:
            r = rows.setdefault(n, [0, 0.0, 0.0, 0.0])
            r[0] += 1
            r[1] += (s - t0) / 1e3
            r[2] += (e - s) / 1e3
        rows.setdefault("(step returns)", [0, 0.0, 0.0, 0.0])
        rows["(step returns)"][0] += 1
        rows["(step returns)"][1] += (t1 - t0) / 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter_ns() - t_begin) / 1e3 / a.steps
    print(f"## host timeline of one train step, mean of {a.steps} (P = {a.gaussians}, s0 = {a.s0}); iteration {total:.1f} us wall\n")
    print("| C-ABI call | calls / step | starts at us (host, from the step's start) | host time inside us |")
    print("|---|---:|---:|---:|")
    for n, r in sorted(rows.items(), key=lambda kv: kv[1][1] / max(kv[1][0], 1)):
        print(f"| `{n}` | {r[0] / a.steps:.2f} | {r[1] / r[0]:.1f} | {r[2] / r[0]:.1f} |")
    return
    for n in names:  # (disabled: a fresh ctypes function object loses its argtypes)
        delattr(L, n)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(a.steps):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("\n## cProfile of the same loop (tottime; the profiler roughly doubles the Python share)\n\n```")
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
    print("```")


if __name__ == "__main__":
    main()
