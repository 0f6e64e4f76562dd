#!/usr/bin/env python3
"""Development tool (GPU): the fixed 8-view batch on one GPU (multiview_batch_step), pipelined and serial, several repetitions,
with Python's cyclic GC on / off -- what bench.py's extra_configs.views8_one_gpu measures, in isolation.
    python tools/views8_probe.py [--reps 5] [--steps 20]"""
import argparse
import gc
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gaussianeditor_amd.multiview as mv  # noqa: E402
from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--no-serial", action="store_true", help="only the pipelined form")
ap.add_argument("--timeline", action="store_true", help="event-stamp the forward and the backward of every view of one extra step per round")
a = ap.parse_args()
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, 1_000_000
sc = synth_scene(P, seed=0, s0=0.01)
ring = ring_cameras(8, W, H)
p1 = {k: sc[k].to(dev) for k in ("xyz", "opacity", "features", "scaling", "rotation")}
rs8 = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx / 2), math.tan(c.FoVy / 2), sc["bg"].to(dev), 1.0, c.world_view_transform.to(dev),
                                     c.full_proj_transform.to(dev), 3, c.camera_center.to(dev), False, False) for c in ring]
G = seed_gradient(H, W, 0).to(dev)


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def dev_allocs():
    st = torch.cuda.memory_stats(dev)
    return st.get("num_device_alloc", 0), st.get("reserved_bytes.all.current", 0) >> 20


# the same two forms alternately, several times over: does the FIRST one measured differ from the later ones?
for rnd in range(a.rounds):
    for gc_on in (True,):
        for name, on in ((("pipelined", True),) if a.no_serial else (("pipelined", True), ("serial", False))):
            mv._VIEW_PIPELINE = on
            b8 = mv.GradBucket(P, 16, dev, sh_exchange="rgb")
            timed(lambda: mv.multiview_batch_step(rs8, p1, [G] * 8, b8), 10, 5)
            a0 = dev_allocs()
            runs = []
            for _ in range(a.reps):
                if not gc_on:
                    gc.collect()
                    gc.disable()
                t = timed(lambda: mv.multiview_batch_step(rs8, p1, [G] * 8, b8), a.steps, 3)
                gc.enable()
                runs.append(8.0 / t)
            runs.sort()
            a1 = dev_allocs()
            print(f"round {rnd} gc {'on ' if gc_on else 'off'} {name:9s}: view-it/s median {runs[len(runs) // 2]:7.1f}  min {runs[0]:7.1f}  max {runs[-1]:7.1f}"
                  f" | hipMalloc calls during the timing {a1[0] - a0[0]}, reserved {a1[1]} MiB", flush=True)
            if a.timeline and on:
                stamps = []
                f0, b0 = mv._view_forward, mv._view_backward

                def stamped(kind, fn):
                    def run(*args, **kw):
                        st = torch.cuda.current_stream(dev)
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        t_host = time.perf_counter()
                        e0.record(st)
                        r = fn(*args, **kw)
                        e1.record(st)
                        stamps.append((kind, st.cuda_stream, e0, e1, t_host, time.perf_counter()))
                        return r
                    return run

                mv._view_forward, mv._view_backward = stamped("fwd", f0), stamped("bwd", b0)
                try:
                    base = torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize(dev)
                    th = time.perf_counter()
                    base.record(torch.cuda.current_stream(dev))
                    mv.multiview_batch_step(rs8, p1, [G] * 8, b8)
                    mv.multiview_batch_step(rs8, p1, [G] * 8, b8)
                    torch.cuda.synchronize(dev)
                finally:
                    mv._view_forward, mv._view_backward = f0, b0
                ids = {}
                line = []
                for kind, sid, e0, e1, h0, h1 in stamps[16:]:
                    ids.setdefault(sid, len(ids))
                    line.append(f"{kind}@S{ids[sid]} gpu {base.elapsed_time(e0) * 1e3:7.0f}-{base.elapsed_time(e1) * 1e3:7.0f} host {(h0 - th) * 1e6:7.0f}-{(h1 - th) * 1e6:7.0f}")
                print("   timeline (us, second stamped step):\n     " + "\n     ".join(line), flush=True)
            del b8
