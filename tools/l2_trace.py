#!/usr/bin/env python3
"""The headline iteration through the drop-in route an unmodified caller uses -- render() (L2) -> GaussianRasterizer (L1
autograd.Function) -> loss.backward() -- as a loop of its own, for a kernel trace (where does the GPU idle while the autograd
engine hands the backward to its thread?) and a launch-thread profile.
    python tools/l2_trace.py [--steps 100] [--profile out.txt] [--l0]
--l0: the same iteration through the L0 entry points (multiview_step without exchange), for comparison under the same trace."""
import argparse
import math
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--profile", default=None)
    ap.add_argument("--l0", action="store_true")
    ap.add_argument("--c3", action="store_true", help="the editor's loop instead: 512 x 512, render() + render(override_color) + backward")
    a = ap.parse_args()
    from gaussianeditor_amd.gaussian_renderer import render
    from gaussianeditor_amd.multiview import GradBucket, multiview_step
    from gaussianeditor_amd.diff_gaussian_rasterization import GaussianRasterizationSettings
    from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene

    dev = torch.device("cuda", 0)
    W, H = (512, 512) if a.c3 else (1920, 1080)
    sc = synth_scene(1_000_000, seed=0, s0=0.01)
    cam = ring_cameras(8, W, H)[0].to(dev)
    bg = sc["bg"].to(dev)
    G = seed_gradient(H, W, 0).to(dev)
    pipe = SimpleNamespace(compute_cov3D_python=False, convert_SHs_python=False, debug=False)

    class PC:
        def __init__(self):
            self.t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items() if isinstance(v, torch.Tensor) and k != "bg"}
            self.active_sh_degree, self.max_sh_degree = 3, 3

        get_xyz = property(lambda s: s.t["xyz"])
        get_opacity = property(lambda s: s.t["opacity"])
        get_scaling = property(lambda s: s.t["scaling"])
        get_rotation = property(lambda s: s.t["rotation"])
        get_features = property(lambda s: s.t["features"])

    pc = PC()
    if a.l0:
        params = {k: pc.t[k].detach() for k in ("xyz", "opacity", "features", "scaling", "rotation")}
        rs = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform,
                                           cam.full_proj_transform, 3, cam.camera_center, False, False)
        bucket = GradBucket(1_000_000, 16, dev, sh_exchange="auto")

        def step():
            multiview_step(rs, params, G, bucket, rows="auto")
    elif a.c3:
        mask = (torch.rand(1_000_000, 1, device=dev) > 0.5).float().repeat(1, 3)

        def step():
            out = render(cam, pc, pipe, bg)
            with torch.no_grad():
                render(cam, pc, pipe, bg, override_color=mask)  # the semantic pass (GassuianEditor.py:183-191)
            (out["render"] * G).sum().backward()
            for v in pc.t.values():
                v.grad = None
    else:
        def step():
            out = render(cam, pc, pipe, bg)
            (out["render"] * G).sum().backward()
            for v in pc.t.values():
                v.grad = None

    import gc
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    gc.disable()
    pr = None
    if a.profile:
        import cProfile
        pr = cProfile.Profile()
        pr.enable()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    if pr is not None:
        import io
        import pstats
        pr.disable()
        s = io.StringIO()
        st = pstats.Stats(pr, stream=s)
        st.sort_stats("cumulative").print_stats(45)
        st.sort_stats("tottime").print_stats(30)
        open(a.profile, "w").write(s.getvalue())
    print(f"{'L0' if a.l0 else ('C3' if a.c3 else 'L2')} route: {1e3 * dt:.4f} ms per iteration, {1 / dt:.1f} it/s")


if __name__ == "__main__":
    main()
