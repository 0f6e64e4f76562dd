"""Development aid: what would launching the path's kernel chains as hipGraphs save?

`tools/microbench/grid_barrier.hip` measured 3.1 us per small dependent kernel launched into a stream against 1.9 us for the
same kernels as nodes of a graph.  This probe replays the REAL chains of the headline view both ways (everything after the
forward's one host readback: gsr_bin + gsr_blend_forward + gsr_blend_backward + gsr_preprocess_backward = 13 kernels), from
a side stream, and compares the time per pass and the results.  (gsr_preprocess reads its counts back on the host, so it
cannot be captured as it stands; its 12 kernels would need a graph built inside the library.)
Measured (profiles/r03_p_launch_floor.md): the graph replays are 4-6 us per pass SLOWER than the stream launches -- the chain's
kernels already follow each other without gaps, and a replay adds its own start-up.  Nothing to gain there.
    python tools/graph_probe.py [--gaussians N] [--s0 S] [--reps K]"""
import argparse
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd import _native  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--s0", type=float, default=0.01)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=40)
a = ap.parse_args()
dev = torch.device("cuda", 0)
P, W, H, M = a.gaussians, a.width, a.height, 16
sc = synth_scene(P, seed=0, s0=a.s0)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(dev).contiguous()  # noqa: E731
xyz, sca, rot, op, sh = d(sc["xyz"]), d(sc["scaling"]), d(sc["rotation"]), d(sc["opacity"]), d(sc["features"])
view, proj, cp, bg = d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center), torch.zeros(3, device=dev)
G = seed_gradient(H, W, 0).to(dev)
p = lambda x: x.data_ptr()  # noqa: E731
L = _native.lib()
side = torch.cuda.Stream(device=dev)
gb, _, ib = _native.scratch_sizes(P, 0, W, H)
geom = torch.empty(gb, dtype=torch.uint8, device=dev)
img = torch.empty(ib, dtype=torch.uint8, device=dev)
radii = torch.empty(P, dtype=torch.int32, device=dev)
color, depth = torch.empty((3, H, W), device=dev), torch.empty((1, H, W), device=dev)
z = torch.empty(P * 11, device=dev)
d_m2, d_col, d_op, d_con = z[:3 * P], z[3 * P:6 * P], z[6 * P:7 * P], z[7 * P:]
d_m3, d_cov = torch.empty(P * 3, device=dev), torch.empty(P * 6, device=dev)
d_sh, d_sc, d_rot = torch.empty(P * M * 3, device=dev), torch.empty(P * 3, device=dev), torch.empty(P * 4, device=dev)
counts = (ctypes.c_int64 * 2)()
with torch.cuda.stream(side):
    s = side.cuda_stream
    _native.check("pre", L.gsr_preprocess(s, P, 3, M, p(xyz), p(sca), 1.0, p(rot), p(op), p(sh), None, None, p(view), p(proj), p(cp),
                                          W, H, tfx, tfy, 0, 0, 0, p(radii), p(geom), counts))
    R, Gi = int(counts[0]), int(counts[1])
    _, bb, _ = _native.scratch_sizes(P, R, W, H, Gi)
    binning = torch.empty(bb, dtype=torch.uint8, device=dev)

    def chain(which):
        if "bin" in which:
            _native.check("bin", L.gsr_bin(s, P, R, Gi, W, H, p(geom), p(binning), p(img)))
        if "fwd" in which:
            _native.check("fwd", L.gsr_blend_forward(s, P, R, W, H, p(bg), p(geom), p(binning), p(img), p(color), p(depth), 0))
        if "bwd" in which:
            _native.check("bwd", L.gsr_blend_backward(s, P, R, W, H, p(bg), p(geom), p(binning), p(img), p(G), p(d_m2), p(d_con),
                                                      p(d_op), p(d_col), 4))
            _native.check("pbw", L.gsr_preprocess_backward(s, P, 3, M, W, H, p(xyz), p(sh), p(sca), 1.0, p(rot), None, p(view),
                                                           p(proj), p(cp), tfx, tfy, p(radii), p(geom), p(d_m2), p(d_con), p(d_col),
                                                           p(d_m3), p(d_cov), p(d_sh), p(d_sc), p(d_rot)))

    def timed(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            e0.record(side)
            for _ in range(reps):
                fn()
            e1.record(side)
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
        return best

    print(f"P={P} {W}x{H} s0={a.s0}: R={R} G={Gi}")
    for which in (("bin",), ("bin", "fwd"), ("bwd",), ("bin", "fwd", "bwd")):
        chain(which)
        side.synchronize()
        want = [t.clone() for t in (color, d_m3, d_sh)]
        direct = timed(lambda: chain(which), a.reps)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            chain(which)
        color.zero_(); d_m3.zero_(); d_sh.zero_()
        graph = timed(g.replay, a.reps)
        side.synchronize()
        pairs = list(zip((color, d_m3, d_sh), want))
        pairs = (pairs[1:] if "fwd" not in which else pairs) if "bwd" in which else (pairs[:1] if "fwd" in which else [])
        # (the blend backward accumulates with float atomics: two runs agree to rounding, not bit for bit)
        same = all(torch.allclose(x, y, rtol=0, atol=1e-5 * float(y.abs().max()) + 1e-30) for x, y in pairs)
        print(f"  {'+'.join(which):12s}: stream launches {direct:7.1f} us per pass, graph replay {graph:7.1f} us per pass "
              f"({direct - graph:+.1f} us); results {'equal' if same else 'DIFFER'}")
