"""Development experiment (GPU): would the colours of K1 (192 B of SH per Gaussian, ~50 us) hide underneath the depth sort and
the binning if they ran on a second stream?  Times, on the headline view, preprocess(skip_color=1) + bin on the main stream
alone, with a full preprocess of the same scene running concurrently on a side stream (a stand-in for a colour-only kernel:
it reads the same 192 MB and more), and the serial sum.

    python tools/overlap_probe.py [--gaussians 1000000] [--s0 0.01]
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
ap.add_argument("--iters", type=int, default=40)
args = ap.parse_args()
dev = torch.device("cuda:0")
W, H, P = 1920, 1080, args.gaussians
sc = synth_scene(P, seed=0, s0=args.s0)
cam = ring_cameras(8, W, H)[0]
d = lambda t: t.to(dev)  # noqa: E731
e = torch.empty(0, device=dev)
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
xyz, op, scl, rot, feat = d(sc["xyz"]), d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), d(sc["features"])
wv, pj, cc = d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center)
radii = torch.empty(P, dtype=torch.int32, device=dev)
radii2 = torch.empty(P, dtype=torch.int32, device=dev)
M = feat.shape[1]


def pre_bin(skip):
    return _C._preprocess_and_bin(dev, P, 3, M, xyz, scl, 1.0, rot, op, feat, e, e, wv, pj, cc, W, H, tfx, tfy, False, skip, radii, 0)


def pre_only():
    from gaussianeditor_amd import _native
    import ctypes
    L = _native.lib()
    sizes = _native.scratch_sizes(P, 0, W, H)
    geom = torch.empty(sizes[0], dtype=torch.uint8, device=dev)
    counts = (ctypes.c_int64 * 2)()
    return geom, counts, L


side = torch.cuda.Stream(device=dev)
geom2, counts2, L = pre_only()
import ctypes  # noqa: E402


def full_pre_on(stream):
    from gaussianeditor_amd import _native
    _native.check("pre", L.gsr_preprocess(stream.cuda_stream, P, 3, M, xyz.data_ptr(), scl.data_ptr(), ctypes.c_float(1.0), rot.data_ptr(),
                                          op.data_ptr(), feat.data_ptr(), None, None, wv.data_ptr(), pj.data_ptr(), cc.data_ptr(), W, H,
                                          ctypes.c_float(tfx), ctypes.c_float(tfy), 0, 0, 0, radii2.data_ptr(), geom2.data_ptr(), counts2))


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    return 1e3 * ts[len(ts) // 2]


hi = torch.cuda.Stream(device=dev, priority=-1)  # the chain on a high-priority stream, the side work on a default one
torch.cuda.set_stream(hi)
main = torch.cuda.current_stream(dev)
t_full = timed(lambda: pre_bin(0), args.iters)
t_skip = timed(lambda: pre_bin(1), args.iters)


acc = torch.empty(P, 3, device=dev)


def stand_in():  # reads the 192 MB of SH coefficients once and writes 12 B per Gaussian, like a colour-only kernel
    torch.sum(feat, dim=1, out=acc)


def overlapped():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        stand_in()
    pre_bin(1)
    main.wait_stream(side)


def overlapped_late():  # the side work starts when the host returns from gsr_preprocess: underneath the rest of the chain
    from gaussianeditor_amd import _native
    st = main.cuda_stream
    gbytes, _, ibytes = _native.scratch_sizes(P, 0, W, H)
    geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
    img = torch.empty(ibytes, dtype=torch.uint8, device=dev)
    counts = (ctypes.c_int64 * 2)()
    _native.check("pre", L.gsr_preprocess(st, P, 3, M, xyz.data_ptr(), scl.data_ptr(), ctypes.c_float(1.0), rot.data_ptr(), op.data_ptr(),
                                          feat.data_ptr(), None, None, wv.data_ptr(), pj.data_ptr(), cc.data_ptr(), W, H,
                                          ctypes.c_float(tfx), ctypes.c_float(tfy), 0, 1, 0, radii.data_ptr(), geom.data_ptr(), counts))
    with torch.cuda.stream(side):  # (no wait: K1 has finished, the host has its counts)
        stand_in()
    R, G = int(counts[0]), int(counts[1])
    _, bbytes, _ = _native.scratch_sizes(P, R, W, H, G)
    binning = torch.empty(bbytes, dtype=torch.uint8, device=dev)
    _native.check("bin", L.gsr_bin(st, P, R, G, W, H, geom.data_ptr(), binning.data_ptr(), img.data_ptr()))
    main.wait_stream(side)


t_both = timed(overlapped, args.iters)
t_side = timed(stand_in, args.iters)
t_serial = timed(lambda: (stand_in(), pre_bin(1)), args.iters)
t_late = timed(overlapped_late, args.iters)
print(f"P={P} s0={args.s0}: preprocess+bin with colours {t_full:.1f} us; without colours {t_skip:.1f} us; "
      f"the stand-in alone {t_side:.1f} us; stand-in then preprocess+bin without colours, one stream {t_serial:.1f} us; "
      f"stand-in on a side stream from the start, joined {t_both:.1f} us; stand-in on a side stream once the host has the "
      f"counts (underneath the last sort pass and the binning), joined {t_late:.1f} us")
