"""Timing experiment: how much of K8+K9 (gsr_preprocess_backward) is the SH gradient store / the SH load?"""
import ctypes
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussianeditor_amd import _native  # noqa: E402
from gaussianeditor_amd.diff_gaussian_rasterization import _C  # noqa: E402
from gaussianeditor_amd.synth import ring_cameras, seed_gradient, synth_scene  # noqa: E402

P, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda:0")
sc = synth_scene(P, seed=0, s0=0.01)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
d = lambda t: t.to(dev)  # noqa: E731
e = torch.empty(0, device=dev)
xyz, sca, rot, sh = d(sc["xyz"]), d(sc["scaling"]), d(sc["rotation"]), d(sc["features"])
vm, pm, cp = d(cam.world_view_transform), d(cam.full_proj_transform), d(cam.camera_center)
R, color, depth, radii, geom, binning, img = _C.rasterize_gaussians(d(sc["bg"]), xyz, e, d(sc["opacity"]), sca, rot, 1.0, e, vm, pm,
                                                                    tfx, tfy, H, W, sh, 3, cp, False, False)
L = _native.lib()
s = torch.cuda.current_stream(dev)
p = lambda t: t.data_ptr()  # noqa: E731
G = seed_gradient(H, W, 0).to(dev)
z = torch.zeros(P * 11, device=dev)
d_m2, d_col, d_op, d_con = z[:3 * P], z[3 * P:6 * P], z[6 * P:7 * P], z[7 * P:]
_native.check("bwd", L.gsr_blend_backward(s.cuda_stream, P, R, W, H, p(d(sc["bg"])), p(geom), p(binning), p(img), p(G), p(d_m2),
                                          p(d_con), p(d_op), p(d_col), 0))
d_m3, d_cov = torch.empty(P * 3, device=dev), torch.empty(P * 6, device=dev)
d_sh, d_sc, d_rot = torch.empty(P * 48, device=dev), torch.empty(P * 3, device=dev), torch.empty(P * 4, device=dev)


def run(name, shs, dsh):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(25):
        if it == 5:
            ev0.record(s)
        _native.check("pbw", L.gsr_preprocess_backward(s.cuda_stream, P, 3, 16 if shs is not None else 0, W, H, p(xyz),
                                                       p(shs) if shs is not None else None, p(sca), 1.0, p(rot), None, p(vm), p(pm),
                                                       p(cp), tfx, tfy, p(radii), p(geom), p(d_m2), p(d_con), p(d_col), p(d_m3),
                                                       p(d_cov), p(dsh) if dsh is not None else None, p(d_sc), p(d_rot)))
    ev1.record(s)
    torch.cuda.synchronize()
    print(f"{name:40s} {ev0.elapsed_time(ev1) / 20 * 1e3:8.1f} us")


run("full (SH load + SH grad store)", sh, d_sh)
run("no SH at all (precomputed-colour path)", None, None)
run("full again", sh, d_sh)
