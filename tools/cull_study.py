"""Development tool (CPU, uses the oracle): how tight is the blend kernels' per-quadrant cull on the headline scene?

For a sample of tiles of synth-v1 (1 M Gaussians, 1080p, view 0) and each 8x8 quadrant: of the list entries a wave walks
(up to the quadrant's largest n_contrib), how many pass (a) the present box test (can_touch_quad), (b) an exact test of
the alpha >= 1/255 ellipse against the quadrant's pixel-centre rectangle, (c) have a pixel that really blends them.

    python tools/cull_study.py [P] [tiles_sampled]
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402
from oracle import cpu as oracle  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s0 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
W, H = 1920, 1080
sc = synth_scene(P, seed=0, s0=s0)
cam = ring_cameras(8, W, H)[0]
tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
n = lambda t: t.numpy()  # noqa: E731
geom = oracle.preprocess(n(sc["xyz"]), n(sc["scaling"]), n(sc["rotation"]), n(sc["opacity"]), n(sc["features"]), None, None,
                         n(cam.world_view_transform), n(cam.full_proj_transform), n(cam.camera_center), W, H, tfx, tfy, 1.0, 3)
binning = oracle.bin_tiles(geom, W, H)
fw = oracle.blend_forward(geom, binning, geom["rgb"], n(sc["bg"]), W, H)
ncontrib = fw["n_contrib"].reshape(H, W)
gx, gy = (W + 15) // 16, (H + 15) // 16
ranges = binning["ranges"]
nonempty = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
rng = np.random.default_rng(0)
tiles = rng.choice(nonempty, size=min(NT, len(nonempty)), replace=False)
co = geom["conic_opacity"].astype(np.float64)
xy = geom["means2D"].astype(np.float64)
tot = dict(walked=0, box=0, exact=0, pixel=0, pix_lanes=0)
for t in tiles:
    ty, tx = divmod(int(t), gx)
    ids = binning["point_list"][ranges[t, 0]:ranges[t, 1]]
    for q in range(4):
        x0, y0 = tx * 16 + 8 * (q & 1), ty * 16 + 8 * (q >> 1)
        if x0 >= W or y0 >= H:
            continue
        nc = ncontrib[y0:min(y0 + 8, H), x0:min(x0 + 8, W)]
        walked = int(nc.max())
        if walked == 0:
            continue
        g = ids[:walked]
        A, B, C, o = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
        mx, my = xy[g, 0], xy[g, 1]
        det = A * C - B * B
        tau2 = 2.0 * np.log(np.maximum(255.0 * o, 1e-30))
        ok_o = o >= 1.0 / 255.0
        with np.errstate(invalid="ignore", divide="ignore"):
            hx = np.sqrt(np.maximum(tau2, 0) * C / det)
            hy = np.sqrt(np.maximum(tau2, 0) * A / det)
        box = ok_o & ~((mx + hx < x0) | (mx - hx > x0 + 7) | (my + hy < y0) | (my - hy > y0 + 7))
        # exact: minimum of the quadratic form over the rectangle [x0, x0+7] x [y0, y0+7]
        # candidates: the centre clamped (if inside: 0), and the minima along the four edges
        def qf(dx, dy):
            return A * dx * dx + 2 * B * dx * dy + C * dy * dy
        xl, xh, yl, yh = x0 - mx, x0 + 7 - mx, y0 - my, y0 + 7 - my  # rectangle relative to the mean
        inside = (xl <= 0) & (xh >= 0) & (yl <= 0) & (yh >= 0)
        best = np.full(len(g), np.inf)
        for yy in (yl, yh):  # horizontal edges: minimise over dx in [xl, xh] with dy fixed: dx* = -B dy / A
            dxs = np.clip(-B * yy / A, xl, xh)
            best = np.minimum(best, qf(dxs, yy))
        for xx in (xl, xh):
            dys = np.clip(-B * xx / C, yl, yh)
            best = np.minimum(best, qf(xx, dys))
        best = np.where(inside, 0.0, best)
        exact = ok_o & (best <= tau2)
        # pixels
        px = np.arange(x0, min(x0 + 8, W))[None, None, :] - mx[:, None, None]
        py = np.arange(y0, min(y0 + 8, H))[None, :, None] - my[:, None, None]
        power = -0.5 * (A[:, None, None] * px * px + C[:, None, None] * py * py) - B[:, None, None] * px * py
        alpha = np.minimum(0.99, o[:, None, None] * np.exp(power))
        hit = (power <= 0) & (alpha >= 1.0 / 255.0)
        # only positions a pixel still evaluates (pos < its n_contrib): the lanes that are not done
        live = np.arange(walked)[:, None, None] < nc[None, :, :]
        pix = (hit & live).any(axis=(1, 2))
        tot["walked"] += walked
        tot["box"] += int(box.sum())
        tot["exact"] += int(exact.sum())
        tot["pixel"] += int(pix.sum())
        tot["pix_lanes"] += int((hit & live).sum())
print(f"P {P} s0 {s0}: tiles sampled {len(tiles)}, R {binning['num_rendered']}")
print({k: v for k, v in tot.items()})
print("passes box / walked %.3f; exact rectangle test / box %.3f; really blended by a pixel / box %.3f; lanes used per box pair %.1f of 64" % (
    tot["box"] / tot["walked"], tot["exact"] / tot["box"], tot["pixel"] / tot["box"], tot["pix_lanes"] / tot["box"]))
