"""Development tool (CPU, uses the oracle): how tight is the blend kernels' per-quadrant cull on the headline scene?

For a sample of tiles of synth-v1 (1 M Gaussians, 1080p, view 0) and each 8x8 quadrant: of the list entries a wave walks
(up to the quadrant's largest n_contrib), how many pass (a) the present box test (can_touch_quad), (b) an exact test of
the alpha >= 1/255 ellipse against the quadrant's pixel-centre rectangle, (c) have a pixel that really blends them.

    python tools/cull_study.py [P] [tiles_sampled] [s0] [--live]

--live adds the study behind `live_pixel_box` (gsr_blend.hip): for every quadrant item of the view, the entries the exact
test keeps against the whole quadrant and against the bounding box of the pixels that are still live when a chunk starts
(approximated by n_contrib: a pixel is live at least up to its last contributor), for the 300 items with the most
evaluated entries, the 20 longest of those, and 600 random items; plus how well the list length predicts an item's work.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402
from oracle import cpu as oracle  # noqa: E402

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
P = int(_pos[0]) if len(_pos) > 0 else 1_000_000
NT = int(_pos[1]) if len(_pos) > 1 else 300
s0 = float(_pos[2]) if len(_pos) > 2 else 0.01
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


def _exact_mask(g, x0, y0, w, h):
    A, B, C, o = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
    mx, my = xy[g, 0], xy[g, 1]
    tau2 = 2.0 * np.log(np.maximum(255.0 * o, 1e-30))
    xl, xh, yl, yh = x0 - mx, x0 + w - mx, y0 - my, y0 + h - my
    cx, cy = np.clip(0, xl, xh), np.clip(0, yl, yh)
    x1, y2 = np.clip(-B * cy / A, xl, xh), np.clip(-B * cx / C, yl, yh)
    q1 = A * x1 * x1 + 2 * B * x1 * cy + C * cy * cy
    q2 = A * cx * cx + 2 * B * cx * y2 + C * y2 * y2
    return (o >= 1.0 / 255.0) & (np.minimum(q1, q2) <= tau2)


def _live_study():
    items = []
    for t in nonempty:
        ty, tx = divmod(int(t), gx)
        ids = binning["point_list"][ranges[t, 0]:ranges[t, 1]]
        for q in range(4):
            x0, y0 = tx * 16 + 8 * (q & 1), ty * 16 + 8 * (q >> 1)
            if x0 >= W or y0 >= H:
                continue
            walked = int(ncontrib[y0:y0 + 8, x0:x0 + 8].max())
            if walked:
                items.append((int(_exact_mask(ids[:walked], x0, y0, 7, 7).sum()), int(t), q, walked, len(ids)))
    items = np.array(items)
    print("quadrant items %d; evaluated entries: total %d, mean %.0f, p90 %.0f, p99 %.0f, max %d" % (
        len(items), items[:, 0].sum(), items[:, 0].mean(), np.percentile(items[:, 0], 90), np.percentile(items[:, 0], 99), items[:, 0].max()))
    print("corr(evaluated, list length) %.3f; corr(evaluated, positions walked) %.3f" % (
        np.corrcoef(items[:, 0], items[:, 4])[0, 1], np.corrcoef(items[:, 0], items[:, 3])[0, 1]))

    def study(sel, label):
        per = []
        for (_v, t, q, _w, _l) in sel:
            ty, tx = divmod(int(t), gx)
            ids = binning["point_list"][ranges[t, 0]:ranges[t, 1]]
            x0, y0 = tx * 16 + 8 * (int(q) & 1), ty * 16 + 8 * (int(q) >> 1)
            nc = ncontrib[y0:y0 + 8, x0:x0 + 8].astype(np.int64)
            full = live = 0
            for c0 in range(0, int(nc.max()), 64):
                g = ids[c0:min(c0 + 64, int(nc.max()))]
                full += int(_exact_mask(g, x0, y0, 7, 7).sum())
                ys, xs = np.nonzero(nc > c0)
                live += int(_exact_mask(g, x0 + xs.min(), y0 + ys.min(), xs.max() - xs.min(), ys.max() - ys.min()).sum())
            per.append((full, live))
        per = np.array(per)
        print("%s %d items: %d entries against the whole quadrant, %d against the live pixels' box (%.3f); the 20 longest: %.3f" % (
            label, len(sel), per[:, 0].sum(), per[:, 1].sum(), per[:, 1].sum() / per[:, 0].sum(), per[:20, 1].sum() / per[:20, 0].sum()))

    study(items[np.argsort(-items[:, 0])[:300]], "the 300 longest")
    study(items[np.random.default_rng(1).choice(len(items), min(600, len(items)), replace=False)], "random")


if "--live" in sys.argv:
    _live_study()
