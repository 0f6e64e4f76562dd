"""Development tool (CPU, oracle): what would a 4x4-pixel decomposition of the blend kernels buy?

Today one wave evaluates an entry for its whole 8x8 quadrant (one (quadrant, entry) pair per entry step, 64 lanes, of which
about a third blend).  Alternative: the four 16-lane DPP rows of a wave own the four 4x4 sub-blocks of the quadrant and walk
their OWN compacted lists, one (sub-block, entry) pair per row and step.  For sampled tiles of the headline view this counts
    E_q   (quadrant, entry) pairs the exact cull keeps (what the kernels evaluate today, live rectangle not modelled)
    E_s   (sub-block, entry) pairs the exact cull keeps against the 4x4 sub-blocks
    S_new wave steps of the row decomposition = sum over quadrants and chunks of max over the 4 rows of their survivors
and the lanes that really blend.  usage: python tools/subblock_study.py [P] [tiles] [s0]"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianeditor_amd.synth import ring_cameras, synth_scene  # noqa: E402
from oracle import cpu as oracle  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 200
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
gx = (W + 15) // 16
ranges = binning["ranges"]
nonempty = np.nonzero(ranges[:, 1] > ranges[:, 0])[0]
tiles = np.random.default_rng(0).choice(nonempty, size=min(NT, len(nonempty)), replace=False)
co = geom["conic_opacity"].astype(np.float64)
xy = geom["means2D"].astype(np.float64)


def exact(g, x0, y0, w, h):
    A, B, C, o = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
    mx, my = xy[g, 0], xy[g, 1]
    tau2 = 2.0 * np.log(np.maximum(255.0 * o, 1e-30))
    xl, xh, yl, yh = x0 - mx, x0 + w - mx, y0 - my, y0 + h - my
    cx, cy = np.clip(0, xl, xh), np.clip(0, yl, yh)
    x1, y2 = np.clip(-B * cy / A, xl, xh), np.clip(-B * cx / C, yl, yh)
    q1 = A * x1 * x1 + 2 * B * x1 * cy + C * cy * cy
    q2 = A * cx * cx + 2 * B * cx * y2 + C * y2 * y2
    return (o >= 1.0 / 255.0) & (np.minimum(q1, q2) <= tau2)


tot = dict(walked=0, Eq=0, Es=0, steps_new=0, steps_new_cum=0, lanes=0)
for t in tiles:
    ty, tx = divmod(int(t), gx)
    ids = binning["point_list"][ranges[t, 0]:ranges[t, 1]]
    for q in range(4):
        x0, y0 = tx * 16 + 8 * (q & 1), ty * 16 + 8 * (q >> 1)
        if x0 + 8 > W or y0 + 8 > H:
            continue
        nc = ncontrib[y0:y0 + 8, x0:x0 + 8]
        walked = int(nc.max())
        if walked == 0:
            continue
        g = ids[:walked]
        A, B, C, o = co[g, 0], co[g, 1], co[g, 2], co[g, 3]
        px = np.arange(x0, x0 + 8)[None, None, :] - xy[g, 0][:, None, None]
        py = np.arange(y0, y0 + 8)[None, :, None] - xy[g, 1][:, None, None]
        power = -0.5 * (A[:, None, None] * px * px + C[:, None, None] * py * py) - B[:, None, None] * px * py
        hit = (power <= 0) & (np.minimum(0.99, o[:, None, None] * np.exp(power)) >= 1.0 / 255.0)
        live = np.arange(walked)[:, None, None] < nc[None, :, :]
        tot["lanes"] += int((hit & live).sum())
        eq = exact(g, x0, y0, 7, 7)
        sub = np.stack([exact(g, x0 + 4 * (r & 1), y0 + 4 * (r >> 1), 3, 3) &
                        (np.arange(walked) < nc[4 * (r >> 1):4 * (r >> 1) + 4, 4 * (r & 1):4 * (r & 1) + 4].max()) for r in range(4)])
        tot["walked"] += walked
        tot["Eq"] += int(eq.sum())
        tot["Es"] += int(sub.sum())
        # per chunk of 64 list positions: rows advance together -> a chunk costs max over the rows of their survivors
        for c0 in range(0, walked, 64):
            tot["steps_new"] += int(sub[:, c0:c0 + 64].sum(axis=1).max())
        tot["steps_new_cum"] += int(sub.sum(axis=1).max())  # if a row could run ahead across chunk boundaries
print(f"P {P} s0 {s0}: {len(tiles)} tiles; positions walked {tot['walked']}")
print(f"(quadrant, entry) pairs E_q = {tot['Eq']}; (4x4 sub-block, entry) pairs E_s = {tot['Es']} = {tot['Es'] / tot['Eq']:.2f} per quadrant pair")
print(f"lanes that blend: {tot['lanes']} = {tot['lanes'] / tot['Eq']:.1f} of 64 per quadrant pair, {tot['lanes'] / tot['Es']:.1f} of 16 per sub-block pair")
print(f"wave entry-steps today: E_q = {tot['Eq']};  row decomposition, rows in step per chunk: {tot['steps_new']} "
      f"({tot['steps_new'] / tot['Eq']:.2f} of today's), rows free to run ahead: {tot['steps_new_cum']} ({tot['steps_new_cum'] / tot['Eq']:.2f}); "
      f"ideal E_s / 4 = {tot['Es'] / 4:.0f} ({tot['Es'] / 4 / tot['Eq']:.2f})")
