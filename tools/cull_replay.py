"""CPU replay of the blend kernels' conservative cull test (gsr_blend.hip: can_touch_quad; the same box bounds the tile
rectangle of GSR_FLAG_TILE_BOUNDS_ALPHA in K1) against the reference's own per-pixel evaluation, in binary32.

For random screen-space Gaussians -- including needles (major sigma up to thousands of pixels, aspect up to several
thousand, any orientation) -- the script builds the 2D covariance, conic and cull box exactly as the kernels do, then
evaluates power / alpha (forward.cu:335-344, with the kernels' fma placement) at pixels OUTSIDE the box and reports
every pixel that the reference would blend (power <= 0 and alpha >= 1/255).  A conservative rule reports none.

    python tools/cull_replay.py [n_gaussians] [--old]     (--old: the round-1 rule, to see the failures it had)
"""
import sys

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def cull_box(conx, cony, conz, o, old=False):
    """Half extents (hx, hy) of the cull box, or None when the rule refuses to cull.  binary32 throughout."""
    if not (o >= f32(1.0 / 255.0)):
        return (f32(-1), f32(-1))  # culled everywhere
    xz = f32(conx * conz)
    det = f32(xz - f32(cony * cony))
    if old:
        if not (det > 0):
            return None
        margin = f32(1.0001)
    else:
        if not (det >= f32(f32(1e-3) * xz)) or not (det > 0):
            return None
        margin = f32(1.001)
    tau2 = f32(f32(2.0) * f32(f32(np.log(f32(255.0) * o).astype(f32) * f32(1.001)) + f32(0.01)))
    inv = f32(f32(f32(1.0) / det) * margin)
    hx = f32(np.sqrt(f32(f32(tau2 * conz) * inv)) + f32(0.01))
    hy = f32(np.sqrt(f32(f32(tau2 * conx) * inv)) + f32(0.01))
    if not (hx == hx) or not (hy == hy):
        return None
    return hx, hy


def replay(n=20000, old=False, seed=0):
    """-> (Gaussians the rule refused to cull, Gaussians with a blended pixel outside their box, largest alpha there)."""
    rng = np.random.default_rng(seed)
    fails = refused = 0
    worst = 0.0
    for _ in range(n):
        s1 = 10 ** rng.uniform(0, 3.5)                     # major sigma, pixels
        aspect = 10 ** rng.uniform(0, 3.7)
        s2 = max(s1 / aspect, 0.0)
        th = rng.uniform(0, np.pi)
        if rng.random() < 0.3:
            th = np.pi / 4 + rng.normal(0, 0.02)           # the worst case: 45 degrees
        c, s = np.cos(th), np.sin(th)
        # covariance as K1 leaves it (low-pass +0.3 on the diagonal), then forward.cu:219-224 in binary32
        cx = f32(c * c * s1 * s1 + s * s * s2 * s2 + 0.3)
        cy = f32(c * s * (s1 * s1 - s2 * s2))
        cz = f32(s * s * s1 * s1 + c * c * s2 * s2 + 0.3)
        det = f32(f32(cx * cz) - f32(cy * cy))
        if det == 0:
            continue
        det_inv = f32(f32(1.0) / det)
        conx, cony, conz = f32(cz * det_inv), f32(-cy * det_inv), f32(cx * det_inv)
        o = f32(rng.uniform(1 / 255, 1.0))
        box = cull_box(conx, cony, conz, o, old)
        if box is None:
            refused += 1
            continue
        hx, hy = box
        # pixels along the major axis (where the level set reaches furthest) and a band around it
        t = np.linspace(-6 * s1, 6 * s1, 4001)
        w = np.linspace(-4 * max(s2, 0.6), 4 * max(s2, 0.6), 9)
        tt, ww = np.meshgrid(t, w)
        dx = np.rint(tt * c - ww * s).astype(f32)          # integer pixel offsets from the mean (mean at a pixel centre)
        dy = np.rint(tt * s + ww * c).astype(f32)
        outside = (np.abs(dx) > hx) | (np.abs(dy) > hy)
        if not outside.any():
            continue
        dx, dy = dx[outside], dy[outside]
        # power = fma(-(B dx), dy, -0.5 * fma(C dy, dy, (A dx) dx))  (gsr_common.h: blend_power)
        a_ = f32(conx * dx) * dx
        s_ = fma(f32(conz * dy), dy, a_.astype(f32))
        power = fma(-f32(cony * dx), dy, f32(-0.5) * s_)
        alpha = np.minimum(f32(0.99), o * np.exp(power.astype(np.float64)).astype(f32))
        bad = (power <= 0) & (alpha >= f32(1.0 / 255.0))
        if bad.any():
            fails += 1
            worst = max(worst, float(alpha[bad].max()))
    return refused, fails, worst


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
    old = "--old" in sys.argv
    refused, fails, worst = replay(n, old)
    rule = "round-1 rule" if old else "current rule"
    print(f"{rule}: {n} Gaussians, {refused} not culled by rule (ill-conditioned), {fails} with a blended pixel outside the box"
          f" (largest alpha there: {worst:.3f})")
    return 1 if (fails and not old) else 0


if __name__ == "__main__":
    sys.exit(main())
