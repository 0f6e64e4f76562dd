"""CPU replay of the conservative cull tests -- the box that bounds the tile rectangle of GSR_FLAG_TILE_BOUNDS_ALPHA in K1
(`cull_box` / `replay`) and the blend kernels' ellipse-against-rectangle test (gsr_blend.hip: can_touch_quad; `cull_rect` /
`replay_rect`) -- against the reference's own per-pixel evaluation, in binary32.

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


def med3(a, b, c):
    return f32(max(min(a, b), min(max(a, b), c)))


def cull_rect(conx, cony, conz, o, mx, my, qx0, qy0, qw, qh):
    """can_touch_quad (gsr_blend.hip) in binary32: False = the rule culls the entry for this rectangle."""
    if not (o >= f32(1.0 / 255.0)):
        return not (o == o)
    A, B, C = conx, cony, conz
    xz = f32(A * C)
    det = f32(xz - f32(B * B))
    if not (det >= f32(f32(1e-3) * xz)) or not (det > 0) or not (A > 0):
        return True
    tau2 = f32(f32(f32(2.0) * f32(f32(np.log(f32(f32(255.0) * o)).astype(f32) * f32(1.001)) + f32(0.01))) * f32(1.001))
    xl, xh = f32(f32(qx0 - f32(0.01)) - mx), f32(f32(f32(qx0 + qw) + f32(0.01)) - mx)
    yl, yh = f32(f32(qy0 - f32(0.01)) - my), f32(f32(f32(qy0 + qh) + f32(0.01)) - my)
    cx, cy = med3(f32(0), xl, xh), med3(f32(0), yl, yh)
    x1 = med3(f32(f32(-f32(B * cy)) * f32(f32(1.0) / A)), xl, xh)
    y2 = med3(f32(f32(-f32(B * cx)) * f32(f32(1.0) / C)), yl, yh)
    q1 = f32(f32(f32(f32(A * x1) * x1) + f32(f32(f32(f32(2.0) * B) * x1) * cy)) + f32(f32(C * cy) * cy))
    q2 = f32(f32(f32(f32(A * cx) * cx) + f32(f32(f32(f32(2.0) * B) * cx) * y2)) + f32(f32(C * y2) * y2))
    return (not (q1 > tau2)) or (not (q2 > tau2))


def replay_rect(n=4000, seed=0, rects_per=40):
    """Random Gaussians (needles included) x random 8x8 / 8x4 pixel rectangles placed around the level set: -> (pairs the
    rule culled, culled pairs in which a pixel of the rectangle would have been blended, largest alpha there, pairs kept)."""
    rng = np.random.default_rng(seed)
    culled = fails = kept = 0
    worst = 0.0
    for _ in range(n):
        s1 = 10 ** rng.uniform(0, 3.0)
        aspect = 10 ** rng.uniform(0, 3.2)
        s2 = s1 / aspect
        th = rng.uniform(0, np.pi)
        if rng.random() < 0.3:
            th = np.pi / 4 + rng.normal(0, 0.02)
        c, s = np.cos(th), np.sin(th)
        cx_ = f32(c * c * s1 * s1 + s * s * s2 * s2 + 0.3)
        cy_ = f32(c * s * (s1 * s1 - s2 * s2))
        cz_ = f32(s * s * s1 * s1 + c * c * s2 * s2 + 0.3)
        det = f32(f32(cx_ * cz_) - f32(cy_ * cy_))
        if det == 0:
            continue
        det_inv = f32(f32(1.0) / det)
        conx, cony, conz = f32(cz_ * det_inv), f32(-cy_ * det_inv), f32(cx_ * det_inv)
        o = f32(rng.uniform(1 / 255, 1.0))
        mx, my = f32(rng.uniform(0, 1920)), f32(rng.uniform(0, 1080))
        lam = np.sqrt(max(2.0 * np.log(255.0 * float(o)), 0.0))  # the level set reaches lam * sigma along each axis
        for _ in range(rects_per):
            # a point near the boundary of the level set, then a rectangle that has it near one of its corners / sides
            phi = rng.uniform(0, 2 * np.pi)
            rr = rng.uniform(0.8, 1.25)
            u, v = rr * lam * np.sqrt(s1 * s1 + 0.3) * np.cos(phi), rr * lam * np.sqrt(s2 * s2 + 0.3) * np.sin(phi)
            bx, by = float(mx) + u * c - v * s, float(my) + u * s + v * c
            qw, qh = f32(7), f32(7 if rng.random() < 0.7 else 3)
            qx0 = f32(np.floor(bx) - rng.integers(0, int(qw) + 1))
            qy0 = f32(np.floor(by) - rng.integers(0, int(qh) + 1))
            touch = cull_rect(conx, cony, conz, o, mx, my, qx0, qy0, qw, qh)
            if touch:
                kept += 1
                continue
            culled += 1
            px = (qx0 + np.arange(int(qw) + 1, dtype=f32))[None, :].repeat(int(qh) + 1, 0)
            py = (qy0 + np.arange(int(qh) + 1, dtype=f32))[:, None].repeat(int(qw) + 1, 1)
            dx, dy = (mx - px).astype(f32), (my - py).astype(f32)  # forward.cu:333: d = xy - pixf
            a_ = f32(conx * dx) * dx
            s_ = fma(f32(conz * dy), dy, a_.astype(f32))
            power = fma(-f32(cony * dx), dy, f32(-0.5) * s_)
            alpha = np.minimum(f32(0.99), o * np.exp(power.astype(np.float64)).astype(f32))
            bad = (power <= 0) & (alpha >= f32(1.0 / 255.0))
            if bad.any():
                fails += 1
                worst = max(worst, float(alpha[bad].max()))
    return culled, fails, worst, kept


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20000
    old = "--old" in sys.argv
    refused, fails, worst = replay(n, old)
    if "--rect" in sys.argv:
        culled, fails, worst, kept = replay_rect(n)
        print(f"ellipse / rectangle rule: {n} Gaussians, {culled} (Gaussian, rectangle) pairs culled, {kept} kept, {fails} culled with a "
              f"blended pixel inside (largest alpha there: {worst:.3f})")
        return 1 if fails else 0
    rule = "round-1 rule" if old else "current rule"
    print(f"{rule}: {n} Gaussians, {refused} not culled by rule (ill-conditioned), {fails} with a blended pixel outside the box"
          f" (largest alpha there: {worst:.3f})")
    return 1 if (fails and not old) else 0


if __name__ == "__main__":
    sys.exit(main())
