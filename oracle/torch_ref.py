"""Float64 *differentiable* restatement of the forward pass (TEST INFRASTRUCTURE).

Purpose: validate the analytic backward of oracle/gsr_oracle.cpp (and hence of
the HIP kernels) against torch.autograd without trusting any hand-derived
gradient.  The continuous maths is re-derived here in float64 from the formulas
in SURVEY.md Appendix A.2-A.5 (DGR/cuda_rasterizer/forward.cu:74-152, 182-255,
261-379); the *discrete* structure -- which Gaussian instance belongs to which
tile, in which order -- is taken from the float32 oracle (`point_list`,
`ranges`), because it is not differentiable anyway.

Conventions that make autograd reproduce the reference's analytic backward
(DGR/cuda_rasterizer/backward.cu) rather than the "true" derivative:
  * alpha = min(0.99, o*G) is differentiated as o*G (straight-through clamp;
    backward.cu:498-503 ignores the clamp);
  * the depth image carries no gradient (DGR/diff_gaussian_rasterization/__init__.py:137
    drops grad_depth);
  * `means2D` is an explicit zero offset added to the projected mean in
    "NDC-scaled pixel units" so that its gradient equals dL_dmean2D
    (backward.cu:545-546: the 0.5*W, 0.5*H factors).
Gaussians whose view-space x/z or y/z is clamped to 1.3*tanfov (forward.cu:82-87)
are differentiated by the reference with the clamped value treated as a constant
in dL/dtz; test scenes keep all Gaussians inside that cone.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def _sh_to_rgb(D: int, shs: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """forward.cu:20-71 (before the +0.5 / clamp). shs (P,M,3), dirs (P,3) unit."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    res = SH_C0 * shs[:, 0]
    if D > 0:
        res = res - SH_C1 * y * shs[:, 1] + SH_C1 * z * shs[:, 2] - SH_C1 * x * shs[:, 3]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * shs[:, 4] + SH_C2[1] * yz * shs[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
               + SH_C2[3] * xz * shs[:, 7] + SH_C2[4] * (xx - yy) * shs[:, 8])
    if D > 2:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * shs[:, 9] + SH_C3[1] * xy * z * shs[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * shs[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * shs[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * shs[:, 13] + SH_C3[5] * z * (xx - yy) * shs[:, 14]
               + SH_C3[6] * x * (xx - 3.0 * yy) * shs[:, 15])
    return res


def render_f64(struct: Dict[str, np.ndarray], means3D, means2D, opacities, scales, rotations, shs, colors_precomp,
               cov3D_precomp, viewmatrix, projmatrix, campos, bg, W: int, H: int, tanfovx: float, tanfovy: float,
               scale_modifier: float = 1.0, sh_degree: int = 0, dL_dimage=None) -> torch.Tensor:
    """Differentiable colour image (3,H,W) in float64.

    `dL_dimage` (3,H,W): the gradient of a loss by the image.  The backward then runs INSIDE this call, tile by tile (a
    tile's graph -- 256 pixels x list length x a dozen float64 tensors -- is freed before the next tile is built: scenes
    with hundreds of thousands of instances fit in memory), the inputs' `.grad` are filled and the returned image is
    detached.

    struct: dict from oracle.cpu.forward() supplying `point_list`, `ranges`, `radii`.
    Tensor arguments are float64 torch tensors (requires_grad as desired); pass
    None for absent optionals exactly like the L1 API.
    """
    dt = torch.float64
    V = viewmatrix.to(dt)  # flat[4c+r] = W2C[r][c]  ->  V[c, r]
    PV = projmatrix.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], dim=1)
    p_view = ph @ V  # row-vector convention: (x,y,z,1) @ W2C^T
    p_hom = ph @ PV
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]

    if cov3D_precomp is None:
        s = scale_modifier * scales
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        Rm = torch.stack([
            1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
            2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
            2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(P, 3, 3)
        Mm = Rm * s[:, None, :]  # R diag(s)
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]],
                            dim=1).reshape(P, 3, 3)

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], dim=1).reshape(P, 2, 3)
    Rv = V[:3, :3].transpose(0, 1)  # W2C rotation
    T = J @ Rv
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c2 = cov2[:, 1, 1] + 0.3
    det = a * c2 - b * b
    conic = torch.stack([c2 / det, -b / det, a / det], dim=1)

    px = ((p_proj[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((p_proj[:, 1] + 1.0) * H - 1.0) * 0.5
    if means2D is not None:
        # d(pixel)/d(means2D) = (0.5 W, 0.5 H): backward.cu:458-461, 545-546
        px = px + 0.5 * W * means2D[:, 0]
        py = py + 0.5 * H * means2D[:, 1]

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(_sh_to_rgb(sh_degree, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    op = opacities.reshape(-1)
    per_gaussian = None
    if dL_dimage is not None:
        # per-Gaussian quantities become leaves of the per-tile graphs; their gradients are pushed to the inputs at the end
        per_gaussian = [px, py, conic, rgb, op]
        px, py, conic, rgb, op = [t.detach().requires_grad_(True) for t in per_gaussian]
        leaves = [px, py, conic, rgb, op]
        dL = dL_dimage.to(dt)
    point_list = torch.from_numpy(struct["point_list"].astype(np.int64))
    ranges = struct["ranges"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    out = torch.zeros(3, H, W, dtype=dt)
    bgd = bg.to(dt)
    for tile in range(gx * gy):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
        xs = torch.arange(tx0, min(tx0 + 16, W), dtype=dt)
        ys = torch.arange(ty0, min(ty0 + 16, H), dtype=dt)
        pyy, pxx = torch.meshgrid(ys, xs, indexing="ij")
        pxx, pyy = pxx.reshape(-1), pyy.reshape(-1)
        npx = pxx.numel()
        if r1 <= r0:
            col = bgd[:, None].expand(3, npx)
        else:
            ids = point_list[r0:r1]
            dx = px[ids][None, :] - pxx[:, None]
            dy = py[ids][None, :] - pyy[:, None]
            cn = conic[ids]
            power = -0.5 * (cn[None, :, 0] * dx * dx + cn[None, :, 2] * dy * dy) - cn[None, :, 1] * dx * dy
            G = torch.exp(power)
            raw = op[ids][None, :] * G
            alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()
            valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
            alpha = torch.where(valid, alpha, torch.zeros_like(alpha))
            one_m = 1.0 - alpha
            Tincl = torch.cumprod(one_m, dim=1)  # T after instance j
            Texcl = torch.cat([torch.ones(npx, 1, dtype=dt), Tincl[:, :-1]], dim=1)
            # termination: first valid instance whose test_T < 1e-4 stops the pixel (not blended)
            stop = valid & (Tincl.detach() < 0.0001)
            stopped = torch.cumsum(stop.to(torch.int64), dim=1) > 0
            live = valid & ~stopped
            w = torch.where(live, alpha * Texcl, torch.zeros_like(alpha))
            Tfinal = torch.where(live, one_m, torch.ones_like(one_m)).prod(dim=1)
            col = (w @ rgb[ids]).transpose(0, 1) + Tfinal[None, :] * bgd[:, None]
        hh, ww = ys.numel(), xs.numel()
        if per_gaussian is not None:
            if col.requires_grad:
                (col.reshape(3, hh, ww) * dL[:, ty0:ty0 + hh, tx0:tx0 + ww]).sum().backward()
            col = col.detach()
        out[:, ty0:ty0 + hh, tx0:tx0 + ww] = col.reshape(3, hh, ww)
    if per_gaussian is not None:
        pairs = [(t, l.grad) for t, l in zip(per_gaussian, leaves) if l.grad is not None and t.requires_grad]
        if pairs:
            torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
    return out
