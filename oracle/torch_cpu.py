"""PyTorch-CPU restatement of the reference's forward render (TEST INFRASTRUCTURE / CPU BASELINE).

SURVEY.md section 8(d): the reference has no CPU render path (`render()` hard-codes "cuda",
gaussiansplatting/gaussian_renderer/__init__.py:62; `_C` is CUDA only), so "the reference's PyTorch-CPU render" is
this: the same maths as DGR/cuda_rasterizer/forward.cu in vectorised float32 torch ops on the host --

    preprocess   forward.cu:155-256  (frustum test, cov3D, cov2D, conic, radius, tile rectangle, SH -> RGB)
    binning      rasterizer_impl.cu:67-125  (u64 keys tile << 32 | depth bits, STABLE argsort, tile ranges)
    blending     forward.cu:261-379  per tile: alpha matrix (pixels x entries), transmittance by cumprod, the
                 1/255 skip and the T < 1e-4 stop expressed as masks

It is timed by bench.py as the second leg of `cpu_baseline` and checked against the C++ oracle in
tests/test_cpu_oracle.py.  It is NOT bit-exact with the oracle (torch's vectorised kernels use FMA and libm's exp);
images agree to ~1e-5, radii except at rounding ties.  Only tests/ and bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435)
TILE = 16


def _sh_to_rgb(D: int, sh: torch.Tensor, d: torch.Tensor) -> torch.Tensor:
    """forward.cu:20-71: sh (P,M,3), d (P,3) unit -> (P,3) before the clamp."""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH_C0 * sh[:, 0]
    if D > 0:
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
               + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
    if D > 2:
        res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
               + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11] + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
               + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
               + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res + 0.5


def preprocess(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
               scale_modifier=1.0, sh_degree=0, colors_precomp=None) -> Dict[str, torch.Tensor]:
    """forward.cu:155-256 for all Gaussians at once.  Matrices are the transposed 4x4 tensors the reference passes."""
    P = means3D.shape[0]
    V, Pm = viewmatrix, projmatrix
    ones = torch.ones(P, 1)
    ph = torch.cat([means3D, ones], 1)
    p_view = (ph @ V)[:, :3]                       # transformPoint4x3 (row vector times the transposed matrix)
    p_hom = ph @ Pm
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    proj = p_hom[:, :2] * p_w[:, None]
    front = p_view[:, 2] > 0.2                     # in_frustum, auxiliary.h:139-164
    # computeCov3D, forward.cu:118-152
    r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(P, 3, 3)
    Mm = R * (scale_modifier * scales)[:, None, :]  # R S
    Sigma = Mm @ Mm.transpose(1, 2)
    # computeCov2D, forward.cu:74-113
    tz = p_view[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(p_view[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(p_view[:, 1] / tz, -limy, limy) * tz
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    J = torch.zeros(P, 2, 3)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -(fx * tx) / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -(fy * ty) / (tz * tz)
    Wr = V[:3, :3].t()                             # world -> view rotation
    T = J @ Wr
    cov = T @ Sigma @ T.transpose(1, 2)
    cx, cy, cz = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = cx * cz - cy * cy
    det_inv = 1.0 / det
    conic = torch.stack([cz * det_inv, -cy * det_inv, cx * det_inv], 1)
    mid = 0.5 * (cx + cz)
    root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root)))
    pix = ((proj[:, 0].double() + 1.0) * W - 1.0) * 0.5
    piy = ((proj[:, 1].double() + 1.0) * H - 1.0) * 0.5
    mean2D = torch.stack([pix, piy], 1).float()
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ri = radius.to(torch.int32)
    # getRect, auxiliary.h:46-56 (float -> int conversions truncate)
    minx = torch.clamp(((mean2D[:, 0] - ri) / TILE).to(torch.int32), 0, gx)
    miny = torch.clamp(((mean2D[:, 1] - ri) / TILE).to(torch.int32), 0, gy)
    maxx = torch.clamp(((mean2D[:, 0] + ri + TILE - 1) / TILE).to(torch.int32), 0, gx)
    maxy = torch.clamp(((mean2D[:, 1] + ri + TILE - 1) / TILE).to(torch.int32), 0, gy)
    tiles = (maxx - minx) * (maxy - miny)
    ok = front & (det != 0) & (tiles > 0) & (ri > 0)
    if colors_precomp is not None:
        rgb = colors_precomp
    else:
        d = means3D - campos[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp(_sh_to_rgb(sh_degree, shs, d), min=0.0)
    return dict(ok=ok, mean2D=mean2D, depth=tz, conic=conic, opacity=opacities.reshape(-1), rgb=rgb,
                radii=torch.where(ok, ri, torch.zeros_like(ri)), rect=torch.stack([minx, miny, maxx, maxy], 1),
                tiles=torch.where(ok, tiles, torch.zeros_like(tiles)))


def render(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, bg, W: int, H: int, tanfovx: float,
           tanfovy: float, scale_modifier: float = 1.0, sh_degree: int = 0, colors_precomp: Optional[torch.Tensor] = None,
           tile_stride: int = 1, stats: Optional[dict] = None):
    """Forward render on the host: (color (3,H,W), depth (1,H,W), radii (P) int32, num_rendered).
    tile_stride > 1 (timing only): blend every tile_stride-th non-empty tile; `stats`, if given, receives the seconds
    spent before blending, in blending, and the tile counts."""
    import time as _time

    _t0 = _time.perf_counter()
    with torch.no_grad():
        g = preprocess(means3D, scales, rotations, opacities, shs, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                       scale_modifier, sh_degree, colors_precomp)
        gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
        vis = g["ok"].nonzero().view(-1)
        rect = g["rect"][vis].to(torch.int64)
        w = rect[:, 2] - rect[:, 0]
        n = g["tiles"][vis].to(torch.int64)
        R = int(n.sum())
        color = bg.view(3, 1, 1).expand(3, H, W).clone()
        depth = torch.zeros(1, H, W)
        if R == 0:
            return color, depth, g["radii"], 0
        # duplicateWithKeys, rasterizer_impl.cu:67-100: rows first inside the rectangle
        owner = torch.repeat_interleave(torch.arange(vis.numel()), n)
        start = torch.cumsum(n, 0) - n
        k = torch.arange(R) - start[owner]
        tile = (rect[owner, 1] + k // w[owner]) * gx + rect[owner, 0] + k % w[owner]
        dbits = g["depth"][vis].view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        keys = (tile << 32) | dbits[owner]
        order = torch.argsort(keys, stable=True)   # cub::DeviceRadixSort::SortPairs is stable
        tile_sorted = tile[order]
        point_list = vis[owner[order]]
        counts = torch.bincount(tile_sorted, minlength=gx * gy)
        ends = torch.cumsum(counts, 0)
        begins = ends - counts
        ys, xs = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
        nonempty = counts.nonzero().view(-1).tolist()
        _t1 = _time.perf_counter()
        if stats is not None:
            stats.update(prepare_s=_t1 - _t0, tiles_nonempty=len(nonempty), tiles_blended=len(nonempty[::tile_stride]))
        for t in nonempty[::tile_stride]:
            ids = point_list[begins[t]:ends[t]]
            tx, ty = t % gx, t // gx
            px = (tx * TILE + xs).reshape(-1)
            py = (ty * TILE + ys).reshape(-1)
            inside = (px < W) & (py < H)
            px, py = px[inside], py[inside]
            m = g["mean2D"][ids]
            dx = m[None, :, 0] - px[:, None].float()
            dy = m[None, :, 1] - py[:, None].float()
            c = g["conic"][ids]
            power = -0.5 * (c[None, :, 0] * dx * dx + c[None, :, 2] * dy * dy) - c[None, :, 1] * dx * dy
            alpha = torch.clamp(g["opacity"][ids][None, :] * torch.exp(power), max=0.99)
            hit = (power <= 0) & (alpha >= 1.0 / 255.0)
            a = torch.where(hit, alpha, torch.zeros_like(alpha))
            t_incl = torch.cumprod(1.0 - a, dim=1)                   # transmittance AFTER each entry
            t_excl = torch.cat([torch.ones(a.shape[0], 1), t_incl[:, :-1]], 1)
            stop = hit & (t_incl < 0.0001)                            # forward.cu:350-355: the entry that would cross is dropped
            alive = torch.cumsum(stop.to(torch.int32), 1) == 0
            wgt = a * t_excl * alive
            col = wgt @ g["rgb"][ids]
            dep = wgt @ g["depth"][ids]
            # final T = transmittance after the last blended entry
            final_T = torch.where(alive, 1.0 - a, torch.ones_like(a)).prod(dim=1)
            color[:, py, px] = (col + final_T[:, None] * bg[None, :]).t()
            depth[0, py, px] = dep
        if stats is not None:
            stats["blend_s"] = _time.perf_counter() - _t1
        return color, depth, g["radii"], R


def _bench_main(argv):
    """`python -m oracle.torch_cpu P W H s0 view nviews tile_stride threads` -> one JSON line (bench.py's second
    cpu_baseline leg, run as a subprocess so that its thread pool and a hard timeout are its own)."""
    import json
    import math
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gaussianeditor_amd.synth import ring_cameras, synth_scene

    P, W, H = int(argv[0]), int(argv[1]), int(argv[2])
    s0, view, nviews, stride, threads = float(argv[3]), int(argv[4]), int(argv[5]), int(argv[6]), int(argv[7])
    torch.set_num_threads(threads)
    sc = synth_scene(P, seed=0, s0=s0, sh_degree=3)
    cam = ring_cameras(nviews, W, H)[view]
    st = {}
    color, _, _, R = render(sc["xyz"], sc["scaling"], sc["rotation"], sc["opacity"], sc["features"], cam.world_view_transform,
                            cam.full_proj_transform, cam.camera_center, sc["bg"], W, H, math.tan(cam.FoVx / 2),
                            math.tan(cam.FoVy / 2), 1.0, 3, tile_stride=stride, stats=st)
    full = st["prepare_s"] + st["blend_s"] * st["tiles_nonempty"] / max(st["tiles_blended"], 1)
    print(json.dumps({"seconds_per_render": full, "prepare_s": st["prepare_s"], "blend_s_sampled": st["blend_s"],
                      "tiles_nonempty": st["tiles_nonempty"], "tiles_blended": st["tiles_blended"], "num_rendered": R,
                      "threads": torch.get_num_threads()}))


if __name__ == "__main__":
    import sys as _sys

    _bench_main(_sys.argv[1:])
