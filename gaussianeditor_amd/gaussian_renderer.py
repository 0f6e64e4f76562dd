"""Mirror of the reference's render boundary, gaussiansplatting/gaussian_renderer/__init__.py
(`render` :45-150, `camera2rasterizer` :21-42): same arguments, same returned dict.

`pc` is duck-typed exactly like the reference uses its GaussianModel: `get_xyz`, `get_opacity`,
`get_scaling`, `get_rotation`, `get_features`, `active_sh_degree`, `max_sh_degree`,
`get_covariance(scaling_modifier)`.  `viewpoint_camera` needs `FoVx, FoVy, image_height,
image_width, world_view_transform, full_proj_transform, camera_center`; `pipe` needs
`compute_cov3D_python`, `convert_SHs_python`.
"""
from __future__ import annotations

import math

import torch

from .diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_SH_C0 = 0.28209479177387814
_SH_C1 = 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH -> RGB in PyTorch for `pipe.convert_SHs_python` (reference: utils/sh_utils.py:57-112,
    degrees 0..3).  sh: (..., C, (deg+1)^2), dirs: (..., 3) unit -> (..., C)."""
    assert 0 <= deg <= 3 and sh.shape[-1] >= (deg + 1) ** 2
    res = _SH_C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _SH_C2[0] * xy * sh[..., 4] + _SH_C2[1] * yz * sh[..., 5]
                   + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] + _SH_C2[3] * xz * sh[..., 7]
                   + _SH_C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + _SH_C3[1] * xy * z * sh[..., 10]
                       + _SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11]
                       + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _SH_C3[5] * z * (xx - yy) * sh[..., 14]
                       + _SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def _settings(cam, bg_color, scale_modifier, sh_degree) -> GaussianRasterizationSettings:
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height),
        image_width=int(cam.image_width),
        tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform,
        sh_degree=sh_degree,
        campos=cam.camera_center,
        prefiltered=False,
        debug=False,
    )


def camera2rasterizer(viewpoint_camera, bg_color: torch.Tensor, sh_degree: int = 0) -> GaussianRasterizer:
    """gaussian_renderer/__init__.py:21-42 (used by GaussianModel.apply_weights, scene/gaussian_model.py:817-832)."""
    return GaussianRasterizer(raster_settings=_settings(viewpoint_camera, bg_color, 1.0, sh_degree))


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None,
           semantic_color=None):
    """gaussian_renderer/__init__.py:45-150.  Background tensor must be on the GPU.

    Extension: `semantic_color` (P,3) adds out["semantic"], the image the reference obtains from a SECOND
    render(..., override_color=semantic_color) of the same camera (threestudio/systems/GassuianEditor.py:183-191,
    webui.py:705-713); here it reuses the preprocessing, sort and tile ranges of this call."""
    xyz = pc.get_xyz
    # dummy (P,3) tensor whose .grad receives the screen-space mean gradient (:60-69).  The reference builds it as
    # `zeros_like(..., requires_grad=True) + 0` and retains the gradient of that non-leaf: an add kernel per render and a
    # clone of the gradient per backward (12 MB each at 10^6 Gaussians, 2 % of an iteration).  A leaf holds the same zeros
    # and receives the same .grad without either.
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device)

    rasterizer = GaussianRasterizer(
        raster_settings=_settings(viewpoint_camera, bg_color, scaling_modifier, pc.active_sh_degree))

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features.float()
    else:
        colors_precomp = override_color

    outs = rasterizer(
        means3D=xyz.float(),
        means2D=screenspace_points.float(),
        shs=shs,
        colors_precomp=colors_precomp,
        opacities=pc.get_opacity.float(),
        scales=None if scales is None else scales.float(),
        rotations=None if rotations is None else rotations.float(),
        cov3D_precomp=cov3D_precomp,
        **({} if semantic_color is None else {"aux_colors": semantic_color.float()}),
    )
    rendered_image, radii, depth = outs[:3]
    out = {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "depth_3dgs": depth,
    }
    if semantic_color is not None:
        out["semantic"] = outs[3]
    return out


def point_cloud_render(viewpoint_camera, xyz, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """gaussian_renderer/__init__.py:156-250 (the web UI's point-cloud view, webui.py:39): every point as a white,
    fully opaque, isotropic Gaussian of scale 0.005 with identity rotation; `pipe` and `override_color` are accepted and
    unused, as in the reference.  Same returned dict as `render` (no `semantic` entry)."""
    del pipe, override_color
    means2D = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        means2D.retain_grad()
    except RuntimeError:
        pass
    rasterizer = GaussianRasterizer(raster_settings=_settings(viewpoint_camera, bg_color, scaling_modifier, 0))
    n = xyz.shape[0]
    rotations = torch.zeros((n, 4), dtype=xyz.dtype, device=xyz.device)
    rotations[:, 0] = 1.0
    image, radii, depth = rasterizer(
        means3D=xyz.float(),
        means2D=means2D.float(),
        shs=None,
        colors_precomp=torch.ones((n, 3), dtype=xyz.dtype, device=xyz.device),
        opacities=torch.ones((n, 1), dtype=torch.float32, device=xyz.device),
        scales=torch.full((n, 3), 0.005, dtype=torch.float32, device=xyz.device),
        rotations=rotations.float(),
        cov3D_precomp=None,
    )
    return {"render": image, "viewspace_points": means2D, "visibility_filter": radii > 0, "radii": radii, "depth_3dgs": depth}
