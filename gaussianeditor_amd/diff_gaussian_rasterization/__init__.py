"""Drop-in replacement for the reference's `diff_gaussian_rasterization` Python package
(DGR/diff_gaussian_rasterization/__init__.py, 364 lines): same public names, same call
signatures, same return arities, same exceptions -- backed by the gfx950 HIP library
instead of the CUDA extension.

Public surface (reference line numbers):
    GaussianRasterizationSettings   :228-240   NamedTuple of 12 fields
    GaussianRasterizer              :243-364   nn.Module with forward / markVisible / apply_weights
    rasterize_gaussians             :26-47
    _RasterizeGaussians             :50-225    autograd.Function

To let unmodified GaussianEditor code `import diff_gaussian_rasterization`, call
`gaussianeditor_amd.install()` once.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C, _reuse
from .. import options as _options

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "_RasterizeGaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _snapshot(args):
    """CPU deep copy of a native call's arguments (reference: cpu_deep_copy_tuple, :18-23)."""
    return tuple(a.cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def _call_native(fn, args, debug: bool, dump_path: str, message: str):
    """Invoke a `_C` entry point; in debug mode dump the inputs on failure, as the
    reference does (:88-107 forward, :180-200 backward)."""
    if not debug:
        return fn(*args)
    saved = _snapshot(args)  # before anything can corrupt them
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print(message)
        raise


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, aux_colors=None):
        rs = raster_settings
        # behaviour flags (include/gsr.h: GSR_FLAG_*) are fixed per render: read once here, reused by the backward
        flags = _options.current_flags()
        # a render no input of which requires a gradient (the viewer, torch.no_grad()) will never see a backward: the
        # preprocessing then leaves out what only the backward reads (GSR_FLAG_FORWARD_ONLY)
        fwd_flags = flags if any(ctx.needs_input_grad) else flags | _options.FLAG_FORWARD_ONLY
        run = lambda fn: (lambda *a: fn(*a, flags=fwd_flags))  # noqa: E731
        # argument order of _C.rasterize_gaussians (rasterize_points.h:17-36)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = _call_native(
            run(_C.rasterize_gaussians), args, rs.debug, "snapshot_fw.dump",
            "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.raster_settings = rs
        ctx.gsr_flags = flags
        ctx.num_rendered = num_rendered
        # view reuse (_reuse.py): a following colour-override render of this view runs the blend kernel on this state
        _reuse.remember(rs, flags, means3D, scales, rotations, opacities, cov3Ds_precomp, num_rendered, geomBuffer,
                        binningBuffer, imgBuffer, radii, depth)
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        # radii / depth never carry a gradient: do not let autograd fill zero tensors for them on every backward
        ctx.set_materialize_grads(False)
        if aux_colors is None:
            return color, radii, depth
        # extension: a second, gradient-free image of the same view with other colours (K6 only)
        aux = _call_native(run(_C.rasterize_gaussians_aux),
                           (rs.bg, aux_colors.detach(), num_rendered, geomBuffer, binningBuffer, imgBuffer, rs.image_height,
                            rs.image_width, rs.debug), rs.debug, "snapshot_fw.dump",
                           "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        ctx.mark_non_differentiable(radii, aux)
        return color, radii, depth, aux

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, grad_aux=None):
        # grad_radii / grad_depth are ignored exactly as in the reference (:137, :155-177):
        # depth is a forward-only output.
        rs = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
         imgBuffer) = ctx.saved_tensors
        if grad_out_color is None:  # only the (gradient-free) depth output was used downstream
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        # argument order of _C.rasterize_gaussians_backward (rasterize_points.h:38-60)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                geomBuffer, ctx.num_rendered, binningBuffer, imgBuffer, rs.debug)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _call_native(
             lambda *a: _C.rasterize_gaussians_backward(*a, flags=ctx.gsr_flags,
                                                        grad_allocator=getattr(ctx, "gsr_grad_allocator", None)),
             args, rs.debug, "snapshot_bw.dump",
             "\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
        # one slot per forward() input (:213-225)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None, None)


class _ReusedRender(torch.autograd.Function):
    """A colour-override render served from the state of the preceding full render of the same view (_reuse.py): the
    blend kernel alone.  Same outputs as `_RasterizeGaussians`; differentiable like it -- a backward through this image
    (nobody in GaussianEditor asks for one) first runs the full forward that was skipped, then the ordinary backward."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                entry):
        rs = raster_settings
        flags = _options.current_flags()
        color = _call_native(lambda *a: _C.rasterize_gaussians_aux(*a, flags=flags),
                             (rs.bg, colors_precomp.detach(), entry.R, entry.geom, entry.binning, entry.img, rs.image_height,
                              rs.image_width, rs.debug), rs.debug, "snapshot_fw.dump",
                             "\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
        radii, depth = entry.radii.clone(), entry.depth.clone()
        ctx.raster_settings = rs
        ctx.gsr_flags = flags
        ctx.save_for_backward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, depth

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth):
        rs, flags = ctx.raster_settings, ctx.gsr_flags
        means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
        fwd = _C.rasterize_gaussians(rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier,
                                     cov3Ds_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height,
                                     rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, flags=flags)
        num_rendered, _, _, radii, geomBuffer, binningBuffer, imgBuffer = fwd
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = _C.rasterize_gaussians_backward(
             rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
             rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos, geomBuffer, num_rendered,
             binningBuffer, imgBuffer, rs.debug, flags=flags)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    if colors_precomp.numel() != 0 and sh.numel() == 0 and means3D.is_cuda:
        # a colour-override render: is it the view the rasterizer rendered last (the reference's second render() of every
        # training view / GUI frame)?  Then the blend kernel alone, on that render's state (_reuse.py)
        entry = _reuse.lookup(raster_settings, _options.current_flags(), means3D, scales, rotations, opacities, cov3Ds_precomp)
        if entry is not None:
            return _ReusedRender.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                       raster_settings, entry)
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


def rasterize_gaussians_with_aux(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                 raster_settings, aux_colors):
    """rasterize_gaussians plus a second image blended with `aux_colors` (P,3): (color, radii, depth, aux_color)."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, aux_colors)


def _absent(like: torch.Tensor) -> torch.Tensor:
    """The reference encodes "not provided" as an empty float32 tensor on "cuda" (:285-295);
    we put it on the device of `means3D`, which is the same thing for every valid call."""
    return torch.empty(0, dtype=torch.float32, device=like.device)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean mask of the points in front of the camera's near plane (:248-256)."""
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, aux_colors=None):
        """Reference signature (:258-309) plus one extension: with `aux_colors` (P,3) a second image of the same
        view, blended with those colours instead, is returned as a fourth value (forward only, no gradient).  It is
        what a second call with colors_precomp=aux_colors would render, at the cost of the blend kernel alone."""
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")  # sic, :271-276
        has_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (has_sr and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")  # :278-283
        shs = _absent(means3D) if shs is None else shs
        colors_precomp = _absent(means3D) if colors_precomp is None else colors_precomp
        scales = _absent(means3D) if scales is None else scales
        rotations = _absent(means3D) if rotations is None else rotations
        cov3D_precomp = _absent(means3D) if cov3D_precomp is None else cov3D_precomp
        if aux_colors is not None:
            return rasterize_gaussians_with_aux(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                                cov3D_precomp, self.raster_settings, aux_colors)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)

    def apply_weights(self, means3D, means2D, opacities, shs=None, weights=None, scales=None, rotations=None,
                      cov3Ds_precomp=None, cnt=None, image_weights=None):
        """Semantic tracing (:311-364): every pixel adds its `image_weights` value(s) to
        `weights[i]` and C to `cnt[i]` for every Gaussian i it would blend.  In place; returns None."""
        assert weights is not None
        assert cnt is not None
        assert image_weights is not None
        rs = self.raster_settings
        shs = _absent(means3D) if shs is None else shs
        scales = _absent(means3D) if scales is None else scales
        rotations = _absent(means3D) if rotations is None else rotations
        cov3Ds_precomp = _absent(means3D) if cov3Ds_precomp is None else cov3Ds_precomp
        # argument order of _C.apply_weights (rasterize_points.h:66-77)
        _C.apply_weights(rs.bg, means3D, weights, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                         rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, shs,
                         rs.sh_degree, rs.campos, rs.prefiltered, image_weights, cnt, rs.debug)
