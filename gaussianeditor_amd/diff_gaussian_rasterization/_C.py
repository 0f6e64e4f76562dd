"""`_C`: the four entry points the reference binds with pybind11
(diff-gaussian-rasterization/ext.cpp:15-20), re-implemented as thin torch glue over the
C ABI of libgsr_hip.so.  Argument order, return arity and error behaviour follow
rasterize_points.cu:35-234; torch owns every allocation (the native library never allocates).

There is no CPU path: tensors must live on a ROCm device ("cuda"), and the import
fails if the HIP library is not built.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import _native, options

_native.lib()  # fail loudly at import time if the extension is missing

NUM_CHANNELS = 3  # config.h:15

#: Allocator for the backward's gradient outputs: fn(name, shape, zero) -> tensor | None, handed to
#: rasterize_gaussians_backward PER CALL (`grad_allocator=`).  gaussianeditor_amd.multiview uses it to make the gradients
#: views of one flat all-reduce bucket.  It travels on the autograd node of the render it belongs to
#: (`attach_grad_allocator`), not in module or thread state: the engine runs the backward of CUDA tensors on its own
#: device thread, and the web UI renders from a second Python thread while a training thread steps.


def attach_grad_allocator(output: torch.Tensor, fn) -> None:
    """Make the backward of the render that produced `output` (the colour image of GaussianRasterizer / render())
    allocate its gradients through `fn` (None removes it)."""
    node = output.grad_fn
    if node is None or not hasattr(node, "raster_settings"):
        raise RuntimeError("attach_grad_allocator: the tensor is not the output of a GaussianRasterizer render")
    node.gsr_grad_allocator = fn


def _flags(flags) -> int:
    return options.current_flags() if flags is None else int(flags)


def _require_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"diff_gaussian_rasterization: `{name}` is on {t.device}; this build only runs on a ROCm GPU "
            "(device 'cuda') through the HIP extension -- there is no CPU fallback.")


def _f32(t: torch.Tensor, name: str, dev: Optional[torch.device] = None) -> torch.Tensor:
    """contiguous float32, 16-byte aligned (the kernels use dwordx4 loads on (P,4)/(P,16,3) rows), and -- with `dev`, the
    device of `means3D` -- on that device: the native call takes raw pointers, so a tensor on the host or on another GPU
    would be a fault inside a kernel instead of an error here (the reference has that hole, rasterize_points.cu:46-48
    checks `means3D` only).  Empty tensors stand for "feature absent" and carry no pointer: they may live anywhere."""
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"diff_gaussian_rasterization: `{name}` must be a tensor, got {type(t).__name__}")
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 for `{name}`, got {t.dtype}")
    if dev is not None and t.numel() != 0 and t.device != dev:
        raise RuntimeError(
            f"diff_gaussian_rasterization: `{name}` is on {t.device} but `means3D` is on {dev}; every tensor of a call "
            "must live on the device of `means3D` (the native library receives raw device pointers)")
    t = t.contiguous()
    if t.data_ptr() % 16 != 0:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def _on_device(t: torch.Tensor, name: str, dev: torch.device, dtype: torch.dtype) -> None:
    """device + dtype check for the non-float arguments (radii, the three saved state buffers, cnt)."""
    if not isinstance(t, torch.Tensor) or t.dtype != dtype:
        raise RuntimeError(f"diff_gaussian_rasterization: `{name}` must be a {dtype} tensor")
    if t.numel() != 0 and t.device != dev:
        raise RuntimeError(
            f"diff_gaussian_rasterization: `{name}` is on {t.device} but `means3D` is on {dev}; every tensor of a call "
            "must live on the device of `means3D` (the native library receives raw device pointers)")


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """empty tensor <=> NULL <=> feature absent (rasterize_points.cu:80-91 / forward.cu:205,241)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device: torch.device) -> int:
    """The current stream of `device` as a raw hipStream_t (torch.cuda.current_stream() builds a Stream object per call: 5 us
    each, a dozen times per view, on a launch thread the pipelined view batch is bound by)."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _check_means(means3D: torch.Tensor) -> None:
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:46-48


def _preprocess_and_bin(dev, P, D, M, means3D, scales, scale_modifier, rotations, opacity, sh, cov3D_precomp, colors,
                        viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy, prefiltered, skip_color, radii, flags):
    L = _native.lib()
    s = _stream(dev)
    gbytes, _, _ = _native.scratch_sizes(P, 0, W, H)
    geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
    counts = (ctypes.c_int64 * 2)()  # num_rendered, and the number of tile-group instances the binning works on
    _native.check("gsr_preprocess", L.gsr_preprocess(
        s, P, D, M, _ptr(means3D), _ptr(scales), scale_modifier, _ptr(rotations), _ptr(opacity), _ptr(sh),
        _ptr(cov3D_precomp), _ptr(colors), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), W, H, tan_fovx, tan_fovy,
        int(bool(prefiltered)), int(skip_color), flags, radii.data_ptr(), geom.data_ptr(), counts))
    R, G = int(counts[0]), int(counts[1])
    # (the image scratch is sized once R is known: its checkpoint pool, 128 MB at 1080p, exists only for views with long lists)
    _, bbytes, ibytes = _native.scratch_sizes(P, R, W, H, G)
    binning = torch.empty(bbytes, dtype=torch.uint8, device=dev)
    img = torch.empty(ibytes, dtype=torch.uint8, device=dev)
    _native.check("gsr_bin", L.gsr_bin(s, P, R, G, W, H, geom.data_ptr(), _ptr(binning), img.data_ptr()))
    return R, geom, binning, img


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, flags=None):
    """RasterizeGaussiansCUDA, rasterize_points.cu:35-95 ->
    (num_rendered, color(3,H,W), depth(1,H,W), radii(P) i32, geomBuffer, binningBuffer, imgBuffer).
    `flags` (extension, keyword): GSR_FLAG_* of include/gsr.h; None = options.current_flags()."""
    flags = _flags(flags)
    _check_means(means3D)
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    u8 = dict(dtype=torch.uint8, device=dev)
    if P == 0:  # rasterize_points.cu:72: everything stays zero / empty
        return (0, torch.zeros((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev),
                torch.zeros((1, H, W), dtype=torch.float32, device=dev), torch.zeros((0,), dtype=torch.int32, device=dev),
                torch.empty(0, **u8), torch.empty(0, **u8), torch.empty(0, **u8))
    M = int(sh.size(1)) if sh.size(0) != 0 else 0
    means3D, opacity = _f32(means3D, "means3D"), _f32(opacity, "opacity", dev)
    background, viewmatrix, projmatrix, campos = (_f32(background, "bg", dev), _f32(viewmatrix, "viewmatrix", dev),
                                                  _f32(projmatrix, "projmatrix", dev), _f32(campos, "campos", dev))
    colors, scales, rotations, cov3D_precomp, sh = (_f32(colors, "colors_precomp", dev), _f32(scales, "scales", dev),
                                                    _f32(rotations, "rotations", dev),
                                                    _f32(cov3D_precomp, "cov3D_precomp", dev), _f32(sh, "sh", dev))
    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        R, geom, binning, img = _preprocess_and_bin(dev, P, int(degree), M, means3D, scales, float(scale_modifier),
                                                    rotations, opacity, sh, cov3D_precomp, colors, viewmatrix,
                                                    projmatrix, campos, W, H, float(tan_fovx), float(tan_fovy),
                                                    prefiltered, 0, radii, flags)
        _native.check("gsr_blend_forward", _native.lib().gsr_blend_forward(
            _stream(dev), P, R, W, H, background.data_ptr(), geom.data_ptr(), _ptr(binning), img.data_ptr(),
            out_color.data_ptr(), out_depth.data_ptr(), flags))
        if debug:
            torch.cuda.synchronize(dev)  # CHECK_CUDA, auxiliary.h:166-173
    return R, out_color, out_depth, radii, geom, binning, img


class PendingForward:
    """What rasterize_gaussians_begin() leaves for rasterize_gaussians_finish(): the view's arguments, its outputs so far and
    the native ticket of its counts."""
    __slots__ = ("dev", "stream", "P", "H", "W", "flags", "debug", "background", "radii", "geom", "ticket", "keep")


def rasterize_gaussians_begin(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                              prefiltered, debug, flags=None) -> "PendingForward":
    """First half of rasterize_gaussians() (extension, no reference counterpart; include/gsr.h gsr_preprocess_begin): K1 and
    the first depth-sort passes are enqueued on the current stream and the call returns WITHOUT waiting for num_rendered.
    rasterize_gaussians_finish(pending), on the same stream, does the rest and returns what rasterize_gaussians() returns;
    between the two the launch thread may enqueue other work (the view batch of multiview.py: the next view's begin comes
    before this view's backward, so the host never waits for a readback).  The argument tensors must stay unchanged until
    finish returned (they are kept alive here)."""
    flags = _flags(flags)
    _check_means(means3D)
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if P == 0:
        raise RuntimeError("rasterize_gaussians_begin: an empty scene has nothing to wait for; call rasterize_gaussians()")
    M = int(sh.size(1)) if sh.size(0) != 0 else 0
    means3D, opacity = _f32(means3D, "means3D"), _f32(opacity, "opacity", dev)
    background, viewmatrix, projmatrix, campos = (_f32(background, "bg", dev), _f32(viewmatrix, "viewmatrix", dev),
                                                  _f32(projmatrix, "projmatrix", dev), _f32(campos, "campos", dev))
    colors, scales, rotations, cov3D_precomp, sh = (_f32(colors, "colors_precomp", dev), _f32(scales, "scales", dev),
                                                    _f32(rotations, "rotations", dev),
                                                    _f32(cov3D_precomp, "cov3D_precomp", dev), _f32(sh, "sh", dev))
    pf = PendingForward()
    pf.dev, pf.P, pf.H, pf.W, pf.flags, pf.debug, pf.background = dev, P, H, W, flags, bool(debug), background
    pf.keep = (means3D, opacity, viewmatrix, projmatrix, campos, colors, scales, rotations, cov3D_precomp, sh)
    pf.radii = torch.empty((P,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        pf.stream = _stream(dev)
        gbytes, _, _ = _native.scratch_sizes(P, 0, W, H)
        pf.geom = torch.empty(gbytes, dtype=torch.uint8, device=dev)
        ticket = ctypes.c_void_p()
        _native.check("gsr_preprocess_begin", _native.lib().gsr_preprocess_begin(
            pf.stream, P, int(degree), M, _ptr(means3D), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(opacity),
            _ptr(sh), _ptr(cov3D_precomp), _ptr(colors), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos), W, H,
            float(tan_fovx), float(tan_fovy), int(bool(prefiltered)), 0, flags, pf.radii.data_ptr(), pf.geom.data_ptr(),
            ctypes.byref(ticket)))
    pf.ticket = ticket
    return pf


def rasterize_gaussians_finish(pf: "PendingForward"):
    """Second half of rasterize_gaussians(): -> (num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer)."""
    if pf.ticket is None:
        raise RuntimeError("rasterize_gaussians_finish: this pending forward was finished already")
    dev, P, H, W, flags = pf.dev, pf.P, pf.H, pf.W, pf.flags
    L = _native.lib()
    with torch.cuda.device(dev):
        s = _stream(dev)
        if s != pf.stream:
            raise RuntimeError("rasterize_gaussians_finish must run on the stream rasterize_gaussians_begin ran on")
        ticket, pf.ticket = pf.ticket, None  # (spent by the native call whatever it returns)
        counts = (ctypes.c_int64 * 2)()
        _native.check("gsr_preprocess_end", L.gsr_preprocess_end(s, P, W, H, pf.geom.data_ptr(), ticket, counts))
        R, G = int(counts[0]), int(counts[1])
        _, bbytes, ibytes = _native.scratch_sizes(P, R, W, H, G)
        binning = torch.empty(bbytes, dtype=torch.uint8, device=dev)
        img = torch.empty(ibytes, dtype=torch.uint8, device=dev)
        out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        _native.check("gsr_bin", L.gsr_bin(s, P, R, G, W, H, pf.geom.data_ptr(), _ptr(binning), img.data_ptr()))
        _native.check("gsr_blend_forward", L.gsr_blend_forward(
            s, P, R, W, H, pf.background.data_ptr(), pf.geom.data_ptr(), _ptr(binning), img.data_ptr(),
            out_color.data_ptr(), out_depth.data_ptr(), flags))
        if pf.debug:
            torch.cuda.synchronize(dev)
    pf.keep = None
    return R, out_color, out_depth, pf.radii, pf.geom, binning, img


def rasterize_gaussians_aux(background, colors, num_rendered, geomBuffer, binningBuffer, imgBuffer, image_height,
                            image_width, debug=False, flags=None):
    """Extension (no reference counterpart): a second image of the view whose state
    (`geomBuffer`, `binningBuffer`, `imgBuffer`, `num_rendered`) an earlier rasterize_gaussians() call returned,
    blended with the per-Gaussian `colors` (P,3) instead -- K6 only, what the reference obtains by running the whole
    forward again with override_color.  Forward only; the saved state of the first render is left intact."""
    flags = _flags(flags)
    _require_cuda(colors, "colors")
    dev = colors.device
    P, H, W = int(colors.size(0)), int(image_height), int(image_width)
    out = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
    if colors.ndimension() != 2 or colors.size(1) != NUM_CHANNELS:
        raise RuntimeError("aux colors must have dimensions (num_points, 3)")
    if P == 0 or geomBuffer.numel() == 0:
        return out.zero_()
    colors, background = _f32(colors, "colors"), _f32(background, "bg", dev)
    for name, buf in (("geomBuffer", geomBuffer), ("binningBuffer", binningBuffer), ("imgBuffer", imgBuffer)):
        _on_device(buf, name, dev, torch.uint8)
    with torch.cuda.device(dev):
        _native.check("gsr_blend_forward_aux", _native.lib().gsr_blend_forward_aux(
            _stream(dev), P, int(num_rendered), W, H, background.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer),
            imgBuffer.data_ptr(), colors.data_ptr(), out.data_ptr(), None, flags))
        if debug:
            torch.cuda.synchronize(dev)
    return out


#: The blend backward's accumulator table kept ACROSS backwards (GSR_FLAG_ACC_SELF_CLEAN, include/gsr.h): one zeroed (P,16)
#: buffer per (device, stream), checked out for the duration of a backward's two native calls and put back only when both were
#: enqueued -- K8+K9 leaves it all zero again in stream order, so the next backward on that stream needs no clear (64 MB in
#: front of every backward at 10^6 Gaussians: 8 of the 14 us of K7's work-list launch).  A backward that fails between the two
#: calls simply does not return its table.  GSR_ACC_PERSIST=0 turns it off (every backward then clears a fresh table).
_ACC_TABLES = {}
_ACC_PERSIST = __import__("os").environ.get("GSR_ACC_PERSIST", "1") != "0"


def _checkout_acc(dev, stream: int, P: int):
    t = _ACC_TABLES.pop((dev.index, stream), None)
    if t is not None and t.numel() == _native.ACC_ROW * P:
        return t
    return torch.zeros((_native.ACC_ROW * P,), dtype=torch.float32, device=dev)  # (dropped: another P -- a densification)


def _alloc(fn, name: str, shape, zero: bool, dev) -> torch.Tensor:
    if fn is not None:
        t = fn(name, tuple(shape), zero)
        if t is not None:
            return t
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32, device=dev)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug, flags=None, grad_allocator=None):
    """RasterizeGaussiansBackwardCUDA, rasterize_points.cu:97-157 ->
    (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D | None, dL_dsh, dL_dscales, dL_drotations).
    `flags` (extension, keyword): the flags the forward of this view ran with; None = options.current_flags().
    `grad_allocator` (extension, keyword): fn(name, shape, zero) -> tensor | None for the gradient outputs, asked for
    "means2D", "opacities", "means3D", "cov3Ds_precomp", "sh", "scales", "rotations" with the tensor's shape.  An allocator
    may answer None to anything (the gradient is then allocated privately).  Special names, all optional:
      "acc_rows"           shape (16 P,), asked FIRST: the blend backward's accumulator table (include/gsr.h: GSR_ACC_*, one
                           64-byte row per Gaussian; workspace, never a gradient -- since round 6 dL_dmeans2D / dL_dopacity
                           are copied out of it by K8+K9).  Asked with zero = False: the blend backward clears it itself
                           (GSR_FLAG_CLEAR_GRADS).  A float32 tensor of 16 P elements, 64-byte aligned, or None;
      "sh_rgb"             shape (P,3): a tensor here asks for the clamp-masked colour gradient INSTEAD of dL_dsh (which is
                           then returned as None; gaussianeditor_amd.multiview rebuilds it after the exchange);
      "after_blend_backward"  a notification, only in the "sh_rgb" mode: `shape` is K7's row mask (uint8 (P,): 1 for every
                           Gaussian whose accumulator row it adds to), K7 has been enqueued and K8+K9 has not; the return
                           value is ignored;
      "row_state"          shape (P,): a uint8 tensor here says that "means2D", "opacities", "means3D", "sh" / "sh_rgb",
                           "scales" and "rotations" were answered with tensors the allocator keeps across calls, with this
                           per-Gaussian state next to them (include/gsr.h: gsr_preprocess_backward_rows): rows that still
                           hold the zeros of an earlier call are not rewritten."""
    flags = _flags(flags)
    dev = means3D.device
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.size(0) != 0 else 0
    if P == 0:
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)  # noqa: E731
        return z(0, 3), z(0, NUM_CHANNELS), z(0, 1), z(0, 3), z(0, 6), z(0, M, 3), z(0, 3), z(0, 4)
    _require_cuda(means3D, "means3D")
    means3D = _f32(means3D, "means3D")
    background, viewmatrix, projmatrix, campos = (_f32(background, "bg", dev), _f32(viewmatrix, "viewmatrix", dev),
                                                  _f32(projmatrix, "projmatrix", dev), _f32(campos, "campos", dev))
    colors, scales, rotations, cov3D_precomp, sh = (_f32(colors, "colors_precomp", dev), _f32(scales, "scales", dev),
                                                    _f32(rotations, "rotations", dev),
                                                    _f32(cov3D_precomp, "cov3D_precomp", dev), _f32(sh, "sh", dev))
    dL_dpix = _f32(dL_dout_color, "dL_dout_color", dev)
    _on_device(radii, "radii", dev, torch.int32)
    for name, buf in (("geomBuffer", geomBuffer), ("binningBuffer", binningBuffer), ("imageBuffer", imageBuffer)):
        _on_device(buf, name, dev, torch.uint8)
    radii = radii.contiguous()
    has_scales = scales.numel() != 0
    grad_alloc = grad_allocator
    # The accumulator table of the blend backward (one 64-byte row per Gaussian; not zero-filled here: K7's own launch clears
    # it -- GSR_FLAG_CLEAR_GRADS; a separate fill cost 8 us at 10^6 Gaussians).  dL_dmeans2D / dL_dopacity / dL_dcolors leave
    # through K8+K9, which writes every row.
    acc = grad_alloc("acc_rows", (_native.ACC_ROW * P,), False) if grad_alloc is not None else None
    if not (isinstance(acc, torch.Tensor) and acc.dtype == torch.float32 and acc.numel() == _native.ACC_ROW * P
            and acc.device == dev and acc.is_contiguous() and acc.data_ptr() % 64 == 0):
        acc = None
    dL_dmeans2D = _alloc(grad_alloc, "means2D", (P, 3), False, dev)
    dL_dopacity = _alloc(grad_alloc, "opacities", (P, 1), False, dev)
    # the gradient of colors_precomp (rasterize_points.cu:133): written only for a caller that passed colours; with SHs the
    # reference's tensor holds the colour accumulator nobody reads -- here a view of that column group
    has_colors = colors.numel() != 0
    # (zeros, not empty, under an allocator: its "row_state" mode leaves rows that are zero again unwritten)
    dL_dcolors = (torch.zeros if grad_alloc is not None else torch.empty)((P, NUM_CHANNELS), dtype=torch.float32, device=dev) \
        if has_colors else None  # (with SHs: set below, once the table is chosen)
    dL_dmeans3D = _alloc(grad_alloc, "means3D", (P, 3), False, dev)
    # the gradient of cov3D_precomp: only where that input exists (the reference fills the tensor with its intermediate
    # dL_dcov3D either way -- 24 B per Gaussian nobody reads when scales / rotations are given; here: None)
    dL_dcov3D = _alloc(grad_alloc, "cov3Ds_precomp", (P, 6), False, dev) if cov3D_precomp.numel() != 0 else None
    # "rgb" exchange mode (multiview.py): an allocator that hands out a (P,3) "sh_rgb" tensor asks for the clamp-masked
    # colour gradient INSTEAD of the (P,M,3) SH gradient; dL_dsh is then returned as None and rebuilt after the exchange
    dL_drgb = grad_alloc("sh_rgb", (P, 3), False) if (grad_alloc is not None and M != 0) else None
    dL_dsh = None if dL_drgb is not None else _alloc(grad_alloc, "sh", (P, M, 3), M == 0, dev)
    dL_dscales = _alloc(grad_alloc, "scales", (P, 3), not has_scales, dev)
    dL_drotations = _alloc(grad_alloc, "rotations", (P, 4), not has_scales, dev)
    row_state = grad_alloc("row_state", (P,), False) if grad_alloc is not None else None
    if row_state is not None and not (isinstance(row_state, torch.Tensor) and row_state.dtype == torch.uint8 and
                                      row_state.numel() == P and row_state.is_contiguous() and row_state.device == dev):
        row_state = None
    # Which table?  The one kept across backwards on this stream: zero on entry, zeroed again by K8+K9 -- no clear (nobody reads
    # the table between K7 and K8+K9: the exchange routes plan their messages from K7's `touched` mask, handed to
    # after_blend_backward).  With an allocator's own table or persistent rows: a table cleared by K7's own launch.
    stream = _stream(dev)
    persist = _ACC_PERSIST and acc is None and row_state is None
    if persist:
        acc = _checkout_acc(dev, stream, P)
        bwd_flags = flags | options.FLAG_ACC_SELF_CLEAN
    else:
        if acc is None:
            acc = torch.empty((_native.ACC_ROW * P,), dtype=torch.float32, device=dev)
        bwd_flags = flags | options.FLAG_CLEAR_GRADS  # (the table is not zero: the blend backward clears it)
    acc = acc.view(P, _native.ACC_ROW)
    if not has_colors:
        dL_dcolors = acc[:, _native.ACC_COLOR:_native.ACC_COLOR + NUM_CHANNELS] if not persist else \
            torch.zeros((0, NUM_CHANNELS), dtype=torch.float32, device=dev)
    L = _native.lib()
    with torch.cuda.device(dev):
        col_out = dL_dcolors.data_ptr() if has_colors else None
        if row_state is not None:  # gradient arrays kept across calls: only the rows that change are written
            # (also when nothing was rendered: the call then only clears the accumulator table)
            touched = torch.empty(((P + 15) // 16) * 16, dtype=torch.uint8, device=dev) if dL_drgb is not None else None
            _native.check("gsr_blend_backward", L.gsr_blend_backward(
                _stream(dev), P, int(R), W, H, background.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer),
                imageBuffer.data_ptr(), dL_dpix.data_ptr(), acc.data_ptr(), _ptr(touched), bwd_flags))
            if dL_drgb is not None:
                grad_alloc("after_blend_backward", touched[:P], False)
            _native.check("gsr_preprocess_backward_rows", L.gsr_preprocess_backward_rows(
                _stream(dev), P, int(degree), M, W, H, means3D.data_ptr(), _ptr(sh), _ptr(scales), float(scale_modifier),
                _ptr(rotations), _ptr(cov3D_precomp), viewmatrix.data_ptr(), projmatrix.data_ptr(), _ptr(campos),
                float(tan_fovx), float(tan_fovy), radii.data_ptr(), geomBuffer.data_ptr(), acc.data_ptr(),
                dL_dmeans2D.data_ptr(), dL_dopacity.data_ptr(), col_out, dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D),
                _ptr(dL_dsh) if dL_drgb is None else None, None if dL_drgb is None else dL_drgb.data_ptr(),
                dL_dscales.data_ptr() if has_scales else None, dL_drotations.data_ptr() if has_scales else None,
                row_state.data_ptr()))
        elif dL_drgb is None:
            _native.check("gsr_backward", L.gsr_backward(
                _stream(dev), P, int(degree), M, int(R), W, H, background.data_ptr(), means3D.data_ptr(), _ptr(sh),
                _ptr(colors), _ptr(scales), float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                viewmatrix.data_ptr(), projmatrix.data_ptr(), _ptr(campos), float(tan_fovx), float(tan_fovy),
                radii.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer), imageBuffer.data_ptr(), dL_dpix.data_ptr(),
                acc.data_ptr(), dL_dmeans2D.data_ptr(), dL_dopacity.data_ptr(), col_out,
                dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D), _ptr(dL_dsh), dL_dscales.data_ptr() if has_scales else None,
                dL_drotations.data_ptr() if has_scales else None, bwd_flags))
        else:
            # (also when nothing was rendered: the call then only clears the accumulator table)
            touched = torch.empty(((P + 15) // 16) * 16, dtype=torch.uint8, device=dev)  # K7 clears it and marks the rows it adds to
            _native.check("gsr_blend_backward", L.gsr_blend_backward(
                _stream(dev), P, int(R), W, H, background.data_ptr(), geomBuffer.data_ptr(), _ptr(binningBuffer),
                imageBuffer.data_ptr(), dL_dpix.data_ptr(), acc.data_ptr(), touched.data_ptr(),
                bwd_flags & ~options.FLAG_ACC_SELF_CLEAN))
            # notification (no allocation): K7 is enqueued, K8+K9 not yet -- multiview.py starts the exchange of the
            # touched-row counts here (from K7's row mask), so that it and the host's wait for it run underneath K8+K9
            grad_alloc("after_blend_backward", touched[:P], False)
            _native.check("gsr_preprocess_backward_rgb", L.gsr_preprocess_backward_rgb(
                _stream(dev), P, int(degree), M, W, H, means3D.data_ptr(), _ptr(sh), _ptr(scales), float(scale_modifier),
                _ptr(rotations), _ptr(cov3D_precomp), viewmatrix.data_ptr(), projmatrix.data_ptr(), _ptr(campos),
                float(tan_fovx), float(tan_fovy), radii.data_ptr(), geomBuffer.data_ptr(), acc.data_ptr(),
                dL_dmeans2D.data_ptr(), dL_dopacity.data_ptr(), dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D), dL_drgb.data_ptr(),
                dL_dscales.data_ptr() if has_scales else None, dL_drotations.data_ptr() if has_scales else None,
                options.FLAG_ACC_SELF_CLEAN if persist else 0))
        if debug:
            torch.cuda.synchronize(dev)
    if persist:  # both halves are enqueued: in stream order the table is all zero again
        _ACC_TABLES[(dev.index, stream)] = acc.view(-1)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def sh_grad_compose(means3D, campos_all, rgb_all, degree, M):
    """Extension for the multi-GPU exchange: dL_dsh (P,M,3) = sum over views v (ascending) of c_k(dir_v) * rgb_all[v],
    from the views' camera centres (N,3) and clamp-masked colour gradients (N,P,3) -- gsr_sh_grad_compose."""
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    P, N = int(means3D.size(0)), int(rgb_all.size(0))
    if rgb_all.shape != (N, P, 3) or campos_all.shape != (N, 3):
        raise RuntimeError("sh_grad_compose: expected campos (N,3) and colour gradients (N,P,3)")
    out = torch.empty((P, int(M), 3), dtype=torch.float32, device=dev)
    if P == 0:
        return out
    means3D, campos_all, rgb_all = _f32(means3D, "means3D"), _f32(campos_all, "campos", dev), _f32(rgb_all, "rgb", dev)
    with torch.cuda.device(dev):
        _native.check("gsr_sh_grad_compose", _native.lib().gsr_sh_grad_compose(
            _stream(dev), P, int(degree), int(M), N, means3D.data_ptr(), campos_all.data_ptr(), rgb_all.data_ptr(),
            out.data_ptr()))
    return out


# --- "touched rows" form of the multi-GPU exchange (gaussianeditor_amd/multiview.py; include/gsr.h: view messages) -----
def _dense_grads(tensors):
    """gsr_dense_grads from [means3D, scales, rotations, means2D, opacities, sh | None] (contiguous float32)."""
    return _native.DenseGrads(*[0 if t is None else t.data_ptr() for t in tensors])


def view_message_words(P, cap):
    n = ctypes.c_int64(0)
    _native.check("gsr_view_message_words", _native.lib().gsr_view_message_words(int(P), int(cap), ctypes.byref(n)))
    return int(n.value)


def view_message_plan(grads5, rgb, readback=True):
    """Marks the rows of this view's gradients (means3D, scales, rotations, means2D, opacities + colour gradient) that are
    not entirely zero.  Returns (plan, count): an int (one host readback), or with readback=False a 1-element int64 device
    tensor (no synchronisation; valid in stream order)."""
    _require_cuda(rgb, "rgb")
    dev, P = rgb.device, int(rgb.size(0))
    if P == 0:
        return (None, None, 0, dev), (0 if readback else torch.zeros(1, dtype=torch.int64, device=dev))
    L = _native.lib()
    mask = torch.empty(P, dtype=torch.uint8, device=dev)
    nbytes = ctypes.c_size_t(0)
    _native.check("gsr_compact_workspace_size", L.gsr_compact_workspace_size(P, ctypes.byref(nbytes)))
    work = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
    count = ctypes.c_int64(0)
    dg = _dense_grads(list(grads5) + [None])
    with torch.cuda.device(dev):
        _native.check("gsr_view_message_plan", L.gsr_view_message_plan(_stream(dev), P, ctypes.byref(dg), rgb.data_ptr(),
                                                                        mask.data_ptr(), work.data_ptr(),
                                                                        ctypes.byref(count) if readback else None))
    return (mask, work, P, dev), (int(count.value) if readback else work[:8].view(torch.int64))


def view_message_plan_blend(touched):
    """The same plan from the blend backward's row mask (uint8 (P,), written by K7 itself), i.e. BEFORE K8+K9 has run --
    gsr_view_message_plan_blend.  The mask IS the plan's mask.  Never synchronises: returns (plan, count) with count a
    1-element int64 device tensor valid in stream order."""
    _require_cuda(touched, "touched")
    dev, P = touched.device, int(touched.numel())
    if P == 0:
        return (None, None, 0, dev), torch.zeros(1, dtype=torch.int64, device=dev)
    if touched.dtype != torch.uint8 or not touched.is_contiguous():
        raise RuntimeError("view_message_plan_blend: expected the contiguous uint8 row mask of the blend backward")
    L = _native.lib()
    nbytes = ctypes.c_size_t(0)
    _native.check("gsr_compact_workspace_size", L.gsr_compact_workspace_size(P, ctypes.byref(nbytes)))
    work = torch.empty(int(nbytes.value), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _native.check("gsr_view_message_plan_blend", L.gsr_view_message_plan_blend(
            _stream(dev), P, touched.data_ptr(), work.data_ptr()))
    return (touched, work, P, dev), work[:8].view(torch.int64)


def view_message_pack(plan, grads5, rgb, campos, cap, message):
    """Writes this view's message (gsr_view_message_words(P, cap) float32 words) into `message`."""
    mask, work, P, dev = plan
    dg = _dense_grads(list(grads5) + [None])
    ptr = lambda t: 0 if t is None else t.data_ptr()  # noqa: E731
    with torch.cuda.device(dev):
        _native.check("gsr_view_message_pack", _native.lib().gsr_view_message_pack(
            _stream(dev), P, ctypes.byref(dg), ptr(rgb), campos.data_ptr(), ptr(mask), ptr(work), int(cap), message.data_ptr()))


def view_messages_accumulate(messages, P, cap, degree, M, means3D, dense, row_valid=None):
    """dense = the sum of the views' messages (rows of `messages`, (n, words)), view 0 first; dense[5] (SH gradient,
    (P,M,3) or None) rebuilt from the colour gradients -- gsr_view_messages_accumulate.
    `row_valid` (uint8 (P,), optional): receives 1 where some view sent a row, 0 elsewhere, and the rows of `dense` of the
    latter are NOT written (gsr_view_messages_accumulate_rows): the consumer must take them as zeros."""
    dev = messages.device
    dg = _dense_grads(dense)
    if row_valid is not None and (row_valid.dtype != torch.uint8 or row_valid.numel() != int(P) or not row_valid.is_contiguous()):
        raise RuntimeError("row_valid must be a contiguous uint8 tensor of P elements")
    with torch.cuda.device(dev):
        _native.check("gsr_view_messages_accumulate", _native.lib().gsr_view_messages_accumulate_rows(
            _stream(dev), int(P), int(degree), int(M), int(messages.size(0)), messages.data_ptr(), int(messages.stride(0)),
            int(cap), means3D.data_ptr(), ctypes.byref(dg), None if row_valid is None else row_valid.data_ptr()))


def mark_visible(means3D, viewmatrix, projmatrix):
    """markVisible, rasterize_points.cu:159-175 -> bool (P)."""
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        _require_cuda(means3D, "means3D")
        dev = means3D.device
        means3D, viewmatrix, projmatrix = (_f32(means3D, "means3D"), _f32(viewmatrix, "viewmatrix", dev),
                                           _f32(projmatrix, "projmatrix", dev))
        with torch.cuda.device(dev):
            _native.check("gsr_mark_visible", _native.lib().gsr_mark_visible(
                _stream(dev), P, means3D.data_ptr(), viewmatrix.data_ptr(), projmatrix.data_ptr(), present.data_ptr()))
    return present


def apply_weights(background, means3D, weights, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                  projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered,
                  image_weights, cnt, debug, flags=None):
    """applyWeightsGaussiansCUDA, rasterize_points.cu:177-234.  `weights` (P,C) float32 and
    `cnt` (P[,1]) int32 are updated in place; returns None."""
    flags = _flags(flags)
    _check_means(means3D)
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    C = int(image_weights.size(0))
    if C not in (1, 2, 3):
        # the reference printf()s and exit(-1)s the process (apply_weights.cu:377-380)
        raise _native.GsrError("gsr_trace_weights", -2, f"Unsupported number of channels: {C}")
    if P == 0:
        return None
    _require_cuda(means3D, "means3D")
    dev = means3D.device
    if weights.dtype != torch.float32 or cnt.dtype != torch.int32:
        raise RuntimeError("apply_weights: weights must be float32 and cnt int32")
    if weights.numel() != P * C or cnt.numel() != P:
        raise RuntimeError("apply_weights: weights must hold P*C and cnt P elements")
    _on_device(weights, "weights", dev, torch.float32)
    _on_device(cnt, "cnt", dev, torch.int32)
    means3D, opacity = _f32(means3D, "means3D"), _f32(opacity, "opacity", dev)
    viewmatrix, projmatrix = _f32(viewmatrix, "viewmatrix", dev), _f32(projmatrix, "projmatrix", dev)
    scales, rotations, cov3D_precomp = (_f32(scales, "scales", dev), _f32(rotations, "rotations", dev),
                                        _f32(cov3D_precomp, "cov3D_precomp", dev))
    image_weights = _f32(image_weights, "image_weights", dev)
    w_work = weights if weights.is_contiguous() else weights.contiguous()
    c_work = cnt if cnt.is_contiguous() else cnt.contiguous()
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        # the reference passes `weights` in the colour slot only to skip the SH evaluation
        # (rasterizer_impl.cu:384, apply_weights.cu:219): no colour is needed at all here.
        R, geom, binning, img = _preprocess_and_bin(dev, P, 0, 0, means3D, scales, float(scale_modifier), rotations,
                                                    opacity, None, cov3D_precomp, None, viewmatrix, projmatrix, None,
                                                    W, H, float(tan_fovx), float(tan_fovy), prefiltered, 1, radii, flags)
        _native.check("gsr_trace_weights", _native.lib().gsr_trace_weights(
            _stream(dev), P, R, W, H, C, geom.data_ptr(), _ptr(binning), img.data_ptr(), image_weights.data_ptr(),
            w_work.data_ptr(), c_work.data_ptr(), flags))
        if debug:
            torch.cuda.synchronize(dev)
    if w_work is not weights:
        weights.copy_(w_work)
    if c_work is not cnt:
        cnt.copy_(c_work)
    return None


def arrays_equal(pairs) -> bool:
    """Extension (include/gsr.h: gsr_arrays_equal): are the (a, b) tensor pairs -- contiguous, same shape and dtype, on one
    ROCm device, at most 8 -- identical bit for bit?  One compare launch; blocks until the device has answered (~10 us)."""
    pairs = [(a, b) for a, b in pairs if a.numel() != 0 or b.numel() != 0]
    if not pairs:
        return True
    if len(pairs) > 8:
        raise RuntimeError("arrays_equal: at most 8 pairs")
    dev = pairs[0][0].device
    for a, b in pairs:
        _require_cuda(a, "arrays_equal argument")
        if a.shape != b.shape or a.dtype != b.dtype or a.device != dev or b.device != dev:
            return False
        if not (a.is_contiguous() and b.is_contiguous()) or (a.numel() * a.element_size()) % 4 != 0:
            raise RuntimeError("arrays_equal: tensors must be contiguous and a multiple of 4 bytes long")
    n = len(pairs)
    A = (ctypes.c_void_p * n)(*[a.data_ptr() for a, _ in pairs])
    B = (ctypes.c_void_p * n)(*[b.data_ptr() for _, b in pairs])
    S = (ctypes.c_size_t * n)(*[a.numel() * a.element_size() for a, _ in pairs])
    eq = ctypes.c_int(0)
    with torch.cuda.device(dev):
        _native.check("gsr_arrays_equal", _native.lib().gsr_arrays_equal(_stream(dev), n, A, B, S, ctypes.byref(eq)))
    return bool(eq.value)
