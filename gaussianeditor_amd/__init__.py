"""MI355X-native differentiable 3D-Gaussian-splatting rasterizer: a drop-in for the
`diff_gaussian_rasterization` extension used by buaacyw/GaussianEditor.

    import gaussianeditor_amd; gaussianeditor_amd.install()
    # from here on `import diff_gaussian_rasterization` resolves to the HIP implementation
"""
import sys as _sys

__version__ = "0.1.0"


def install() -> None:
    """Register the drop-in under the reference's module name so that unmodified reference
    code (gaussian_renderer/__init__.py:14-17, scene/gaussian_model.py:32) imports it."""
    from . import diff_gaussian_rasterization as _dgr

    _sys.modules["diff_gaussian_rasterization"] = _dgr
    _sys.modules["diff_gaussian_rasterization._C"] = _dgr._C
    # the other two hard imports of scene/gaussian_model.py (:24, :26)
    from . import simple_knn as _knn

    _sys.modules["simple_knn"] = _knn
    _sys.modules["simple_knn._C"] = _knn._C
    try:  # a real `plyfile` installation wins
        import plyfile as _ply  # noqa: F401
    except ImportError:
        from .compat import plyfile as _ply

        _sys.modules["plyfile"] = _ply


def set_tile_bounds(mode: str) -> None:
    """Opt-in binning rule (process-wide, include/gsr.h: GSR_OPT_TILE_BOUNDS).

    "reference" (default): every Gaussian is binned into the reference's square of side 2 ceil(3 sigma_max); the
    internal state (num_rendered, instance lists, n_contrib) equals the reference's bit for bit.
    "alpha": only into the tiles its alpha >= 1/255 level set can reach.  Images, depths, radii, traced weights and
    gradients are unchanged (every dropped instance would have been skipped at each pixel); fewer instances are
    sorted and walked.  Scratch buffers of a view must be produced and consumed under one setting."""
    from . import _native

    if mode not in ("reference", "alpha"):
        raise ValueError('tile bounds: "reference" or "alpha"')
    _native.check("gsr_set_option", _native.lib().gsr_set_option(1, 1 if mode == "alpha" else 0))


def get_tile_bounds() -> str:
    import ctypes

    from . import _native

    v = ctypes.c_int(0)
    _native.check("gsr_get_option", _native.lib().gsr_get_option(1, ctypes.byref(v)))
    return "alpha" if v.value else "reference"
