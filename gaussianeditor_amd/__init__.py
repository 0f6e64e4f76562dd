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
    """Opt-in binning rule: the default of the GSR_FLAG_TILE_BOUNDS_ALPHA flag (include/gsr.h) for renders started
    from now on (a render's backward always reuses the flags of its forward; `options.override` is per thread).

    "reference" (default): every Gaussian is binned into the reference's square of side 2 ceil(3 sigma_max); the
    internal state (num_rendered, instance lists, n_contrib) equals the reference's bit for bit.
    "alpha": only into the tiles its alpha >= 1/255 level set can reach.  Images, depths, radii, traced weights and
    gradients are unchanged (every dropped instance would have been skipped at each pixel); fewer instances are
    sorted and walked."""
    from . import options

    if mode not in ("reference", "alpha"):
        raise ValueError('tile bounds: "reference" or "alpha"')
    f = options.default_flags() & ~options.FLAG_TILE_BOUNDS_ALPHA  # the default only: never a thread's override()
    options.set_default_flags(f | (options.FLAG_TILE_BOUNDS_ALPHA if mode == "alpha" else 0))


def get_tile_bounds() -> str:
    from . import options

    return "alpha" if options.default_flags() & options.FLAG_TILE_BOUNDS_ALPHA else "reference"


def set_fast_exp(on: bool) -> None:
    """Opt-in: the default of GSR_FLAG_FAST_EXP (include/gsr.h) -- exp() of the blend loops on the hardware's
    v_exp_f32 instead of the exactly specified polynomial.  Results stay within the 1e-5 parity bar; n_contrib /
    final_T are then no longer bit-identical to the CPU oracle."""
    from . import options

    f = options.default_flags() & ~options.FLAG_FAST_EXP
    options.set_default_flags(f | (options.FLAG_FAST_EXP if on else 0))


def get_fast_exp() -> bool:
    from . import options

    return bool(options.default_flags() & options.FLAG_FAST_EXP)


def set_view_reuse(on: bool) -> None:
    """View reuse (default on; `GSR_VIEW_REUSE=0` in the environment turns it off): a colour-override render of the view
    the rasterizer rendered last -- the reference's second `render(..., override_color=...)` of every training view and
    GUI frame -- runs the blend kernel alone on that render's state, after PROVING that camera, positions, scales,
    rotations and opacities are the first render's (diff_gaussian_rasterization/_reuse.py).  Off: every render runs in full."""
    from .diff_gaussian_rasterization import _reuse

    _reuse.set_view_reuse(on)


def get_view_reuse() -> bool:
    from .diff_gaussian_rasterization import _reuse

    return _reuse.view_reuse()
