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
