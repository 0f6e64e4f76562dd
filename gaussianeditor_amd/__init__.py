"""MI355X-native differentiable 3D-Gaussian-splatting rasterizer: a drop-in for
the `diff_gaussian_rasterization` extension used by buaacyw/GaussianEditor."""

__version__ = "0.1.0"
