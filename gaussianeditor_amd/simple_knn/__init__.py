"""Drop-in for the reference's `simple_knn` package (gaussiansplatting/submodules/simple-knn): only `_C.distCUDA2`
exists there and only `gaussiansplatting/scene/gaussian_model.py:26,288-291` uses it.  SURVEY.md section 8(f) rank 1."""
from . import _C  # noqa: F401
