"""3DGS point-cloud files in the reference's layout: one command from a `point_cloud.ply` to the GPU render path.

The reference stores a trained scene as a binary little-endian PLY with ONE `vertex` element whose properties are all
float32, in this order (gaussiansplatting/scene/gaussian_model.py:396-445, `save_ply`):

    x y z   nx ny nz (zeros)   f_dc_0..2   f_rest_0..(3 (M - 1) - 1)   opacity   scale_0..2   rot_0..3

`f_dc` / `f_rest` are written CHANNEL-MAJOR (the (P, M', 3) tensors transposed to (P, 3, M') and flattened), `opacity` is
the logit and `scale_*` the log of what the rasterizer consumes, `rot_*` the unnormalised quaternion (w, x, y, z).
`load_ply` (:455-533) reads by property NAME, sorts `f_rest_*`, `scale_*`, `rot_*` by their numeric suffix, and derives
the SH degree from the number of `f_rest_*` columns.  `load_gaussians_ply` / `save_gaussians_ply` restate exactly that on
top of whichever `plyfile` is importable (the real package, or gaussianeditor_amd.compat.plyfile); `activated()` applies
the model's activations (:96-130: exp, sigmoid, normalize, cat) so that the result is what `render()` / the rasterizer take.
BASELINE configs[1] ("Mip-NeRF360 bicycle .ply, 1080p forward") is then `python bench.py --ply <file>`.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

__all__ = ["load_gaussians_ply", "save_gaussians_ply", "activated"]


def _plyfile():
    try:
        import plyfile  # the real package wins
    except ImportError:
        from .compat import plyfile
    return plyfile


def load_gaussians_ply(path: str, device="cpu") -> Dict[str, torch.Tensor]:
    """-> xyz (P,3), f_dc (P,1,3), f_rest (P,M-1,3), opacity (P,1) [logit], scaling (P,3) [log], rotation (P,4), float32,
    as `GaussianModel.load_ply` leaves them (gaussian_model.py:455-533), plus "max_sh_degree" (int)."""
    v = _plyfile().PlyData.read(path).elements[0]
    col = lambda name: np.asarray(v[name])  # noqa: E731
    names = [p.name for p in v.properties]
    xyz = np.stack((col("x"), col("y"), col("z")), axis=1)
    opacity = col("opacity")[..., np.newaxis]
    f_dc = np.zeros((xyz.shape[0], 3, 1))
    for c in range(3):
        f_dc[:, c, 0] = col(f"f_dc_{c}")
    by_suffix = lambda prefix: sorted([n for n in names if n.startswith(prefix)], key=lambda n: int(n.split("_")[-1]))  # noqa: E731
    rest_names = by_suffix("f_rest_")
    degree = int(((len(rest_names) + 3) / 3) ** 0.5 - 1)  # :479-480
    if 3 * (degree + 1) ** 2 - 3 != len(rest_names):
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* columns do not make a whole number of SH bands")
    f_rest = np.zeros((xyz.shape[0], len(rest_names)))
    for i, n in enumerate(rest_names):
        f_rest[:, i] = col(n)
    f_rest = f_rest.reshape((xyz.shape[0], 3, (degree + 1) ** 2 - 1))  # (P, F * coeffs) -> (P, F, coeffs), :485-487
    scale_names, rot_names = by_suffix("scale_"), by_suffix("rot")
    scales = np.stack([col(n) for n in scale_names], axis=1)
    rots = np.stack([col(n) for n in rot_names], axis=1)
    t = lambda a: torch.tensor(a, dtype=torch.float, device=device)  # noqa: E731
    return {"xyz": t(xyz), "f_dc": t(f_dc).transpose(1, 2).contiguous(), "f_rest": t(f_rest).transpose(1, 2).contiguous(),
            "opacity": t(opacity), "scaling": t(scales), "rotation": t(rots), "max_sh_degree": degree}


def save_gaussians_ply(path: str, xyz, f_dc, f_rest, opacity, scaling, rotation) -> None:
    """`GaussianModel.save_ply` (gaussian_model.py:410-445) for raw (un-activated) tensors of those shapes."""
    pf = _plyfile()
    n = lambda t: t.detach().cpu().numpy().astype(np.float32)  # noqa: E731
    xyz_ = n(xyz)
    dc = n(f_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    rest = n(f_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
    attrs = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(dc.shape[1])] + \
            [f"f_rest_{i}" for i in range(rest.shape[1])] + ["opacity"] + [f"scale_{i}" for i in range(scaling.shape[1])] + \
            [f"rot_{i}" for i in range(rotation.shape[1])]  # construct_list_of_attributes, :396-408
    table = np.concatenate((xyz_, np.zeros_like(xyz_), dc, rest, n(opacity).reshape(-1, 1), n(scaling), n(rotation)), axis=1)
    elements = np.empty(xyz_.shape[0], dtype=[(a, "f4") for a in attrs])
    for i, a in enumerate(attrs):
        elements[a] = table[:, i]
    pf.PlyData([pf.PlyElement.describe(elements, "vertex")]).write(path)


def activated(raw: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The getters of GaussianModel (gaussian_model.py:96-130) on the loaded tensors: what the rasterizer consumes, with the
    key names of gaussianeditor_amd.synth.synth_scene (xyz, scaling, rotation, opacity, features (P,M,3))."""
    return {"xyz": raw["xyz"], "scaling": torch.exp(raw["scaling"]), "rotation": torch.nn.functional.normalize(raw["rotation"]),
            "opacity": torch.sigmoid(raw["opacity"]), "features": torch.cat((raw["f_dc"], raw["f_rest"]), dim=1).contiguous()}
