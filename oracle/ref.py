"""ctypes front-end of oracle/_ref/libgsr_ref*.so: the REFERENCE's own rasterizer sources compiled for
gfx950 (oracle/ref_build/Makefile).  TEST INFRASTRUCTURE ONLY; needs a GPU.

Two builds exist: "fma" (hipcc's default floating-point contraction, the analogue of nvcc's default
-fmad=true) and "nofma" (-ffp-contract=off, one rounding per operation -- the semantics the CPU oracle
specifies).  All tensors are torch CUDA tensors; intermediates are copied out of the reference's own
GeometryState / BinningState / ImageState chunks.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs: Dict[str, ctypes.CDLL] = {}


def lib_path(variant: str = "nofma") -> str:
    return os.path.join(_HERE, "_ref", "libgsr_ref_nofma.so" if variant == "nofma" else "libgsr_ref.so")


def available(variant: str = "nofma") -> bool:
    return os.path.exists(lib_path(variant))


def _lib(variant: str) -> ctypes.CDLL:
    if variant not in _libs:
        L = ctypes.CDLL(lib_path(variant))
        L.gsrref_create.restype = ctypes.c_void_p
        L.gsrref_destroy.argtypes = [ctypes.c_void_p]
        _libs[variant] = L
    return _libs[variant]


def _p(t: Optional[torch.Tensor]):
    return ctypes.c_void_p(0 if t is None or t.numel() == 0 else t.data_ptr())


class Reference:
    """One context = the three scratch chunks of one forward call (kept for export / backward)."""

    def __init__(self, variant: str = "nofma", device: str = "cuda:0"):
        self.L = _lib(variant)
        self.dev = torch.device(device)
        self.h = ctypes.c_void_p(self.L.gsrref_create())

    def __del__(self):
        try:
            self.L.gsrref_destroy(self.h)
        except Exception:
            pass

    def _d(self, t, dtype=torch.float32):
        return None if t is None else t.to(self.dev, dtype).contiguous()

    def forward(self, means3D, scales, rotations, opacities, shs, colors_precomp, cov3D_precomp, viewmatrix, projmatrix,
                campos, bg, W, H, tanfovx, tanfovy, scale_modifier=1.0, sh_degree=0) -> Dict[str, torch.Tensor]:
        d = self._d
        self.inp = dict(means3D=d(means3D), scales=d(scales), rotations=d(rotations), opacities=d(opacities), shs=d(shs),
                        colors=d(colors_precomp), cov3D=d(cov3D_precomp), view=d(viewmatrix), proj=d(projmatrix),
                        campos=d(campos), bg=d(bg))
        i = self.inp
        P = i["means3D"].shape[0]
        M = 0 if i["shs"] is None else i["shs"].shape[1]
        self.P, self.M, self.W, self.H, self.D = P, M, W, H, sh_degree
        self.tf = (float(tanfovx), float(tanfovy), float(scale_modifier))
        f32 = dict(dtype=torch.float32, device=self.dev)
        color, depth = torch.empty((3, H, W), **f32), torch.empty((1, H, W), **f32)
        radii = torch.empty(P, dtype=torch.int32, device=self.dev)
        R = self.L.gsrref_forward(self.h, P, sh_degree, M, _p(i["bg"]), W, H, _p(i["means3D"]), _p(i["shs"]), _p(i["colors"]),
                                  _p(i["opacities"]), _p(i["scales"]), ctypes.c_float(scale_modifier), _p(i["rotations"]),
                                  _p(i["cov3D"]), _p(i["view"]), _p(i["proj"]), _p(i["campos"]), ctypes.c_float(tanfovx),
                                  ctypes.c_float(tanfovy), 0, _p(color), _p(depth), _p(radii))
        if R < 0:
            raise RuntimeError("reference forward failed")
        self.R, self.radii = R, radii
        T = ((W + 15) // 16) * ((H + 15) // 16)
        out = dict(means2D=torch.zeros((P, 2), **f32), depths=torch.zeros(P, **f32), cov3D=torch.zeros((P, 6), **f32),
                   rgb=torch.zeros((P, 3), **f32), conic_opacity=torch.zeros((P, 4), **f32),
                   tiles_touched=torch.zeros(P, dtype=torch.int32, device=self.dev),
                   clamped=torch.zeros((P, 3), dtype=torch.uint8, device=self.dev),
                   point_offsets=torch.zeros(P, dtype=torch.int32, device=self.dev),
                   keys=torch.zeros(R, dtype=torch.int64, device=self.dev),
                   point_list=torch.zeros(R, dtype=torch.int32, device=self.dev),
                   ranges=torch.zeros((T, 2), dtype=torch.int32, device=self.dev), final_T=torch.zeros(H * W, **f32),
                   n_contrib=torch.zeros(H * W, dtype=torch.int32, device=self.dev))
        o = out
        if self.L.gsrref_export(self.h, _p(o["means2D"]), _p(o["depths"]), _p(o["cov3D"]), _p(o["rgb"]),
                                _p(o["conic_opacity"]), _p(o["tiles_touched"]), _p(o["clamped"]), _p(o["point_offsets"]),
                                _p(o["keys"]), _p(o["point_list"]), _p(o["ranges"]), _p(o["final_T"]), _p(o["n_contrib"])) != 0:
            raise RuntimeError("reference export failed")
        out.update(color=color, depth=depth, radii=radii, num_rendered=R)
        return out

    def backward(self, dL_dpix) -> Dict[str, torch.Tensor]:
        i, P, M = self.inp, self.P, self.M
        f32 = dict(dtype=torch.float32, device=self.dev)
        g = dict(dL_dmeans2D=torch.empty((P, 3), **f32), dL_dconic=torch.empty((P, 4), **f32),
                 dL_dopacity=torch.empty((P, 1), **f32), dL_dcolors=torch.empty((P, 3), **f32),
                 dL_dmeans3D=torch.empty((P, 3), **f32), dL_dcov3D=torch.empty((P, 6), **f32),
                 dL_dsh=torch.empty((P, M, 3), **f32), dL_dscales=torch.empty((P, 3), **f32),
                 dL_drotations=torch.empty((P, 4), **f32))
        dpix = self._d(dL_dpix)
        rc = self.L.gsrref_backward(self.h, self.D, M, _p(i["bg"]), _p(i["means3D"]), _p(i["shs"]), _p(i["colors"]),
                                    _p(i["scales"]), ctypes.c_float(self.tf[2]), _p(i["rotations"]), _p(i["cov3D"]),
                                    _p(i["view"]), _p(i["proj"]), _p(i["campos"]), ctypes.c_float(self.tf[0]),
                                    ctypes.c_float(self.tf[1]), _p(self.radii), _p(dpix), _p(g["dL_dmeans2D"]),
                                    _p(g["dL_dconic"]), _p(g["dL_dopacity"]), _p(g["dL_dcolors"]), _p(g["dL_dmeans3D"]),
                                    _p(g["dL_dcov3D"]), _p(g["dL_dsh"]) if M else ctypes.c_void_p(0), _p(g["dL_dscales"]),
                                    _p(g["dL_drotations"]))
        if rc != 0:
            raise RuntimeError("reference backward failed")
        return g

    def apply_weights(self, means3D, scales, rotations, opacities, viewmatrix, projmatrix, campos, W, H, tanfovx, tanfovy,
                      image_weights, weights, cnt, scale_modifier=1.0):
        d = self._d
        m, s, r, o = d(means3D), d(scales), d(rotations), d(opacities)
        v, pj, cp, iw = d(viewmatrix), d(projmatrix), d(campos), d(image_weights)
        P = m.shape[0]
        radii = torch.empty(P, dtype=torch.int32, device=self.dev)
        rc = self.L.gsrref_apply_weights(self.h, P, W, H, _p(m), _p(weights), _p(o), _p(s), ctypes.c_float(scale_modifier),
                                         _p(r), ctypes.c_void_p(0), _p(v), _p(pj), _p(cp), ctypes.c_float(tanfovx),
                                         ctypes.c_float(tanfovy), _p(iw), _p(radii), _p(cnt), int(iw.shape[0]))
        if rc != 0:
            raise RuntimeError("reference apply_weights failed")


# ---- the reference's simple-knn (oracle/_ref/libknn_ref*.so) ----
_knn_libs: Dict[str, ctypes.CDLL] = {}


def knn_lib_path(variant: str = "nofma") -> str:
    return os.path.join(_HERE, "_ref", "libknn_ref_nofma.so" if variant == "nofma" else "libknn_ref.so")


def knn_available(variant: str = "nofma") -> bool:
    return os.path.exists(knn_lib_path(variant))


def knn_mean_dist2(points: torch.Tensor, variant: str = "nofma") -> torch.Tensor:
    """SimpleKNN::knn of the reference (simple_knn.cu:185-221) on a (P,3) float32 CUDA tensor -> (P,) float32."""
    if variant not in _knn_libs:
        L = ctypes.CDLL(knn_lib_path(variant))
        L.gsrref_knn.restype = ctypes.c_int
        L.gsrref_knn.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _knn_libs[variant] = L
    pts = points.contiguous()
    out = torch.zeros(pts.shape[0], dtype=torch.float32, device=pts.device)
    torch.cuda.synchronize()
    rc = _knn_libs[variant].gsrref_knn(int(pts.shape[0]), pts.data_ptr(), out.data_ptr())
    if rc != 0:
        raise RuntimeError(f"reference SimpleKNN::knn failed ({rc})")
    return out
