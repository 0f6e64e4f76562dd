"""Deterministic synthetic scenes and cameras for tests and bench.py.

`synth_scene` is SURVEY.md section 8(d) `synth-v1(P, seed)`; `ring_cameras` is
`ring-v1(K, W, H)`.  Camera matrices follow the reference's host conventions
exactly (citations relative to /root/reference/gaussiansplatting):

* ``world_view_transform = getWorld2View2(R, T).T``      scene/cameras.py:92,
  utils/graphics_utils.py:40-51
* ``projection_matrix    = getProjectionMatrix(...).T``  scene/cameras.py:93,
  utils/graphics_utils.py:67-87
* ``full_proj_transform  = world_view_transform @ projection_matrix``  scene/cameras.py:94
* ``camera_center        = world_view_transform.inverse()[3, :3]``     scene/cameras.py:95

All tensors are produced on the CPU (float32) so that they are bit-identical
on every machine; callers move them to the GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch

__all__ = ["Camera", "synth_scene", "ring_cameras", "look_at_camera", "seed_gradient"]


@dataclass
class Camera:
    """Duck-type of the reference's ``Simple_Camera`` (scene/cameras.py:59-98):
    exactly the attributes ``render()`` / ``camera2rasterizer()`` read."""

    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # (4,4) = W2C^T
    full_proj_transform: torch.Tensor  # (4,4) = (P @ W2C)^T
    camera_center: torch.Tensor  # (3,)
    znear: float = 0.01
    zfar: float = 100.0

    def to(self, device) -> "Camera":
        return Camera(
            self.image_height,
            self.image_width,
            self.FoVx,
            self.FoVy,
            self.world_view_transform.to(device),
            self.full_proj_transform.to(device),
            self.camera_center.to(device),
            self.znear,
            self.zfar,
        )


def _world2view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """utils/graphics_utils.py:40-51 with translate=0, scale=1 (the two
    inversions of the original cancel for those defaults up to float64
    round-off; we keep them so the float32 result is identical)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    cam_center = C2W[:3, 3]
    C2W[:3, 3] = (cam_center + np.array([0.0, 0.0, 0.0])) * 1.0
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def _projection(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """utils/graphics_utils.py:67-87."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4, dtype=torch.float32)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_camera(eye, target, W: int, H: int, fovy_deg: float = 60.0, znear: float = 0.01, zfar: float = 100.0) -> Camera:
    """COLMAP-style camera (x right, y down, z forward) at `eye` looking at
    `target`, world up = +y pointing *down* in camera space (i.e. the camera's
    y axis is -world_up projected)."""
    eye = np.asarray(eye, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    world_up = np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, world_up)
    if np.linalg.norm(right) < 1e-8:
        right = np.array([1.0, 0.0, 0.0])
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    # camera-to-world rotation, columns = camera axes in world coordinates
    R_c2w = np.stack([right, down, fwd], axis=1)
    # the reference stores R = C2W rotation (qvec2rotmat(q)^T, dataset_readers.py:85)
    R = R_c2w
    T = -R_c2w.T @ eye  # W2C translation
    fovy = math.radians(fovy_deg)
    fovx = 2.0 * math.atan(math.tan(fovy / 2.0) * W / H)
    wv = torch.tensor(_world2view(R, T)).transpose(0, 1).contiguous()
    proj = _projection(znear, zfar, fovx, fovy).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return Camera(H, W, fovx, fovy, wv, full, center, znear, zfar)


def ring_cameras(K: int, W: int, H: int, radius: float = 4.0, elevation_deg: float = 15.0, fovy_deg: float = 60.0) -> List[Camera]:
    """ring-v1(K, W, H): K views at azimuth 360*k/K, elevation 15 deg, distance
    4 from the origin, looking at the origin."""
    cams = []
    el = math.radians(elevation_deg)
    for k in range(K):
        az = 2.0 * math.pi * k / K
        eye = [radius * math.cos(el) * math.sin(az), -radius * math.sin(el), -radius * math.cos(el) * math.cos(az)]
        cams.append(look_at_camera(eye, [0.0, 0.0, 0.0], W, H, fovy_deg))
    return cams


def synth_scene(P: int, seed: int = 0, s0: float = 0.01, sh_degree: int = 3) -> Dict[str, torch.Tensor]:
    """synth-v1(P, seed): the *activated* quantities the rasterizer consumes
    (scene/gaussian_model.py:221-258 getters): xyz (P,3), scaling (P,3) > 0,
    rotation (P,4) unit, opacity (P,1) in (0,1), features (P,M,3)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    M = (sh_degree + 1) ** 2
    xyz = 2.0 * torch.rand(P, 3, generator=g) - 1.0
    scaling = torch.exp(math.log(s0) + 0.35 * torch.randn(P, 3, generator=g))
    rotation = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=-1)
    opacity = torch.sigmoid(-2.0 + 6.0 * torch.rand(P, 1, generator=g))
    f_dc = 0.5 * torch.randn(P, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(P, M - 1, 3, generator=g)
    features = torch.cat([f_dc, f_rest], dim=1).contiguous()
    return {
        "xyz": xyz.contiguous(),
        "scaling": scaling.contiguous(),
        "rotation": rotation.contiguous(),
        "opacity": opacity.contiguous(),
        "features": features,
        "active_sh_degree": sh_degree,
        "bg": torch.zeros(3, dtype=torch.float32),
    }


def seed_gradient(H: int, W: int, seed: int = 0) -> torch.Tensor:
    """dL/dpixels for `loss = (color * G).sum()`, G = randn(3,H,W, seed+1000)/N."""
    g = torch.Generator(device="cpu").manual_seed(seed + 1000)
    return (torch.randn(3, H, W, generator=g) / float(H * W)).contiguous()
