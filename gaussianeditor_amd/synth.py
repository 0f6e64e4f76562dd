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

__all__ = ["Camera", "synth_scene", "synth_scene_v2", "ring_cameras", "look_at_camera", "seed_gradient"]


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


def _quat_from_frames(t1: torch.Tensor, t2: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    """Unit quaternions (r, x, y, z) of the rotations whose matrix columns are (t1, t2, n) (orthonormal, right-handed),
    in the convention the rasterizer builds R from (forward.cu:118-132)."""
    m00, m01, m02 = t1[:, 0], t2[:, 0], n[:, 0]
    m10, m11, m12 = t1[:, 1], t2[:, 1], n[:, 1]
    m20, m21, m22 = t1[:, 2], t2[:, 2], n[:, 2]
    # the numerically safe four-branch form: divide by the largest of (r, x, y, z)
    q = torch.stack([
        torch.stack([1 + m00 + m11 + m22, m21 - m12, m02 - m20, m10 - m01], 1),
        torch.stack([m21 - m12, 1 + m00 - m11 - m22, m01 + m10, m02 + m20], 1),
        torch.stack([m02 - m20, m01 + m10, 1 - m00 + m11 - m22, m12 + m21], 1),
        torch.stack([m10 - m01, m02 + m20, m12 + m21, 1 - m00 - m11 + m22], 1)], 1)  # (P, 4 candidates, 4)
    best = torch.stack([q[:, 0, 0], q[:, 1, 1], q[:, 2, 2], q[:, 3, 3]], 1).argmax(1)
    q = q[torch.arange(q.shape[0]), best]
    return torch.nn.functional.normalize(q, dim=-1)


def synth_scene_v2(P: int, seed: int = 0, sh_degree: int = 3) -> Dict[str, torch.Tensor]:
    """synth-v2(P, seed): a scene that LOOKS like a trained capture to the rasterizer, next to synth-v1's uniform cube (which
    covers a fifth of the ring views' tiles and lets a tenth of its Gaussians ever receive a gradient).  Stands in for what
    BASELINE configs[1] / configs[2] would exercise (threestudio/systems/GassuianEditor.py:165-207):

    * Gaussians lie ON SURFACES -- a dome of radius 8 around the ring cameras (every pixel of every ring view sees it: no
      empty tile), a ground disk, three spherical shells that cut through the ground and one another, a tilted wall through
      the origin -- 35 / 25 / 25 / 15 % of them;
    * they are DISKS: two tangent scales log-normal around 0.9 x the surface's sample spacing, the normal scale a tenth of
      that, the thin axis along the surface normal (jittered by ~6 degrees), a random rotation in the tangent plane;
    * opacity is BIMODAL: 65 % in [0.85, 0.99], 35 % in [0.02, 0.3];
    * colour: a smooth function of the position in the DC term + small higher-order terms.
    Same dictionary as synth_scene (activated quantities)."""
    g = torch.Generator(device="cpu").manual_seed(seed + 7919)
    M = (sh_degree + 1) ** 2
    n_dome = int(0.35 * P)
    n_ground = int(0.25 * P)
    n_shell = int(0.25 * P)
    n_wall = P - n_dome - n_ground - n_shell

    def unit(n):
        return torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1)

    pos, nrm, spacing = [], [], []
    # dome: radius 8, normals pointing inwards
    d = unit(n_dome)
    pos.append(8.0 * d)
    nrm.append(-d)
    spacing.append(torch.full((n_dome,), math.sqrt(4 * math.pi * 64.0 / max(n_dome, 1))))
    # ground: the disk of radius 3 in the plane y = 0.8 (the ring cameras sit at y = -1.04 and look down on it)
    r = 3.0 * torch.sqrt(torch.rand(n_ground, generator=g))
    a = 2 * math.pi * torch.rand(n_ground, generator=g)
    pos.append(torch.stack([r * torch.cos(a), torch.full((n_ground,), 0.8), r * torch.sin(a)], 1))
    nrm.append(torch.tensor([0.0, -1.0, 0.0]).expand(n_ground, 3))
    spacing.append(torch.full((n_ground,), math.sqrt(math.pi * 9.0 / max(n_ground, 1))))
    # three shells that intersect the ground and each other
    centres = torch.tensor([[0.0, 0.2, 0.0], [0.9, 0.45, 0.5], [-0.8, 0.1, -0.6]])
    radii = torch.tensor([0.8, 0.55, 0.65])
    which = torch.randint(0, 3, (n_shell,), generator=g)
    d = unit(n_shell)
    pos.append(centres[which] + radii[which, None] * d)
    nrm.append(d)
    area = float((4 * math.pi * radii ** 2).sum())
    spacing.append(torch.full((n_shell,), math.sqrt(area / max(n_shell, 1))))
    # a wall through the origin, tilted by 30 degrees about the vertical, 4 wide and 2.4 high
    u = 4.0 * (torch.rand(n_wall, generator=g) - 0.5)
    v = 2.4 * (torch.rand(n_wall, generator=g) - 0.5) - 0.4
    ca, sa = math.cos(math.radians(30.0)), math.sin(math.radians(30.0))
    pos.append(torch.stack([u * ca, v, u * sa], 1))
    nrm.append(torch.tensor([-sa, 0.0, ca]).expand(n_wall, 3))
    spacing.append(torch.full((n_wall,), math.sqrt(4.0 * 2.4 / max(n_wall, 1))))
    xyz = torch.cat(pos, 0)
    n = torch.cat(nrm, 0)
    sp = torch.cat(spacing, 0)
    # a fixed shuffle: neighbours in memory are not neighbours in space (a trained model's order is not spatial either)
    perm = torch.randperm(P, generator=g)
    xyz, n, sp = xyz[perm].contiguous(), n[perm], sp[perm]
    # disk frames: thin axis = the jittered normal, tangents rotated at random
    n = torch.nn.functional.normalize(n + 0.1 * torch.randn(P, 3, generator=g), dim=-1)
    helper = torch.where((n[:, 1].abs() < 0.9)[:, None], torch.tensor([0.0, 1.0, 0.0]).expand(P, 3),
                         torch.tensor([1.0, 0.0, 0.0]).expand(P, 3))
    t1 = torch.nn.functional.normalize(torch.cross(helper, n, dim=1), dim=-1)
    t2 = torch.cross(n, t1, dim=1)
    ang = 2 * math.pi * torch.rand(P, generator=g)
    c, s_ = torch.cos(ang)[:, None], torch.sin(ang)[:, None]
    t1, t2 = c * t1 + s_ * t2, -s_ * t1 + c * t2
    rotation = _quat_from_frames(t1, t2, n)
    tang = 0.9 * sp[:, None] * torch.exp(0.3 * torch.randn(P, 2, generator=g))
    scaling = torch.cat([tang, 0.1 * tang.mean(1, keepdim=True)], 1)
    hi = torch.rand(P, generator=g) < 0.65
    opacity = torch.where(hi, 0.85 + 0.14 * torch.rand(P, generator=g), 0.02 + 0.28 * torch.rand(P, generator=g))[:, None]
    base = 0.5 + 0.35 * torch.stack([torch.sin(1.3 * xyz[:, 0] + 0.4), torch.sin(1.7 * xyz[:, 1] + 1.1), torch.sin(0.9 * xyz[:, 2] + 2.3)], 1)
    f_dc = ((base - 0.5) / 0.28209479177387814)[:, None, :] + 0.1 * torch.randn(P, 1, 3, generator=g)
    f_rest = 0.05 * torch.randn(P, M - 1, 3, generator=g)
    return {
        "xyz": xyz,
        "scaling": scaling.contiguous(),
        "rotation": rotation.contiguous(),
        "opacity": opacity.contiguous(),
        "features": torch.cat([f_dc, f_rest], 1).contiguous(),
        "active_sh_degree": sh_degree,
        "bg": torch.zeros(3, dtype=torch.float32),
    }


def seed_gradient(H: int, W: int, seed: int = 0) -> torch.Tensor:
    """dL/dpixels for `loss = (color * G).sum()`, G = randn(3,H,W, seed+1000)/N."""
    g = torch.Generator(device="cpu").manual_seed(seed + 1000)
    return (torch.randn(3, H, W, generator=g) / float(H * W)).contiguous()
