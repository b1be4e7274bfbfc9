"""Camera matrices in the conventions the rasterizer operator expects.

Restates (does not import) the reference recipe:
  * getWorld2View2 / getProjectionMatrix  [REF /root/reference/utils/graphics_utils.py:38-79]
  * Camera.world_view_transform = W2C^T, full_proj_transform = (P @ W2C)^T,
    camera_center = inverse(world_view_transform)[3,:3], znear=0.01, zfar=100
    [REF /root/reference/scene/cameras.py:53-71]
Pinned against the reference functions by tests/golden/camera_golden.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

ZNEAR = 0.01
ZFAR = 100.0


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """W2C 4x4 (float32) from a C2W rotation `R` and W2C translation `t` (reference convention)."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    # the reference round-trips through C2W (translate=0, scale=1): keep the same float path
    Rt = np.linalg.inv(np.linalg.inv(Rt))
    return Rt.astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """Perspective matrix with clip.w = view z and depth mapped to [0, 1]."""
    tan_y = math.tan(fovy / 2)
    tan_x = math.tan(fovx / 2)
    top, right = tan_y * znear, tan_x * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def focal2fov(focal: float, pixels: int) -> float:
    return 2 * math.atan(pixels / (2 * focal))


@dataclass
class SimpleCamera:
    """The subset of scene.cameras.Camera that gaussian_renderer.render reads
    [REF /root/reference/gaussian_renderer/__init__.py:36-49; utils/point_utils.py:10-14]."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    full_proj_transform: torch.Tensor   # [4,4] = (P @ W2C)^T
    camera_center: torch.Tensor         # [3]
    znear: float = ZNEAR
    zfar: float = ZFAR

    def to(self, device) -> "SimpleCamera":
        return SimpleCamera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                            self.world_view_transform.to(device), self.full_proj_transform.to(device),
                            self.camera_center.to(device), self.znear, self.zfar)


def make_camera(width: int, height: int, fovx: float, fovy: float, R: np.ndarray = None,
                t: np.ndarray = None) -> SimpleCamera:
    R = np.eye(3) if R is None else R
    t = np.zeros(3) if t is None else t
    wvt_t = torch.tensor(world_to_view(R, t)).transpose(0, 1)   # same (strided) tensor the reference inverts
    proj = projection_matrix(ZNEAR, ZFAR, fovx, fovy).transpose(0, 1)
    full = (wvt_t.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt_t.inverse()[3, :3].contiguous()
    wvt = wvt_t.contiguous()
    return SimpleCamera(int(width), int(height), float(fovx), float(fovy), wvt, full, center)


def yaw_rotation(deg: float) -> np.ndarray:
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)
