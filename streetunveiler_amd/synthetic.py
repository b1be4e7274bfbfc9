"""Seeded synthetic surfel scenes for the benchmark and the parity tests.

Camera at the origin looking +z, fx = fy = 0.8*W; Gaussians: z ~ U(1,50),
x,y ~ U(-1.1,1.1)*z*tan(FoV/2), per-axis scale z*exp(U(log 5e-4, log 5e-3)),
random unit quaternions, opacity sigmoid(N(0,1.5^2)), SH DC ~ N(0,1), rest
~ N(0,0.1^2), background 0.  Everything is drawn on the CPU with
torch.Generator().manual_seed(seed) and moved afterwards, so the CPU oracle
and the GPU path see identical bits.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .camera import SimpleCamera, focal2fov, make_camera, yaw_rotation


def synthetic_camera(width: int, height: int, index: int = None, n_cams: int = 8) -> SimpleCamera:
    fx = fy = 0.8 * width
    fovx, fovy = focal2fov(fx, width), focal2fov(fy, height)
    if index is None:
        return make_camera(width, height, fovx, fovy)
    # 8-camera batch: camera k yawed by (k - 3.5) * 5 degrees, same Gaussians
    R = yaw_rotation((index - (n_cams - 1) / 2.0) * 5.0)
    return make_camera(width, height, fovx, fovy, R=R)


def synthetic_gaussians(P: int, width: int, height: int, seed: int = 0, sh_coeffs: int = 16,
                        scale_lo: float = 5e-4, scale_hi: float = 5e-3) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    cam = synthetic_camera(width, height)
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    z = torch.rand(P, generator=g) * 49.0 + 1.0
    x = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * tx
    y = (torch.rand(P, generator=g) * 2.2 - 1.1) * z * ty
    means3D = torch.stack([x, y, z], dim=1).contiguous()
    lo, hi = math.log(scale_lo), math.log(scale_hi)
    scales = (z[:, None] * torch.exp(torch.rand(P, 2, generator=g) * (hi - lo) + lo)).contiguous()
    q = torch.randn(P, 4, generator=g)
    rotations = (q / q.norm(dim=1, keepdim=True)).contiguous()
    opacities = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5).contiguous()
    shs = torch.randn(P, sh_coeffs, 3, generator=g)
    shs[:, 1:] *= 0.1
    return dict(means3D=means3D.float(), scales=scales.float(), rotations=rotations.float(),
                opacities=opacities.float(), shs=shs.float().contiguous())


def posed_scene(P: int, width: int, height: int, seed: int = 0, scale_lo: float = 5e-4, scale_hi: float = 5e-3, spread: float = 10.0,
                sh_coeffs: int = 16, near_third: bool = False, behind_fraction: float = 0.0, focal_range=(0.55, 1.4)):
    """The benchmark scene seen by a camera in GENERAL position: a random rotation (any yaw / pitch / roll), a centre drawn from
    U(-spread, spread)^3, fx in [0.55, 1.4] W and fy = fx * U(0.8, 1.25) (so FoVx and FoVy are unrelated), the Gaussians drawn in that
    camera's frame exactly as `synthetic_gaussians` draws them and moved to world coordinates (float64, then rounded).  The benchmark
    cameras of SURVEY 8d all sit at the origin and only yaw: with them `campos`, the translation row of the view matrix and two of
    the three rotation axes are zeros / ones that a wrong term could hide behind.  -> (camera, gaussians)"""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g, dtype=torch.float64); q = (q / q.norm()).tolist()
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]], dtype=np.float64)     # camera -> world
    c = ((torch.rand(3, generator=g, dtype=torch.float64) * 2 - 1) * spread).numpy()
    fx = float(torch.rand(1, generator=g) * (focal_range[1] - focal_range[0]) + focal_range[0]) * width   # (0.55 .. 1.4 W: FoVx 85 .. 39 degrees)
    fy = fx * float(torch.rand(1, generator=g) * 0.45 + 0.8)
    cam = make_camera(width, height, focal2fov(fx, width), focal2fov(fy, height), R=R, t=-R.T @ c)
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    zc = torch.rand(P, generator=g) * 49.0 + 1.0
    xc = (torch.rand(P, generator=g) * 2.2 - 1.1) * zc * tx
    yc = (torch.rand(P, generator=g) * 2.2 - 1.1) * zc * ty
    if near_third:   # a third of the Gaussians around / behind the near plane (view depth -0.1 .. 0.4)
        zc = zc.clone(); zc[: P // 3] = torch.rand(P // 3, generator=g) * 0.5 - 0.1
    if behind_fraction > 0:   # the camera INSIDE the cloud, as on a street: the last `behind_fraction` of the Gaussians mirrored behind it
        nb = int(P * behind_fraction)
        if nb: zc = zc.clone(); zc[P - nb:] = -zc[P - nb:]
    local = torch.stack([xc, yc, zc], dim=1).double()
    means3D = (local @ torch.tensor(R).t() + torch.tensor(c)).float().contiguous()
    lo, hi = math.log(scale_lo), math.log(scale_hi)
    scales = (zc[:, None] * torch.exp(torch.rand(P, 2, generator=g) * (hi - lo) + lo)).contiguous()
    qq = torch.randn(P, 4, generator=g)
    rotations = (qq / qq.norm(dim=1, keepdim=True)).contiguous()
    opacities = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5).contiguous()
    shs = torch.randn(P, sh_coeffs, 3, generator=g)
    shs[:, 1:] *= 0.1
    return cam, dict(means3D=means3D, scales=scales.float(), rotations=rotations.float(), opacities=opacities.float(), shs=shs.float().contiguous())


def posed_rig(P: int, width: int, height: int, n_cams: int, seed: int = 0, scale_lo: float = 5e-4, scale_hi: float = 5e-3, spread: float = 10.0,
              jitter: float = 2.0, behind_fraction: float = 0.0):
    """`n_cams` cameras in general position looking at ONE set of Gaussians (a multi-camera step: streetunveiler_amd.parallel): camera 0
    and the Gaussians are `posed_scene(seed)`; camera k > 0 is camera 0 moved by up to `jitter` units sideways / backwards and turned by
    up to ~6 degrees about each axis -- every camera has its own centre (the yawed benchmark batch shares one: the origin), which is what
    the per-view directions of the factored SH-gradient exchange depend on.  -> (cameras, gaussians)"""
    import numpy as np
    cam0, g = posed_scene(P, width, height, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi, spread=spread, behind_fraction=behind_fraction)
    W2C = cam0.world_view_transform.t().double().numpy()          # [R^T | t]
    R0, c0 = W2C[:3, :3].T, cam0.camera_center.double().numpy()
    gen = torch.Generator().manual_seed(seed + 7919)
    cams = [cam0]
    for _ in range(1, n_cams):
        a = ((torch.rand(3, generator=gen, dtype=torch.float64) * 2 - 1) * 0.1).tolist()
        cx, sx, cy, sy, cz, sz = math.cos(a[0]), math.sin(a[0]), math.cos(a[1]), math.sin(a[1]), math.cos(a[2]), math.sin(a[2])
        Rs = (np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]))
        off = ((torch.rand(3, generator=gen, dtype=torch.float64) * 2 - 1) * jitter).numpy()
        off[2] = -abs(off[2])                                       # backwards, never into the scene
        R, c = R0 @ Rs, c0 + R0 @ off
        cams.append(make_camera(width, height, cam0.FoVx, cam0.FoVy, R=R, t=-R.T @ c))
    return cams, g


def clustered_gaussians(P: int, width: int, height: int, fraction: float = 0.5, seed: int = 0) -> Dict[str, torch.Tensor]:
    """A street-like, NON-uniform variant of the benchmark scene: `fraction` of the Gaussians is squeezed into four screen regions
    (dense facades / vegetation) and made translucent, so the tile lists are heavy-tailed (1920x1080, 3 M Gaussians: list length
    p50 ~830, p99 ~27 k, max ~41 k against a uniform ~1 640) and walked deep.  One blend wave per tile makes the longest lists the
    tail of K6 / K7: tools/clustered_scene.py times it, tests/test_gpu_fullsize.py checks it against the oracle."""
    g = synthetic_gaussians(P, width, height, seed=seed)
    cam = synthetic_camera(width, height)
    gen = torch.Generator().manual_seed(9)
    n = int(P * fraction)
    tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    centres = torch.tensor([[-0.6, 0.1], [0.1, -0.3], [0.55, 0.4], [0.8, -0.5]])
    which = torch.randint(0, 4, (n,), generator=gen)
    z = g["means3D"][:n, 2]
    off = torch.randn(n, 2, generator=gen) * 0.06
    g["means3D"][:n, 0] = (centres[which, 0] + off[:, 0]) * z * tx
    g["means3D"][:n, 1] = (centres[which, 1] + off[:, 1]) * z * ty
    g["opacities"][:n] *= 0.3      # translucent clutter: lists are walked deep
    return g


def synthetic_upstream_grads(width: int, height: int, seed: int = 1, aux: bool = True):
    """dL_dcolor ~ N(0,1)[3,H,W], dL_dallmap ~ N(0,1)[7,H,W]; aux=False keeps colour + alpha only (C2)."""
    g = torch.Generator().manual_seed(seed)
    dcolor = torch.randn(3, height, width, generator=g)
    dallmap = torch.randn(7, height, width, generator=g)
    if not aux:
        keep = torch.zeros(7, 1, 1)
        keep[1] = 1.0
        dallmap = dallmap * keep
    return dcolor.contiguous(), dallmap.contiguous()
