#!/usr/bin/env python
"""Fixture kit for whoever holds the CUDA fork of `diff_surfel_rasterization` (the native module the reference imports at
/root/reference/gaussian_renderer/__init__.py:11 and calls at :129-138; an empty submodule in the reference tree, .gitmodules:9-12).

STAND-ALONE: needs numpy, torch with CUDA and the installed fork -- nothing of this repository.  Copy this one file to a CUDA box and run

    python make_cuda_fixtures.py [--out DIR]          # writes DIR/cuda_fork_<scene>.npz (default: the current directory)

then drop the .npz files into `tests/golden/` of this repository.  `tests/test_gpu_cuda_fork_fixtures.py` picks them up:
the HIP kernels AND the CPU oracle are then compared with what the real CUDA rasterizer produced (integers bit-exact, floats to
1e-4 at robust pixels, the per-row gradient bars of tests/gpu_util.py); without the files the test SKIPS and says so.

Every scene is seeded numpy, small enough for the CPU oracle (seconds), and is rendered exactly as the reference renders:
GaussianRasterizationSettings in the keyword order of gaussian_renderer/__init__.py:39-52, GaussianRasterizer(...)(means3D, means2D,
shs | colors_precomp, opacities, scales, rotations, cov3D_precomp=None), backward with seeded upstream gradients on BOTH outputs
(`color` and `allmap`).  Recorded per scene: every input tensor, the settings, the upstream gradients, `color`, `radii`, `allmap`, and the
gradients of means3D, means2D, opacities, scales, rotations and shs / colors_precomp.

Scenes (why each): `small` the seeded benchmark-style scene; `posed` a camera with its own centre, a full rotation and FoVx != FoVy;
`clones` thousands of equal depth keys per tile (stable sort order); `precomp` a colors_precomp call (the semantic passes of the
reference, gaussian_renderer/__init__.py:327-460); `culled` every Gaussian behind the camera or off-screen (empty lists, zero gradients);
`nonunit` un-normalised quaternions (the rasterizer must not normalise: scene/gaussian_model.py does); `truncating` an image size for which
int(focal * tanfov * 2) != W in float32 (backward viewport); `modifier` scale_modifier = 0.7.
"""
import argparse
import math
import os
import sys

import numpy as np


def fov_from_focal(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def projection(znear, zfar, fovx, fovy):
    """The reference's getProjectionMatrix (utils/graphics_utils.py:51-79), as data flow: a symmetric frustum, z_sign = 1."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    Pm = np.zeros((4, 4), np.float64)
    Pm[0, 0] = 2.0 * znear / (2 * right); Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0; Pm[2, 2] = zfar / (zfar - znear); Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def camera(width, height, fx, fy, R=None, centre=None):
    """R: camera -> world rotation, centre: camera position.  Returns what scene/cameras.py:59-71 hands the rasterizer:
    world_view_transform (TRANSPOSED 4x4), full_proj_transform (transposed), camera_center, FoVx, FoVy."""
    R = np.eye(3) if R is None else np.asarray(R, np.float64)
    c = np.zeros(3) if centre is None else np.asarray(centre, np.float64)
    w2c = np.eye(4); w2c[:3, :3] = R.T; w2c[:3, 3] = -R.T @ c
    fovx, fovy = fov_from_focal(fx, width), fov_from_focal(fy, height)
    wvt = w2c.T.astype(np.float32)
    proj = projection(0.01, 100.0, fovx, fovy).T.astype(np.float32)
    full = (wvt.astype(np.float32) @ proj).astype(np.float32)
    centre32 = np.linalg.inv(wvt.astype(np.float64))[3, :3].astype(np.float32)
    return dict(width=width, height=height, fovx=fovx, fovy=fovy, viewmatrix=wvt, projmatrix=full, campos=centre32)


def quat_to_R(q):
    r, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                     [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                     [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]], np.float64)


def gaussians(rng, P, cam, R=None, centre=None, scale_lo=2e-3, scale_hi=3e-2, unit_quaternions=True, sh_coeffs=16):
    """Drawn in the camera's frame (z in 1..50, x / y across 1.1 x the frustum), moved to world coordinates."""
    tx, ty = math.tan(cam["fovx"] / 2), math.tan(cam["fovy"] / 2)
    z = rng.uniform(1.0, 50.0, P)
    x = rng.uniform(-1.1, 1.1, P) * z * tx
    y = rng.uniform(-1.1, 1.1, P) * z * ty
    local = np.stack([x, y, z], 1)
    R = np.eye(3) if R is None else R
    c = np.zeros(3) if centre is None else np.asarray(centre, np.float64)
    means3D = (local @ R.T + c).astype(np.float32)
    scales = (z[:, None] * np.exp(rng.uniform(math.log(scale_lo), math.log(scale_hi), (P, 2)))).astype(np.float32)
    q = rng.normal(size=(P, 4))
    if unit_quaternions:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    else:
        q *= rng.uniform(0.5, 2.0, (P, 1)) / np.linalg.norm(q, axis=1, keepdims=True)
    opacities = (1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, (P, 1))))).astype(np.float32)
    shs = rng.normal(size=(P, sh_coeffs, 3)); shs[:, 1:] *= 0.1
    return dict(means3D=means3D, scales=scales, rotations=q.astype(np.float32), opacities=opacities, shs=shs.astype(np.float32))


def scenes():
    out = []
    W, H = 160, 96
    # small: the benchmark-style camera at the origin
    rng = np.random.default_rng(7)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    out.append(("small", cam, gaussians(rng, 4000, cam), dict(bg=[0.1, 0.2, 0.3], sh_degree=3)))
    # posed: own centre, full rotation, fx != fy
    rng = np.random.default_rng(11)
    q = rng.normal(size=4); q /= np.linalg.norm(q); R = quat_to_R(q); c = rng.uniform(-15, 15, 3)
    cam = camera(176, 112, 0.7 * 176, 0.95 * 176, R, c)
    out.append(("posed", cam, gaussians(rng, 4000, cam, R, c), dict(bg=[0.0, 0.0, 0.0], sh_degree=3)))
    # clones: 40 depth planes x 150 Gaussians at bit-identical depth, all in front of the camera
    rng = np.random.default_rng(13)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    g = gaussians(rng, 6000, cam)
    planes = rng.uniform(2.0, 30.0, 40).astype(np.float32)
    g["means3D"][:, 2] = planes[np.arange(6000) % 40]
    g["means3D"][3000:] = g["means3D"][:3000]           # ... and half of them exact clones of the other half (position only)
    out.append(("clones", cam, g, dict(bg=[0.3, 0.1, 0.0], sh_degree=2)))
    # precomp: colours handed over, no SH
    rng = np.random.default_rng(17)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    g = gaussians(rng, 3000, cam)
    g["colors_precomp"] = rng.uniform(0.0, 1.0, (3000, 3)).astype(np.float32); del g["shs"]
    out.append(("precomp", cam, g, dict(bg=[0.0, 0.0, 0.0], sh_degree=0)))
    # culled: everything behind the camera or far off-screen
    rng = np.random.default_rng(19)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    g = gaussians(rng, 500, cam)
    g["means3D"][:250, 2] *= -1.0
    g["means3D"][250:, 0] += 1.0e4
    out.append(("culled", cam, g, dict(bg=[0.5, 0.5, 0.5], sh_degree=3)))
    # nonunit: quaternion norms 0.5 .. 2
    rng = np.random.default_rng(23)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    out.append(("nonunit", cam, gaussians(rng, 3000, cam, unit_quaternions=False), dict(bg=[0.0, 0.1, 0.0], sh_degree=1)))
    # truncating: first (W, fx) of a small grid whose float32 focal * tanfov * 2 truncates below W
    rng = np.random.default_rng(29)
    pick = None
    for w in range(150, 260):
        for k in range(55, 140, 3):
            fx = np.float32(w * k / 100.0)
            tan = np.float32(math.tan(fov_from_focal(float(fx), w) * 0.5))
            foc = np.float32(w) / (np.float32(2.0) * tan)
            if int(np.float32(foc * tan * np.float32(2.0))) != w:
                pick = (w, float(fx)); break
        if pick:
            break
    if pick:
        w, fx = pick
        cam = camera(w, 100, fx, fx)
        out.append(("truncating", cam, gaussians(rng, 3000, cam), dict(bg=[0.0, 0.0, 0.0], sh_degree=3)))
    # modifier: scale_modifier != 1
    rng = np.random.default_rng(31)
    cam = camera(W, H, 0.8 * W, 0.8 * W)
    out.append(("modifier", cam, gaussians(rng, 3000, cam), dict(bg=[0.2, 0.2, 0.2], sh_degree=3, scale_modifier=0.7)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=".")
    ap.add_argument("--inputs-only", action="store_true", help="write the scenes without rendering them (no fork needed: to inspect what would be rendered)")
    ap.add_argument("--self-test-with-drop-in", action="store_true",
                    help="render with WHATEVER diff_surfel_rasterization is importable (the MI355X repository's own test of this kit: files are "
                         "named selftest_*.npz and pin nothing)")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if not a.inputs_only:
        import torch
        assert torch.cuda.is_available(), "needs a GPU"
        assert a.self_test_with_drop_in or torch.version.cuda is not None, "this script renders with the CUDA fork: it needs a CUDA build of torch"
        import diff_surfel_rasterization as dsr
        native = getattr(getattr(dsr, "_C", None), "__file__", "") or ""
        assert a.self_test_with_drop_in or native.endswith((".so", ".pyd")), ("`diff_surfel_rasterization._C` is not a compiled extension (%r): this must be the CUDA fork, not the "
                                                 "drop-in package of the MI355X repository" % native)
        from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    for name, cam, g, opt in scenes():
        rng = np.random.default_rng(1000 + len(name))
        H, W = cam["height"], cam["width"]
        dL_dcolor = rng.normal(size=(3, H, W)).astype(np.float32)
        dL_dallmap = rng.normal(size=(7, H, W)).astype(np.float32)
        rec = dict(g)
        rec.update(viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"], bg=np.asarray(opt["bg"], np.float32),
                   tanfovx=np.float64(math.tan(cam["fovx"] * 0.5)), tanfovy=np.float64(math.tan(cam["fovy"] * 0.5)),
                   image_width=np.int64(W), image_height=np.int64(H), sh_degree=np.int64(opt["sh_degree"]),
                   scale_modifier=np.float64(opt.get("scale_modifier", 1.0)), dL_dcolor=dL_dcolor, dL_dallmap=dL_dallmap)
        if not a.inputs_only:
            dev = "cuda"
            t = {k: torch.tensor(v, device=dev).requires_grad_() for k, v in g.items()}
            means2D = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
            settings = GaussianRasterizationSettings(
                image_height=int(H), image_width=int(W), tanfovx=float(rec["tanfovx"]), tanfovy=float(rec["tanfovy"]),
                bg=torch.tensor(rec["bg"], device=dev), scale_modifier=float(rec["scale_modifier"]),
                viewmatrix=torch.tensor(cam["viewmatrix"], device=dev), projmatrix=torch.tensor(cam["projmatrix"], device=dev),
                sh_degree=int(opt["sh_degree"]), campos=torch.tensor(cam["campos"], device=dev), prefiltered=False, debug=False)
            color, radii, allmap = GaussianRasterizer(raster_settings=settings)(
                means3D=t["means3D"], means2D=means2D, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"), opacities=t["opacities"],
                scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
            ((color * torch.tensor(dL_dcolor, device=dev)).sum() + (allmap * torch.tensor(dL_dallmap, device=dev)).sum()).backward()
            torch.cuda.synchronize()
            rec.update(out_color=color.detach().cpu().numpy(), out_radii=radii.cpu().numpy().astype(np.int32), out_allmap=allmap.detach().cpu().numpy())
            for k, v in list(t.items()) + [("means2D", means2D)]:
                rec["grad_" + k] = (torch.zeros_like(v) if v.grad is None else v.grad).cpu().numpy()
            rec["fork"] = np.array([native, torch.__version__, str(torch.version.cuda), torch.cuda.get_device_name(0)])
        path = os.path.join(a.out, ("selftest_%s.npz" if a.self_test_with_drop_in else "cuda_fork_%s.npz" if not a.inputs_only else "cuda_fork_inputs_%s.npz") % name)
        np.savez_compressed(path, **rec)
        print("wrote", path, "P =", g["means3D"].shape[0], "%dx%d" % (W, H), file=sys.stderr)


if __name__ == "__main__":
    main()
