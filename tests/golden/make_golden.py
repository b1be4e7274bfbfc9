"""Generates the committed golden fixtures.  Run HERE (build container) only:

    python tests/golden/make_golden.py

G1/G2 import the reference's own python (read-only, from /root/reference) and
record inputs + outputs as data; nothing of the reference travels with the
repo.  G3/G4 freeze the CPU oracle's per-stage outputs on small seeded scenes
(regression pins for the oracle itself; gradients there are cross-checked
against float64 autograd when the fixture is made).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def g1_g2_from_reference():
    sys.path.insert(0, "/root/reference")
    from utils.sh_utils import eval_sh, RGB2SH, SH2RGB                     # noqa: E402
    from utils.graphics_utils import getProjectionMatrix, getWorld2View2, focal2fov, fov2focal  # noqa: E402
    sys.path.pop(0)

    # G1: eval_sh for deg 0..3 (reference layout: sh [..., C, coeffs], dirs [..., 3])
    g = torch.Generator().manual_seed(0)
    sh = torch.randn(64, 3, 16, generator=g)
    dirs = torch.randn(64, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    out = {"sh": sh.numpy(), "dirs": dirs.numpy()}
    for deg in range(4):
        out[f"rgb_deg{deg}"] = eval_sh(deg, sh, dirs).numpy()
        out[f"color_deg{deg}"] = torch.clamp_min(eval_sh(deg, sh, dirs) + 0.5, 0.0).numpy()  # gaussian_renderer:81-82
    rgb = torch.rand(16, 3, generator=g)
    out["rgb2sh_in"] = rgb.numpy(); out["rgb2sh_out"] = RGB2SH(rgb).numpy(); out["sh2rgb_out"] = SH2RGB(RGB2SH(rgb)).numpy()
    np.savez_compressed(os.path.join(HERE, "sh_golden.npz"), **out)

    # G2: camera matrices exactly as scene/cameras.py:59-71 builds them
    from streetunveiler_amd.camera import yaw_rotation
    cams = {}
    k = 0
    for (W, H) in [(256, 256), (1920, 1080), (3840, 2160), (48, 40)]:
        for yaw, tvec in [(0.0, (0, 0, 0)), (-17.5, (0, 0, 0)), (7.5, (0.3, -0.2, 1.5))]:
            fx = fy = 0.8 * W
            fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
            R = yaw_rotation(yaw); t = np.array(tvec, dtype=np.float64)
            w2c = getWorld2View2(R, t, np.array([0.0, 0.0, 0.0]), 1.0)
            wvt = torch.tensor(w2c).transpose(0, 1)
            proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
            full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
            center = wvt.inverse()[3, :3]
            cams[f"c{k}_meta"] = np.array([W, H, yaw, *tvec, fovx, fovy, fov2focal(fovx, W)], dtype=np.float64)
            cams[f"c{k}_R"] = R; cams[f"c{k}_t"] = t
            cams[f"c{k}_wvt"] = wvt.numpy(); cams[f"c{k}_proj"] = proj.numpy()
            cams[f"c{k}_full"] = full.numpy(); cams[f"c{k}_center"] = center.numpy()
            k += 1
    cams["n"] = np.array(k)
    np.savez_compressed(os.path.join(HERE, "camera_golden.npz"), **cams)


def g3_g4_from_oracle():
    from oracle import surfel_oracle as so
    from oracle.torch64 import forward_backward64
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads

    # G3: 64 Gaussians, 32x32, SH deg 3, every stage + all gradients
    W = H = 32
    cam = synthetic_camera(W, H, index=2)
    g = synthetic_gaussians(64, W, H, seed=3, scale_lo=0.01, scale_hi=0.15)
    bg = np.array([0.3, 0.1, 0.7], np.float32)
    fwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                               shs=g["shs"].numpy(), viewmatrix=cam.world_view_transform.numpy(),
                               projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=bg,
                               image_width=W, image_height=H, sh_degree=3)
    dc, da = synthetic_upstream_grads(W, H)
    grads = so.rasterize_backward(fwd, dc.numpy(), da.numpy())
    _, g64 = forward_backward64(fwd, dc.numpy(), da.numpy())
    for k, v in g64.items():
        err = np.abs(grads[k] - v).max() / (np.abs(v).max() + 1e-30)
        assert err < 2e-4, (k, err)
    out = {f"in_{k}": v.numpy() for k, v in g.items()}
    out.update(in_view=cam.world_view_transform.numpy(), in_proj=cam.full_proj_transform.numpy(),
               in_campos=cam.camera_center.numpy(), in_bg=bg, in_dL_dcolor=dc.numpy(), in_dL_dallmap=da.numpy())
    for k in ["radii", "means2D", "depths", "transMat", "normal_opacity", "rgb", "clamped", "tiles_touched", "rect",
              "keys", "point_list", "ranges", "color", "allmap", "final_T", "n_contrib"]:
        out[f"fwd_{k}"] = fwd[k]
    out["fwd_num_rendered"] = np.array(fwd["num_rendered"])
    for k, v in grads.items():
        out[f"bwd_{k}"] = v
    for k, v in g64.items():
        out[f"bwd64_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"), **out)

    # G4: BASELINE config C1 (10k Gaussians, 256x256, SH deg 0, forward only): checksums
    W = H = 256
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(10000, W, H, seed=0)
    fwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(),
                               shs=g["shs"].numpy(), viewmatrix=cam.world_view_transform.numpy(),
                               projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
                               bg=np.zeros(3, np.float32), image_width=W, image_height=H, sh_degree=0)
    np.savez_compressed(
        os.path.join(HERE, "c1_checksums.npz"),
        num_rendered=np.array(fwd["num_rendered"]), radii_sum=np.array(fwd["radii"].astype(np.int64).sum()),
        visible=np.array((fwd["radii"] > 0).sum()), tiles_sum=np.array(fwd["tiles_touched"].astype(np.int64).sum()),
        point_list_crc=np.array(int(np.bitwise_xor.reduce(fwd["point_list"].astype(np.uint64) * np.arange(1, fwd["num_rendered"] + 1, dtype=np.uint64)))),
        color_sum=fwd["color"].astype(np.float64).sum(axis=(1, 2)), allmap_sum=fwd["allmap"].astype(np.float64).sum(axis=(1, 2)),
        color_probe=fwd["color"][:, ::37, ::41].copy(), allmap_probe=fwd["allmap"][:, ::37, ::41].copy(),
        radii_probe=fwd["radii"][::97].copy())


if __name__ == "__main__":
    g1_g2_from_reference()
    g3_g4_from_oracle()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
