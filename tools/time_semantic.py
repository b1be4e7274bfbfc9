#!/usr/bin/env python
"""render_semantic's 6 class channels as ONE 6-channel pass vs the reference's two 3-channel passes,
forward+backward at the C3 size (3M Gaussians, 1920x1080).  python tools/time_semantic.py [steps]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cam = synthetic_camera(W, H)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items() if k != "shs"}
sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(1)).to(dev)
onehot = (sem.view(-1, 1) == torch.arange(6, device=dev).view(1, -1)).float()
dc3, da = [t.to(dev) for t in synthetic_upstream_grads(W, H)]
dc6 = torch.cat([dc3, dc3.flip(0)], 0).contiguous()
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
bg6 = torch.tensor([0, 0, 0, 0, 1.0, 0], device=dev)
def settings(bg):
    return GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform.to(dev),
                                         cam.full_proj_transform.to(dev), 0, cam.camera_center.to(dev), False, False)
def call(bg, colors, dc, with_allmap):
    c, r, a = GaussianRasterizer(settings(bg))(means3D=g["means3D"], means2D=m2d, colors_precomp=colors, opacities=g["opacities"],
                                               scales=g["scales"], rotations=g["rotations"])
    if with_allmap: torch.autograd.backward([c, a], [dc, da])
    else: torch.autograd.backward([c], [dc])
def one_pass():
    call(bg6, onehot, dc6, True)
def two_passes():
    call(bg6[:3].contiguous(), onehot[:, :3].contiguous(), dc6[:3], True)
    call(bg6[3:].contiguous(), onehot[:, 3:].contiguous(), dc6[3:], False)
for name, fn in (("two 3-channel passes (reference call pattern)", two_passes), ("one 6-channel pass", one_pass)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps): fn()
    t1.record(); torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / steps:.3f} ms per fwd+bwd")
