#!/usr/bin/env python
"""render() (operator + fused map post-processing + the python around it) vs the bare operator, fwd+bwd at the C3 size."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd.gaussian_renderer import PipelineParams, SurfelModel, render
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], None, 3, 3)
bg = torch.zeros(3, device=dev); pipe = PipelineParams()
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center, False, False)
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
def full():
    for t in g.values(): t.grad = None
    o = render(cam, pc, pipe, bg)
    (o["render"].sum() + o["rend_dist"].sum() + o["rend_alpha"].sum() + (o["rend_normal"] * o["surf_normal"]).sum()).backward()
def bare():
    for t in g.values(): t.grad = None
    c, r, a = GaussianRasterizer(s)(means3D=g["means3D"], means2D=m2d, shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
    (c.sum() + a.sum()).backward()
for name, fn in (("bare operator + sum losses", bare), ("render() + regulariser-style losses", full)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): fn()
    t1.record(); torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / 10:.3f} ms fwd+bwd")
