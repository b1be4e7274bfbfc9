#!/usr/bin/env python
"""one training iteration's render() + render_semantic() of the same view -- the reference's three
rasterizations (rgb + 2 x 3 class channels), our two (rgb + one 6-channel pass), and ONE 9-channel pass -- forward+backward at the
C3 size.  python tools/time_rgb_and_semantic.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streetunveiler_amd.gaussian_renderer import PipelineParams, SurfelModel, render, render_and_semantic, render_semantic
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(1)).to(dev)
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], sem, 3, 3)
bg = torch.zeros(3, device=dev); pipe = PipelineParams()
def two_calls():
    for t in g.values(): t.grad = None
    a = render(cam, pc, pipe, bg); b = render_semantic(cam, pc, pipe, bg)
    (a["render"].sum() + a["rend_dist"].sum() + b["render_semantics"].sum()).backward()
def one_call():
    for t in g.values(): t.grad = None
    o = render_and_semantic(cam, pc, pipe, bg)
    (o["render"].sum() + o["rend_dist"].sum() + o["render_semantics"].sum()).backward()
for name, fn in (("render() + render_semantic() (rgb pass + one 6-channel pass)", two_calls), ("render_and_semantic() (one 9-channel pass)", one_call)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): fn()
    t1.record(); torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / 10:.3f} ms fwd+bwd")
