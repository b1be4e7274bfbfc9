#!/usr/bin/env python
"""render_with_mask with boolean-indexed inputs (the reference's way) vs the mask inside the operator, fwd+bwd
at the C3 size with 60 % of the Gaussians kept.  python tools/time_mask.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streetunveiler_amd.gaussian_renderer import PipelineParams, SurfelModel, render_with_mask
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], None, 3, 3)
m = (torch.rand(P, generator=torch.Generator().manual_seed(0)) < 0.6).to(dev)
bg = torch.zeros(3, device=dev)
def step(fused):
    for t in g.values(): t.grad = None
    out = render_with_mask(cam, pc, PipelineParams(fused_mask=fused), bg, m)
    (out["render"].sum() + out["rend_dist"].sum()).backward()
for name, fused in (("boolean-indexed inputs (reference call pattern)", False), ("mask inside the operator", True)):
    for _ in range(3): step(fused)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): step(fused)
    t1.record(); torch.cuda.synchronize()
    print(f"{name}: {t0.elapsed_time(t1) / 10:.3f} ms per render_with_mask fwd+bwd")
