#!/usr/bin/env python
"""The reference's per-class distortion renders [REF train.py:94-103] at the C3 size: five class-filtered render() calls
(boolean-indexed, and with the mask handed to the operator) vs render_class_distortions (one pass)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streetunveiler_amd.gaussian_renderer import PipelineParams, SurfelModel, render, render_class_distortions
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H, dev = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 1920, 1080, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(0)).to(dev)
sem[sem == 4] = 2   # the reference prunes the sky Gaussians before training
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], sem, 3, 3)
bg = torch.zeros(3, device=dev)
classes = [0, 1, 2, 3, 5]
def zero():
    for t in g.values(): t.grad = None
def loop(pipe):
    zero(); loss = 0
    for k in classes:
        loss = loss + render(cam, pc, pipe, bg, semantic_filter_bit=1 << k, reverse_semantic=True)["rend_dist"].mean()
    loss.backward()
def one():
    zero(); render_class_distortions(cam, pc, PipelineParams(), bg, classes)["rend_dist"].mean(dim=(1, 2, 3)).sum().backward()
for name, fn in [("five render() calls, boolean-indexed inputs (reference pattern)", lambda: loop(PipelineParams())),
                 ("five render() calls, mask inside the operator", lambda: loop(PipelineParams(fused_mask=True))),
                 ("render_class_distortions (one pass)", one)]:
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms fwd+bwd")
