#!/usr/bin/env python
"""The fused training-step pattern alone (render_and_semantic + render_class_distortions, fwd + bwd: streetunveiler_amd/train_pattern.py
fused_pattern -- the two rasterizations that stand for the reference's eight of a late iteration [REF /root/reference/train.py:84-109]) on the
C3 scene, N iterations: the command tools/profile_train_step.sh puts under rocprofv3.  Prints ms per iteration (HIP events).
    python tools/train_step_bench.py [--steps 6] [--warmup 2] [--tile 16 16]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streetunveiler_amd.gaussian_renderer import SurfelModel
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
from streetunveiler_amd.train_pattern import fused_pattern, make_weights

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--gaussians", type=int, default=3_000_000); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
a = ap.parse_args()
P, W, H, dev = a.gaussians, a.width, a.height, "cuda:0"
cam = synthetic_camera(W, H).to(dev)
g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(0)).to(dev); sem[sem == 4] = 2   # (the reference prunes the sky Gaussians)
pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], sem, 3, 3)
w = make_weights(H, W, dev); bg = torch.zeros(3, device=dev)


def step():
    for t in g.values():
        t.grad = None
    fused_pattern(cam, pc, bg, w)["loss"].backward()


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    step()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"train_step_fused_2_calls_ms": round(e0.elapsed_time(e1) / a.steps, 4), "steps": a.steps, "gaussians": P, "width": W, "height": H}))
