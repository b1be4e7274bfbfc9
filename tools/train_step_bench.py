#!/usr/bin/env python
"""The training-step pattern alone -- what stands for the reference's eight rasterizations of a late iteration [REF /root/reference/train.py:84-109]:
render_train_view (9-channel render + per-class distortion pass on ONE plan; streetunveiler_amd/train_pattern.py one_plan_pattern), or with
--pattern fused the two separate rasterizations -- fwd + bwd on the C3 scene, N iterations: the command tools/profile_train_step.sh puts
under rocprofv3.  Prints ms per iteration (HIP events).
    python tools/train_step_bench.py [--steps 6] [--warmup 2] [--pattern one_plan|fused]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from streetunveiler_amd.gaussian_renderer import SurfelModel
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
from streetunveiler_amd.train_pattern import fused_pattern, make_weights, one_plan_pattern

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=6); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--pattern", choices=["one_plan", "fused"], default="one_plan", help="one_plan: render_train_view (one K1 / binning / K8); fused: the two rasterizations")
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
    (one_plan_pattern if a.pattern == "one_plan" else fused_pattern)(cam, pc, bg, w)["loss"].backward()


for _ in range(a.warmup):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.steps):
    step()
e1.record(); torch.cuda.synchronize()
print(json.dumps({"train_step_ms": round(e0.elapsed_time(e1) / a.steps, 4), "pattern": a.pattern, "steps": a.steps, "gaussians": P, "width": W, "height": H}))
