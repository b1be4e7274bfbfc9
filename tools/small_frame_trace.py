#!/usr/bin/env python
"""The forward of a SMALL frame (10 k Gaussians, 256x256, no_grad) N times: the command to put under `rocprofv3 --kernel-trace --stats` to see
the launch chain whose length is that frame's floor (tools/notes_round6_measured.md).  Prints us per forward (wall clock around a synchronize).
    python tools/small_frame_trace.py [--gaussians 10000] [--size 256] [--iters 200] [--capacity]"""
import argparse, json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import _C
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians

ap = argparse.ArgumentParser()
ap.add_argument("--gaussians", type=int, default=10_000); ap.add_argument("--size", type=int, default=256); ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--capacity", action="store_true", help="SR_FLAG_BINNING_CAPACITY: no host read-back")
a = ap.parse_args()
dev, P, S = "cuda:0", a.gaussians, a.size
cam = synthetic_camera(S, S); g = {k: v.to(dev) for k, v in synthetic_gaussians(P, S, S, seed=0).items()}
e = torch.empty(0, device=dev)
args = (torch.zeros(3, device=dev), g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev),
        math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), S, S, g["shs"], 3, cam.camera_center.to(dev), False, False)
with torch.no_grad():
    D = int(_C.rasterize_gaussians(*args)[0])
    kw = dict(forward_only=True, binning_capacity=int(1.5 * D) + 1024) if a.capacity else dict(forward_only=True)
    for _ in range(10):
        _C.rasterize_gaussians(*args, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.iters):
        _C.rasterize_gaussians(*args, **kw)
    torch.cuda.synchronize()
print(json.dumps({"gaussians": P, "size": S, "duplicates": D, "capacity": a.capacity, "us_per_forward": round((time.perf_counter() - t0) * 1e6 / a.iters, 1)}))
