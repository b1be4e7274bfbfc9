import math, os, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
lib = _lib.load()
cam = synthetic_camera(W, H); g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
dc, da = [t.to(dev) for t in synthetic_upstream_grads(W, H)]
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                  cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
def step():
    for t in list(g.values()) + [m2d]: t.grad = None
    c, r, a = GaussianRasterizer(s)(means3D=g["means3D"], means2D=m2d, shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
    torch.autograd.backward([c, a], [dc, da])
for timing in (0, 1, 0, 1):
    lib.sr_set_stage_timing(timing)
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30 * 1e3
    print(f"stage timing {timing}: {dt:.4f} ms/step")
    if timing: _lib.stage_stats()
lib.sr_set_stage_timing(0)
