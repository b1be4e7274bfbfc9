import math, os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
lib = _lib.load()
cam = synthetic_camera(W, H); g = {k: v.to(dev) for k, v in synthetic_gaussians(P, W, H).items()}
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                  cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
m2d = torch.zeros(P, 3, device=dev)
def step():
    with torch.no_grad():
        GaussianRasterizer(s)(means3D=g["means3D"], means2D=m2d, shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
for _ in range(3): step()
torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
for _ in range(10): step()
torch.cuda.synchronize()
print({k: round(ms / n, 4) for k, (ms, n) in _lib.stage_stats().items() if n})
