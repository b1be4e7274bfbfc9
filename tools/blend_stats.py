#!/usr/bin/env python
"""Counts what the forward blend actually does on the C3 bench scene (culling efficiency, lane utilisation)."""
import ctypes as C, math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
P, W, H = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000, 1920, 1080
dev = "cuda:0"
lib = _lib.load()
cam = synthetic_camera(W, H); g = {k: v.to(dev) for k, v in synthetic_gaussians(P, W, H).items()}
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                  cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
for cull in (True, False):
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    GaussianRasterizer(s, quadrant_cull=cull, blend_counters=out)(means3D=g["means3D"], means2D=torch.zeros(P, 3, device=dev), shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
    torch.cuda.synchronize()
    st = out.tolist()
    print(f"cull={cull}: staged={st[0]:,} kept={st[1]:,} ({st[1]/max(st[0],1):.1%}) quad_tests={st[2]:,} ({st[2]/max(st[1],1):.2f}/kept) "
          f"tests_with_valid={st[3]:,} ({st[3]/max(st[2],1):.1%}) valid_pairs={st[4]:,} lanes/valid_test={st[4]/max(st[3],1):.1f} lanes/test={st[4]/max(st[2],1):.1f} tests_hitting_rows0-3={st[5]:,} tests_hitting_rows4-7={st[6]:,}")
