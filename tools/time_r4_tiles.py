#!/usr/bin/env python
"""The reference's documented operating point (`-r 4`: 480x320 frames [REF README.md:195-207]) by tile shape: fwd+bwd ms and per-stage ms.
python tools/time_r4_tiles.py [P] [W] [H]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
P, W, H = (int(x) for x in (sys.argv[1:4] + ["1500000", "480", "320"][len(sys.argv) - 1:]))
dev = "cuda:0"; lib = _lib.load()
cam = synthetic_camera(W, H); g = synthetic_gaussians(P, W, H, seed=0)
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                  cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
dc, da = (x.to(dev) for x in synthetic_upstream_grads(W, H, seed=1))
for tile, bk in [(None, "one_wave"), (None, "coop"), (None, None), ((8, 8), None), ((16, 8), None), ((32, 8), None), ((32, 16), None)]:
    r = GaussianRasterizer(s, tile=tile, backward_kernel=bk)
    def step():
        for v in list(t.values()) + [m2]: v.grad = None
        c, radii, am = r(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, am], [dc, da])
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3 / 20
    lib.sr_set_stage_timing(1)
    for _ in range(3): step()
    torch.cuda.synchronize(); st = _lib.stage_stats(); lib.sr_set_stage_timing(0)
    print(tile or (16, 16), bk or "", f"{ms:.3f} ms/step", {k: round(v / max(n, 1), 3) for k, (v, n) in st.items() if n}, flush=True)
