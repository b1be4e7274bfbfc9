#!/usr/bin/env python
"""Does the number of tiles relative to the chip's wave slots matter?  K6 / K7 run one (two) single-wave workgroups per tile; the chip holds
3 072 K7 waves (3 per SIMD) and 6 144 K6 waves at a time, i.e. 2.66 "rounds" at 1920x1080.  Frames of different heights at the SAME
Gaussian density (P ~ H): blend time per tile against tiles / 3 072.   python tools/tail_quantisation.py -> gpurun_out/tail_quantisation.json"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
dev, W = "cuda:0", 1920
lib = _lib.load()
rows = []
for H in [int(a) for a in sys.argv[1:]] or [816, 976, 1024, 1080, 1152, 1216, 1232, 1312, 1440, 1632]:
    P = int(round(3_000_000 * H / 1080))
    cam = synthetic_camera(W, H); g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
    dc, da = [t.to(dev) for t in synthetic_upstream_grads(W, H)]
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    def step():
        for t in list(g.values()) + [m2d]: t.grad = None
        c, r, a = GaussianRasterizer(s)(means3D=g["means3D"], means2D=m2d, shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
        torch.autograd.backward([c, a], [dc, da])
    for _ in range(3): step()
    torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
    for _ in range(10): step()
    torch.cuda.synchronize()
    st = {k: ms / n for k, (ms, n) in _lib.stage_stats().items() if n}; lib.sr_set_stage_timing(0)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    row = dict(H=H, P=P, tiles=tiles, k7_rounds=round(tiles / 3072, 3), blend_fwd_ms=round(st["blend_fwd"], 4), blend_bwd_ms=round(st["blend_bwd"], 4),
               fwd_ns_per_tile=round(st["blend_fwd"] * 1e6 / tiles, 1), bwd_ns_per_tile=round(st["blend_bwd"] * 1e6 / tiles, 1))
    rows.append(row); print(json.dumps(row), flush=True)
    del g, m2d, dc, da
    torch.cuda.empty_cache()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(what=__doc__, rows=rows), open("gpurun_out/tail_quantisation.json", "w"), indent=1)
