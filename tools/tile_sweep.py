#!/usr/bin/env python
"""BASELINE config 5's tile-size sweep on one GPU: stage times of one fwd+bwd per tile shape.
python tools/tile_sweep.py [gaussians width height]   (default: the C5 scene, 6 M Gaussians at 3840x2160)"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
P, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (6_000_000, 3840, 2160)
dev = "cuda:0"
lib = _lib.load()
cam = synthetic_camera(W, H); g = {k: v.to(dev).requires_grad_() for k, v in synthetic_gaussians(P, W, H).items()}
dc, da = [t.to(dev) for t in synthetic_upstream_grads(W, H)]
s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                  cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
rows = []
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ.get("SR_TILES", "8x8,16x8,16x16,32x8,32x16").split(",")]
for tile in shapes:
    def step():
        for t in list(g.values()) + [m2d]: t.grad = None
        c, r, a = GaussianRasterizer(s, tile=tile)(means3D=g["means3D"], means2D=m2d, shs=g["shs"], opacities=g["opacities"], scales=g["scales"], rotations=g["rotations"])
        torch.autograd.backward([c, a], [dc, da])
    e = torch.empty(0, device=dev)
    with torch.no_grad():
        D = _C.rasterize_gaussians(s.bg, g["means3D"].detach(), e, g["opacities"].detach(), g["scales"].detach(), g["rotations"].detach(), 1.0, e,
                                   s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, H, W, g["shs"].detach(), 3, s.campos, False, False, tile=tile)[0]
    for _ in range(3): step()
    torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10): step()
    t1.record(); torch.cuda.synchronize()
    st = {k: round(ms / n, 4) for k, (ms, n) in _lib.stage_stats().items() if n}; lib.sr_set_stage_timing(0)
    ms = t0.elapsed_time(t1) / 10
    row = dict(tile=f"{tile[0]}x{tile[1]}", duplicates_D=int(D), ms_per_step=round(ms, 3), msplats_per_s=round(P / ms / 1e3, 1),
               binning_ms=round(sum(st[k] for k in ("depth_sort", "scan", "expand_x", "expand_y", "ranges")), 3), **st)
    rows.append(row); print(json.dumps(row), flush=True)
out = dict(scene=f"{P} Gaussians, {W}x{H}, SH 3, fwd+bwd, 1 GPU", rows=rows)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/tile_sweep_{W}x{H}.json", "w"), indent=1)
