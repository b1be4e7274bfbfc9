#!/usr/bin/env python
"""Stage times on a NON-uniform scene: a street-like distribution in which a fraction of the Gaussians is squeezed into a
few screen regions (dense facades / vegetation), so tile-list lengths are heavy-tailed.  One wave per tile makes the longest
lists the tail of the blend kernels; this tool shows how much.  python tools/clustered_scene.py [fraction_clustered]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import clustered_gaussians, synthetic_camera, synthetic_upstream_grads
P, W, H, dev = 3_000_000, 1920, 1080, "cuda:0"
frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
lib = _lib.load()
cam = synthetic_camera(W, H)
g = clustered_gaussians(P, W, H, frac)
tx, ty = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
dc, da = [x.to(dev) for x in synthetic_upstream_grads(W, H)]
s = GaussianRasterizationSettings(H, W, tx, ty, torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                  cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
e = torch.empty(0, device=dev)
with torch.no_grad():
    D, _, _, radii, geom, binning, img = _C.rasterize_gaussians(s.bg, t["means3D"].detach(), e, t["opacities"].detach(), t["scales"].detach(), t["rotations"].detach(), 1.0, e,
                                                                 s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, H, W, t["shs"].detach(), 3, s.campos, False, False)
    r = _C.binning_view(binning, P, D, W, H)["ranges"].cpu().numpy().astype(np.int64)
    ln = r[:, 1] - r[:, 0]
    nc = _C.image_view(img, W, H)["n_contrib"][0].cpu().numpy()
print(f"D={D}, list length mean {ln.mean():.0f}, p50 {np.percentile(ln, 50):.0f}, p99 {np.percentile(ln, 99):.0f}, max {ln.max()}; "
      f"deepest contributor per pixel mean {nc.mean():.0f}, max {nc.max()}")
def step():
    for x in list(t.values()) + [m2d]: x.grad = None
    c, rr, a = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([c, a], [dc, da])
for _ in range(3): step()
torch.cuda.synchronize(); lib.sr_set_stage_timing(1)
for _ in range(10): step()
torch.cuda.synchronize()
st = _lib.stage_stats(); lib.sr_set_stage_timing(0)
print({k: round(ms / n_, 4) for k, (ms, n_) in st.items() if n_})
