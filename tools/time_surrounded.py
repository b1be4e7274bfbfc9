"""Stage times of one forward + backward when the camera stands INSIDE the cloud (synthetic.posed_scene(behind_fraction=...)): what a street
scene looks like to the operator -- most of the P Gaussians culled -- next to the benchmark's 86 %-visible frame.
python tools/time_surrounded.py [out.json] [--gaussians 3000000] [--behind 0,0.5,0.8]   (GPU box)"""
import argparse, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import posed_scene, synthetic_upstream_grads
ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?"); ap.add_argument("--gaussians", type=int, default=3_000_000); ap.add_argument("--behind", default="0,0.5,0.8")
a = ap.parse_args()
dev, W, H, P = "cuda:0", 1920, 1080, a.gaussians
lib = _lib.load()
dc, da = [t.to(dev) for t in synthetic_upstream_grads(W, H, seed=1)]
res = {}
for frac in [float(x) for x in a.behind.split(",")]:
    cam, g = posed_scene(P, W, H, seed=7, spread=25.0, behind_fraction=frac)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0, cam.world_view_transform.to(dev),
                                      cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    def step():
        for v in list(t.values()) + [m2]: v.grad = None
        c, r, am = GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        torch.autograd.backward([c, am], [dc, da])
        return r
    for _ in range(3): r = step()
    V = int((r > 0).sum())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): step()
    e1.record(); torch.cuda.synchronize()
    wall = e0.elapsed_time(e1) / 20
    lib.sr_set_stage_timing(1)
    for _ in range(10): step()
    torch.cuda.synchronize()
    st = {k: round(ms / n, 4) for k, (ms, n) in _lib.stage_stats().items() if n}
    lib.sr_set_stage_timing(0)
    res[f"behind_{frac}"] = dict(P=P, visible=V, visible_frac=round(V / P, 3), ms_per_step=round(wall, 4), msplats_per_s=round(P / wall / 1e3, 1), stage_ms=st,
                                 per_P_stages_ms=round(sum(st.get(k, 0) for k in ("preprocess", "depth_sort", "scan", "preprocess_bwd")), 4))
    del t, m2
    torch.cuda.empty_cache()
print(json.dumps(res, indent=1))
if a.out: json.dump(res, open(a.out, "w"), indent=1)
