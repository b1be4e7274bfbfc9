"""Forward-only rendering (SR_FLAG_FORWARD_ONLY: what the operator does under torch.no_grad()) against the training forward, per stage and
end to end, on the C3 and the C5 scene; plus the state bytes the mode does not write.     python tools/fwd_only_time.py [out.json] [--configs c3,c5] [--modes training_forward,forward_only]   (GPU box)"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizationSettings, _C
from streetunveiler_amd import _lib
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians
dev = "cuda:0"
lib = _lib.load()
out = {}
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("out", nargs="?"); ap.add_argument("--configs", default="c3,c5"); ap.add_argument("--modes", default="training_forward,forward_only")
cli = ap.parse_args()
SCENES = {"c3": (3_000_000, 1920, 1080), "c5": (6_000_000, 3840, 2160)}
for tag, (P, W, H) in {k: SCENES[k] for k in cli.configs.split(",")}.items():
    cam = synthetic_camera(W, H); g = {k: v.to(dev) for k, v in synthetic_gaussians(P, W, H).items()}
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), 3, cam.camera_center.to(dev), False, False)
    e = torch.empty(0, device=dev)
    res = {}
    for mode, fo in [(m, m == "forward_only") for m in cli.modes.split(",")]:
        def step():
            return _C.rasterize_gaussians(s.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                                          H, W, g["shs"], 3, s.campos, False, False, forward_only=fo)
        for _ in range(3): r = step()
        D, V = r[0], int((r[3] > 0).sum())
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): step()
        b.record(); torch.cuda.synchronize()
        wall = a.elapsed_time(b) / 20
        lib.sr_set_stage_timing(1)
        for _ in range(10): step()
        torch.cuda.synchronize()
        st = {k: round(ms / n, 4) for k, (ms, n) in _lib.stage_stats().items() if n}
        lib.sr_set_stage_timing(0)
        res[mode] = dict(ms_per_frame=round(wall, 4), frames_per_s=round(1e3 / wall, 1), stage_ms=st)
    res["state_bytes_not_written"] = dict(sh_jacobian=36 * V, image_state=20 * W * H, hit_masks=2 * D, total=36 * V + 20 * W * H + 2 * D)
    res["scene"] = dict(P=P, W=W, H=H, D=D, visible=V)
    if "training_forward" in res and "forward_only" in res:
        res["speedup"] = round(res["training_forward"]["ms_per_frame"] / res["forward_only"]["ms_per_frame"], 4)
    out[tag] = res
    del g
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))
if cli.out:
    json.dump(out, open(cli.out, "w"), indent=1)
