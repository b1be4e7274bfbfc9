#!/usr/bin/env python
"""Which Gaussian holds the worst gradient row of a full-size parity run, and why: HIP, the float32 oracle and the float64 arbiter side by
side under the kernels' own decisions.   python tools/worst_row.py [--camera 7] [--tensor dL_dmeans3D] [--gaussians 3000000]   [--posed 7,25]   (GPU box)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
from tests.gpu_util import forced_f64_reference, gradient_row_errors, run_hip, run_hip_raw, run_oracle

ap = argparse.ArgumentParser()
ap.add_argument("--camera", type=int, default=7); ap.add_argument("--tensor", default="dL_dmeans3D"); ap.add_argument("--gaussians", type=int, default=3_000_000)
ap.add_argument("--posed", default=None, help="seed,spread: a camera in general position (synthetic.posed_scene) instead of the benchmark camera")
a = ap.parse_args()
W, H, P = 1920, 1080, a.gaussians
cam = synthetic_camera(W, H, index=a.camera); g = synthetic_gaussians(P, W, H, seed=0); bg = np.zeros(3, np.float32)
if a.posed:
    from streetunveiler_amd.synthetic import posed_scene
    seed, spread = a.posed.split(",")
    cam, g = posed_scene(P, W, H, seed=int(seed), spread=float(spread))
dc, da = synthetic_upstream_grads(W, H, seed=1)
fwd, _ = run_oracle(g, cam, bg, 3)
raw = run_hip_raw(g, cam, bg, 3, decisions=True)
out = run_hip(g, cam, bg, 3, dc, da)
_, f64, b64 = forced_f64_reference(g, cam, bg, 3, dc, da, base=fwd, raw=raw)
_, f32, b32 = forced_f64_reference(g, cam, bg, 3, dc, da, base=fwd, raw=raw, f64=False)
vis = np.ones(P, bool)
e_hip = gradient_row_errors(out, b64, vis, (g, cam))[a.tensor]
e_o32 = gradient_row_errors({k: v for k, v in b32.items()}, b64, vis, (g, cam))[a.tensor]
order = np.argsort(-np.where(fwd["radii"] > 0, e_hip, 0))[:5]
ref = b64.get(a.tensor + "64", b64[a.tensor]).reshape(P, -1)
rep = []
for i in order:
    i = int(i)
    rep.append(dict(gaussian=i, err_hip=float(e_hip[i]), err_oracle_f32=float(e_o32[i]), hip=np.asarray(out[a.tensor]).reshape(P, -1)[i].tolist(),
                    oracle_f32=np.asarray(b32[a.tensor]).reshape(P, -1)[i].tolist(), f64=ref[i].tolist(), tensor_max=float(np.abs(ref).max()),
                    radius=int(fwd["radii"][i]), tiles_touched=int(fwd["tiles_touched"][i]), depth=float(fwd["depths"][i]), opacity=float(fwd["normal_opacity"][i, 3]),
                    means2D=fwd["means2D"][i].tolist(), transMat=fwd["transMat"][i].tolist(), scales=g["scales"][i].tolist(),
                    dL_dtransMat64=b64["dL_dtransMat64"][i].tolist(), dT_hip_vs_f64="n/a"))
print(json.dumps(rep, indent=1))
print("rows over 1e-2:", int((e_hip[fwd["radii"] > 0] > 1e-2).sum()), "over 5e-3:", int((e_hip[fwd["radii"] > 0] > 5e-3).sum()), "| oracle f32 rows over 1e-2:", int((e_o32[fwd["radii"] > 0] > 1e-2).sum()))
