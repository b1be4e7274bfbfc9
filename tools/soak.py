"""Soak: a few scenes rendered forward + backward thousands of times in a round-robin, every result compared bit for bit with the first one of
its scene -- a race or a stale buffer anywhere in the chain (device-side item counts of the compacting sort included) shows as a mismatch.
python tools/soak.py [iterations per scene = 1500]   (GPU box)"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diff_surfel_rasterization import GaussianRasterizer
from streetunveiler_amd.synthetic import posed_scene, synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
from tests.gpu_util import DEV, settings_for
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
NAMES = ("means3D", "opacities", "scales", "rotations", "shs")
scenes = []
for k, (P, W, H, tile, kind) in enumerate([(60_000, 640, 360, None, "bench"), (25_000, 333, 201, (32, 16), "posed"), (120_000, 960, 540, (8, 8), "inside"), (900, 64, 48, None, "posed")]):
    if kind == "bench": cam, g = synthetic_camera(W, H, index=k), synthetic_gaussians(P, W, H, seed=k, scale_lo=2e-3, scale_hi=2e-2)
    else: cam, g = posed_scene(P, W, H, seed=k, scale_lo=2e-3, scale_hi=2e-2, spread=15.0, behind_fraction=0.7 if kind == "inside" else 0.0)
    dc, da = [t.to(DEV) for t in synthetic_upstream_grads(W, H, seed=k)]
    scenes.append(dict(cam=cam, g={n: g[n].to(DEV) for n in NAMES}, dc=dc, da=da, tile=tile, P=P))
def step(sc):
    t = {n: sc["g"][n].detach().requires_grad_() for n in NAMES}
    m2 = torch.zeros(sc["P"], 3, device=DEV, requires_grad=True)
    c, r, a = GaussianRasterizer(settings_for(sc["cam"], [0.1, 0.2, 0.3], 3), tile=sc["tile"])(means3D=t["means3D"], means2D=m2, opacities=t["opacities"], scales=t["scales"],
                                                                                         rotations=t["rotations"], shs=t["shs"])
    grads = torch.autograd.grad((c * sc["dc"]).sum() + (a * sc["da"]).sum(), [t[n] for n in NAMES] + [m2])
    return [c.detach(), a.detach(), r] + list(grads)
ref = [step(sc) for sc in scenes]
torch.cuda.synchronize()
t0 = time.time(); bad = 0
for it in range(N):
    for k, sc in enumerate(scenes):
        out = step(sc)
        if it % 8 == 0 or it == N - 1:   # (comparing costs a sync: every 8th round, which still sees every buffer reused many times)
            for j, (a, b) in enumerate(zip(ref[k], out)):
                if not torch.equal(a, b):
                    bad += 1; print(f"MISMATCH iteration {it} scene {k} output {j}: max |d| {float((a.double() - b.double()).abs().max()):.3e}", flush=True)
torch.cuda.synchronize()
print(f"soak: {N} x {len(scenes)} fwd+bwd steps in {time.time() - t0:.1f} s, {bad} mismatches")
sys.exit(1 if bad else 0)
