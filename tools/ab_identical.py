#!/usr/bin/env python
"""Are two builds of the library bit-identical in their results?  Renders one scene fwd + bwd with each (a subprocess per build,
SURFEL_RASTER_LIB) and compares every output and gradient bit for bit.
    gpurun -- 'python tools/ab_identical.py ab/libA.so ab/libB.so [P W H]'"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

if sys.argv[1] == "--worker":
    import torch
    from tests.gpu_util import run_hip
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
    P, W, H = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    cam = synthetic_camera(W, H, index=3); g = synthetic_gaussians(P, W, H, seed=5)
    dc, da = synthetic_upstream_grads(W, H)
    out = run_hip(g, cam, [0.1, 0.2, 0.3], 3, dc, da)
    np.savez(sys.argv[2], **{k: v for k, v in out.items() if v is not None})
    sys.exit(0)

a, b = sys.argv[1], sys.argv[2]
size = sys.argv[3:6] if len(sys.argv) >= 6 else ["400000", "1280", "720"]
res = []
for i, lib in enumerate((a, b)):
    out = f"/tmp/ab_identical_{i}.npz"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", out] + size, check=True, cwd=ROOT,
                   env=dict(os.environ, SURFEL_RASTER_LIB=os.path.abspath(lib)))
    res.append(dict(np.load(out)))
bad = [k for k in res[0] if not np.array_equal(res[0][k], res[1][k], equal_nan=True)]
for k in res[0]:
    d = np.abs(res[0][k].astype(np.float64) - res[1][k].astype(np.float64)).max()
    print(f"{k:16s} {'identical' if k not in bad else 'DIFFERS'}  max |a - b| = {d:.3e}")
print("BIT-IDENTICAL" if not bad else f"NOT identical: {bad}")
sys.exit(1 if bad else 0)
