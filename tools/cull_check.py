#!/usr/bin/env python
"""Quadrant culling must be invisible: one fuzz scene with culling on and off (and the row-mapped kernel), per-pixel state compared.
FUZZ_BIG=3 python tools/cull_check.py 113"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.fuzz_parity import make_scene
from tests.gpu_util import run_hip_raw, run_oracle
for seed in [int(a) for a in sys.argv[1:]]:
    sc = make_scene(seed)
    g, cam, bg, deg, colors, tile = (sc[k] for k in ("g", "cam", "bg", "deg", "colors", "tile"))
    t = tile if tile != (16, 16) else None
    on = run_hip_raw(g, cam, bg, deg, colors=colors, tile=t)
    off = run_hip_raw(g, cam, bg, deg, colors=colors, tile=t, quadrant_cull=False)
    fwd, _ = run_oracle(g, cam, bg, deg, colors=colors, tile=tile)
    a, b = on["img"]["n_contrib"].view(np.uint32), off["img"]["n_contrib"].view(np.uint32)
    d = np.argwhere(a[0] != b[0])
    print(sc["tag"], "pixels whose last contributor depends on the culling:", len(d))
    for (y, x) in d[:10]:
        print(f"   ({x},{y}) culled {a[0][y, x]} unculled {b[0][y, x]} oracle32 {fwd['n_contrib'][0][y, x]}")
        # which list entries were dropped?  the tile's list and the splats' screen positions
        tw, th = tile
        tid = (y // th) * ((cam.image_width + tw - 1) // tw) + x // tw
        r0, r1 = fwd["ranges"][tid]
        ids = fwd["point_list"][r0:r1]
        lo, hi = int(min(a[0][y, x], b[0][y, x])), int(max(a[0][y, x], b[0][y, x]))
        for k in range(lo, min(hi, lo + 6)):
            gid = int(ids[k]); print(f"      entry {k + 1}: gaussian {gid} centre {fwd['means2D'][gid]} radius {fwd['radii'][gid]} opacity {float(g['opacities'][gid]):.4f} depth {fwd['depths'][gid]:.4f}")
