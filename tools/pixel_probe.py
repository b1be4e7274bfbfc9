#!/usr/bin/env python
"""Every list entry of one pixel: the kernels' decisions (sr_debug_pair_decisions) against a float64 evaluation of the oracle's formulas; prints
the entries where they differ.  FUZZ_BIG=3 python tools/pixel_probe.py <seed> <x> <y>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.fuzz_parity import make_scene
from tests.gpu_util import run_hip_raw, run_oracle
seed, x, y = (int(a) for a in sys.argv[1:4])
sc = make_scene(seed)
g, cam, bg, deg, colors, tile = (sc[k] for k in ("g", "cam", "bg", "deg", "colors", "tile"))
fwd, _ = run_oracle(g, cam, bg, deg, colors=colors, tile=tile)
raw = run_hip_raw(g, cam, bg, deg, colors=colors, tile=tile if tile != (16, 16) else None, decisions=True)
tw, th = tile
QX, QY = tw // 8, th // 8
tx, ty = x // tw, y // th
tid = ty * ((cam.image_width + tw - 1) // tw) + tx
r0, r1 = (int(v) for v in fwd["ranges"][tid])
lx, ly = x - tx * tw, y - ty * th
q = (ly // 8) * QX + lx // 8; lane = (ly % 8) * 8 + lx % 8
valid = raw["decisions"]["valid"].reshape(-1, QX * QY); use3d = raw["decisions"]["use3d"].reshape(-1, QX * QY)
T = 1.0; shown = 0
for pos in range(r0, r1):
    gid = int(fwd["point_list"][pos])
    T9 = fwd["transMat"][gid].astype(np.float64); c = fwd["means2D"][gid].astype(np.float64); opa = float(fwd["normal_opacity"][gid][3])
    Tu, Tv, Tw = T9[0:3], T9[3:6], T9[6:9]
    p = np.cross(x * Tw - Tu, y * Tw - Tv)
    hv, hu = bool((int(valid[pos, q]) >> lane) & 1), bool((int(use3d[pos, q]) >> lane) & 1)
    if p[2] == 0: ov, ou, alpha, rho3d, rho2d, depth = False, False, 0.0, np.inf, 0.0, 0.0
    else:
        s = p[:2] / p[2]; rho3d = s @ s; d = c - np.array([x, y], np.float64); rho2d = 2.0 * (d @ d)
        ou = rho3d <= rho2d
        depth = s[0] * Tw[0] + s[1] * Tw[1] + Tw[2] if ou else Tw[2]
        alpha = min(0.99, opa * np.exp(-0.5 * min(rho3d, rho2d)))
        ov = (not depth < 0.2) and (not alpha < 1.0 / 255.0)
    if hv != ov or (hv and hu != ou):
        shown += 1
        print(f"entry {pos - r0 + 1}: gaussian {gid} kernels valid={hv} 3d={hu} | float64 valid={ov} 3d={ou} alpha {alpha:.6g} rho3d {rho3d:.6g} rho2d {rho2d:.6g} depth {depth:.6g} p.z {p[2]:.3g} T before {T:.4g}")
    if ov:
        if T * (1 - alpha) < 1e-4: break
        T *= 1 - alpha
print(f"{shown} differing entries among {r1 - r0}; float64 final T {T:.6e}; kernels' final T {raw['img']['final_T'][0][y, x]:.6e}")
