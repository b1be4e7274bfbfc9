#!/usr/bin/env python
"""One (pixel, list entry) pair of a fuzz scene under the microscope: the oracle's ray-splat evaluation (global pixel coordinates) and the
kernels' (tile-local, staged cross products), each in float32 and float64.  FUZZ_BIG=3 python tools/pair_probe.py <seed> <x> <y> <entry (1-based)>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.fuzz_parity import make_scene
from tests.gpu_util import run_oracle
seed, x, y, entry = (int(a) for a in sys.argv[1:5])
sc = make_scene(seed)
g, cam, bg, deg, colors, tile = (sc[k] for k in ("g", "cam", "bg", "deg", "colors", "tile"))
fwd, _ = run_oracle(g, cam, bg, deg, colors=colors, tile=tile)
tw, th = tile
tx, ty = x // tw, y // th
tid = ty * ((cam.image_width + tw - 1) // tw) + tx
r0, r1 = fwd["ranges"][tid]
gid = int(fwd["point_list"][r0 + entry - 1])
T9 = fwd["transMat"][gid]; c = fwd["means2D"][gid]; opa = float(fwd["normal_opacity"][gid][3])
print(f"gaussian {gid}: centre {c}, radius {fwd['radii'][gid]}, opacity {opa:.5f}, depth {fwd['depths'][gid]:.5f}\n transMat {T9}")
for dt in (np.float32, np.float64):
    Tu, Tv, Tw = (T9[0:3].astype(dt), T9[3:6].astype(dt), T9[6:9].astype(dt))
    px, py = dt(x), dt(y)
    k = px * Tw - Tu; l = py * Tw - Tv
    p = np.cross(k, l).astype(dt)
    s = p[:2] / p[2]
    rho3d = s[0] * s[0] + s[1] * s[1]
    d = np.array([c[0] - px, c[1] - py], dt); rho2d = dt(2.0) * (d[0] * d[0] + d[1] * d[1])
    rho = min(rho3d, rho2d)
    depth = s[0] * Tw[0] + s[1] * Tw[1] + Tw[2] if rho3d <= rho2d else Tw[2]
    alpha = min(0.99, opa * np.exp(-0.5 * rho))
    print(f" oracle form {dt.__name__}: p {p}, s {s}, rho3d {rho3d:.7g} rho2d {rho2d:.7g} depth {depth:.7g} alpha {alpha:.7g}")
    # the kernels' form: tile-local origin at the tile centre, staged A = Tv' x Tw, B = Tw x Tu', C = Tu' x Tv'
    Xc, Yc = dt(tx * tw + tw // 2), dt(ty * th + th // 2)
    Tu_, Tv_ = (Tu - Xc * Tw).astype(dt), (Tv - Yc * Tw).astype(dt)
    A, B, C = np.cross(Tv_, Tw).astype(dt), np.cross(Tw, Tu_).astype(dt), np.cross(Tu_, Tv_).astype(dt)
    xl, yl = px - Xc, py - Yc
    pp = (xl * A + yl * B + C).astype(dt)
    s2 = pp[:2] / pp[2]
    r3 = s2[0] * s2[0] + s2[1] * s2[1]
    print(f" kernel form {dt.__name__}: A {A} B {B} C {C}\n    pp {pp}, s {s2}, rho3d {r3:.7g}, alpha {min(0.99, opa * np.exp(-0.5 * min(r3, rho2d))):.7g}")
