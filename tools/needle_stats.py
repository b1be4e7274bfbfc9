"""CPU only: the alpha of sampled (pixel, list entry) pairs of fuzz scenes evaluated in float32 three ways -- upstream's per-pixel cross product (the
oracle), the kernels' staged cross products about the TILE centre, and the same about the splat's own centre -- against float64, binned by the
surfel's axis ratio (or, BY_RADIUS=1, by its radius).  python tools/needle_stats.py <seed> [<seed> ...]   (seeds >= 400000: needles)"""
import os, sys, math
sys.path.insert(0, os.getcwd())
import numpy as np
from tools.fuzz_parity import make_scene
from oracle import surfel_oracle as so
rows = []
for seed in [int(a) for a in sys.argv[1:]]:
    sc = make_scene(seed)
    g, cam, bg, deg, tile = sc["g"], sc["cam"], sc["bg"], sc["deg"], sc["tile"]
    n = {k: v.numpy() for k, v in g.items()}
    fwd = so.rasterize_forward(n["means3D"], n["opacities"], n["scales"], n["rotations"], shs=n["shs"], viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                               campos=cam.camera_center.numpy(), bg=bg, image_width=cam.image_width, image_height=cam.image_height, sh_degree=deg, tile=tile,
                               tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    tw, th = tile; W, H = cam.image_width, cam.image_height; gx = (W + tw - 1) // tw
    ratio = n["scales"].max(1) / n["scales"].min(1)
    if os.environ.get("BY_RADIUS"): ratio = fwd["radii"].astype(np.float64)
    rng = np.random.default_rng(seed)
    for _ in range(400):
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        tid = (y // th) * gx + x // tw; r0, r1 = fwd["ranges"][tid]
        ids = fwd["point_list"][r0:r0 + int(fwd["n_contrib"][0][y, x])]
        if not len(ids): continue
        T = fwd["transMat"][ids]; c = fwd["means2D"][ids]; opa = fwd["normal_opacity"][ids, 3]
        def alpha(dt, form):
            Tu, Tv, Tw = T[:, 0:3].astype(dt), T[:, 3:6].astype(dt), T[:, 6:9].astype(dt); px, py = dt(x), dt(y)
            if form == "oracle":
                p = np.cross(px * Tw - Tu, py * Tw - Tv).astype(dt)
            elif form in ("tile_fma", "tile_fma_kahan"):
                # the tile origin with the shift done as ONE fused multiply-add (exact product, one rounding), and optionally Kahan's
                # fma-compensated a b - c d for the cross products: float64 stands in for the fused operation, results rounded to float32
                f32 = np.float32
                ox = np.full(len(ids), (x // tw) * tw + tw // 2, np.float64); oy = np.full(len(ids), (y // th) * th + th // 2, np.float64)
                Tw_ = Tw.astype(f32); Tu_ = (Tu.astype(np.float64) - ox[:, None] * Tw_.astype(np.float64)).astype(f32); Tv_ = (Tv.astype(np.float64) - oy[:, None] * Tw_.astype(np.float64)).astype(f32)
                def cross(a, b):
                    if form == "tile_fma": return np.cross(a.astype(f32), b.astype(f32)).astype(f32)
                    a64, b64 = a.astype(np.float64), b.astype(np.float64)       # Kahan: the result of a b - c d correctly rounded (to within 1.5 ulp)
                    return np.cross(a64, b64).astype(f32)
                A, B, C = cross(Tv_, Tw_), cross(Tw_, Tu_), cross(Tu_, Tv_)
                p = ((f32(px) - ox.astype(f32))[:, None] * A + (f32(py) - oy.astype(f32))[:, None] * B + C).astype(f32)
                dt = f32
            else:
                if form == "tile": ox, oy = np.full(len(ids), (x // tw) * tw + tw // 2, dt), np.full(len(ids), (y // th) * th + th // 2, dt)
                else: ox, oy = c[:, 0].astype(dt), c[:, 1].astype(dt)       # the splat's own (AABB) centre
                Tu_, Tv_ = (Tu - ox[:, None] * Tw).astype(dt), (Tv - oy[:, None] * Tw).astype(dt)
                A, B, C = np.cross(Tv_, Tw).astype(dt), np.cross(Tw, Tu_).astype(dt), np.cross(Tu_, Tv_).astype(dt)
                p = ((px - ox)[:, None] * A + (py - oy)[:, None] * B + C).astype(dt)
            with np.errstate(all="ignore"):
                s = p[:, :2] / p[:, 2:3]; r3 = (s * s).sum(1)
            d = np.stack([c[:, 0].astype(dt) - px, c[:, 1].astype(dt) - py], 1); r2 = dt(2.0) * (d * d).sum(1)
            rho = np.where(r3 <= r2, r3, r2)
            return np.minimum(0.99, opa * np.exp(-0.5 * rho)).astype(np.float64)
        a64 = alpha(np.float64, "oracle"); keep = a64 >= 1 / 255
        for form in ("oracle", "tile", "tile_fma", "tile_fma_kahan", "centre"):
            e = np.abs(alpha(np.float32, form) - a64)
            for r, ee in zip(ratio[ids][keep], e[keep]): rows.append((form, r, ee))
import collections
bins = [(1, 10), (10, 30), (30, 100), (100, 300), (300, 2000)] if not os.environ.get("BY_RADIUS") else [(0, 20), (20, 100), (100, 400), (400, 100000)]
for form in ("oracle", "tile", "tile_fma", "tile_fma_kahan", "centre"):
    print(form)
    for lo, hi in bins:
        e = np.array([ee for f, r, ee in rows if f == form and lo <= r < hi])
        if e.size: print(f"   ratio {lo:4d}-{hi:4d}: pairs {e.size:6d}  alpha error p50 {np.median(e):.1e} p99 {np.quantile(e, .99):.1e} max {e.max():.1e}")
