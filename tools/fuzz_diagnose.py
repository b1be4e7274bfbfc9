#!/usr/bin/env python
"""One scene of tools/fuzz_parity.py in detail: the HIP operator AND the float32 oracle against the free-running float64 reference, per
output, robust / non-robust elements apart.  python tools/fuzz_diagnose.py <seed> [<seed> ...]   (FUZZ_BIG as in fuzz_parity.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.fuzz_parity import make_scene
from tests.gpu_util import forced_f64_reference, free_f64_reference, gradient_row_errors, run_hip, run_hip_raw, run_oracle

for seed in [int(a) for a in sys.argv[1:]]:
    sc = make_scene(seed)
    g, cam, bg, deg, dc, da, colors, tile = (sc[k] for k in ("g", "cam", "bg", "deg", "dc", "da", "colors", "tile"))
    print(sc["tag"])
    fwd, bwd = run_oracle(g, cam, bg, deg, dc, da, colors=colors, tile=tile)
    out = run_hip(g, cam, bg, deg, dc, da, colors=colors, tile=tile if tile != (16, 16) else None)
    xfwd, xbwd, margins = free_f64_reference(g, cam, bg, deg, dc, da, colors=colors, tile=tile, base=fwd)
    rob_px = margins["pixel"] > 1.0
    for name, hip, o32, ref in [("color", out["color"], fwd["color"], xfwd["color"])] + [(f"allmap[{c}]", out["allmap"][c], fwd["allmap"][c], xfwd["allmap"][c]) for c in (0, 1, 2, 3, 4, 6)]:
        m = np.broadcast_to(rob_px, np.shape(ref))
        e = lambda a: np.abs(np.asarray(a, np.float64) - ref) / (1.0 + np.abs(ref))
        eh, eo = e(hip), e(o32)
        print(f"  {name:10s} robust max: hip {eh[m].max(initial=0):.2e} oracle32 {eo[m].max(initial=0):.2e} | non-robust max: hip {eh[~m].max(initial=0):.2e} oracle32 {eo[~m].max(initial=0):.2e}")
    raw0 = run_hip_raw(g, cam, bg, deg, colors=colors, tile=tile if tile != (16, 16) else None)
    nc = raw0["img"]["n_contrib"].view(np.uint32).reshape(2, *rob_px.shape)
    print(f"  robust pixels {int(rob_px.sum())} of {rob_px.size}: last contributor differs from float64 in hip {int((nc[0] != xfwd['n_contrib'][0])[rob_px].sum())}, "
          f"oracle32 {int((fwd['n_contrib'][0] != xfwd['n_contrib'][0])[rob_px].sum())}; longest list {int(xfwd['n_contrib'][0].max())}")
    vis = xfwd["radii"] > 0
    rob_g = vis & (margins["gaussian"] > 1.0)
    eh = gradient_row_errors(out, xbwd, np.ones_like(vis), (g, cam)); eo = gradient_row_errors({k: (v if out.get(k) is not None else None) for k, v in bwd.items()}, xbwd, np.ones_like(vis), (g, cam))
    for key in eh:
        ref = xbwd.get(key + "64", xbwd.get(key)); P = ref.shape[0]
        r = np.asarray(ref, np.float64).reshape(P, -1)
        loose = lambda a: np.abs(np.asarray(a, np.float64).reshape(P, -1) - r).max(1) / (np.abs(r).max() + 1e-30)
        lh, lo = loose(out[key]), loose(bwd[key])
        q = lambda x: (float(np.quantile(x, 0.999)) if x.size else 0.0, float(x.max(initial=0)))
        print(f"  {key:14s} robust rows p99.9/max: hip {q(eh[key][rob_g])[0]:.2e}/{q(eh[key][rob_g])[1]:.2e} oracle32 {q(eo[key][rob_g])[0]:.2e}/{q(eo[key][rob_g])[1]:.2e}"
              f" | non-robust max of tensor scale: hip {lh[vis & ~rob_g].max(initial=0):.2e} oracle32 {lo[vis & ~rob_g].max(initial=0):.2e}")
    # the same with the kernels' decisions forced on both checkers (float64 = the reference, float32 = what the arithmetic itself loses)
    raw = run_hip_raw(g, cam, bg, deg, colors=colors, tile=tile if tile != (16, 16) else None, decisions=True)
    _, sfwd, sb64 = forced_f64_reference(g, cam, bg, deg, dc, da, tile=tile, colors=colors, base=fwd, raw=raw)
    _, _, sb32 = forced_f64_reference(g, cam, bg, deg, dc, da, tile=tile, colors=colors, base=fwd, raw=raw, f64=False)
    eh = gradient_row_errors(out, sb64, vis, (g, cam)); eo = gradient_row_errors({k: (v if out.get(k) is not None else None) for k, v in sb32.items()}, sb64, vis, (g, cam))
    for key in eh:
        q = lambda x: (float(np.quantile(x, 0.999)) if x.size else 0.0, float(x.max(initial=0)))
        print(f"  forced {key:14s} rows p99.9/max: hip {q(eh[key])[0]:.2e}/{q(eh[key])[1]:.2e} oracle32 {q(eo[key])[0]:.2e}/{q(eo[key])[1]:.2e}")
        if eh[key].size and key in eo and eh[key].max() > 5e-3:   # which Gaussian, and is it the float32 oracle's bad row too?
            ids = np.flatnonzero(vis); ih, io = int(np.argmax(eh[key])), int(np.argmax(eo[key])); i = int(ids[ih])
            R = lambda d, k: np.asarray(d.get(k + "64", d.get(k)), np.float64).reshape(vis.size, -1)
            print(f"      worst hip row: Gaussian {i} (oracle32 there {eo[key][ih]:.2e}; oracle32's worst row: Gaussian {int(ids[io])}, hip there {eh[key][io]:.2e}); radius {int(fwd['radii'][i])} tiles {int(fwd['tiles_touched'][i])}"
                  f" depth {float(fwd['depths'][i]):.3f} centre {fwd['means2D'][i].tolist()} scales {g['scales'][i].tolist()}\n      hip {R(out, key)[i].tolist()}\n      o32 {R(sb32, key)[i].tolist()}\n      f64 {R(sb64, key)[i].tolist()}")
    bad = np.argwhere((nc[0] != xfwd["n_contrib"][0]) & rob_px)
    ft = lambda d: np.asarray(d["final_T"]).reshape(-1, *rob_px.shape)[0] if "final_T" in d else np.full(rob_px.shape, np.nan)
    for (y, x) in bad[:5]:
        print(f"    pixel ({x},{y}): margin {margins['pixel'][y, x]:.3g}, last contributor hip {nc[0][y, x]} oracle32 {fwd['n_contrib'][0][y, x]} float64 {xfwd['n_contrib'][0][y, x]},"
              f" final T hip {ft(raw0['img'])[y, x]:.6e} oracle32 {ft(fwd)[y, x]:.6e} float64 {ft(xfwd)[y, x]:.6e}")
    eh = np.abs(np.asarray(out["color"], np.float64) - xfwd["color"]) / (1.0 + np.abs(xfwd["color"]))
    eh = np.where(np.broadcast_to(rob_px, eh.shape), eh, 0.0)
    c, y, x = np.unravel_index(np.argmax(eh), eh.shape)
    print(f"    worst robust colour element: channel {c} pixel ({x},{y}) hip {out['color'][c, y, x]:.7f} oracle32 {fwd['color'][c, y, x]:.7f} float64 {xfwd['color'][c, y, x]:.7f};"
          f" last contributor hip {nc[0][y, x]} oracle32 {fwd['n_contrib'][0][y, x]} float64 {xfwd['n_contrib'][0][y, x]}; margin {margins['pixel'][y, x]:.3g};"
          f" final T hip {ft(raw0['img'])[y, x]:.6e} oracle32 {ft(fwd)[y, x]:.6e} float64 {ft(xfwd)[y, x]:.6e}")
