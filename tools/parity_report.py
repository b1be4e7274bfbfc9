#!/usr/bin/env python
"""Error distribution of the HIP operator against the CPU oracle, two ways:
  * FORCED decisions: the oracle blends with the hard decisions the HIP kernels took (which pairs contribute, by which path, where
    each pixel stops, which contributor is its median: sr_debug_pair_decisions + n_contrib) and computes the values itself.  What is
    left is rounding -- this is the strict bar (images 1e-4, gradient rows relative to their own magnitude).
  * FREE-running oracle: the same comparison with the oracle's own decisions; the difference between the two tables is exactly the
    set of decisions taken within rounding noise of a threshold (alpha >= 1/255, T' < 1e-4, rho3d <= rho2d, T > 0.5), also split by
    the oracle's decision margins (so_render_margins).
Writes one JSON (default gpurun_out/r02_parity.json).

    python tools/parity_report.py [--gaussians 500000 --width 1920 --height 1080] [--out FILE]      (GPU box)
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def quantiles(e):
    e = np.asarray(e, np.float64).ravel()
    if e.size == 0:
        return dict(n=0)
    return dict(n=int(e.size), max=float(e.max()), p999=float(np.quantile(e, 0.999)), p99=float(np.quantile(e, 0.99)), p50=float(np.quantile(e, 0.5)))


def report(P, W, H, seed=0, deg=3, aux=True, scale_lo=5e-4, scale_hi=5e-3, eps=None):
    from oracle import surfel_oracle as so
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
    from tests.gpu_util import run_hip, run_hip_raw
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi)
    dc, da = synthetic_upstream_grads(W, H, seed=1, aux=aux)
    bg = np.zeros(3, np.float32)
    t0 = time.time()
    hip = run_hip(g, cam, bg, deg, dc, da)
    t1 = time.time()
    fwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), shs=g["shs"].numpy(),
                               viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                               campos=cam.camera_center.numpy(), bg=bg, image_width=W, image_height=H, sh_degree=deg)
    bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy())
    m = so.render_margins(fwd, eps)
    t2 = time.time()
    # forced-decision oracle
    raw = run_hip_raw(g, cam, bg, deg, decisions=True)
    forced = dict(valid=raw["decisions"]["valid"], use3d=raw["decisions"]["use3d"], n_contrib=raw["img"]["n_contrib"].view(np.uint32))
    ffwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), shs=g["shs"].numpy(),
                                viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                                campos=cam.camera_center.numpy(), bg=bg, image_width=W, image_height=H, sh_degree=deg, forced=forced)
    fbwd = so.rasterize_backward(ffwd, dc.numpy(), da.numpy())
    t3 = time.time()
    # the arbiter: the same blend, same float32 per-Gaussian inputs, same forced decisions, evaluated in double precision
    dfwd = so.rasterize_forward(g["means3D"].numpy(), g["opacities"].numpy(), g["scales"].numpy(), g["rotations"].numpy(), shs=g["shs"].numpy(),
                                viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
                                campos=cam.camera_center.numpy(), bg=bg, image_width=W, image_height=H, sh_degree=deg, forced=forced, f64=True)
    dbwd = so.rasterize_backward(dfwd, dc.numpy(), da.numpy())
    t4 = time.time()
    rob_px = m["pixel"] > 1.0
    rob_med = rob_px & (m["median"] > 1.0)
    vis = fwd["radii"] > 0
    rob_g = vis & (m["gaussian"] > 1.0)
    out = {"config": dict(gaussians=P, width=W, height=H, sh_degree=deg, aux_gradients=aux, seed=seed, D=int(fwd["num_rendered"]), visible=int(vis.sum())),
           "eps": m["eps"], "seconds": dict(hip=round(t1 - t0, 2), oracle=round(t2 - t1, 2), forced_oracle=round(t3 - t2, 2), forced_f64=round(t4 - t3, 2)),
           "decisions_differ": dict(pixels_last_contributor=int((raw["img"]["n_contrib"].view(np.uint32)[0] != fwd["n_contrib"][0]).sum()),
                                    pixels_median_contributor=int((raw["img"]["n_contrib"].view(np.uint32)[1] != fwd["n_contrib"][1]).sum()),
                                    pixels_total=int(W * H)),
           "robust_fraction": dict(pixels=float(rob_px.mean()), pixels_incl_median=float(rob_med.mean()), visible_gaussians=float(rob_g.sum() / max(1, vis.sum()))),
           "radii_equal": bool(np.array_equal(hip["radii"], fwd["radii"])),
           "definition": "image error = |hip - oracle| / (1 + |oracle|) per element; gradient error = max_j |hip - oracle|[row, j] / "
                         "(max_j |oracle|[row, j] + 1e-3 * tensor max) per Gaussian row",
           "images": {}, "gradients": {}, "forced": {"images": {}, "gradients": {}}, "forced_vs_f64": {"images": {}, "gradients": {}}}
    vis_rows = fwd["radii"] > 0
    for name, a, b, d in [("color", hip["color"], ffwd["color"], dfwd["color"])] + [(f"allmap[{c}]", hip["allmap"][c], ffwd["allmap"][c], dfwd["allmap"][c]) for c in range(7)]:
        eh = np.abs(a.astype(np.float64) - d) / (1.0 + np.abs(d)); eo = np.abs(b.astype(np.float64) - d) / (1.0 + np.abs(d))
        out["forced_vs_f64"]["images"][name] = dict(hip=quantiles(eh), oracle_f32=quantiles(eo), hip_over_1e4=int((eh > 1e-4).sum()), oracle_f32_over_1e4=int((eo > 1e-4).sum()))
    for key in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"):
        ref = dbwd[key].reshape(P, -1).astype(np.float64)
        tmax = np.abs(ref).max()
        rows = lambda x: (np.abs(x.reshape(P, -1).astype(np.float64) - ref).max(1) / (np.abs(ref).max(1) + 1e-3 * tmax))[vis_rows]
        out["forced_vs_f64"]["gradients"][key] = dict(tensor_max=float(tmax), hip_rows=quantiles(rows(hip[key])), oracle_f32_rows=quantiles(rows(fbwd[key])))
    assert np.array_equal(ffwd["n_contrib"], forced["n_contrib"]), "forced oracle must reproduce the forced stop / median positions"
    for name, a, b in [("color", hip["color"], ffwd["color"])] + [(f"allmap[{c}]", hip["allmap"][c], ffwd["allmap"][c]) for c in range(7)]:
        err = np.abs(a.astype(np.float64) - b) / (1.0 + np.abs(b))
        out["forced"]["images"][name] = dict(all=quantiles(err), over_1e4=int((err > 1e-4).sum()))
    for key in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D"):
        ref = fbwd[key].reshape(P, -1).astype(np.float64); got = hip[key].reshape(P, -1).astype(np.float64)
        tmax = np.abs(ref).max()
        row = np.abs(got - ref).max(1) / (np.abs(ref).max(1) + 1e-3 * tmax)
        out["forced"]["gradients"][key] = dict(tensor_max=float(tmax), rows=quantiles(row[fwd["radii"] > 0]),
                                                rel_tensor_max=quantiles(np.abs(got - ref).max(1)[fwd["radii"] > 0] / (tmax + 1e-30)))
    chans = [("color", hip["color"], fwd["color"], rob_px)] + [(f"allmap[{c}]", hip["allmap"][c:c + 1], fwd["allmap"][c:c + 1], rob_med if c == 5 else rob_px) for c in range(7)]
    for name, a, b, rob in chans:
        err = np.abs(a.astype(np.float64) - b) / (1.0 + np.abs(b))
        r = np.broadcast_to(rob, err.shape)
        out["images"][name] = dict(robust=quantiles(err[r]), non_robust=quantiles(err[~r]), over_1e4_robust=int((err[r] > 1e-4).sum()), over_1e4_non_robust=int((err[~r] > 1e-4).sum()))
    for name, key in [("dL_dmeans3D", "dL_dmeans3D"), ("dL_dopacity", "dL_dopacity"), ("dL_dscales", "dL_dscales"), ("dL_drotations", "dL_drotations"),
                      ("dL_dsh", "dL_dsh"), ("dL_dmeans2D", "dL_dmeans2D")]:
        ref = bwd[key].reshape(P, -1).astype(np.float64)
        got = hip[name].reshape(P, -1).astype(np.float64)
        tmax = np.abs(ref).max()
        row = np.abs(got - ref).max(1) / (np.abs(ref).max(1) + 1e-3 * tmax)
        tens = np.abs(got - ref).max(1) / (tmax + 1e-30)
        out["gradients"][name] = dict(tensor_max=float(tmax), rows_robust=quantiles(row[rob_g]), rows_non_robust=quantiles(row[vis & ~rob_g]),
                                      rel_tensor_max_robust=quantiles(tens[rob_g]), rel_tensor_max_non_robust=quantiles(tens[vis & ~rob_g]),
                                      invisible_rows_nonzero=int((np.abs(got[~vis]).max(1) > 0).sum()) if (~vis).any() else 0)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500_000); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-aux", action="store_true"); ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r02_parity.json"))
    a = ap.parse_args()
    r = report(a.gaussians, a.width, a.height, aux=not a.no_aux)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(r, open(a.out, "w"), indent=1)
    print(json.dumps({k: r[k] for k in ("config", "robust_fraction", "radii_equal", "decisions_differ", "seconds")}))
    for k, v in r["forced"]["images"].items():
        print("FORCED", k, v["all"], "| >1e-4:", v["over_1e4"])
    for k, v in r["forced"]["gradients"].items():
        print("FORCED", k, "rows", v["rows"])
    for k, v in r["forced_vs_f64"]["images"].items():
        print("vs F64", k, "hip", v["hip"]["max"], v["hip"]["p999"], ">1e-4:", v["hip_over_1e4"], "| oracle_f32", v["oracle_f32"]["max"], v["oracle_f32"]["p999"], ">1e-4:", v["oracle_f32_over_1e4"])
    for k, v in r["forced_vs_f64"]["gradients"].items():
        print("vs F64", k, "hip rows max/p999/p99", v["hip_rows"]["max"], v["hip_rows"]["p999"], v["hip_rows"]["p99"], "| oracle_f32", v["oracle_f32_rows"]["max"], v["oracle_f32_rows"]["p999"], v["oracle_f32_rows"]["p99"])
    for k, v in r["images"].items():
        print(k, "robust", v["robust"], "| >1e-4: robust", v["over_1e4_robust"], "non-robust", v["over_1e4_non_robust"])
    for k, v in r["gradients"].items():
        print(k, "rows robust", v["rows_robust"], "| non-robust", v["rows_non_robust"])
