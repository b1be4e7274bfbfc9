#!/usr/bin/env python
"""Error distribution of the HIP operator against the CPU oracle, three ways (one scene, K1 + binning of the oracle shared):
  * FORCED decisions vs the float64 arbiter: the checker blends with the hard decisions the HIP kernels took (sr_debug_pair_decisions +
    n_contrib) in double precision, and runs K8 in double precision too ("<name>64"): what is left is the kernels' rounding.  The
    float32 oracle, forced the same way, is measured against the same arbiter (which of the two float32 implementations is noisier).
  * FREE-running float64 reference: the checker takes every decision itself (exactly evaluated quantities on the float32
    per-Gaussian state); its float64 decision margins (so_render_margins_f64) split pixels / Gaussians into ROBUST (no decision
    within the noise allowance of a threshold: two correct implementations must agree to rounding, and take the same decisions)
    and non-robust.  This is the independent check of the kernels' DECISIONS.
  * FREE-running float32 oracle: the classic comparison, split by the oracle's own float32 margins (kept for continuity with
    profiles/r02_parity_c2.json).
Also: what sets the level of dL_dscales / dL_drotations -- K8's float32 arithmetic (float32 K8 on exact sums vs float64 K8) or the
conditioning of the chain with respect to its float32 inputs (rows measured against the magnitude of the terms they sum).

    python tools/parity_report.py [--gaussians 500000 --width 1920 --height 1080] [--out FILE]      (GPU box)
"""
import math
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

GRADS = ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh", "dL_dmeans2D")


def quantiles(e):
    e = np.asarray(e, np.float64).ravel()
    if e.size == 0:
        return dict(n=0)
    return dict(n=int(e.size), max=float(e.max()), p999=float(np.quantile(e, 0.999)), p99=float(np.quantile(e, 0.99)), p50=float(np.quantile(e, 0.5)))


def row_err(got, ref):
    """max_j |got - ref| / (max_j |ref| + 1e-3 tensor max) per row."""
    P = ref.shape[0]
    r = np.asarray(ref, np.float64).reshape(P, -1); a = np.asarray(got, np.float64).reshape(P, -1)
    return np.abs(a - r).max(1) / (np.abs(r).max(1) + 1e-3 * np.abs(r).max())


def truncating_sizes():
    """Upstream's backward takes the image size as int(focal * tanfov * 2), focal = size / (2 tanfov), in float32 (SR_BACKWARD_WH_FROM_FOCAL,
    shipped = 1): for which sizes that is size - 1.  tanfov arrives as the reference computes it, math.tan(fov / 2) with
    fov = 2 atan(size / (2 focal)) [REF /root/reference/utils/graphics_utils.py:81-85, gaussian_renderer/__init__.py:35-36]."""
    def truncates(size, focal):
        t = np.float32(math.tan(0.5 * (2.0 * math.atan(size / (2.0 * focal)))))
        f = np.float32(size) / (np.float32(2.0) * t)
        return int(np.float32(np.float32(f * t) * np.float32(2.0))) == size - 1
    named = {"C1 256x256 fx=0.8W": [(256, 204.8), (256, 204.8)], "C2/C3/C4 1920x1080 fx=0.8W": [(1920, 1536.0), (1080, 1536.0)],
             "C5 3840x2160 fx=0.8W": [(3840, 3072.0), (2160, 3072.0)],
             "Waymo front camera 1920x1280 f~2060": [(1920, 2060.0), (1280, 2060.0)], "the same at -r 4 (480x320)": [(480, 515.0), (320, 515.0)]}
    out = {"named": {k: {"width_truncates": bool(truncates(*v[0])), "height_truncates": bool(truncates(*v[1]))} for k, v in named.items()}}
    sizes = list(range(64, 4097)); factors = [0.5 + 0.01 * i for i in range(151)]   # focal = factor * size
    hits = sum(truncates(sz, f * sz) for sz in sizes for f in factors)
    out["grid"] = dict(sizes="64..4096", focal_over_size="0.50..2.00 step 0.01", pairs=len(sizes) * len(factors), truncating=int(hits),
                       fraction=hits / (len(sizes) * len(factors)))
    out["effect"] = "where it truncates, K8's viewport uses W - 1 (H - 1): dL_dmeans2D (the densification proxy) scales by (W - 1) / W, the dL/dT chain shifts by half a pixel"
    return out


def upstream_semantics(P, W, H, fwd, raw, hip_color):
    """What the two upstream rules the shipped build follows (include/surfel_switches.h: SR_REFERENCE_PZ_SKIP = 1, SR_BACKWARD_WH_FROM_FOCAL = 1)
    change on this scene: the (pixel, splat) pairs the per-pair `if (p.z == 0) continue` removes -- in the oracle's (k x l).z and in the
    kernels' staged cross product (counted against the pz_zero_through_filter build of the kernels, run in a subprocess) -- and the image sizes
    upstream's int(focal * tanfov * 2) truncates."""
    import subprocess, tempfile
    from oracle import surfel_oracle as so
    from streetunveiler_amd import build as sb
    out = {"oracle_pz_census": so.pz_zero_census(fwd), "oracle_pz_census_up_to_the_kernels_last_contributor": so.pz_zero_census(fwd, raw["img"]["n_contrib"].view(np.uint32))}
    lib = sb.build_variant("pz_zero_through_filter")   # (idempotent: compiles only if the variant is older than a source)
    with tempfile.TemporaryDirectory() as d:
        f = os.path.join(d, "v.npz")
        env = dict(os.environ, SURFEL_RASTER_LIB=lib)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dump_decisions.py"), str(P), str(W), str(H), f], env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        v = np.load(f)
        assert int(v["switches"]) == 64
        only_variant = v["valid"] & ~raw["decisions"]["valid"]; only_shipped = raw["decisions"]["valid"] & ~v["valid"]
        pop = lambda a: int(np.unpackbits(a.view(np.uint8)).sum())
        dcol = np.abs(v["color"].astype(np.float64) - hip_color)
        out["kernels"] = dict(pairs_the_pz_skip_removes=pop(only_variant), pairs_valid_only_with_the_skip=pop(only_shipped),
                              pixels_whose_colour_differs=int((dcol.max(0) > 0).sum()), max_colour_difference=float(dcol.max()),
                              pixels_with_other_last_contributor=int((v["n_contrib"][0] != raw["img"]["n_contrib"].view(np.uint32)[0]).sum()),
                              note="pairs counted over the WHOLE lists (the decision dump does not stop at the pixel's last contributor)")
    out["backward_image_size"] = truncating_sizes()
    return out


def report(P, W, H, seed=0, deg=3, aux=True, scale_lo=5e-4, scale_hi=5e-3, eps=None):
    from oracle import surfel_oracle as so
    from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians, synthetic_upstream_grads
    from tests.gpu_util import k8_term_magnitudes, run_hip, run_hip_raw
    cam = synthetic_camera(W, H)
    g = synthetic_gaussians(P, W, H, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi)
    dc, da = synthetic_upstream_grads(W, H, seed=1, aux=aux)
    bg = np.zeros(3, np.float32)
    n = lambda k: g[k].numpy()
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(), bg=bg,
              image_width=W, image_height=H, sh_degree=deg, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    orc = lambda **k: so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), shs=n("shs"), **kw, **k)
    sec = {}
    t = time.time(); hip = run_hip(g, cam, bg, deg, dc, da); raw = run_hip_raw(g, cam, bg, deg, decisions=True); sec["hip"] = round(time.time() - t, 2)
    t = time.time(); fwd = orc(); bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy()); m32 = so.render_margins(fwd, eps); sec["oracle_f32"] = round(time.time() - t, 2)
    forced = dict(valid=raw["decisions"]["valid"], use3d=raw["decisions"]["use3d"], n_contrib=raw["img"]["n_contrib"].view(np.uint32))
    t = time.time(); ffwd = orc(forced=forced, reuse=fwd); fbwd = so.rasterize_backward(ffwd, dc.numpy(), da.numpy()); sec["forced_f32"] = round(time.time() - t, 2)
    t = time.time(); dfwd = orc(forced=forced, f64=True, reuse=fwd); dbwd = so.rasterize_backward(dfwd, dc.numpy(), da.numpy()); sec["forced_f64"] = round(time.time() - t, 2)
    t = time.time(); xfwd = orc(f64=True, reuse=fwd); xbwd = so.rasterize_backward(xfwd, dc.numpy(), da.numpy()); m64 = so.render_margins(xfwd, eps, f64=True, kernel_decisions=raw["decisions"]); sec["free_f64"] = round(time.time() - t, 2)
    assert np.array_equal(ffwd["n_contrib"], forced["n_contrib"]) and np.array_equal(dfwd["n_contrib"], forced["n_contrib"])
    hip_nc = raw["img"]["n_contrib"].view(np.uint32)
    vis = fwd["radii"] > 0
    rob_px = m64["pixel"] > 1.0; rob_med = rob_px & (m64["median"] > 1.0); rob_g = vis & (m64["gaussian"] > 1.0)
    out = {"config": dict(gaussians=P, width=W, height=H, sh_degree=deg, aux_gradients=aux, seed=seed, D=int(fwd["num_rendered"]), visible=int(vis.sum())),
           "eps": m64["eps"], "seconds": sec, "radii_equal": bool(np.array_equal(hip["radii"], fwd["radii"])),
           "binning_bit_exact": bool(raw["D"] == fwd["num_rendered"] and np.array_equal(raw["bin"]["point_list"].view(np.uint32), fwd["point_list"])
                                     and np.array_equal(raw["bin"]["ranges"].view(np.uint32), fwd["ranges"])),
           "definition": "image error = |a - b| / (1 + |b|) per element; gradient error = max_j |a - b|[row, j] / (max_j |b|[row, j] + 1e-3 * tensor max) "
                         "per visible Gaussian row; '64' references run K8 in float64 as well as the blend"}
    out["build_switches"] = dict(kernels=int(__import__("streetunveiler_amd._lib", fromlist=["load"]).load().sr_build_switches()), oracle=int(so.build_switches()),
                                 meaning="0 = every named switch at upstream's value (include/surfel_switches.h)")
    if seed == 0 and scale_lo == 5e-4 and scale_hi == 5e-3 and deg == 3:   # (the subprocess renders the benchmark scene)
        out["upstream_semantics"] = upstream_semantics(P, W, H, fwd, raw, hip["color"])
    # every per-pair decision of the kernels (contribute or not, which path), pixel by pixel up to where the float64 walk stops, against the
    # float64 checker's own: at robust pixels the "forced decisions" of section 1 are the arbiter's own decisions
    dis = m64["disagree"]
    out["pair_decisions_vs_free_f64"] = dict(pairs_differing_at_robust_pixels=int(dis[rob_px].sum()), pairs_differing_at_non_robust_pixels=int(dis[~rob_px].sum()),
                                             non_robust_pixels=int((~rob_px).sum()), non_robust_pixels_with_a_differing_pair=int((dis[~rob_px] > 0).sum()))
    # ---- 1. forced decisions vs the float64 arbiter (blend AND K8 in double) ----
    fz = {"images": {}, "gradients": {}}
    for name, a, b, d in [("color", hip["color"], ffwd["color"], dfwd["color"])] + [(f"allmap[{c}]", hip["allmap"][c], ffwd["allmap"][c], dfwd["allmap"][c]) for c in range(7)]:
        eh = np.abs(a.astype(np.float64) - d) / (1.0 + np.abs(d)); eo = np.abs(b.astype(np.float64) - d) / (1.0 + np.abs(d))
        fz["images"][name] = dict(hip=quantiles(eh), oracle_f32=quantiles(eo), hip_over_1e4=int((eh > 1e-4).sum()), oracle_f32_over_1e4=int((eo > 1e-4).sum()))
    for key in GRADS:
        ref = dbwd.get(key + "64", dbwd[key])
        fz["gradients"][key] = dict(tensor_max=float(np.abs(ref).max()), hip_rows=quantiles(row_err(hip[key], ref)[vis]), oracle_f32_rows=quantiles(row_err(fbwd[key], ref)[vis]),
                                    f32_k8_on_exact_sums_rows=quantiles(row_err(dbwd[key], ref)[vis]))
    # what sets the level of scales / rotations: rows against the magnitude of the terms K8 sums
    ms, mr, _mm = k8_term_magnitudes(g, cam, dbwd["dL_dtransMat64"])
    for key, mag in (("dL_dscales", ms), ("dL_drotations", mr)):
        ref = dbwd[key + "64"].reshape(P, -1)
        res = np.abs(ref).max(1) + 1e-3 * np.abs(ref).max()
        for who, got in (("hip", hip[key]), ("oracle_f32", fbwd[key])):
            d = np.abs(np.asarray(got, np.float64).reshape(P, -1) - ref).max(1)
            fz["gradients"][key][who + "_rows_vs_term_magnitude"] = quantiles((d / (mag + 1e-3 * np.abs(ref).max()))[vis])
        fz["gradients"][key]["cancellation_factor_rows"] = quantiles((mag / res)[vis])
    out["forced_vs_f64"] = fz
    # ---- 2. free-running float64 reference, float64 margins ----
    fr = {"robust_fraction": dict(pixels=float(rob_px.mean()), pixels_incl_median=float(rob_med.mean()), visible_gaussians=float(rob_g.sum() / max(1, vis.sum()))),
          "decisions": dict(robust_pixels_with_other_last_contributor=int((hip_nc[0] != xfwd["n_contrib"][0])[rob_px].sum()),
                            robust_pixels_with_other_median=int((hip_nc[1] != xfwd["n_contrib"][1])[rob_med].sum()),
                            pixels_with_other_last_contributor=int((hip_nc[0] != xfwd["n_contrib"][0]).sum()),
                            pixels_with_other_median=int((hip_nc[1] != xfwd["n_contrib"][1]).sum()), pixels_total=int(W * H)),
          "images": {}, "gradients": {}}
    for name, a, b, rob in [("color", hip["color"], xfwd["color"], rob_px)] + [(f"allmap[{c}]", hip["allmap"][c], xfwd["allmap"][c], rob_med if c == 5 else rob_px) for c in range(7)]:
        err = np.abs(a.astype(np.float64) - b) / (1.0 + np.abs(b)); r = np.broadcast_to(rob, err.shape)
        fr["images"][name] = dict(robust=quantiles(err[r]), non_robust=quantiles(err[~r]), over_1e4_robust=int((err[r] > 1e-4).sum()), over_1e4_non_robust=int((err[~r] > 1e-4).sum()))
    for key in GRADS:
        ref = xbwd.get(key + "64", xbwd[key]); e = row_err(hip[key], ref)
        tens = np.abs(np.asarray(hip[key], np.float64).reshape(P, -1) - np.asarray(ref, np.float64).reshape(P, -1)).max(1) / (np.abs(ref).max() + 1e-30)
        fr["gradients"][key] = dict(rows_robust=quantiles(e[rob_g]), rows_non_robust=quantiles(e[vis & ~rob_g]), rel_tensor_max_non_robust=quantiles(tens[vis & ~rob_g]),
                                    invisible_rows_nonzero=int((np.abs(np.asarray(hip[key]).reshape(P, -1)[~vis]).max(1) > 0).sum()) if (~vis).any() else 0)
    out["free_vs_f64"] = fr
    # ---- 3. free-running float32 oracle, its own float32 margins (as in round 2) ----
    r32 = m32["pixel"] > 1.0; r32m = r32 & (m32["median"] > 1.0); r32g = vis & (m32["gaussian"] > 1.0)
    f3 = {"robust_fraction": dict(pixels=float(r32.mean()), pixels_incl_median=float(r32m.mean()), visible_gaussians=float(r32g.sum() / max(1, vis.sum()))),
          "decisions_differ": dict(pixels_last_contributor=int((hip_nc[0] != fwd["n_contrib"][0]).sum()), pixels_median_contributor=int((hip_nc[1] != fwd["n_contrib"][1]).sum())),
          "images": {}, "gradients": {}}
    for name, a, b, rob in [("color", hip["color"], fwd["color"], r32)] + [(f"allmap[{c}]", hip["allmap"][c], fwd["allmap"][c], r32m if c == 5 else r32) for c in range(7)]:
        err = np.abs(a.astype(np.float64) - b) / (1.0 + np.abs(b)); r = np.broadcast_to(rob, err.shape)
        f3["images"][name] = dict(robust=quantiles(err[r]), over_1e4_robust=int((err[r] > 1e-4).sum()), over_1e4_non_robust=int((err[~r] > 1e-4).sum()))
    for key in GRADS:
        e = row_err(hip[key], bwd[key]); f3["gradients"][key] = dict(rows_robust=quantiles(e[r32g]), rows_non_robust=quantiles(e[vis & ~r32g]))
    out["free_vs_f32_oracle"] = f3
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=500_000); ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-aux", action="store_true"); ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_parity.json"))
    a = ap.parse_args()
    r = report(a.gaussians, a.width, a.height, aux=not a.no_aux)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(r, open(a.out, "w"), indent=1)
    print(json.dumps({k: r[k] for k in ("config", "radii_equal", "binning_bit_exact", "seconds", "build_switches")}))
    print("UPSTREAM SEMANTICS", json.dumps(r.get("upstream_semantics")))
    for k, v in r["forced_vs_f64"]["images"].items():
        print("FORCED vs F64", k, "hip max/p999", v["hip"]["max"], v["hip"]["p999"], ">1e-4:", v["hip_over_1e4"], "| oracle_f32", v["oracle_f32"]["max"], ">1e-4:", v["oracle_f32_over_1e4"])
    for k, v in r["forced_vs_f64"]["gradients"].items():
        print("FORCED vs F64", k, "hip rows max/p999/p99", v["hip_rows"]["max"], v["hip_rows"]["p999"], v["hip_rows"]["p99"], "| oracle_f32", v["oracle_f32_rows"]["max"], v["oracle_f32_rows"]["p999"],
              "| f32 K8 on exact sums", v["f32_k8_on_exact_sums_rows"]["max"], v["f32_k8_on_exact_sums_rows"]["p999"])
        if "hip_rows_vs_term_magnitude" in v:
            print("      vs term magnitude: hip", v["hip_rows_vs_term_magnitude"], "| oracle_f32", v["oracle_f32_rows_vs_term_magnitude"], "| cancellation", v["cancellation_factor_rows"])
    f = r["free_vs_f64"]
    print("FREE vs F64", f["robust_fraction"], f["decisions"])
    for k, v in f["images"].items():
        print("FREE vs F64", k, "robust", v["robust"], "| >1e-4: robust", v["over_1e4_robust"], "non-robust", v["over_1e4_non_robust"], "non-robust max", v["non_robust"].get("max"))
    for k, v in f["gradients"].items():
        print("FREE vs F64", k, "rows robust", v["rows_robust"], "| non-robust", v["rows_non_robust"], "| non-robust / tensor max", v["rel_tensor_max_non_robust"].get("max"))
