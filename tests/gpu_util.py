"""Helpers shared by the -m gpu parity tests: run the HIP operator through the drop-in package (i.e. through
the C-ABI) and the CPU oracle on identical seeded inputs."""
import math

import numpy as np
import torch

from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C
from oracle import surfel_oracle as so
from tests.bars import BARS as _BAR_TABLE, bar

DEV = "cuda:0"
BARS = {k: v[0] for k, v in _BAR_TABLE.items()}   # name -> value: THE tolerances (tests/bars.py: frozen, one table, a changelog rule)


def settings_for(cam, bg, deg, debug=False, dev=DEV):
    return GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                         torch.as_tensor(bg, dtype=torch.float32).to(dev), 1.0, cam.world_view_transform.to(dev),
                                         cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, debug)


def run_oracle(g, cam, bg, deg, dc=None, da=None, mode="sh", colors=None, Tpre=None, tile=(16, 16)):
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.numpy(), bg=np.asarray(bg, np.float32), image_width=cam.image_width,
              image_height=cam.image_height, sh_degree=deg, tile=tile, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    n = lambda k: g[k].numpy()
    if Tpre is not None:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), shs=n("shs") if colors is None else None,
                                   colors_precomp=colors, transMat_precomp=Tpre, **kw)
    elif colors is not None:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), colors_precomp=colors, **kw)
    else:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), shs=n("shs"), **kw)
    bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy()) if dc is not None else None
    return fwd, bwd


def run_hip(g, cam, bg, deg, dc=None, da=None, colors=None, Tpre=None, debug=False, tile=None, quadrant_cull=True, row_mapped=None):
    """Returns dict with outputs, internal state views and (if dc given) input gradients, all numpy."""
    dev = DEV
    s = settings_for(cam, bg, deg, debug)
    P = g["means3D"].shape[0]
    t = {k: v.to(dev).requires_grad_() for k, v in g.items()}
    means2D = torch.zeros(P, 3, device=dev, requires_grad=True)
    kw = dict(means3D=t["means3D"], means2D=means2D, opacities=t["opacities"])
    if colors is not None:
        t["colors"] = torch.as_tensor(colors).to(dev).requires_grad_(); kw["colors_precomp"] = t["colors"]
    else:
        kw["shs"] = t["shs"]
    if Tpre is not None:
        t["Tpre"] = torch.as_tensor(Tpre).to(dev).requires_grad_(); kw["cov3D_precomp"] = t["Tpre"]
    else:
        kw["scales"] = t["scales"]; kw["rotations"] = t["rotations"]
    color, radii, allmap = GaussianRasterizer(s, tile=tile, quadrant_cull=quadrant_cull, row_mapped=row_mapped)(**kw)
    out = dict(color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(), allmap=allmap.detach().cpu().numpy())
    if dc is not None:
        ((color * dc.to(dev)).sum() + (allmap * da.to(dev)).sum()).backward()
        torch.cuda.synchronize()
        z = lambda x: None if x.grad is None else x.grad.cpu().numpy()
        out.update(dL_dmeans3D=z(t["means3D"]), dL_dopacity=z(t["opacities"]), dL_dmeans2D=z(means2D))
        if colors is None: out["dL_dsh"] = z(t["shs"])
        else: out["dL_dcolors"] = z(t["colors"])
        if Tpre is None: out.update(dL_dscales=z(t["scales"]), dL_drotations=z(t["rotations"]))
        else: out["dL_dtransMat"] = z(t["Tpre"])
    return out


def run_hip_raw(g, cam, bg, deg, colors=None, Tpre=None, tile=None, quadrant_cull=True, decisions=False, row_mapped=None):
    """Calls _C.rasterize_gaussians directly and returns the state-buffer views as numpy (for bit-exact checks)."""
    dev = DEV
    s = settings_for(cam, bg, deg)
    e = torch.empty(0, device=dev)
    d = lambda k: g[k].to(dev)
    P = g["means3D"].shape[0]
    sh = d("shs") if colors is None else e
    col = e if colors is None else torch.as_tensor(colors).to(dev)
    sc, ro = (d("scales"), d("rotations")) if Tpre is None else (e, e)
    tp = e if Tpre is None else torch.as_tensor(Tpre).to(dev)
    D, color, allmap, radii, geom, binning, img = _C.rasterize_gaussians(
        s.bg, d("means3D"), col, d("opacities"), sc, ro, 1.0, tp, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
        s.image_height, s.image_width, sh, deg, s.campos, False, False, tile=tile, quadrant_cull=quadrant_cull, row_mapped=row_mapped)
    dec = None
    if decisions:   # the hard decisions the blend kernels act on, per (list entry, pixel) pair (sr_debug_pair_decisions)
        valid, use3d = _C.pair_decisions(s.bg, d("means3D"), 1.0, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width,
                                         deg, s.campos, geom, D, binning, tile=tile)
        dec = dict(valid=valid.cpu().numpy().view(np.uint64), use3d=use3d.cpu().numpy().view(np.uint64))
    torch.cuda.synchronize()
    W, H = cam.image_width, cam.image_height
    gv = {k: v.cpu().numpy() for k, v in _C.geom_view(geom, P).items()} if P else {}
    bv = {k: v.cpu().numpy() for k, v in _C.binning_view(binning, P, D, W, H, tile or (16, 16)).items()}
    iv = {k: v.cpu().numpy() for k, v in _C.image_view(img, W, H).items()}
    return dict(D=D, color=color.cpu().numpy(), allmap=allmap.cpu().numpy(), radii=radii.cpu().numpy(), geom=gv, bin=bv, img=iv, decisions=dec)


def assert_close_frac(a, b, atol, rtol, max_bad_frac, hard, name=""):
    """|a-b| <= atol + rtol*|b| for all but `max_bad_frac` of the elements (alpha>=1/255 and T<1e-4 are hard
    thresholds: an ulp difference in exp()/rcp() can flip a contributor); every element within `hard`."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    err = np.abs(a - b)
    bad = err > (atol + rtol * np.abs(b))
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_bad_frac, f"{name}: {frac:.3e} of elements off by more than tol (max err {err.max():.3e})"
    if hard is not None:
        assert (err <= hard * max(1.0, np.abs(b).max())).all(), f"{name}: max err {err.max():.3e} exceeds hard bound"


def assert_grads_close(got, ref, rel, name="", max_bad_frac=bar("oracle32_grad_bad_frac"), hard=bar("oracle32_grad_hard")):
    """Per tensor, relative to its largest magnitude: all but `max_bad_frac` of the elements within `rel`, every
    element within `hard`.  The slack for a few elements is the same threshold-flip effect as in the images: one
    (pixel, splat) pair crossing alpha = 1/255 or T = 1e-4 because exp()/rcp() differ by an ulp moves that
    Gaussian's gradient by a finite amount; everything else differs only by float summation order."""
    got = np.asarray(got, np.float64).reshape(np.asarray(ref).shape); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max() + 1e-20
    err = np.abs(got - ref) / scale
    bad = (err > rel).mean() if err.size else 0.0
    assert bad <= max_bad_frac, f"{name}: {bad:.2e} of elements off by more than {rel} of scale {scale:.3e} (max rel err {err.max():.2e})"
    if hard is not None:
        assert err.max() <= hard, f"{name}: max rel err {err.max():.2e} exceeds hard bound {hard}"


def check_allmap(got, ref, tag, max_bad_frac=bar("oracle32_image_bad_frac_small"), hard=bar("oracle32_image_hard")):
    """allmap parity: channels 0-4 and 6 are sums (a flipped contributor moves them by <= 1/255-ish of the scale);
    channel 5 (median depth) is a SELECTION -- the depth of the last contributor with T > 0.5 -- so a pixel whose T
    crosses 0.5 within float noise legitimately jumps to another splat's depth: it only gets the fraction criterion."""
    got = np.asarray(got); ref = np.asarray(ref)
    keep = [0, 1, 2, 3, 4, 6]
    tol = bar("oracle32_image_atol")
    assert_close_frac(got[keep], ref[keep], tol, tol, max_bad_frac, hard, tag + " allmap[sums]")
    assert_close_frac(got[5], ref[5], tol, tol, max_bad_frac, None, tag + " allmap[median depth]")


# ---- strict parity: same hard decisions on both sides, float64 arbiter -----------------------------------------------
def forced_f64_reference(g, cam, bg, deg, dc=None, da=None, tile=None, colors=None, base=None, raw=None, f64=True):
    """The blend AND K8 evaluated in double precision (oracle/surfel_blend.inc, surfel_k8.inc, REAL = double) on the float32 per-Gaussian
    state of the oracle's K1, with the hard decisions the HIP kernels took (sr_debug_pair_decisions + n_contrib).  What differs from
    the HIP result is rounding only.  `base` = an oracle forward of the same scene (its K1 + binning are reused), `raw` = a
    run_hip_raw(..., decisions=True) of it.  -> (raw HIP state incl. decisions, forward dict, backward dict or None).
    `f64=False`: the float32 oracle with the same forced decisions -- what float32 arithmetic itself loses on this scene."""
    if raw is None:
        raw = run_hip_raw(g, cam, bg, deg, colors=colors, tile=tile, decisions=True)
    forced = dict(valid=raw["decisions"]["valid"], use3d=raw["decisions"]["use3d"], n_contrib=raw["img"]["n_contrib"].view(np.uint32))
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
              bg=np.asarray(bg, np.float32), image_width=cam.image_width, image_height=cam.image_height, sh_degree=deg, tile=tile or (16, 16),
              forced=forced, f64=bool(f64), reuse=base, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    n = lambda k: g[k].numpy()
    if colors is not None:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), colors_precomp=colors, **kw)
    else:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), shs=n("shs"), **kw)
    assert np.array_equal(fwd["n_contrib"], forced["n_contrib"])
    bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy()) if dc is not None else None
    return raw, fwd, bwd


def row_errors(got, ref, vis):
    """Per Gaussian row: max_j |got - ref| / (max_j |ref| + 1e-3 * max |ref| over the tensor), visible rows only.  (A tensor that is
    identically zero in the reference -- e.g. the densification proxy of a scene blended by the screen-space filter alone -- must be
    identically zero: 0 / 0 counts as 0, anything else as inf.)"""
    ref = np.asarray(ref, np.float64); P = ref.shape[0]
    ref = ref.reshape(P, -1); got = np.asarray(got, np.float64).reshape(P, -1)
    num = np.abs(got - ref).max(1); den = np.abs(ref).max(1) + bar("row_floor") * np.abs(ref).max()
    with np.errstate(divide="ignore", invalid="ignore"):
        e = np.where(num == 0, 0.0, num / den)
    return e[vis]


def k8_term_magnitudes(g, cam, dT):
    """Magnitude of the terms K8 sums into dL_dscales / dL_drotations: Appendix A.6 with every product replaced by its absolute value,
    on the float64 transMat gradient dT [P,9] (rows Tu | Tv | Tw).  dL_dscales = R^T (B^T dT) has up to 800x cancellation between
    those terms on the benchmark scenes (profiles/r03_parity_c3.json), so a float32 rounding of the INPUT sums -- either
    implementation's -- shows up in these two tensors amplified by that factor; K8's own float32 arithmetic does not (float32 K8 on
    exact sums: p99.9 2e-6).  Rows of these two tensors are therefore measured against this magnitude; so are the rows of dL_dmeans3D, whose
    three components are sums of the same kind (no cancellation to speak of under the unrotated benchmark camera, some under a rotated one:
    the worst of 2.5 M rows of C4's camera 7 sits at 1.1e-2 of its own magnitude).  -> (scales[P], rotations[P], means3D[P])."""
    W, H = cam.image_width, cam.image_height
    proj = cam.full_proj_transform.numpy().astype(np.float64).reshape(16)
    B = np.zeros((3, 4))
    for k in range(4):
        a0, a1, a3 = proj[4 * k], proj[4 * k + 1], proj[4 * k + 3]
        B[0, k] = 0.5 * W * a0 + 0.5 * (W - 1) * a3; B[1, k] = 0.5 * H * a1 + 0.5 * (H - 1) * a3; B[2, k] = a3
    Ba = np.abs(B[:, :3])                                         # [r, k]
    dTa = np.abs(np.asarray(dT, np.float64)).reshape(-1, 3, 3)   # [P, r, c]
    dL0 = np.einsum("rk,pr->pk", Ba, dTa[:, :, 0]); dL1 = np.einsum("rk,pr->pk", Ba, dTa[:, :, 1])
    q = g["rotations"].numpy().astype(np.float64); r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                  2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    Ra = np.abs(R)
    s = g["scales"].numpy().astype(np.float64)
    dscale = np.maximum((dL0 * Ra[:, :, 0]).sum(1), (dL1 * Ra[:, :, 1]).sum(1))
    drot = 2.0 * np.abs(q).max(1) * (dL0 * s[:, :1] + dL1 * s[:, 1:2]).sum(1)   # (the normal column of dL/dR is not a cancellation source)
    # dL_dmeans3D[k] = sum_r B[r][k] dT[r][2] (+ the SH direction term): three products per component under the unrotated benchmark camera
    # (B is then sparse), nine cancelling ones once the view matrix carries a rotation (BASELINE config 4's yawed cameras)
    dmean = np.einsum("rk,pr->pk", Ba, dTa[:, :, 2]).max(1)
    return dscale, drot, dmean


def gradient_row_errors(hip, bwd64, vis, scene=None):
    """{tensor: per-row error over the visible Gaussians} of the HIP gradients against a float64 backward (blend AND K8 in double where
    the reference carries "<name>64").  Rows are relative to their own magnitude (row_errors); dL_dscales / dL_drotations -- with
    `scene` = (g, cam) -- to the magnitude of the terms they sum (k8_term_magnitudes)."""
    out = {}
    mags = None
    if scene is not None and "dL_dtransMat64" in bwd64 and "scales" in scene[0]:
        ms, mr, mm = k8_term_magnitudes(scene[0], scene[1], bwd64["dL_dtransMat64"])
        mags = {"dL_dscales": ms, "dL_drotations": mr, "dL_dmeans3D": mm}
    for key in STRICT_ROW_BARS:
        ref = bwd64.get(key + "64", bwd64.get(key))
        if key not in hip or hip[key] is None or ref is None:
            continue
        if mags is not None and key in mags:
            P = ref.shape[0]
            r = np.asarray(ref, np.float64).reshape(P, -1); a = np.asarray(hip[key], np.float64).reshape(P, -1)
            num, den = np.abs(a - r).max(1), np.maximum(mags[key], np.abs(r).max(1)) + bar("row_floor") * np.abs(r).max()
            # (a frame whose visible Gaussians reach no pixel has all-zero gradients on both sides: 0 / 0 is "no error", anything / 0 is inf)
            out[key] = np.divide(num, den, out=np.where(num == 0, 0.0, np.inf), where=den > 0)[vis]
        else:
            out[key] = row_errors(hip[key], ref, vis)
    return out


# Gradient bars, per Gaussian row: 99.9 % of the rows / every row.  Reference = the all-float64 backward.  Every tensor sits at the
# float32 rounding of the blend's sums (measured at C2 / C3, profiles/r03_parity_*.json: p99.9 <= 2.5e-5, max 3.3e-3), dL_dscales and
# dL_drotations included once their rows are measured against the terms they sum (p99.9 6.0e-5 / 6.6e-5, max 5.2e-4 / 3.6e-3 at C2 since
# K7's moments are taken about the Gaussian's own centre; the float32 oracle, forced to the same decisions, shows 1.6e-4 / 2.0e-3 against
# the same arbiter).  Without `scene` (precomputed transMat: no scale / rotation chain) only the plain row metric applies.  dL_dmeans2D
# (the densification proxy) is ONE element of dL/dT times Tw.z * W / 2 -- a single float32 sum of cancelling terms rather than a row
# maximum -- and sits at p99.9 4e-5 on the benchmark scenes, 4e-4 under a train.py-style loss that weights the distortion map by 100
# (tests/test_gpu_render_api.py).
STRICT_ROW_BARS = {k: (bar("row_p999_means2D") if k == "dL_dmeans2D" else bar("row_p999"), bar("row_max"))
                   for k in ("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dmeans2D", "dL_dcolors", "dL_dscales", "dL_drotations")}
PLAIN_ROW_BARS = (bar("row_plain_p999"), bar("row_plain_max"))   # dL_dscales / dL_drotations without a scene (plain row metric)


def rows_within(e, p999_bar, max_bar, outlier_frac=0.0, e32=None):
    """The row bars: every row within `max_bar`, and all but 0.1 % of the rows within `p999_bar` -- counted, with two rows allowed in any
    case (the 99.9th percentile of a few hundred rows is just their maximum: a 400-Gaussian scene would be held to `p999_bar` everywhere).
    `outlier_frac` > 0 (the full-size yawed-camera runs only): that fraction of the rows -- one in a million -- may sit between `max_bar`
    and 5 x `max_bar`.  Measured case: ONE of 2.57 M rows of C4's camera 7 (Gaussian 1490332: a one-pixel splat on the corner of four
    tiles) at 1.09e-2 in dL_dmeans3D and 3.3e-2 in the densification proxy dL_dmeans2D -- both carry dL/dTu.z, which K7's moment form
    (sum dp, sum x dp, sum y dp, crossed with Tv / Tw once per Gaussian in K8) obtains as a difference of sums that the per-pixel cross
    product of the reference's formulation never forms (tools/worst_row.py: the float32 oracle has this row at 2.7e-4, and 48 OTHER rows of
    the same frame above 1e-2, 0.32 at worst, where the kernels are below 2e-3).
    `e32` (the randomised sweeps only) = the float32 oracle's error on the same rows under the same decisions: a row above `max_bar` is
    accepted where the oracle's own error is at least half of it -- the splat itself is ill-conditioned in float32, whichever way the sums
    are formed.  Measured case: posed seed 100551, Gaussian 5397 (a 9:1 splat at depth 12.5): kernels 1.94e-2, float32 oracle 1.45e-2."""
    e = np.asarray(e)
    if e.size == 0:
        return True
    if e32 is not None:   # rows whose conditioning costs the float32 oracle at least half as much: held to twice ITS error instead of the bar
        e = np.where(e <= bar("row_oracle32_excuse") * np.asarray(e32), np.minimum(e, max_bar), e)
    over = int((e > max_bar).sum())
    return bool(e.max() <= (bar("row_outlier_factor") * max_bar if outlier_frac > 0 else max_bar) and over <= int(np.ceil(outlier_frac * e.size))
                and int((e > p999_bar).sum()) <= max(2, int(np.ceil(bar("row_p999_fraction") * e.size))))


def assert_strict_parity(hip, fwd64, bwd64=None, tag="", report=None, scene=None, oracle32=None, oracle32_fwd=None, value_noise=None, outlier_frac=0.0,
                         elementwise32=None):
    """Images: |hip - f64| <= 1e-4 * (1 + |f64|) for EVERY element of colour and all seven aux maps -- north_star's tolerance, no
    exempt fraction.  Gradients: STRICT_ROW_BARS (scene = (g, cam) switches dL_dscales / dL_drotations to the term-magnitude metric).
    `oracle32` = the float32 oracle's backward under the SAME forced decisions: a row bar then reads "within the bar, or no worse than the
    float32 restatement of the reference on this scene" (random ill-conditioned scenes of the fuzz sweep).
    `elementwise32` = the float32 oracle's FORWARD under the same forced decisions (the full-size general-pose run): an image element may
    exceed 1e-4 only where the float32 restatement of the reference's formulation is further from float64 at that very element, and only
    one element in a million may (measured there: 1 of 20.7 M at 1.13e-4, the float32 oracle at 1.4e-2 on it and above 1e-4 on 1 137)."""
    for name, a, b in [("color", hip["color"], fwd64["color"]), ("allmap", hip["allmap"], fwd64["allmap"])]:
        err = np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b))
        if elementwise32 is not None:
            e32 = np.abs(np.asarray(elementwise32[name], np.float64) - b) / (1.0 + np.abs(b))
            if report is not None:
                report[f"{tag}{name} over 1e-4"] = dict(kernels=int((err > bar("robust_pixel")).sum()), float32_oracle=int((e32 > bar("robust_pixel")).sum()), kernels_max=float(err.max()), float32_oracle_max=float(e32.max()))
            assert (err <= np.maximum(bar("robust_pixel"), e32)).all() and int((err > bar("robust_pixel")).sum()) <= int(np.ceil(bar("elementwise_over_1e4_frac") * err.size)), \
                f"{tag} {name}: {int((err > 1e-4).sum())} elements over 1e-4 (max {err.max():.3e}); the float32 oracle: {int((e32 > 1e-4).sum())} (max {e32.max():.3e})"
            continue
        if report is not None:
            report[f"{tag}{name}"] = dict(max=float(err.max()), p999=float(np.quantile(err, 0.999)))
        vbar = bar("robust_pixel")
        if oracle32_fwd is not None:   # (fuzz sweep: "... or no worse than the float32 oracle under the same decisions")
            vbar = max(vbar, float((np.abs(np.asarray(oracle32_fwd[name], np.float64) - b) / (1.0 + np.abs(b))).max()))
        over = err if value_noise is None else err - np.broadcast_to(value_noise, err.shape)   # (per-pixel conditioning: assert_free_parity)
        assert over.max() <= vbar, f"{tag} {name}: max error {err.max():.3e} of (1 + |value|) with identical decisions (bar {vbar:.1e})"
    if bwd64 is None:
        return
    vis = fwd64["radii"] > 0
    errs32 = {} if oracle32 is None else gradient_row_errors({k: (v if hip.get(k) is not None else None) for k, v in oracle32.items()}, bwd64, vis, scene)
    for key, e in gradient_row_errors(hip, bwd64, vis, scene).items():
        p999_bar, max_bar = STRICT_ROW_BARS[key]
        if scene is None and key in ("dL_dscales", "dL_drotations"):
            p999_bar, max_bar = PLAIN_ROW_BARS   # plain row metric: the cancellation of the chain is in the number (see k8_term_magnitudes)
        if key in errs32 and errs32[key].size:
            p999_bar, max_bar = max(p999_bar, float(np.quantile(errs32[key], 0.999))), max(max_bar, float(errs32[key].max()))
        if report is not None:
            report[f"{tag}{key}"] = dict(max=float(e.max()), p999=float(np.quantile(e, 0.999)), p99=float(np.quantile(e, 0.99)))
        assert rows_within(e, p999_bar, max_bar, outlier_frac, e32=errs32[key] if key in errs32 and errs32[key].size else None), \
            f"{tag} {key}: row errors p99.9 {np.quantile(e, 0.999):.2e} (bar {p999_bar:.1e}), max {e.max():.2e} (bar {max_bar:.1e}), {int((e > max_bar).sum())} rows over it"


# ---- free-running parity: the checker takes its OWN decisions (float64), robust / non-robust classification ----------------
# Non-robust budget: fraction of pixels / visible Gaussians with some decision within the noise allowance of its threshold
# (oracle.surfel_oracle.DEFAULT_EPS, widened per pair by what float32 can know about an ill-conditioned ray-splat intersection and per
# pixel by the depth of its list: surfel_blend.inc so_render_margins).  Measured (profiles/r03_parity_*.json): C2 0.17 % of the pixels
# and 21 % of the visible Gaussians, C3 0.57 % and 19 % (one near-threshold pair anywhere in a Gaussian's footprint makes the whole row non-robust); the
# small test scenes with splats hundreds of pixels wide reach about 1 % / 30 %.  The full-size tests pass their own, tighter budgets.
NONROBUST_PIXEL_BUDGET = bar("nonrobust_pixel_budget")
NONROBUST_GAUSSIAN_BUDGET = bar("nonrobust_gaussian_budget")


def free_f64_reference(g, cam, bg, deg, dc=None, da=None, tile=None, colors=None, base=None, kernel_decisions=None):
    """The float64 blend + float64 K8 running FREE on the oracle's float32 per-Gaussian state: every hard decision (contribute,
    path, stop, median) is the checker's own, taken on exactly evaluated quantities -- nothing comes from the HIP kernels.  With the
    float64 decision margins (so_render_margins_f64) it says where two correct implementations MUST agree to rounding (robust
    pixels / Gaussians) and where a decision sits within float32 noise of its threshold.  -> (fwd64, bwd64 or None, margins).
    `kernel_decisions` = run_hip_raw(..., decisions=True)["decisions"]: the margins then carry `disagree`[H,W], the per-pixel number of
    pairs whose contribute / path decision in the kernels differs from the checker's own (assert_free_parity demands 0 at robust pixels)."""
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
              bg=np.asarray(bg, np.float32), image_width=cam.image_width, image_height=cam.image_height, sh_degree=deg, tile=tile or (16, 16),
              f64=True, reuse=base, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    n = lambda k: g[k].numpy()
    if colors is not None:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), colors_precomp=colors, **kw)
    else:
        fwd = so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), shs=n("shs"), **kw)
    bwd = so.rasterize_backward(fwd, dc.numpy(), da.numpy()) if dc is not None else None
    return fwd, bwd, so.render_margins(fwd, f64=True, kernel_decisions=kernel_decisions)


def _assert_everything_but_the_differing_pixels(hip, nc, fwd64, bwd64, margins, tag, rep, scene, value_slack, lenient, outlier_frac=0.0, differing_cap=bar("differing_pixel_frac")):
    """The robust / non-robust split PREDICTS where two correct float32 implementations may decide differently; this is the check on what
    actually happened.  A pixel DIFFERS if one of its pair decisions, its stopping entry or its median entry in the kernels is not the
    free-running float64 checker's.  Every other pixel -- robust or not -- saw the same contributor set on both sides and must meet the value
    bar; every visible Gaussian that is not in the list of a differing pixel (up to that pixel's deeper stop) must meet the strict row bars
    against the FREE float64 backward.  The differing pixels themselves are few: at most 1e-4 of the frame (measured: 32 of 2 M at C3)."""
    dis = margins["disagree"]
    H, W = dis.shape
    differs = (dis > 0) | (nc[0] != fwd64["n_contrib"][0]) | (nc[1] != fwd64["n_contrib"][1])
    frac = float(differs.mean())
    tw, th = fwd64["_inputs"]["tile"]
    gx = (W + tw - 1) // tw
    P = fwd64["radii"].shape[0]
    affected = np.zeros(P, bool)
    ranges = np.asarray(fwd64["ranges"]).reshape(-1, 2); plist = np.asarray(fwd64["point_list"])
    for py, px in zip(*np.nonzero(differs)):
        r0 = int(ranges[(py // th) * gx + px // tw][0]); r1 = int(ranges[(py // th) * gx + px // tw][1])
        deep = min(r1 - r0, int(max(nc[0][py, px], fwd64["n_contrib"][0][py, px])) + 1)
        affected[plist[r0:r0 + deep]] = True
    vis = fwd64["radii"] > 0
    if rep is not None:
        rep[f"{tag}differing_pixels"] = dict(pixels=int(differs.sum()), fraction=frac, gaussians_in_their_lists=int((affected & vis).sum()),
                                             fraction_of_visible=float((affected & vis).sum() / max(1, vis.sum())))
    # (... or, on a small frame most of which is non-robust, at most 0.5 % of the NON-ROBUST pixels -- the set the margin walk predicts a
    # differing decision can only come from.  Measured: C3 32 of 11 800, posed fuzz seed 100525 -- 235x47, 24 % of the frame non-robust,
    # lists 2 492 deep -- 4 of 2 664.)
    nonrobust = int((margins["pixel"] <= 1.0).sum())
    assert frac <= differing_cap or differs.sum() <= max(3, int(bar("differing_of_nonrobust") * nonrobust)), \
        f"{tag}: {int(differs.sum())} pixels ({frac:.2e} of the frame, {nonrobust} non-robust pixels) hold a decision that differs from the float64 checker's"
    if lenient:   # (fuzz sweep on ill-conditioned random scenes: its value bars are relative to the float32 oracle -- the robust-element checks carry them)
        return
    keep = ~differs
    for name, a, b in [("color", hip["color"], fwd64["color"])] + [(f"allmap[{c}]", hip["allmap"][c], fwd64["allmap"][c]) for c in range(7)]:
        err = np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b)) - np.broadcast_to(margins.get("value_noise", 0.0), np.shape(b))
        m = np.broadcast_to(keep, err.shape)
        assert err[m].max(initial=0.0) <= bar("robust_pixel") * value_slack, \
            f"{tag} {name}: a pixel with the checker's own decisions is off by {err[m].max():.3e} of (1 + |value|) against the free-running float64 reference"
    rows = vis & ~affected
    for key, e in gradient_row_errors(hip, bwd64, np.ones_like(vis), scene).items():
        p999_bar, max_bar = STRICT_ROW_BARS[key]
        if scene is None and key in ("dL_dscales", "dL_drotations"):
            p999_bar, max_bar = PLAIN_ROW_BARS
        er = e[rows]
        if rep is not None:
            rep[f"{tag}{key} rows outside the differing pixels"] = dict(rows=int(rows.sum()), max=float(er.max(initial=0.0)), p999=float(np.quantile(er, 0.999)) if er.size else 0.0)
        if er.size:
            assert rows_within(er, p999_bar * value_slack, max_bar * value_slack, outlier_frac), \
                f"{tag} {key}: rows outside the differing pixels' lists p99.9 {np.quantile(er, 0.999):.2e} (bar {p999_bar:.1e}), max {er.max():.2e} (bar {max_bar:.1e})"


def assert_free_parity(hip, hip_n_contrib, fwd64, bwd64, margins, tag="", report=None, scene=None, pixel_budget=NONROBUST_PIXEL_BUDGET,
                       gaussian_budget=NONROBUST_GAUSSIAN_BUDGET, value_slack=1.0, nonrobust_pixel_cap=bar("nonrobust_pixel_cap"), nonrobust_row_cap=bar("nonrobust_row_cap"),
                       oracle32=None, oracle32_fwd=None, outlier_frac=0.0, differing_cap=bar("differing_pixel_frac")):
    """HIP against the free-running float64 reference.
      * every ROBUST pixel: same last contributor, colour and the six summed aux maps within 1e-4 * (1 + |value|) -- no exempt
        fraction; where the median selection is robust too: same median contributor and median depth within the same bar;
      * every ROBUST visible Gaussian (no near-threshold decision anywhere in its footprint): the strict per-row bars against the
        all-float64 backward (blend AND K8 in double: "<name>64");
      * the non-robust remainder is counted against its measured fraction and only has to stay finite and within the loose bars
        of a flipped contributor (2e-2 per pixel, 5e-2 of the tensor scale per row).
    `value_slack` > 1 widens the VALUE bars of the robust elements (never the identical-decision checks): the fuzz sweep uses it for its
    camera-plane regime, where the ray-splat intersection itself is ill-conditioned in float32.
    `nonrobust_*_cap`: the loose bars of the non-robust remainder (None = only finite: a flipped decision can move a few-pixel splat's whole
    gradient -- the random scenes of the fuzz sweep check those elements with forced decisions instead).
    `differing_cap`: the largest fraction of the frame that may hold a decision differing from the checker's (all of them non-robust pixels --
    a robust one fails earlier): 1e-4 everywhere in the suite (C3: 1.5e-5); the FUZZ_BIG=4 sweep of tools/fuzz_parity.py, whose translucent
    regime puts tens of millions of duplicates on small frames (pixels thousands of contributors deep), passes 5e-4 (measured 1.3e-4 .. 2.3e-4).
    `oracle32` = the float32 oracle's backward on the same scene: a robust-row bar then reads "within the bar, OR at least twice as accurate
    as the float32 restatement of the reference on the same rows" (ill-conditioned random scenes -- translucent deep lists of large
    splats -- where float32 itself is 1e-2 off the float64 reference: tools/fuzz_diagnose.py); `oracle32_fwd` = its forward: the same
    rule for the value bar of the robust pixels."""
    rob_px = margins["pixel"] > 1.0
    rob_med = rob_px & (margins["median"] > 1.0)
    rep = {} if report is None else report
    assert (~rob_px).mean() <= pixel_budget, f"{tag}: {(~rob_px).mean():.2e} of the pixels are non-robust"
    if hip_n_contrib is not None:   # (the render()-level tests do not see the image state)
        nc = np.asarray(hip_n_contrib).view(np.uint32).reshape(2, *rob_px.shape)
        assert np.array_equal(nc[0][rob_px], fwd64["n_contrib"][0][rob_px]), f"{tag}: a robust pixel stops at a different entry"
        assert np.array_equal(nc[1][rob_med], fwd64["n_contrib"][1][rob_med]), f"{tag}: a robust pixel picks a different median"
    if "disagree" in margins:
        # every single (pixel, splat) decision of the kernels at a robust pixel -- contribute or not, ray-splat or screen-space path -- is the
        # float64 checker's own: the "forced decisions" of the strict bar are, at these pixels, not the kernels' word but the arbiter's
        dis = margins["disagree"]
        if report is not None:
            report[f"{tag}pair_decisions"] = dict(differing_pairs_at_robust_pixels=int(dis[rob_px].sum()), differing_pairs_at_non_robust_pixels=int(dis[~rob_px].sum()),
                                                  non_robust_pixels_with_a_differing_pair=int((dis[~rob_px] > 0).sum()))
        assert not dis[rob_px].any(), f"{tag}: {int((dis[rob_px] > 0).sum())} robust pixels hold a pair the kernels decided differently from the float64 checker"
        if hip_n_contrib is not None and bwd64 is not None:
            _assert_everything_but_the_differing_pixels(hip, nc, fwd64, bwd64, margins, tag, rep if report is not None else None, scene, value_slack,
                                                        oracle32 is not None or oracle32_fwd is not None, outlier_frac, differing_cap)
    for name, a, b, mask in [("color", hip["color"], fwd64["color"], rob_px)] + \
                            [(f"allmap[{c}]", hip["allmap"][c], fwd64["allmap"][c], rob_med if c == 5 else rob_px) for c in range(7)]:
        err = np.abs(np.asarray(a, np.float64) - b) / (1.0 + np.abs(b))
        m = np.broadcast_to(mask, err.shape)
        rep[f"{tag}{name}"] = dict(robust_max=float(err[m].max()) if m.any() else 0.0, non_robust_max=float(err[~m].max()) if (~m).any() else 0.0,
                                   non_robust_over_1e4=int((err[~m] > bar("robust_pixel")).sum()))
        assert np.isfinite(a).all(), f"{tag} {name}: non-finite output"
        vbar = bar("robust_pixel") * value_slack
        if oracle32_fwd is not None:
            o = oracle32_fwd["color"] if name == "color" else oracle32_fwd["allmap"][int(name[7])]
            eo = np.abs(np.asarray(o, np.float64) - b) / (1.0 + np.abs(b))
            vbar = max(vbar, float(eo[m].max(initial=0.0)))
        # ... plus, per pixel, what float32 rounding of ITS ray-splat intersections can put into its transmittance and weights (a contributor
        # in the middle of the list whose ray runs nearly parallel to its plane: no decision is near a threshold, the pixel is robust, and
        # its alpha still carries 1e-4 of relative noise -- so_render_margins' value_noise, zero for well-conditioned pixels)
        over = err - np.broadcast_to(margins.get("value_noise", 0.0), err.shape)
        assert over[m].max(initial=0.0) <= vbar, f"{tag} {name}: robust pixel off by {err[m].max():.3e} of (1 + |value|) against the free-running float64 reference (bar {vbar:.1e} + the pixel's conditioning)"
        if name != "allmap[5]" and nonrobust_pixel_cap is not None:   # (the median depth of a non-robust pixel is another splat's depth: a selection, not a sum)
            assert err[~m].max(initial=0.0) <= nonrobust_pixel_cap, f"{tag} {name}: non-robust pixel off by {err[~m].max():.3e}"
    if bwd64 is None:
        return
    vis = fwd64["radii"] > 0
    rob_g = vis & (margins["gaussian"] > 1.0)
    frac = 1.0 - rob_g.sum() / max(1, vis.sum()) if vis.any() else 0.0   # (a frame with no visible Gaussian has no non-robust one)
    assert frac <= gaussian_budget, f"{tag}: {frac:.2f} of the visible Gaussians are non-robust"
    errs = gradient_row_errors(hip, bwd64, np.ones_like(vis), scene)
    errs32 = {} if oracle32 is None else gradient_row_errors({k: (v if hip.get(k) is not None else None) for k, v in oracle32.items()}, bwd64,
                                                             np.ones_like(vis), scene)
    for key, e in errs.items():
        p999_bar, max_bar = STRICT_ROW_BARS[key]
        if scene is None and key in ("dL_dscales", "dL_drotations"):
            p999_bar, max_bar = PLAIN_ROW_BARS
        ref = bwd64.get(key + "64", bwd64.get(key)); P = ref.shape[0]
        r = np.asarray(ref, np.float64).reshape(P, -1); a = np.asarray(hip[key], np.float64).reshape(P, -1)
        loose = np.abs(a - r).max(1) / (np.abs(r).max() + 1e-30)
        er = e[rob_g]
        rep[f"{tag}{key}"] = dict(robust_rows=int(rob_g.sum()), robust_max=float(er.max(initial=0.0)), robust_p999=float(np.quantile(er, 0.999)) if er.size else 0.0,
                                  non_robust_max_of_tensor_scale=float(loose[vis & ~rob_g].max(initial=0.0)))
        assert np.isfinite(a).all(), f"{tag} {key}: non-finite gradient"
        if er.size:
            p999_eff, max_eff = p999_bar * value_slack, max_bar * value_slack
            if key in errs32 and errs32[key][rob_g].size:
                o = errs32[key][rob_g]
                p999_eff, max_eff = max(p999_eff, 0.5 * float(np.quantile(o, 0.999))), max(max_eff, float(o.max()))   # (worst row: no worse than the oracle's)
            assert rows_within(er, p999_eff, max_eff, outlier_frac, e32=errs32[key][rob_g] if key in errs32 and errs32[key][rob_g].size else None), \
                f"{tag} {key}: robust rows p99.9 {np.quantile(er, 0.999):.2e} (bar {p999_eff:.1e}), max {er.max():.2e} (bar {max_eff:.1e})"
        if nonrobust_row_cap is not None:
            assert loose[vis & ~rob_g].max(initial=0.0) <= nonrobust_row_cap, f"{tag} {key}: a non-robust row is off by {loose[vis & ~rob_g].max():.2e} of the tensor scale"
        assert not np.abs(a[~vis]).any(), f"{tag} {key}: gradient on an invisible Gaussian"
