#!/usr/bin/env python
"""Randomised self-consistency sweep of the shared-geometry extensions over random scenes, tile shapes and regimes
(tools/fuzz_parity.py's generator): each extension against the plain operator calls of THIS build it replaces -- which the parity
sweeps hold against the oracle:
  * 9 channels (SH colour + six precomputed) == the 3-channel SH call + the 6-channel call: colour / allmap / radii bit for bit,
    gradients = the sum of the two calls' gradients;
  * per-class distortion pass == `allmap[6]` of the operator on each class subset, gradients = the sum over the subsets;
  * one plan (forward_with_class_distortions, 9 channels + class maps on one K1 / binning / K8) == those two calls: maps bit for bit, gradients
    = their sum;
  * mask= == boolean-indexing every input first: images bit for bit, gradients scattered back;
  * fused activations == torch's sigmoid / exp / normalize in front of the operator (images bit for bit? no: K1 evaluates them itself --
    to 1e-5), gradients w.r.t. the raw parameters.
python tools/fuzz_extensions.py [n_scenes] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from diff_surfel_rasterization import GaussianRasterizer
from tests.gpu_util import DEV, assert_close_frac, assert_grads_close, settings_for
from tools.fuzz_parity import make_scene

NAMES = ("means3D", "opacities", "scales", "rotations")


def leaves(g, idx=None, extra=()):
    sel = (lambda a: a) if idx is None else (lambda a: a[idx])
    t = {k: sel(g[k].to(DEV)).clone().requires_grad_() for k in NAMES + tuple(extra)}
    t["means2D"] = torch.zeros(t["means3D"].shape[0], 3, device=DEV, requires_grad=True)
    return t


def geo(t):
    return dict(means3D=t["means3D"], means2D=t["means2D"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])


def grads(t):
    return {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in t.items()}


def close(a, b, rel, name):
    assert_grads_close(a.detach().cpu().numpy(), b.detach().cpu().numpy(), rel, name, max_bad_frac=0.0, hard=rel)


def f64_class_gradients(g, cam, tile, cls, n_cls, gd):
    """Float64 (free-running) backward of the class-filtered renders, summed over the classes: the truth both float32 paths approximate."""
    from tests.gpu_util import free_f64_reference, run_oracle
    P = g["means3D"].shape[0]; W, H = cam.image_width, cam.image_height
    keys = dict(means3D="dL_dmeans3D", opacities="dL_dopacity", scales="dL_dscales", rotations="dL_drotations", means2D="dL_dmeans2D")
    out = {n: torch.zeros((P, 3) if n == "means2D" else tuple(g[n].shape), dtype=torch.float64) for n in keys}
    bg = np.zeros(3, np.float32)
    for k in range(n_cls):
        idx = (cls == k).cpu()
        if not bool(idx.any()):
            continue
        sub = {n: g[n][idx] for n in g}
        cols = np.zeros((int(idx.sum()), 3), np.float32)
        da = torch.zeros(7, H, W); da[6] = gd[k].cpu()
        fwd, _ = run_oracle(sub, cam, bg, 0, colors=cols, tile=tile)
        _, xb, _ = free_f64_reference(sub, cam, bg, 0, torch.zeros(3, H, W), da, colors=cols, tile=tile, base=fwd)
        for n, key in keys.items():
            out[n][idx] += torch.as_tensor(np.asarray(xb.get(key + "64", xb[key]), np.float64)).reshape(out[n][idx].shape)
    return out


def one(seed):
    sc = make_scene(seed)
    g, cam, deg, P = sc["g"], sc["cam"], sc["deg"], sc["P"]
    tile = sc["tile"]      # (every shape of the sweep, 32x16 included since round 5)
    tl = None if tile == (16, 16) else tile
    W, H = cam.image_width, cam.image_height
    rng = np.random.default_rng(seed + 1)
    gen = torch.Generator().manual_seed(seed)
    bg9 = rng.random(9).astype(np.float32)
    cols = torch.as_tensor(rng.random((P, 6)).astype(np.float32)).to(DEV)
    gc = torch.randn(9, H, W, generator=gen).to(DEV); ga = torch.randn(7, H, W, generator=gen).to(DEV)
    # ---- 9 channels ----
    t9 = leaves(g, extra=("shs",)); c9 = cols.clone().requires_grad_()
    o9 = GaussianRasterizer(settings_for(cam, bg9, deg), tile=tl)(shs=t9["shs"], extra_colors=c9, **geo(t9))
    torch.autograd.backward([o9[0], o9[2]], [gc, ga])
    t3 = leaves(g, extra=("shs",))
    o3 = GaussianRasterizer(settings_for(cam, bg9[:3], deg), tile=tl)(shs=t3["shs"], **geo(t3))
    torch.autograd.backward([o3[0], o3[2]], [gc[:3].contiguous(), ga])
    t6 = leaves(g); c6 = cols.clone().requires_grad_()
    o6 = GaussianRasterizer(settings_for(cam, bg9[3:], 0), tile=tl)(colors_precomp=c6, **geo(t6))
    torch.autograd.backward([o6[0]], [gc[3:].contiguous()])
    assert torch.equal(o9[0][:3], o3[0]) and torch.equal(o9[0][3:], o6[0]) and torch.equal(o9[2], o3[2]) and torch.equal(o9[1], o3[1]), "9-channel forward"
    g9, g3, g6 = grads(t9), grads(t3), grads(t6)
    for k in NAMES + ("means2D",):
        close(g9[k], g3[k] + g6[k], 3e-5, f"9ch d{k}")
    close(g9["shs"], g3["shs"], 3e-5, "9ch dshs"); close(c9.grad, c6.grad, 3e-5, "9ch dextra")
    # ---- per-class pass ----
    n_cls = int(rng.integers(1, 7))
    cls = torch.randint(-1, n_cls + 1, (P,), generator=gen).to(DEV)
    gd = (torch.rand(n_cls, H, W, generator=gen) + 0.5).to(DEV)
    s0 = settings_for(cam, np.zeros(3, np.float32), 0)
    tc = leaves(g)
    dist, radii = GaussianRasterizer(s0, tile=tl).class_distortions(tc["means3D"], tc["means2D"], tc["opacities"], tc["scales"], tc["rotations"], cls, n_cls)
    (dist * gd).sum().backward()
    gcl = grads(tc)
    sums = {k: torch.zeros_like(v) for k, v in gcl.items()}
    for k in range(n_cls):
        idx = cls == k
        if not bool(idx.any()):
            assert not dist[k].any(); continue
        ts = leaves(g, idx)
        _, r, am = GaussianRasterizer(s0, tile=tl)(colors_precomp=torch.zeros(int(idx.sum()), 3, device=DEV), **geo(ts))
        assert float((am[6].detach() - dist[k].detach()).abs().max()) <= 1e-6 * max(1.0, float(am[6].detach().abs().max())), f"class {k} map"
        assert torch.equal(r, radii[idx]), "class radii"
        (am[6] * gd[k]).sum().backward()
        for n, v in grads(ts).items():
            sums[n][idx] += v
    for n in sums:
        if float(sums[n].abs().max()) > 0:
            # (the class backward sums the distortion terms in another order than K7 -- psi = a2 + m (m a0 - 2 a1) literally -- and the terms
            # cancel to the variance of m along the ray: a few 1e-4 of the tensor's scale on ill-conditioned random scenes)
            # (... and where the class maps are ~0 -- hardly any overlap along the rays -- the whole gradient tensor is that cancellation's
            # rounding noise (1e-5 and less; the class pass and the subset render are then equally far -- 1e-1 of that scale -- from a float64
            # evaluation, and closer to each other than to it): errors are taken relative to max(scale, 1e-3))
            err = float((gcl[n] - sums[n]).abs().max()) / max(float(sums[n].abs().max()), 1e-3)
            if err > 6e-4:   # ... or, measured: the class pass is no further from the float64 backward of the subset renders than those renders are
                # (up to the factor between two float32 summation orders of the same cancelling terms: 2.09x at seed 30703 of round 5's 1 000-scene
                # sweep -- 2.9e-6 against 1.4e-6 on a tensor whose largest entry is 1.5e-4 -- the one scene in 2 000 above 2x; 3x is the bar)
                ref = f64_class_gradients(g, cam, tile, cls, n_cls, gd)[n]
                e_class = float((gcl[n].double().cpu() - ref).abs().max()); e_sub = float((sums[n].double().cpu() - ref).abs().max())
                assert e_class <= 3.0 * e_sub + 1e-12, f"class d{n}: {err:.2e} of max(scale, 1e-3) from the subset renders; {e_class:.2e} vs their {e_sub:.2e} from float64 (tensor max {float(ref.abs().max()):.2e})"
    # ---- one plan: the 9-channel render AND the class pass on one K1 / binning / K8 (sr_class_*_shared) ----
    tp = leaves(g, extra=("shs",)); cp = cols.clone().requires_grad_()
    op = GaussianRasterizer(settings_for(cam, bg9, deg), tile=tl).forward_with_class_distortions(
        means3D=tp["means3D"], means2D=tp["means2D"], opacities=tp["opacities"], scales=tp["scales"], rotations=tp["rotations"], classes=cls, n_classes=n_cls,
        shs=tp["shs"], extra_colors=cp)
    torch.autograd.backward([op[0], op[2], op[3]], [gc, ga, gd])
    assert torch.equal(op[0], o9[0]) and torch.equal(op[2], o9[2]) and torch.equal(op[1], o9[1]) and torch.equal(op[3], dist.detach()), "one-plan forward"
    gp = grads(tp)
    for k in NAMES + ("means2D",):
        want = g9[k] + gcl[k]
        err = float((gp[k] - want).abs().max()) / max(float(want.abs().max()), 1e-3)
        assert err <= 1e-4, f"one plan d{k}: {err:.2e} of max(scale, 1e-3) from the two calls' sum"
    close(gp["shs"], g9["shs"], 3e-5, "one plan dshs"); close(cp.grad, c9.grad, 3e-5, "one plan dextra")
    # ---- mask ----
    m = (torch.rand(P, generator=gen) < 0.6).to(DEV)
    tm = leaves(g, extra=("shs",))
    om = GaussianRasterizer(settings_for(cam, bg9[:3], deg), tile=tl)(shs=tm["shs"], mask=m, **geo(tm))
    torch.autograd.backward([om[0], om[2]], [gc[:3].contiguous(), ga])
    if bool(m.any()):
        ti = leaves(g, m, extra=("shs",))
        oi = GaussianRasterizer(settings_for(cam, bg9[:3], deg), tile=tl)(shs=ti["shs"], **geo(ti))
        torch.autograd.backward([oi[0], oi[2]], [gc[:3].contiguous(), ga])
        assert torch.equal(om[0], oi[0]) and torch.equal(om[2], oi[2]) and torch.equal(om[1][m], oi[1]) and not om[1][~m].any(), "mask forward"
        gm, gi = grads(tm), grads(ti)
        for k in gm:
            assert torch.equal(gm[k][m], gi[k]) and not gm[k][~m].any(), f"mask d{k}"
    # ---- fused activations ----
    raw = dict(opacities=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)), scales=torch.log(g["scales"]), rotations=g["rotations"] * (0.5 + torch.rand(P, 1, generator=gen)))
    ta = {k: (raw[k] if k in raw else g[k]).to(DEV).clone().requires_grad_() for k in NAMES + ("shs",)}; ta["means2D"] = torch.zeros(P, 3, device=DEV, requires_grad=True)
    oa = GaussianRasterizer(settings_for(cam, bg9[:3], deg), tile=tl, fused_activations=True)(shs=ta["shs"], **geo(ta))
    torch.autograd.backward([oa[0], oa[2]], [gc[:3].contiguous(), ga])
    tb = {k: (raw[k] if k in raw else g[k]).to(DEV).clone().requires_grad_() for k in NAMES + ("shs",)}; tb["means2D"] = torch.zeros(P, 3, device=DEV, requires_grad=True)
    ob = GaussianRasterizer(settings_for(cam, bg9[:3], deg), tile=tl)(shs=tb["shs"], means3D=tb["means3D"], means2D=tb["means2D"], opacities=torch.sigmoid(tb["opacities"]),
                                                                       scales=torch.exp(tb["scales"]), rotations=torch.nn.functional.normalize(tb["rotations"]))
    torch.autograd.backward([ob[0], ob[2]], [gc[:3].contiguous(), ga])
    assert_close_frac(oa[0].detach().cpu().numpy(), ob[0].detach().cpu().numpy(), 1e-4, 1e-4, 2e-3, None, "fused activations colour")
    ga_, gb_ = grads(ta), grads(tb)
    for k in ga_:
        assert_grads_close(ga_[k].cpu().numpy(), gb_[k].cpu().numpy(), 2e-3, f"fused activations d{k}", max_bad_frac=5e-3, hard=None)
    return sc["tag"] + f" classes={n_cls}"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    bad = 0
    for k in range(n):
        try:
            tag = one(seed0 + k)
            print(f"ok   scene {k} {tag}", flush=True)
        except Exception as e:   # noqa
            bad += 1
            print(f"FAIL scene {k} (seed {seed0 + k}): {type(e).__name__}: {str(e)[:300]}", flush=True)
    print(f"{n - bad}/{n} scenes consistent")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
