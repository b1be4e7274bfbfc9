"""GPU (-m gpu): the reference's operator surface (render / render_with_mask / render_semantic[_with_mask],
SURVEY 8a rows A1-A4) on top of the HIP rasterizer, end to end against the CPU oracle."""
import math
import os

import numpy as np
import pytest
import torch

from oracle.postprocess_torch import postprocess_allmap as postprocess_allmap_torch
from streetunveiler_amd.gaussian_renderer import (PipelineParams, SurfelModel, postprocess_allmap, render, render_semantic,
                                                  render_semantic_with_mask, render_with_mask)
from streetunveiler_amd.synthetic import synthetic_camera, synthetic_gaussians

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(P, W, H, seed, dev, requires_grad=False, gaussians=None):
    g = synthetic_gaussians(P, W, H, seed=seed, scale_lo=3e-3, scale_hi=5e-2) if gaussians is None else gaussians
    sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(seed))
    t = {k: v.to(dev) for k, v in g.items()}
    if requires_grad:
        for v in t.values():
            v.requires_grad_()
    return g, sem, SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(dev), 3, 3), t


def _oracle(g, cam, bg, deg, idx=None, colors=None):
    from oracle import surfel_oracle as so
    sel = (lambda a: a) if idx is None else (lambda a: a[idx])
    kw = dict(viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.numpy(), bg=np.asarray(bg, np.float32), image_width=cam.image_width,
              image_height=cam.image_height, sh_degree=deg, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2))
    n = lambda k: sel(g[k].numpy())
    if colors is None:
        return so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), shs=n("shs"), **kw)
    return so.rasterize_forward(n("means3D"), n("opacities"), n("scales"), n("rotations"), colors_precomp=colors, **kw)


@pytest.mark.parametrize("posed", [False, True])
def test_render_dict_and_backward_through_regularisers(monkeypatch, posed):
    from tests.gpu_util import assert_close_frac, assert_free_parity, free_f64_reference
    import streetunveiler_amd.gaussian_renderer as gr
    P, W, H = 6000, 208, 120
    cam, gp = synthetic_camera(W, H, index=2), None
    if posed:   # a camera in general position (own centre, full rotation, FoVx != FoVy): the maps' back-projection uses all of it
        from streetunveiler_amd.synthetic import posed_scene
        cam, gp = posed_scene(P, W, H, seed=21, scale_lo=3e-3, scale_hi=5e-2, spread=12.0)
    g, sem, pc, t = _model(P, W, H, 21, DEV, requires_grad=True, gaussians=gp)
    bg = torch.tensor([0.2, 0.3, 0.1])
    pipe = PipelineParams(depth_ratio=0.0)
    seen = {}
    fused_maps = gr.postprocess_allmap

    def spy(cam_, pipe_, allmap):   # the operator's allmap output: keep it and the gradient the loss sends back into it
        allmap.retain_grad(); seen["allmap"] = allmap
        return fused_maps(cam_, pipe_, allmap)

    monkeypatch.setattr(gr, "postprocess_allmap", spy)
    out = render(cam.to(DEV), pc, pipe, bg.to(DEV))
    assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii", "rend_alpha", "rend_normal", "rend_dist",
                        "surf_depth", "surf_normal", "surf_point"}
    out["render"].retain_grad()
    fwd = _oracle(g, cam, bg.numpy(), 3)
    np.testing.assert_array_equal(out["radii"].cpu().numpy(), fwd["radii"])
    np.testing.assert_array_equal(out["visibility_filter"].cpu().numpy(), fwd["radii"] > 0)
    assert_close_frac(out["render"].detach().cpu().numpy(), fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, "render")
    # the same post-processing applied to the oracle's allmap (CPU)
    ref = postprocess_allmap_torch(cam, pipe.depth_ratio, torch.tensor(fwd["allmap"]))
    for k in ["rend_alpha", "rend_normal", "rend_dist", "surf_depth"]:
        assert_close_frac(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), 1e-4, 1e-4, 1e-3, 5e-2, k)

    def loss_fn(o, image):   # train.py-style: image term + normal consistency + distortion + alpha
        normal_error = (1 - (o["rend_normal"] * o["surf_normal"]).sum(dim=0))[None]
        return image.square().mean() + 0.05 * normal_error.mean() + 100.0 * o["rend_dist"].mean() + 0.1 * o["rend_alpha"].mean()

    loss_fn(out, out["render"]).backward()
    torch.cuda.synchronize()
    # The operator's backward, fed with exactly the upstream gradients autograd handed it (through the fused map kernels, whose own
    # backward tests/test_reference_render_golden.py pins), against the free-running float64 reference: identical bars as the
    # full-size tests -- 1e-4 at every robust pixel, strict rows on every robust Gaussian.
    dc, da = out["render"].grad.cpu(), seen["allmap"].grad.cpu()
    hip = dict(color=out["render"].detach().cpu().numpy(), allmap=seen["allmap"].detach().cpu().numpy(),
               dL_dmeans3D=t["means3D"].grad.cpu().numpy(), dL_dopacity=t["opacities"].grad.cpu().numpy(), dL_dscales=t["scales"].grad.cpu().numpy(),
               dL_drotations=t["rotations"].grad.cpu().numpy(), dL_dsh=t["shs"].grad.cpu().numpy(), dL_dmeans2D=out["viewspace_points"].grad.cpu().numpy())
    xfwd, xbwd, margins = free_f64_reference(g, cam, bg.numpy(), 3, dc, da, base=fwd)
    assert_free_parity(hip, None, xfwd, xbwd, margins, tag="render() ", scene=(g, cam))


def test_semantic_filter_mask_and_python_sh_path():
    from tests.gpu_util import assert_close_frac
    P, W, H = 4000, 160, 96
    cam = synthetic_camera(W, H)
    g, sem, pc, t = _model(P, W, H, 5, DEV)
    bg = torch.zeros(3)
    pipe = PipelineParams()
    # semantic_filter_bit keeps classes whose bit is set (reverse_semantic=True) or clear (False)
    bit = (1 << 1) | (1 << 4)
    for reverse in (True, False):
        out = render(cam.to(DEV), pc, pipe, bg.to(DEV), semantic_filter_bit=bit, reverse_semantic=reverse)
        keep = ((1 << sem.numpy()) & bit) != 0
        keep = keep if reverse else ~keep
        fwd = _oracle(g, cam, bg.numpy(), 3, idx=keep)
        assert out["radii"].shape[0] == int(keep.sum())
        np.testing.assert_array_equal(out["radii"].cpu().numpy(), fwd["radii"])
        assert_close_frac(out["render"].detach().cpu().numpy(), fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, "masked render")
    # explicit boolean mask
    mask = torch.rand(P, generator=torch.Generator().manual_seed(1)) > 0.4
    out = render_with_mask(cam.to(DEV), pc, pipe, bg.to(DEV), mask.to(DEV))
    fwd = _oracle(g, cam, bg.numpy(), 3, idx=mask.numpy())
    assert_close_frac(out["render"].detach().cpu().numpy(), fwd["color"], 1e-4, 1e-4, 2e-4, 2e-2, "render_with_mask")
    # python SH fallback == native SH path
    a = render(cam.to(DEV), pc, PipelineParams(convert_SHs_python=False), bg.to(DEV))["render"]
    b = render(cam.to(DEV), pc, PipelineParams(convert_SHs_python=True), bg.to(DEV))["render"]
    assert_close_frac(a.detach().cpu().numpy(), b.detach().cpu().numpy(), 1e-5, 1e-5, 1e-4, 1e-2, "convert_SHs_python")
    with pytest.raises(NotImplementedError):
        render(cam.to(DEV), pc, PipelineParams(compute_cov3D_python=True), bg.to(DEV))


def _oracle_semantics(g, sem, cam, idx=None):
    """render_semantics exactly as the reference assembles it: two 3-channel one-hot passes, sky = background
    [REF gaussian_renderer/__init__.py:346-369, 417-446]."""
    P = sem.shape[0]
    expect = []
    for i in (0, 3):
        onehot = np.zeros((P, 3), np.float32)
        for c in range(3):
            onehot[sem.numpy() == i + c, c] = 1.0
        bg = np.zeros(3, np.float32)
        if i <= 4 < i + 3:
            bg[4 - i] = 1.0     # sky is background
        expect.append(_oracle(g, cam, bg, 3, idx=idx, colors=onehot if idx is None else onehot[idx])["color"])
    return np.concatenate(expect, 0)


def _check_derived_semantic_maps(out, expect, H, W, lead_axis):
    """semantic_uncertainty / semantic_rgb against what the reference computes from the class map with torch.topk(k=2) and
    argmax [REF gaussian_renderer/__init__.py:448-452, 586-590; utils/semantic_utils.py:128-135]: exactly on the operator's own
    class map, and within the image tolerance on the oracle's."""
    from tests.gpu_util import assert_close_frac
    prob = out["render_semantics"].detach()
    tv, _ = torch.topk(prob, k=2, dim=0)
    ref_unc = 1.0 - (tv[0] - tv[1])
    assert out["semantic_uncertainty"].shape == ((1, H, W) if lead_axis else (H, W))
    assert torch.equal(out["semantic_uncertainty"].detach().reshape(H, W), ref_unc)
    colour = torch.tensor([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [255, 0, 255], [0, 255, 255]], device=prob.device)
    assert torch.equal(out["semantic_rgb"], colour[torch.argmax(prob, dim=0)].permute(2, 0, 1) / 255.0)
    etv, _ = torch.topk(torch.tensor(expect), k=2, dim=0)
    assert_close_frac(out["semantic_uncertainty"].detach().reshape(H, W).cpu().numpy(), (1.0 - (etv[0] - etv[1])).numpy(), 2e-4, 0, 4e-4, 4e-2, "semantic_uncertainty")
    # the class colour agrees with the oracle's wherever the oracle's decision is not within float noise of a tie
    e = torch.tensor(expect)
    clear = (etv[0] - etv[1]) > 1e-3
    same = (torch.argmax(prob, dim=0).cpu() == torch.argmax(e, dim=0))[clear]
    assert float(same.float().mean()) > 0.9995


def test_render_semantic_six_classes():
    from tests.gpu_util import assert_close_frac
    P, W, H = 4000, 160, 96
    cam = synthetic_camera(W, H, index=5)
    g, sem, pc, t = _model(P, W, H, 9, DEV)
    out = render_semantic(cam.to(DEV), pc, PipelineParams(), torch.zeros(3, device=DEV))
    assert out["render_semantics"].shape == (6, H, W) and out["semantic_rgb"].shape == (3, H, W) and out["semantic_uncertainty"].shape == (H, W)
    expect = _oracle_semantics(g, sem, cam)
    assert_close_frac(out["render_semantics"].detach().cpu().numpy(), expect, 1e-4, 1e-4, 2e-4, 2e-2, "render_semantics")
    _check_derived_semantic_maps(out, expect, H, W, lead_axis=False)
    # class probabilities + background sum to one wherever the sky class absorbs the leftover transmittance
    np.testing.assert_allclose(out["render_semantics"].detach().sum(0).cpu().numpy(), 1.0, atol=2e-3)
    # semantic_filter_bit variants select by class bit like render() does
    bit = (1 << 0) | (1 << 3) | (1 << 5)
    for reverse in (True, False):
        keep = ((1 << sem.numpy()) & bit) != 0
        keep = keep if reverse else ~keep
        o = render_semantic(cam.to(DEV), pc, PipelineParams(), torch.zeros(3, device=DEV), semantic_filter_bit=bit, reverse_semantic=reverse)
        e = _oracle_semantics(g, sem, cam, idx=keep)
        assert_close_frac(o["render_semantics"].detach().cpu().numpy(), e, 1e-4, 1e-4, 2e-4, 2e-2, "render_semantics, filter bit")
        _check_derived_semantic_maps(o, e, H, W, lead_axis=False)


def test_render_semantic_with_mask_against_the_oracle_on_the_subset():
    """A4 [REF gaussian_renderer/__init__.py:462-598]: boolean-indexed inputs, two 3-channel passes, [1,H,W] uncertainty --
    oracle on the masked subset, plain and with the mask handed to the operator (fused_mask), forward and backward."""
    from oracle import surfel_oracle as so
    from tests.gpu_util import assert_close_frac, assert_grads_close
    P, W, H = 5000, 176, 104
    cam = synthetic_camera(W, H, index=1)
    g, sem, _, _ = _model(P, W, H, 17, DEV)
    m = torch.rand(P, generator=torch.Generator().manual_seed(2)) > 0.45
    expect = _oracle_semantics(g, sem, cam, idx=m.numpy())
    cls_w = torch.tensor([1.0, -0.5, 0.3, 0.8, -1.2, 0.6]).view(6, 1, 1)
    # oracle gradients: the two passes' gradients add (same geometry, colours are constants)
    ref = None
    for i in (0, 3):
        onehot = np.zeros((P, 3), np.float32)
        for c in range(3):
            onehot[sem.numpy() == i + c, c] = 1.0
        bg = np.zeros(3, np.float32)
        if i <= 4 < i + 3:
            bg[4 - i] = 1.0
        fwd = _oracle(g, cam, bg, 3, idx=m.numpy(), colors=onehot[m.numpy()])
        bwd = so.rasterize_backward(fwd, np.broadcast_to(cls_w[i:i + 3].numpy(), (3, H, W)).astype(np.float32).copy(), np.zeros((7, H, W), np.float32))
        ref = bwd if ref is None else {k: ref[k] + bwd[k] for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations")}
    for fused in (False, True):
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        pc = SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(DEV), 3, 3)
        out = render_semantic_with_mask(cam.to(DEV), pc, PipelineParams(fused_mask=fused), torch.zeros(3, device=DEV), m.to(DEV))
        assert set(out) == {"render_semantics", "semantic_rgb", "semantic_uncertainty"}
        assert_close_frac(out["render_semantics"].detach().cpu().numpy(), expect, 1e-4, 1e-4, 2e-4, 2e-2, "masked render_semantics")
        _check_derived_semantic_maps(out, expect, H, W, lead_axis=True)
        (out["render_semantics"] * cls_w.to(DEV)).sum().backward()
        for name, key in (("dL_dmeans3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dscales", "scales"), ("dL_drotations", "rotations")):
            got = t[key].grad.cpu().numpy()
            assert not got[~m.numpy()].any(), name                  # masked-out Gaussians get no gradient
            assert_grads_close(got[m.numpy()], ref[name], 2e-3, "masked semantic " + name)
        assert t["shs"].grad is None or not t["shs"].grad.any()      # colours are the one-hot class channels, not the SHs


def test_fused_postprocess_matches_torch_restatement():
    """csrc/postprocess.hip forward + backward against the plain-torch restatement (float64 on the CPU)."""
    W, H = 200, 120
    _postprocess_against_torch(synthetic_camera(W, H, index=6), W, H)
    # ... and a camera in general position: its own centre, rotation about all three axes, fx != fy (the back-projection's intrinsics and
    # camera-to-world transform are then no longer diagonal / identity-like)
    from streetunveiler_amd.synthetic import posed_scene
    _postprocess_against_torch(posed_scene(1, W, H, seed=12, spread=20.0)[0], W, H)
    with pytest.raises(Exception, match="no CPU path"):
        postprocess_allmap(synthetic_camera(W, H), PipelineParams(), torch.zeros(7, H, W))


def _postprocess_against_torch(cam, W, H):
    g = torch.Generator().manual_seed(4)
    allmap = torch.rand(7, H, W, generator=g)
    allmap[0] = allmap[0] * 20 + 1; allmap[5] = allmap[5] * 20 + 1
    allmap[1] = allmap[1] * 0.98 + 0.01
    allmap[:, :5, :9] = 0.0                       # empty pixels: alpha == 0, 0/0 -> nan_to_num -> 0
    grads = {k: torch.randn(c, H, W, generator=g) for k, c in [("rend_normal", 3), ("surf_depth", 1), ("surf_normal", 3), ("surf_point", 3)]}
    for ratio in (0.0, 1.0, 0.4):
        a64 = allmap.double().requires_grad_()
        ref = postprocess_allmap_torch(cam, ratio, a64)
        sum((ref[k] * grads[k].double()).sum() for k in grads).backward()
        a_gpu = allmap.to(DEV).requires_grad_()
        out = postprocess_allmap(cam.to(DEV), PipelineParams(depth_ratio=ratio), a_gpu)
        sum((out[k] * grads[k].to(DEV)).sum() for k in grads).backward()
        for k in ["rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "surf_point"]:
            np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref[k].detach().numpy(), rtol=2e-4, atol=2e-4, err_msg=k)
        gref = torch.nan_to_num(a64.grad, 0.0, 0.0, 0.0).numpy()   # torch gives NaN (0 * inf) at alpha == 0; the kernel gives 0
        ggpu = a_gpu.grad.cpu().numpy()
        assert np.isfinite(ggpu).all()
        scale = np.abs(gref).max()
        assert np.abs(ggpu - gref).max() <= 2e-4 * scale, np.abs(ggpu - gref).max() / scale


def test_fused_postprocess_matches_reference_fixture(golden_dir):
    """csrc/postprocess.hip against tests/golden/reference_render_golden.npz: the dict the reference's own render() /
    utils.point_utils.depth_to_normal returned for these allmaps, and the gradient its autograd returned."""
    from streetunveiler_amd.camera import SimpleCamera
    z = np.load(os.path.join(golden_dir, "reference_render_golden.npz"))
    maps = ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "surf_point")
    for ci in range(int(z["n_cases"])):
        pre = f"c{ci}_"
        W, H, yaw, tx, ty, tz, ratio, fovx, fovy = z[pre + "meta"]
        W, H = int(W), int(H)
        cam = SimpleCamera(W, H, float(fovx), float(fovy), torch.tensor(z[pre + "wvt"]).to(DEV), torch.tensor(z[pre + "full"]).to(DEV),
                           torch.tensor(z[pre + "center"]).to(DEV))
        for run in ("render", "render_mask"):
            a = torch.tensor(z[pre + run + "_allmap"]).to(DEV).requires_grad_()
            out = postprocess_allmap(cam, PipelineParams(depth_ratio=float(ratio)), a)
            for k in maps:
                ref = z[pre + run + "_" + k]
                # where alpha == 0 the reference's surf_normal is 0 * NaN-free values; everything is finite in the fixture
                np.testing.assert_allclose(out[k].detach().cpu().numpy(), ref, rtol=1e-4, atol=1e-4 * max(1.0, np.abs(ref).max()), err_msg=f"{pre}{run} {k}")
            sum((out[k] * torch.tensor(z[pre + "up_" + k]).to(DEV)).sum() for k in maps).backward()
            ref = z[pre + run + "_allmap_grad"]
            got = a.grad.cpu().numpy()
            fin = np.isfinite(ref)          # the reference's autograd yields NaN (0 * inf) at alpha == 0; the kernel keeps the finite terms
            assert np.isfinite(got).all() and fin.mean() > 0.8
            scale = np.abs(ref[fin]).max()
            assert np.abs(got[fin] - ref[fin]).max() <= 2e-4 * scale, (pre, run, np.abs(got[fin] - ref[fin]).max() / scale)


def test_fused_activations_and_ply_checkpoint(tmp_path):
    """raw _opacity/_scaling/_rotation straight into the operator (sigmoid/exp/normalize fused into K1, adjoints
    into K8) == the reference's torch activations in front of it; and a PLY checkpoint drives the same render."""
    from tests.gpu_util import assert_close_frac, assert_grads_close, check_allmap
    P, W, H = 6000, 224, 128
    cam = synthetic_camera(W, H, index=3).to(DEV)
    g = synthetic_gaussians(P, W, H, seed=12, scale_lo=3e-3, scale_hi=5e-2)
    gen = torch.Generator().manual_seed(5)
    raw = dict(scaling=torch.log(g["scales"]), opacity=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)),
               rotation=g["rotations"] * (0.25 + 3 * torch.rand(P, 1, generator=gen)))
    sem = torch.randint(0, 6, (P,), generator=gen)
    bg = torch.tensor([0.2, 0.1, 0.3], device=DEV)
    results = {}
    for fused in (False, True):
        leaves = {k: v.to(DEV).requires_grad_() for k, v in dict(xyz=g["means3D"], features=g["shs"], **raw).items()}
        pc = SurfelModel(leaves["xyz"], leaves["scaling"], leaves["rotation"], leaves["opacity"], leaves["features"], sem.to(DEV), 3, 3, raw=True)
        out = render(cam, pc, PipelineParams(fused_activations=fused, depth_ratio=0.3), bg)
        loss = (out["render"] * torch.linspace(0.5, 1.5, W, device=DEV)).sum() + out["rend_alpha"].sum() + out["rend_dist"].sum() * 10 \
            + (out["rend_normal"] * out["surf_normal"]).sum() + out["surf_depth"].mean()
        loss.backward()
        results[fused] = (out, {k: v.grad.detach().cpu().numpy() for k, v in leaves.items()}, pc)
    a, b = results[False], results[True]
    assert float((a[0]["radii"] != b[0]["radii"]).float().mean()) < 1e-3      # an ulp of exp() may move a ceil()
    assert_close_frac(b[0]["render"].detach().cpu().numpy(), a[0]["render"].detach().cpu().numpy(), 1e-4, 1e-4, 2e-4, 2e-2, "fused render")
    for k in ("rend_alpha", "rend_dist", "surf_depth", "rend_normal"):
        assert_close_frac(b[0][k].detach().cpu().numpy(), a[0][k].detach().cpu().numpy(), 1e-4, 1e-4, 5e-4, None, "fused " + k)
    for k in a[1]:
        assert np.abs(a[1][k]).max() > 0, k
        assert_grads_close(b[1][k], a[1][k], 2e-3, "fused d" + k)
    # PLY round trip of the raw checkpoint, then the same (fused) render from the file
    path = os.path.join(tmp_path, "point_cloud.ply")
    b[2].save_ply(path)
    pc2 = SurfelModel.from_ply(path, device=DEV)
    for name in ("_xyz", "_scaling", "_rotation", "_opacity", "_features"):
        assert torch.equal(getattr(pc2, name).detach(), getattr(b[2], name).detach()), name
    assert torch.equal(pc2.get_semantics.cpu(), sem.to(torch.int32))
    out2 = render(cam, pc2, PipelineParams(fused_activations=True, depth_ratio=0.3), bg)
    assert torch.equal(out2["render"], b[0]["render"]) and torch.equal(out2["radii"], b[0]["radii"])


def test_mask_inside_the_operator_equals_boolean_indexing():
    """SURVEY 8f N1 (class masks without gathered copies): `mask=` on the operator / PipelineParams.fused_mask give the images of
    the reference's boolean-indexed call bit for bit, with full-size radii and gradients (zero where masked out)."""
    P, W, H = 9000, 256, 144
    cam = synthetic_camera(W, H, index=1).to(DEV)
    g, sem, _, _ = _model(P, W, H, 21, DEV)
    bg = torch.tensor([0.1, 0.0, 0.2], device=DEV)
    m = (torch.rand(P, generator=torch.Generator().manual_seed(3)) > 0.4).to(DEV)
    wgt = torch.linspace(0.2, 1.8, W, device=DEV)
    res = {}
    for fused in (False, True):
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        pc = SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(DEV), 3, 3)
        out = render_with_mask(cam, pc, PipelineParams(fused_mask=fused, depth_ratio=0.5), bg, m)
        ((out["render"] * wgt).sum() + out["rend_dist"].sum() * 5 + out["rend_alpha"].sum() + out["surf_depth"].mean()).backward()
        res[fused] = (out, {k: v.grad.clone() for k, v in t.items()})
    a, b = res[False], res[True]
    assert torch.equal(a[0]["render"], b[0]["render"]) and torch.equal(a[0]["rend_dist"], b[0]["rend_dist"])
    assert b[0]["radii"].shape == (P,) and a[0]["radii"].shape == (int(m.sum()),)
    assert torch.equal(b[0]["radii"][m], a[0]["radii"]) and not b[0]["radii"][~m].any()
    assert torch.equal(b[0]["visibility_filter"][m], a[0]["visibility_filter"]) and not b[0]["visibility_filter"][~m].any()
    for k in a[1]:
        assert torch.equal(a[1][k], b[1][k]), k          # same records, same sums: bit-identical gradients
        assert not b[1][k][~m].any()
    # semantic filter path of render(): bit selects classes, kept out when reverse_semantic is False
    for fused in (False, True):
        t = {k: v.to(DEV) for k, v in g.items()}
        pc = SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(DEV), 3, 3)
        res[fused] = render(cam, pc, PipelineParams(fused_mask=fused), bg, semantic_filter_bit=0b000110, reverse_semantic=False)["render"]
    assert torch.equal(res[False], res[True])
    # and the 6-class semantic render with a mask
    pc = SurfelModel(*(g[k].to(DEV) for k in ("means3D", "scales", "rotations", "opacities", "shs")), sem.to(DEV), 3, 3)
    s0 = render_semantic_with_mask(cam, pc, PipelineParams(), torch.zeros(3, device=DEV), m)["render_semantics"]
    s1 = render_semantic_with_mask(cam, pc, PipelineParams(fused_mask=True), torch.zeros(3, device=DEV), m)["render_semantics"]
    assert torch.equal(s0, s1)


def test_render_and_semantic_in_one_pass():
    """SH colour + six class channels blended in ONE rasterization (9 channels) == render() and render_semantic()."""
    from streetunveiler_amd.gaussian_renderer import render_and_semantic
    P, W, H = 8000, 224, 128
    cam = synthetic_camera(W, H, index=4).to(DEV)
    g, sem, _, _ = _model(P, W, H, 31, DEV)
    bg = torch.tensor([0.3, 0.2, 0.1], device=DEV)
    wgt = torch.linspace(0.5, 1.5, W, device=DEV)
    cls_w = torch.tensor([1.0, -0.5, 0.3, 0.8, -1.2, 0.6], device=DEV).view(6, 1, 1)

    def fresh():
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        return t, SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(DEV), 3, 3)

    def loss_rgb(o): return (o["render"] * wgt).sum() + o["rend_dist"].sum() * 3 + o["rend_alpha"].sum()
    def loss_sem(o): return (o["render_semantics"] * cls_w).sum()
    t1, pc1 = fresh()
    o1 = render_and_semantic(cam, pc1, PipelineParams(), bg)
    (loss_rgb(o1) + loss_sem(o1)).backward()
    t2, pc2 = fresh()
    o2a = render(cam, pc2, PipelineParams(), bg)
    o2b = render_semantic(cam, pc2, PipelineParams(), torch.zeros(3, device=DEV))
    (loss_rgb(o2a) + loss_sem(o2b)).backward()
    assert o1["render"].shape == (3, H, W) and o1["render_semantics"].shape == (6, H, W)
    assert torch.equal(o1["render"], o2a["render"]) and torch.equal(o1["render_semantics"], o2b["render_semantics"])
    for k in ("rend_alpha", "rend_dist", "surf_depth", "rend_normal"):
        assert torch.equal(o1[k], o2a[k]), k
    assert torch.equal(o1["radii"], o2a["radii"])
    from tests.gpu_util import assert_grads_close
    for k in t1:
        assert float(t2[k].grad.abs().max()) > 0
        assert_grads_close(t1[k].grad.cpu().numpy(), t2[k].grad.cpu().numpy(), 2e-5, "one pass d" + k, max_bad_frac=0.0, hard=2e-5)


def test_extensions_compose():
    """9-channel pass + mask inside the operator + fused activations on a raw (checkpoint-style) model == the plain composition
    of the reference-style calls."""
    from streetunveiler_amd.gaussian_renderer import render_and_semantic
    from tests.gpu_util import assert_close_frac
    P, W, H = 7000, 208, 120
    cam = synthetic_camera(W, H, index=6).to(DEV)
    g = synthetic_gaussians(P, W, H, seed=44, scale_lo=3e-3, scale_hi=5e-2)
    gen = torch.Generator().manual_seed(8)
    sem = torch.randint(0, 6, (P,), generator=gen)
    m = (torch.rand(P, generator=gen) > 0.3).to(DEV)
    raw = dict(xyz=g["means3D"], features=g["shs"], scaling=torch.log(g["scales"]), opacity=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)),
               rotation=g["rotations"] * 1.7)
    bg = torch.tensor([0.05, 0.1, 0.15], device=DEV)
    def model():
        t = {k: v.to(DEV) for k, v in raw.items()}
        return SurfelModel(t["xyz"], t["scaling"], t["rotation"], t["opacity"], t["features"], sem.to(DEV), 3, 3, raw=True)
    plain = PipelineParams()
    a = render_with_mask(cam, model(), plain, bg, m)
    s = render_semantic_with_mask(cam, model(), plain, torch.zeros(3, device=DEV), m)
    o = render_and_semantic(cam, model(), PipelineParams(fused_mask=True, fused_activations=True), bg, mask=m)
    assert_close_frac(o["render"].detach().cpu().numpy(), a["render"].detach().cpu().numpy(), 1e-4, 1e-4, 2e-4, 2e-2, "composed render")
    assert_close_frac(o["render_semantics"].detach().cpu().numpy(), s["render_semantics"].detach().cpu().numpy(), 1e-4, 1e-4, 2e-4, 2e-2, "composed semantics")
    assert o["radii"].shape == (P,) and not o["radii"][~m].any()
    assert float((o["radii"][m] != a["radii"]).float().mean()) < 1e-3


def test_class_distortions_one_pass_equals_the_per_class_renders():
    """render_class_distortions == the reference's loop `render(..., semantic_filter_bit=1 << k, reverse_semantic=True)
    ["rend_dist"]` [REF train.py:94-103] -- against the CPU oracle on each class subset (forward and the summed backward), and against
    this build's own per-class render() calls."""
    from oracle import surfel_oracle as so
    from streetunveiler_amd.gaussian_renderer import render_class_distortions
    from tests.gpu_util import assert_close_frac, assert_grads_close
    P, W, H = 9000, 240, 136
    cam = synthetic_camera(W, H, index=3)
    g, sem, _, _ = _model(P, W, H, 41, DEV)
    classes = [0, 1, 2, 3, 5]          # every concerned class but the sky
    gen = torch.Generator().manual_seed(9)
    g_dist = torch.rand(len(classes), 1, H, W, generator=gen) + 0.5      # upstream gradients of the five maps
    bg = torch.zeros(3)

    def fresh():
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        return t, SurfelModel(t["means3D"], t["scales"], t["rotations"], t["opacities"], t["shs"], sem.to(DEV), 3, 3)

    t1, pc1 = fresh()
    out = render_class_distortions(cam.to(DEV), pc1, PipelineParams(), bg.to(DEV))
    assert out["rend_dist"].shape == (len(classes), 1, H, W) and out["radii"].shape == (P,)
    (out["rend_dist"] * g_dist.to(DEV)).sum().backward()
    # (a) oracle on every class subset
    ref = {k: np.zeros_like(g[n].numpy()) for k, n in (("dL_dmeans3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dscales", "scales"), ("dL_drotations", "rotations"))}
    ref2d = np.zeros((P, 3), np.float32)
    vis_any = np.zeros(P, bool)
    for j, k in enumerate(classes):
        idx = (sem.numpy() == k)
        fwd = _oracle(g, cam, bg.numpy(), 3, idx=idx)
        assert_close_frac(out["rend_dist"][j, 0].detach().cpu().numpy(), fwd["allmap"][6], 1e-5, 1e-4, 2e-4, 2e-2, f"class {k} rend_dist")
        da = np.zeros((7, H, W), np.float32); da[6] = g_dist[j, 0].numpy()
        bwd = so.rasterize_backward(fwd, np.zeros((3, H, W), np.float32), da)
        for name in ref:
            ref[name][idx] += bwd[name]
        ref2d[idx] += bwd["dL_dmeans2D"]
        vis_any[idx] |= fwd["radii"] > 0
        np.testing.assert_array_equal(out["radii"].cpu().numpy()[idx], fwd["radii"])
    for name, key in (("dL_dmeans3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dscales", "scales"), ("dL_drotations", "rotations")):
        assert np.abs(ref[name]).max() > 0
        assert_grads_close(t1[key].grad.cpu().numpy(), ref[name], 2e-3, "class pass " + name)
    assert_grads_close(out["viewspace_points"].grad.cpu().numpy(), ref2d, 2e-3, "class pass dL_dmeans2D")
    assert not t1["means3D"].grad[torch.tensor(sem.numpy() == 4)].any()      # the sky class is in no chain
    # (b) the reference's call pattern on this build: five class-filtered render() calls
    t2, pc2 = fresh()
    loss = 0
    for j, k in enumerate(classes):
        o = render(cam.to(DEV), pc2, PipelineParams(), bg.to(DEV), semantic_filter_bit=1 << k, reverse_semantic=True)
        d = o["rend_dist"]
        assert float((d - out["rend_dist"][j]).detach().abs().max()) <= 1e-6 * max(1.0, float(d.detach().abs().max())), k
        loss = loss + (d * g_dist[j].to(DEV)).sum()
    loss.backward()
    for key in ("means3D", "opacities", "scales", "rotations"):
        assert_grads_close(t1[key].grad.cpu().numpy(), t2[key].grad.cpu().numpy(), 2e-5, "one pass vs five renders d" + key, max_bad_frac=0.0, hard=2e-5)


@pytest.mark.parametrize("tile", [(8, 8), (16, 8), (32, 8), (32, 16)])
def test_class_distortions_on_other_tile_shapes(tile):
    """The per-class pass on the other tile shapes of BASELINE config 5's sweep: every class map equals `allmap[6]` of the operator called
    WITH THE SAME TILE on the class subset (the reference's call pattern), its gradients the sum of those calls' gradients, and both agree
    with the 16x16 pass (oracle-checked above) to float summation order."""
    from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    from tests.gpu_util import assert_close_frac, assert_grads_close, settings_for
    P, W, H = 7000, 232, 136
    cam = synthetic_camera(W, H, index=1)
    g = synthetic_gaussians(P, W, H, seed=57, scale_lo=3e-3, scale_hi=6e-2)
    n_cls = 5
    cls = torch.randint(-1, n_cls + 1, (P,), generator=torch.Generator().manual_seed(3))    # -1 and n_cls: in no class
    g_dist = (torch.rand(n_cls, H, W, generator=torch.Generator().manual_seed(4)) + 0.5).to(DEV)
    s = settings_for(cam, np.zeros(3, np.float32), 0)
    names = ("means3D", "opacities", "scales", "rotations")

    def one_pass(tl):
        t = {k: g[k].to(DEV).requires_grad_() for k in names}
        m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
        dist, radii = GaussianRasterizer(s, tile=tl).class_distortions(t["means3D"], m2d, t["opacities"], t["scales"], t["rotations"], cls.to(DEV), n_cls)
        (dist * g_dist).sum().backward()
        return dist.detach(), radii, {k: t[k].grad for k in names}, m2d.grad

    dist, radii, grads, g2d = one_pass(tile)
    assert dist.shape == (n_cls, H, W)
    # (a) the subset renders with the same tile
    sums = {k: torch.zeros_like(grads[k]) for k in names}
    sum2d = torch.zeros_like(g2d)
    for k in range(n_cls):
        idx = (cls == k).to(DEV)
        t = {n: g[n].to(DEV)[idx].clone().requires_grad_() for n in names}
        m2d = torch.zeros(int(idx.sum()), 3, device=DEV, requires_grad=True)
        cols = torch.zeros(int(idx.sum()), 3, device=DEV)
        _, r, allmap = GaussianRasterizer(s, tile=tile)(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], colors_precomp=cols,
                                                         scales=t["scales"], rotations=t["rotations"])
        d = allmap[6]
        assert float((d.detach() - dist[k]).abs().max()) <= 1e-6 * max(1.0, float(d.detach().abs().max())), k
        assert torch.equal(r, radii[idx])
        (d * g_dist[k]).sum().backward()
        for n in names:
            sums[n][idx] += t[n].grad
        sum2d[idx] += m2d.grad
    for n in names:
        assert float(sums[n].abs().max()) > 0
        assert_grads_close(grads[n].cpu().numpy(), sums[n].cpu().numpy(), 2e-5, f"class pass {tile} vs subset renders d{n}", max_bad_frac=0.0, hard=2e-5)
    assert_grads_close(g2d.cpu().numpy(), sum2d.cpu().numpy(), 2e-5, f"class pass {tile} vs subset renders dmeans2D", max_bad_frac=0.0, hard=2e-5)
    assert not grads["means3D"][(cls < 0) | (cls >= n_cls)].any()
    # (b) the 16x16 pass
    dist16, radii16, grads16, g2d16 = one_pass(None)
    both = (radii > 0) & (radii16 > 0)   # (a splat beyond the image's last pixel can still reach the padding of the coarser tile grid)
    assert torch.equal(radii[both], radii16[both]) and float(both.float().mean()) > 0.5
    # (another tiling = other tile-local coordinates: a pair at the alpha >= 1/255 threshold can fall the other way)
    assert_close_frac(dist.cpu().numpy(), dist16.cpu().numpy(), 2e-6, 1e-3, 1e-3, 2e-2, f"class pass {tile} vs 16x16 rend_dist")   # (a cancelling sum: m^2 A + M2 - 2 m M1)
    for n in names:
        assert_grads_close(grads[n].cpu().numpy(), grads16[n].cpu().numpy(), 1e-3, f"class pass {tile} vs 16x16 d{n}")


def test_training_iteration_eight_calls_equal_two_rasterizations():
    """The reference's late training iteration rasterizes one view eight times [REF train.py:84-109]: render_semantic (two 3-channel
    passes), five class-filtered renders on boolean-indexed copies, render.  streetunveiler_amd.train_pattern issues those eight calls
    through the drop-in operator and, beside them, this build's two rasterizations (render_and_semantic + render_class_distortions):
    same maps, same parameter gradients -- what bench.py's `train_step` section times at the C3 size."""
    from streetunveiler_amd.gaussian_renderer import SurfelModel
    from streetunveiler_amd.train_pattern import compare_and_time
    W, H, P = 208, 128, 6000
    cam = synthetic_camera(W, H, index=2).to(DEV)
    g = {k: v.to(DEV).requires_grad_() for k, v in synthetic_gaussians(P, W, H, seed=21, scale_lo=2e-3, scale_hi=3e-2).items()}
    sem = torch.randint(0, 6, (P,), generator=torch.Generator().manual_seed(3)).to(DEV)
    sem[sem == 4] = 2
    pc = SurfelModel(g["means3D"], g["scales"], g["rotations"], g["opacities"], g["shs"], sem, 3, 3)
    r = compare_and_time(cam, pc, torch.tensor([0.1, 0.2, 0.3], device=DEV), list(g.values()), iters=1, warmup=1)
    d = r["max_abs_difference_of_maps"]
    assert d["render"] == 0.0 and d["render_semantics"] == 0.0 and d["rend_dist"] == 0.0 and d["rend_normal"] == 0.0, d   # same blend, bit for bit
    assert d["class_dist"] <= 2e-6, d
    assert r["max_gradient_difference_of_tensor_scale"] <= 5e-5, r
    assert r["reference_8_calls_ms"] > 0 and r["fused_2_calls_ms"] > 0
    # ... and as ONE plan (render_train_view: one K1, one binning, one K8 -- the class backward adds into the colour pass's records)
    d1 = r["one_plan_max_abs_difference_of_maps"]
    assert d1["render"] == 0.0 and d1["render_semantics"] == 0.0 and d1["rend_dist"] == 0.0 and d1["rend_normal"] == 0.0, d1
    assert d1["class_dist"] <= 2e-6, d1
    assert r["one_plan_max_gradient_difference_of_tensor_scale"] <= 5e-5, r
    assert r["one_plan_ms"] > 0


def test_shared_plan_class_pass_equals_the_two_calls_on_every_tile_shape():
    """`forward_with_class_distortions` (sr_class_forward_shared / sr_class_backward_shared) against the operator + `class_distortions` called
    separately: colour, allmap, radii and the class maps bit-identical, every gradient to float summation order -- with SH colours, with
    precomputed colours, with the 9-channel pass, on four tile shapes; a Gaussian only a class chain reaches (occluded in the full render)
    still gets its gradient."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import assert_grads_close, settings_for
    P, W, H = 5000, 216, 130
    cam = synthetic_camera(W, H, index=1)
    g = synthetic_gaussians(P, W, H, seed=33, scale_lo=3e-3, scale_hi=8e-2)
    g["opacities"] = (g["opacities"] * 0.5 + 0.5).contiguous()   # opaque enough that the full render saturates where single classes do not
    n_cls = 4
    cls = torch.randint(-1, n_cls, (P,), generator=torch.Generator().manual_seed(5))
    gen = torch.Generator().manual_seed(6)
    six = torch.rand(P, 6, generator=gen)
    names = ("means3D", "opacities", "scales", "rotations", "shs")
    for tile, mode in [(None, "sh"), ((8, 8), "nine"), ((32, 16), "nine"), ((16, 8), "col3"), ((32, 8), "sh")]:
        nc = 9 if mode == "nine" else 3
        bg = torch.linspace(0.05, 0.6, nc)
        s = settings_for(cam, bg.numpy(), 3 if mode != "col3" else 0)
        w_c = torch.randn(nc, H, W, generator=gen).to(DEV); w_a = torch.randn(7, H, W, generator=gen).to(DEV); w_d = (torch.rand(n_cls, H, W, generator=gen) + 0.5).to(DEV)

        def leaves():
            t = {k: g[k].to(DEV).clone().requires_grad_() for k in names}
            t["m2d"] = torch.zeros(P, 3, device=DEV, requires_grad=True)
            t["six"] = six.to(DEV).clone().requires_grad_()
            return t

        def colour_kw(t):
            if mode == "sh":
                return dict(shs=t["shs"])
            if mode == "nine":
                return dict(shs=t["shs"], extra_colors=t["six"])
            return dict(colors_precomp=t["six"][:, :3].contiguous())

        a = leaves()
        r = GaussianRasterizer(s, tile=tile)
        c1, rad1, al1, d1 = r.forward_with_class_distortions(means3D=a["means3D"], means2D=a["m2d"], opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"],
                                                             classes=cls.to(DEV), n_classes=n_cls, **colour_kw(a))
        ((c1 * w_c).sum() + (al1 * w_a).sum() + 50.0 * (d1 * w_d).sum()).backward()
        b = leaves()
        c2, rad2, al2 = r(means3D=b["means3D"], means2D=b["m2d"], opacities=b["opacities"], scales=b["scales"], rotations=b["rotations"], **colour_kw(b))
        d2, rad3 = r.class_distortions(b["means3D"], b["m2d"], b["opacities"], b["scales"], b["rotations"], cls.to(DEV), n_cls)
        ((c2 * w_c).sum() + (al2 * w_a).sum() + 50.0 * (d2 * w_d).sum()).backward()
        assert torch.equal(c1, c2) and torch.equal(al1, al2) and torch.equal(rad1, rad2) and torch.equal(d1, d2) and torch.equal(rad1, rad3), (tile, mode)
        for k in a:
            if a[k].grad is None:
                assert b[k].grad is None or not b[k].grad.any(), (tile, mode, k)
                continue
            assert float(b[k].grad.abs().max()) > 0, (tile, mode, k)
            assert_grads_close(a[k].grad.cpu().numpy(), b[k].grad.cpu().numpy(), 2e-5, f"one plan {tile} {mode} d{k}", max_bad_frac=1e-4, hard=1e-3)
        # the class chains reach duplicates the saturated full render never blended: gradients only the class pass produces
        only_dist = GaussianRasterizer(s, tile=tile)
        e = leaves()
        dd, _ = only_dist.class_distortions(e["means3D"], e["m2d"], e["opacities"], e["scales"], e["rotations"], cls.to(DEV), n_cls)
        (dd * w_d).sum().backward()
        f = leaves()
        cc, _, aa = only_dist(means3D=f["means3D"], means2D=f["m2d"], opacities=f["opacities"], scales=f["scales"], rotations=f["rotations"], **colour_kw(f))
        ((cc * w_c).sum() + (aa * w_a).sum()).backward()
        class_only = (e["means3D"].grad.abs().sum(1) > 0) & (f["means3D"].grad.abs().sum(1) == 0)
        assert class_only.any(), "the scene has no Gaussian that only a class chain reaches"
        assert (a["means3D"].grad[class_only].abs().sum(1) > 0).all()


def test_shared_plan_class_pass_with_fused_activations_and_mask():
    """The one-plan form under the other opt-in extensions: raw parameters with the activations fused into K1 / K8 (SR_ACT_*) and a
    Gaussian mask read by K1 -- against the same two separate calls of this build."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import assert_grads_close, settings_for
    P, W, H = 4000, 200, 120
    cam = synthetic_camera(W, H, index=6)
    g = synthetic_gaussians(P, W, H, seed=44, scale_lo=3e-3, scale_hi=6e-2)
    raw = dict(means3D=g["means3D"], shs=g["shs"], opacities=torch.logit(g["opacities"].clamp(1e-4, 1 - 1e-4)), scales=torch.log(g["scales"]),
               rotations=g["rotations"] * 1.7)
    n_cls = 5
    cls = torch.randint(0, n_cls, (P,), generator=torch.Generator().manual_seed(2)).to(DEV)
    mask = (torch.rand(P, generator=torch.Generator().manual_seed(3)) < 0.7).to(DEV)
    s = settings_for(cam, np.array([0.3, 0.2, 0.1], np.float32), 3)
    gen = torch.Generator().manual_seed(4)
    w_c = torch.randn(3, H, W, generator=gen).to(DEV); w_a = torch.randn(7, H, W, generator=gen).to(DEV); w_d = (torch.rand(n_cls, H, W, generator=gen) + 0.5).to(DEV)

    def leaves():
        t = {k: v.to(DEV).clone().requires_grad_() for k, v in raw.items()}
        t["m2d"] = torch.zeros(P, 3, device=DEV, requires_grad=True)
        return t

    r = GaussianRasterizer(s, fused_activations=True)
    a = leaves()
    c1, rad1, al1, d1 = r.forward_with_class_distortions(means3D=a["means3D"], means2D=a["m2d"], opacities=a["opacities"], scales=a["scales"], rotations=a["rotations"],
                                                         classes=cls, n_classes=n_cls, shs=a["shs"], mask=mask)
    ((c1 * w_c).sum() + (al1 * w_a).sum() + 20.0 * (d1 * w_d).sum()).backward()
    b = leaves()
    c2, rad2, al2 = r(means3D=b["means3D"], means2D=b["m2d"], opacities=b["opacities"], scales=b["scales"], rotations=b["rotations"], shs=b["shs"], mask=mask)
    d2, _ = r.class_distortions(b["means3D"], b["m2d"], b["opacities"], b["scales"], b["rotations"], cls, n_cls, mask=mask)
    ((c2 * w_c).sum() + (al2 * w_a).sum() + 20.0 * (d2 * w_d).sum()).backward()
    assert torch.equal(c1, c2) and torch.equal(al1, al2) and torch.equal(rad1, rad2) and torch.equal(d1, d2)
    assert not rad1[~mask].any() and (rad1[mask] > 0).any()
    for k in a:
        assert float(b[k].grad.abs().max()) > 0, k
        assert not a[k].grad[~mask].any(), k
        assert_grads_close(a[k].grad.cpu().numpy(), b[k].grad.cpu().numpy(), 2e-5, f"one plan, fused activations + mask, d{k}", max_bad_frac=1e-4, hard=1e-3)


def test_class_passes_on_empty_and_fully_culled_scenes():
    """Edge cases of the per-class pass and of its one-plan form: no Gaussians at all, and Gaussians that are all behind the camera
    (P > 0, D = 0) -- zero distortion maps, the background colour, zero gradients, no kernel faults."""
    from diff_surfel_rasterization import GaussianRasterizer
    from tests.gpu_util import settings_for
    W, H = 70, 45
    cam = synthetic_camera(W, H)
    bg = np.array([0.2, 0.4, 0.6], np.float32)
    s = settings_for(cam, bg, 3)
    for P, behind in ((0, False), (300, True)):
        g = synthetic_gaussians(max(P, 1), W, H, seed=2)
        g = {k: v[:P].clone() for k, v in g.items()}
        if behind:
            g["means3D"][:, 2] *= -1
        t = {k: v.to(DEV).requires_grad_() for k, v in g.items()}
        cls = torch.zeros(P, dtype=torch.int32, device=DEV)
        m2d = torch.zeros(P, 3, device=DEV, requires_grad=True)
        r = GaussianRasterizer(s)
        dist, radii = r.class_distortions(t["means3D"], m2d, t["opacities"], t["scales"], t["rotations"], cls, 3)
        assert dist.shape == (3, H, W) and not dist.any() and not radii.any()
        c, rad, a, d1 = r.forward_with_class_distortions(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                                          classes=cls, n_classes=3, shs=t["shs"])
        assert not d1.any() and not a.any() and not rad.any()
        assert torch.allclose(c, torch.as_tensor(bg).to(DEV)[:, None, None].expand(3, H, W))
        (c.sum() + a.sum() + d1.sum() + dist.sum()).backward()
        for k, v in t.items():
            assert v.grad is None or not v.grad.any(), k
