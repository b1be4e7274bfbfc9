"""CPU: fixtures produced by the reference's OWN python (tests/golden/make_golden.py g56: gaussian_renderer/__init__.py,
utils/point_utils.py, utils/semantic_utils.py executed around a linear stand-in for the un-vendored native rasterizer) pin
  * oracle/postprocess_torch.py -- the checker of the fused HIP post-processing, forward AND autograd gradient;
  * the python mirror of render / render_with_mask / render_semantic / render_semantic_with_mask: masking by
    boolean index and by semantic bit, one-hot class channels, per-pass backgrounds, the top-2 margin that replaces torch.topk
    (exact ties included) and the argmax colour lookup.
The -m gpu counterpart (tests/test_gpu_render_api.py::test_fused_postprocess_matches_reference_fixture) runs the HIP kernels on the
same fixture."""
import os

import numpy as np
import pytest
import torch

from oracle.postprocess_torch import postprocess_allmap as postprocess_allmap_torch
import streetunveiler_amd.gaussian_renderer as gr
from streetunveiler_amd.camera import SimpleCamera

MAPS = ("rend_alpha", "rend_normal", "rend_dist", "surf_depth", "surf_normal", "surf_point")


def load_case(z, ci):
    pre = f"c{ci}_"
    W, H, yaw, tx, ty, tz, ratio, fovx, fovy = z[pre + "meta"]
    cam = SimpleCamera(int(W), int(H), float(fovx), float(fovy), torch.tensor(z[pre + "wvt"]), torch.tensor(z[pre + "full"]), torch.tensor(z[pre + "center"]))
    return pre, cam, float(ratio)


@pytest.fixture(scope="module")
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_render_golden.npz"))


def test_postprocess_checker_matches_the_reference_code(golden):
    for ci in range(int(golden["n_cases"])):
        pre, cam, ratio = load_case(golden, ci)
        for run in ("render", "render_mask"):
            allmap = torch.tensor(golden[pre + run + "_allmap"], requires_grad=True)
            out = postprocess_allmap_torch(cam, ratio, allmap)
            for k in MAPS:
                np.testing.assert_allclose(out[k].detach().numpy(), golden[pre + run + "_" + k], rtol=1e-6, atol=1e-6, err_msg=f"{pre}{run} {k}")
            sum((out[k] * torch.tensor(golden[pre + "up_" + k])).sum() for k in MAPS).backward()
            ref = golden[pre + run + "_allmap_grad"]
            # at alpha == 0 the reference's autograd yields NaN (0 * inf); compared where it is finite
            fin = np.isfinite(ref)
            assert fin.mean() > 0.8
            np.testing.assert_array_equal(np.isfinite(allmap.grad.numpy()), fin)
            np.testing.assert_allclose(allmap.grad.numpy()[fin], ref[fin], rtol=1e-5, atol=1e-6 * np.abs(ref[fin]).max())


class LinearRasterizer:
    """The generator's linear stand-in with the drop-in operator's call surface (any number of colour channels, `mask=`)."""
    weight = attr7 = None
    H = W = 0

    def __init__(self, raster_settings, fused_activations=False, tile=None):
        self.s = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                 mask=None, extra_colors=None):
        cls = LinearRasterizer
        ids = means3D[:, 0].detach().round().long()
        w = torch.tensor(cls.weight, dtype=torch.float64)[:, ids]
        if mask is not None:
            w = w * mask.view(1, -1).double()
        col = colors_precomp if colors_precomp is not None else shs[:, 0, :]
        if extra_colors is not None:
            col = torch.cat([col, extra_colors], 1)
        T = 1.0 - w.sum(1, keepdim=True)
        img = (w @ col.double() + T * self.s.bg.double().view(1, -1)).float().t().reshape(-1, cls.H, cls.W)
        allmap = (w @ torch.tensor(cls.attr7, dtype=torch.float64)[ids]).float().t().reshape(7, cls.H, cls.W)
        radii = torch.ones(ids.shape[0], dtype=torch.int32)
        radii[ids % 5 == 0] = 0
        if mask is not None:
            radii = radii * mask.to(torch.int32)
        return img, radii, allmap


@pytest.fixture()
def mirror_on_cpu(monkeypatch):
    monkeypatch.setattr(gr, "GaussianRasterizer", LinearRasterizer)
    monkeypatch.setattr(gr, "postprocess_allmap", lambda cam, pipe, allmap: postprocess_allmap_torch(cam, pipe.depth_ratio, allmap))


def _model(golden, pre):
    sem = torch.tensor(golden[pre + "sem"])
    N = sem.shape[0]
    xyz = torch.cat([torch.arange(N, dtype=torch.float32).view(-1, 1), torch.zeros(N, 2)], 1)
    return gr.SurfelModel(xyz, torch.ones(N, 2), torch.ones(N, 4), torch.ones(N, 1), torch.tensor(golden[pre + "features"]), sem, 3, 3)


@pytest.mark.parametrize("fused_mask", [False, True])
def test_render_mirror_matches_the_reference_code(golden, mirror_on_cpu, fused_mask):
    for ci in range(int(golden["n_cases"])):
        pre, cam, ratio = load_case(golden, ci)
        LinearRasterizer.weight, LinearRasterizer.attr7 = golden[pre + "weight"], golden[pre + "attr7"]
        LinearRasterizer.H, LinearRasterizer.W = cam.image_height, cam.image_width
        pc = _model(golden, pre)
        pipe = gr.PipelineParams(depth_ratio=ratio, fused_mask=fused_mask)
        bg = torch.tensor(golden[pre + "bg"])
        mask = torch.tensor(golden[pre + "mask"])
        runs = {"render": lambda: gr.render(cam, pc, pipe, bg),
                "render_mask": lambda: gr.render_with_mask(cam, pc, pipe, bg, mask),
                "render_bit_rev": lambda: gr.render(cam, pc, pipe, bg, semantic_filter_bit=0b010110, reverse_semantic=True),
                "render_bit_fwd": lambda: gr.render(cam, pc, pipe, bg, semantic_filter_bit=0b010110, reverse_semantic=False)}
        for name, fn in runs.items():
            out = fn()
            np.testing.assert_allclose(out["render"].numpy(), golden[pre + name + "_render"], rtol=0, atol=1e-6, err_msg=name)
            if fused_mask and name != "render":
                # opt-in extension: full-size radii / visibility (zero where masked out) instead of subset-size
                keep = {"render_mask": mask.numpy(),
                        "render_bit_rev": ((1 << golden[pre + "sem"]) & 0b010110) != 0,
                        "render_bit_fwd": ((1 << golden[pre + "sem"]) & 0b010110) == 0}[name]
                np.testing.assert_array_equal(out["radii"].numpy()[keep], golden[pre + name + "_radii"])
                assert not out["radii"].numpy()[~keep].any()
            else:
                np.testing.assert_array_equal(out["radii"].numpy(), golden[pre + name + "_radii"])
                np.testing.assert_array_equal(out["visibility_filter"].numpy(), golden[pre + name + "_visibility_filter"])
                assert out["viewspace_points"].shape[0] == int(golden[pre + name + "_n"])
            if name in ("render", "render_mask"):
                for k in MAPS:
                    np.testing.assert_allclose(out[k].numpy(), golden[pre + name + "_" + k], rtol=1e-6, atol=1e-6, err_msg=f"{name} {k}")


@pytest.mark.parametrize("fused_mask", [False, True])
def test_semantic_mirror_matches_the_reference_code(golden, mirror_on_cpu, fused_mask):
    """A3 / A4: render_semantics, and the outputs the reference derives with torch.topk(k=2) and argmax
    [REF gaussian_renderer/__init__.py:448-452, 586-590; utils/semantic_utils.py:128-135], incl. exact ties."""
    for ci in range(int(golden["n_cases"])):
        pre, cam, ratio = load_case(golden, ci)
        LinearRasterizer.weight, LinearRasterizer.attr7 = golden[pre + "weight"], golden[pre + "attr7"]
        LinearRasterizer.H, LinearRasterizer.W = cam.image_height, cam.image_width
        pc = _model(golden, pre)
        pipe = gr.PipelineParams(depth_ratio=ratio, fused_mask=fused_mask)
        bg = torch.tensor(golden[pre + "bg"])
        mask = torch.tensor(golden[pre + "mask"])
        runs = {"semantic": lambda: gr.render_semantic(cam, pc, pipe, bg),
                "semantic_mask": lambda: gr.render_semantic_with_mask(cam, pc, pipe, bg, mask),
                "semantic_bit_rev": lambda: gr.render_semantic(cam, pc, pipe, bg, semantic_filter_bit=0b100101, reverse_semantic=True),
                "semantic_bit_fwd": lambda: gr.render_semantic(cam, pc, pipe, bg, semantic_filter_bit=0b100101, reverse_semantic=False)}
        for name, fn in runs.items():
            out = fn()
            assert set(out) == {"render_semantics", "semantic_rgb", "semantic_uncertainty"}
            np.testing.assert_allclose(out["render_semantics"].numpy(), golden[pre + name + "_render_semantics"], rtol=0, atol=1e-6, err_msg=name)
            # the derived maps, from the reference's own class map so that the comparison is exact (ties: rows 3W..4W)
            prob = torch.tensor(golden[pre + name + "_render_semantics"])
            unc, best = gr._top2_margin(prob)
            want_unc = golden[pre + name + "_semantic_uncertainty"]
            assert out["semantic_uncertainty"].shape == want_unc.shape, name     # [1,H,W] from render_semantic_with_mask, [H,W] otherwise
            np.testing.assert_array_equal(unc.numpy(), want_unc.reshape(unc.shape))
            np.testing.assert_array_equal((gr._SEMANTIC_COLOR[best].permute(2, 0, 1) / 255.0).numpy(), golden[pre + name + "_semantic_rgb"])
            np.testing.assert_allclose(out["semantic_uncertainty"].numpy(), want_unc, rtol=0, atol=2e-6)
            same = out["render_semantics"].argmax(0) == prob.argmax(0)
            np.testing.assert_array_equal(out["semantic_rgb"].numpy()[:, same.numpy()], golden[pre + name + "_semantic_rgb"][:, same.numpy()])
            assert same.float().mean() > 0.95


def test_top2_margin_equals_topk_on_random_and_tied_maps():
    g = torch.Generator().manual_seed(0)
    for _ in range(20):
        p = torch.rand(6, 17, 23, generator=g)
        p[:, :3] = p[:1, :3]                      # all six equal
        p[2, 3:6] = p[4, 3:6] = 2.0               # two-way tie for the maximum
        p[1, 6:8] = float("-inf")
        tv, ti = torch.topk(p, k=2, dim=0)
        unc, best = gr._top2_margin(p)
        assert torch.equal(unc, 1.0 - (tv[0] - tv[1]))
        assert torch.equal(p.gather(0, best.unsqueeze(0))[0], tv[0])
        assert torch.equal(best, torch.argmax(p, dim=0))
