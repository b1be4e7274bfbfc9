"""The rasterizer calls of ONE late training iteration of the reference, two ways.

After iteration 27 500 the reference's loop [REF /root/reference/train.py:84-109] renders the same view eight times:
`render_semantic` = two 3-channel one-hot passes [REF gaussian_renderer/__init__.py:417-444], five class-filtered `render(...,
semantic_filter_bit=1 << k, reverse_semantic=True)["rend_dist"]` on boolean-indexed copies of every parameter tensor [REF :89-105], and
the colour `render`.  `reference_pattern` issues exactly those eight operator calls through this build's drop-in operator;
`fused_pattern` gets the same maps from TWO rasterizations: `render_and_semantic` (SH colour + six class channels in one 9-channel
pass) and `render_class_distortions` (one transmittance chain per class).  bench.py times both (`train_step` in its JSON line) and
reports how far the resulting maps are apart; tests/test_gpu_render_api.py checks the equivalence on a small scene.
"""
from __future__ import annotations

from typing import Dict

import torch

from diff_surfel_rasterization import GaussianRasterizer

from .gaussian_renderer import (PipelineParams, _geometry_inputs, _screenspace_points, _settings, concerned_classes_ind_map,
                                concerned_classes_list, render, render_and_semantic, render_class_distortions, render_train_view)

LAMBDA_DIST = 100.0   # opt.lambda_dist's order of magnitude; any fixed weight serves the comparison


def _semantic_two_passes(cam, pc, pipe):
    """render_semantic as the reference does it: `for i in range(0, 6, 3)` -- one rasterization per three one-hot class channels."""
    dev = pc.get_xyz.device
    n_cls = len(concerned_classes_list)
    bg_prob = torch.zeros(n_cls, device=dev)
    bg_prob[concerned_classes_ind_map["sky"]] = 1.0
    one_hot = (pc.get_semantics.view(-1, 1) == torch.arange(n_cls, device=dev).view(1, -1)).float()
    parts = []
    for i in range(0, n_cls, 3):
        screenspace_points = _screenspace_points(pc)
        rasterizer = GaussianRasterizer(raster_settings=_settings(cam, pc, pipe, bg_prob[i:i + 3].contiguous(), 1.0))
        means3D, means2D, opacity, scales, rotations, cov = _geometry_inputs(pc, pipe, screenspace_points, None, 1.0)
        img, _, _ = rasterizer(means3D=means3D, means2D=means2D, shs=None, colors_precomp=one_hot[:, i:i + 3].contiguous(), opacities=opacity,
                               scales=scales, rotations=rotations, cov3D_precomp=cov)
        parts.append(img)
    return torch.cat(parts, dim=0)


def _loss(maps: Dict[str, torch.Tensor], weights: Dict[str, torch.Tensor]) -> torch.Tensor:
    return ((maps["render"] * weights["render"]).sum() + (maps["render_semantics"] * weights["semantics"]).sum()
            + maps["rend_dist"].mean() + (maps["rend_normal"] * weights["normal"]).sum()
            + LAMBDA_DIST * sum(d.mean() for d in maps["class_dist"]))


def reference_pattern(cam, pc, bg, weights) -> Dict[str, torch.Tensor]:
    """Eight operator calls [REF train.py:84-109]: 2 (semantic) + 5 (class-filtered renders, boolean-indexed inputs) + 1 (colour)."""
    pipe = PipelineParams()
    maps = {"render_semantics": _semantic_two_passes(cam, pc, pipe)}
    maps["class_dist"] = [render(cam, pc, pipe, bg, semantic_filter_bit=1 << k, reverse_semantic=True)["rend_dist"]
                          for k, name in enumerate(concerned_classes_list) if name != "sky"]
    main = render(cam, pc, pipe, bg)
    maps.update(render=main["render"], rend_dist=main["rend_dist"], rend_normal=main["rend_normal"])
    maps["loss"] = _loss(maps, weights)
    return maps


def fused_pattern(cam, pc, bg, weights) -> Dict[str, torch.Tensor]:
    """The same maps from two rasterizations: one 9-channel pass, one per-class distortion pass."""
    pipe = PipelineParams()
    both = render_and_semantic(cam, pc, pipe, bg)
    dist = render_class_distortions(cam, pc, pipe, bg)["rend_dist"]
    maps = dict(render=both["render"], render_semantics=both["render_semantics"], rend_dist=both["rend_dist"], rend_normal=both["rend_normal"],
                class_dist=[dist[j] for j in range(dist.shape[0])])
    maps["loss"] = _loss(maps, weights)
    return maps


def one_plan_pattern(cam, pc, bg, weights) -> Dict[str, torch.Tensor]:
    """The same maps from ONE plan: preprocess, binning and the per-Gaussian backward run once for the 9-channel render AND the per-class
    distortion pass (`render_train_view`)."""
    out = render_train_view(cam, pc, PipelineParams(), bg)
    dist = out["class_rend_dist"]
    maps = dict(render=out["render"], render_semantics=out["render_semantics"], rend_dist=out["rend_dist"], rend_normal=out["rend_normal"],
                class_dist=[dist[j] for j in range(dist.shape[0])])
    maps["loss"] = _loss(maps, weights)
    return maps


def make_weights(H: int, W: int, device, seed: int = 5) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    return {"render": torch.randn(3, H, W, generator=g).to(device), "semantics": torch.randn(6, H, W, generator=g).to(device),
            "normal": torch.randn(3, H, W, generator=g).to(device) * 0.1}


def compare_and_time(cam, pc, bg, leaves, iters: int = 3, warmup: int = 1) -> Dict:
    """ms per forward + backward of either pattern, and the largest differences between their maps / parameter gradients."""
    dev = pc.get_xyz.device
    weights = make_weights(int(cam.image_height), int(cam.image_width), dev)
    out, grads, keep = {}, {}, {}
    for name, fn in (("reference_8_calls", reference_pattern), ("fused_2_calls", fused_pattern), ("one_plan", one_plan_pattern)):
        def step():
            for t in leaves:
                t.grad = None
            m = fn(cam, pc, bg, weights)
            m["loss"].backward()
            return m
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(iters):
            m = step()
        t1.record(); torch.cuda.synchronize()
        out[name + "_ms"] = round(t0.elapsed_time(t1) / iters, 3)
        keep[name] = {k: (v.detach() if torch.is_tensor(v) else [x.detach() for x in v]) for k, v in m.items() if k != "loss"}
        grads[name] = [t.grad.detach().clone() for t in leaves]
        del m
    a, b = keep["reference_8_calls"], keep["fused_2_calls"]
    diff = {k: float((a[k] - b[k]).abs().max()) for k in ("render", "render_semantics", "rend_dist", "rend_normal")}
    diff["class_dist"] = max(float((x - y).abs().max()) for x, y in zip(a["class_dist"], b["class_dist"]))
    out["max_abs_difference_of_maps"] = diff
    out["max_gradient_difference_of_tensor_scale"] = max(float((x - y).abs().max() / (x.abs().max() + 1e-30)) for x, y in zip(grads["reference_8_calls"], grads["fused_2_calls"]))
    c = keep["one_plan"]
    d1 = {k: float((a[k] - c[k]).abs().max()) for k in ("render", "render_semantics", "rend_dist", "rend_normal")}
    d1["class_dist"] = max(float((x - y).abs().max()) for x, y in zip(a["class_dist"], c["class_dist"]))
    out["one_plan_max_abs_difference_of_maps"] = d1
    out["one_plan_max_gradient_difference_of_tensor_scale"] = max(float((x - y).abs().max() / (x.abs().max() + 1e-30)) for x, y in zip(grads["reference_8_calls"], grads["one_plan"]))
    out["speedup"] = round(out["reference_8_calls_ms"] / out["fused_2_calls_ms"], 2)
    out["one_plan_speedup"] = round(out["reference_8_calls_ms"] / out["one_plan_ms"], 2)
    out["pattern"] = ("one late training iteration of the reference (train.py:84-109): render + render_semantic (2 passes) + 5 class-filtered renders = "
                      "8 operator calls with boolean-indexed inputs, against render_and_semantic + render_class_distortions = 2 rasterizations, against render_train_view = "
                      "both on ONE preprocess / binning / per-Gaussian backward; "
                      "fwd+bwd incl. the allmap post-processing and the loss kernels, untimed extra section")
    return out
