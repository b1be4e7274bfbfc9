"""`diff_surfel_rasterization._C` -- the three native entry points the reference's autograd shim calls,
implemented on the C-ABI of libsurfel_raster.so (HIP, gfx950).

Call shapes follow the reference extension (SURVEY.md 8b "Native signatures"; the only call site of
the package is /root/reference/gaussian_renderer/__init__.py:11,129-138):

    rasterize_gaussians(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier,
                        transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree,
                        campos, prefiltered, debug)
        -> (num_rendered, color[3,H,W], allmap[7,H,W], radii[P] i32, geomBuffer, binningBuffer, imgBuffer)
    rasterize_gaussians_backward(bg, means3D, radii, colors_precomp, scales, rotations, scale_modifier,
                        transMat_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dcolor, dL_dallmap,
                        sh, degree, campos, geomBuffer, num_rendered, binningBuffer, imgBuffer, debug)
        -> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dtransMat[P,9],
            dL_dsh[P,M,3], dL_dscales[P,2], dL_drotations[P,4])
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]

Empty tensors stand for "not provided", as in the reference.  No CPU fallback: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from streetunveiler_amd import _lib as L


# Tracing: with SURFEL_ROCTX=1 every operator entry point is bracketed by a roctx range (libroctx64: rocprofv3 --marker-trace
# shows "surfel:forward" / "surfel:backward" / "surfel:class_forward" / ... around the kernels of one call).  Off by default: no library is
# loaded and the context manager is a no-op.
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        import os
        if os.environ.get("SURFEL_ROCTX") == "1":
            for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
                try:
                    lib = C.CDLL(name)
                    lib.roctxRangePushA.argtypes = [C.c_char_p]; lib.roctxRangePushA.restype = C.c_int
                    lib.roctxRangePop.restype = C.c_int
                    _roctx = lib
                    break
                except OSError:
                    continue
    return _roctx


class _range:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        lib = _roctx_lib()
        self.on = bool(lib)
        if self.on:
            lib.roctxRangePushA(("surfel:" + self.name).encode())
        return self

    def __exit__(self, *exc):
        if self.on:
            _roctx.roctxRangePop()
        return False


def _ptr(t: torch.Tensor):
    return None if t is None or t.numel() == 0 else C.c_void_p(t.data_ptr())


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if t is None:
        return None
    if not t.is_cuda:
        raise L.SurfelRasterError(f"{name} must be a CUDA (ROCm) tensor; the surfel rasterizer has no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# SURFEL_EXTRA_FLAGS: SrFrame.flags bits OR-ed into every call of this process (A/B runs of opt-in switches such as SR_FLAG_ONE_SWEEP_SORT
# under an unchanged caller; results are bit-identical by the switches' own contracts).
_EXTRA_FLAGS = int(os.environ.get("SURFEL_EXTRA_FLAGS", "0"), 0)


def _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, degree, campos, prefiltered, debug, tile=None,
           quadrant_cull=True, blend_counters=None, ballot_ranking=False, row_mapped=None, forward_only=False, no_precomp_color_grad=False,
           binning_capacity=None, backward_kernel=None):
    keep = [_f32c(bg, "bg"), _f32c(viewmatrix, "viewmatrix"), _f32c(projmatrix, "projmatrix"), _f32c(campos, "campos"), blend_counters]
    if blend_counters is not None and (blend_counters.dtype != torch.int64 or blend_counters.numel() < 16 or not blend_counters.is_cuda):
        raise L.SurfelRasterError("blend_counters must be a CUDA (ROCm) int64 tensor with 16 entries")
    fr = L.SrFrame(int(H), int(W), float(tan_fovx), float(tan_fovy), float(scale_modifier), int(degree),
                   int(bool(prefiltered)), int(bool(debug)), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]),
                   int(tile[0]) if tile else 0, int(tile[1]) if tile else 0, (0 if quadrant_cull else L.SR_FLAG_NO_QUADRANT_CULL) | (L.SR_FLAG_BALLOT_RANKING if ballot_ranking else 0) |
                   (0 if row_mapped is None else (L.SR_FLAG_ROW_MAPPED_FORWARD if row_mapped else L.SR_FLAG_QUADRANT_MAPPED_FORWARD)) |
                   (L.SR_FLAG_FORWARD_ONLY if forward_only else 0) | (L.SR_FLAG_NO_PRECOMP_COLOR_GRAD if no_precomp_color_grad else 0) |
                   (L.SR_FLAG_BINNING_CAPACITY if binning_capacity is not None else 0) |
                   ({None: 0, "one_wave": L.SR_FLAG_ONE_WAVE_BACKWARD, "coop": L.SR_FLAG_COOP_BACKWARD, "rows": L.SR_FLAG_ROW_BACKWARD}[backward_kernel]) | _EXTRA_FLAGS,
                   _ptr(blend_counters))
    return fr, keep


def _mask(mask, P, dev):
    """[P] bool / uint8 tensor or None -> contiguous uint8 storage on `dev` (torch.bool is one byte per element)."""
    if mask is None:
        return None
    if mask.numel() != P or mask.dtype not in (torch.bool, torch.uint8):
        raise L.SurfelRasterError("mask must be a bool tensor with one entry per Gaussian")
    if not mask.is_cuda:
        raise L.SurfelRasterError("mask must be a CUDA (ROCm) tensor")
    return mask.contiguous().view(torch.uint8)


def _channels(colors_precomp):
    """3 as in the reference, or 6: two 3-channel passes over the same geometry folded into one."""
    if colors_precomp is None or colors_precomp.numel() == 0:
        return 3
    nc = int(colors_precomp.shape[-1])
    if colors_precomp.ndim != 2 or nc not in (3, 6):
        raise L.SurfelRasterError("colors_precomp must have dimensions (num_points, 3) or (num_points, 6)")
    return nc


def _gaussians(means3D, opacities, scales, rotations, sh, colors_precomp, transMat_precomp, activations=0, mask=None):
    P = int(means3D.shape[0])
    M = int(sh.shape[1]) if sh is not None and sh.numel() else 0
    NC = _channels(colors_precomp)
    if M and NC == 6:
        NC = 9   # SH colour + six precomputed channels in one pass
    elif M and colors_precomp is not None and colors_precomp.numel():
        raise L.SurfelRasterError("Please provide exactly one of either SHs or precomputed colors!")
    g = L.SrGaussians(P, M, NC, int(activations), _ptr(means3D), _ptr(opacities), _ptr(scales), _ptr(rotations), _ptr(sh),
                      _ptr(colors_precomp), _ptr(transMat_precomp), _ptr(mask))
    return g


def rasterize_gaussians(bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, transMat_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, activations=0, tile=None, mask=None, quadrant_cull=True, blend_counters=None,
                        ballot_ranking=False, row_mapped=None, forward_only=False, classes=None, n_classes=0, binning_capacity=None, backward_kernel=None):
    """`classes` [P] integer tensor + `n_classes` (extension, SURVEY 8f N1 in full): the per-class distortion pass runs on the plan AND the
    binning of this very render (sr_class_forward_shared); the return tuple then ends with (dist[n_classes,H,W], class_state) and
    rasterize_gaussians_backward takes `class_state` / `dL_ddist` to return the gradients of colour, allmap and distortion maps from ONE K8.
    `forward_only=True` (SR_FLAG_FORWARD_ONLY): no backward will follow -- what the reference's inference callers do under
    torch.no_grad() [REF /root/reference/render.py:68; utils/mesh_utils.py:82-100].  color / allmap / radii are bit-identical; the state
    only a backward reads is not written (imgBuffer comes back empty, the SH direction Jacobian and the hit masks stay unwritten), so the
    returned buffers must not be handed to rasterize_gaussians_backward.
    `tile` = (width, height) of the binning tile, default the reference's 16x16 (BASELINE config 5 sweeps 8x8, 16x8,
    16x16, 32x8, 32x16); the backward must be given the same shape.  `quadrant_cull=False` / `blend_counters` (int64[16], device):
    per-call SrFrame.flags / SrFrame.blend_counters (tests and profiling; results are identical).  `ballot_ranking=True`
    (SR_FLAG_BALLOT_RANKING): the binning of this call ranks with match-any ballots, the fallback of the LDS-atomic ranking.
    `row_mapped=True` / `False` (SR_FLAG_ROW_MAPPED_FORWARD / SR_FLAG_QUADRANT_MAPPED_FORWARD): force one of the two forward blend kernels
    (bit-identical results); None: the device picks per frame.
    `binning_capacity=N` (SR_FLAG_BINNING_CAPACITY, round 6): the forward WITHOUT the host read-back of D -- the binning buffer is sized for N
    duplicates, the returned `num_rendered` is N (hand it to the backward as usual, with the same `binning_capacity`), nothing in the call
    waits for the GPU, and the whole call can be captured into a HIP graph.  Whether the frame fitted is decided on the device:
    `forward_status(geom, P)` -> device int32 [D, visible, overflow]; overflow = 1 means nothing was rendered (background image, zero gradients) and
    the frame has to be rendered again with at least D items.  Results of a frame that fits are bit-identical to the default mode."""
    lib = L.load()
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise L.SurfelRasterError("means3D must have dimensions (num_points, 3)")
    means3D = _f32c(means3D, "means3D"); opacities = _f32c(opacities, "opacities")
    colors_precomp = _f32c(colors_precomp, "colors_precomp"); scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations"); transMat_precomp = _f32c(transMat_precomp, "transMat_precomp")
    sh = _f32c(sh, "sh")
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    with torch.cuda.device(dev), _range("forward"):
        if binning_capacity is not None and (classes is not None or int(binning_capacity) < 0):
            raise L.SurfelRasterError("binning_capacity: a non-negative number of duplicates; not with the shared-plan class pass")
        fr, keep = _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, degree, campos, prefiltered, debug, tile,
                          quadrant_cull, blend_counters, ballot_ranking, row_mapped, forward_only, binning_capacity=binning_capacity,
                          backward_kernel=backward_kernel)   # ("one_wave" / "coop" also force the forward's band / cooperative kernel)
        mask = _mask(mask, P, dev)
        g = _gaussians(means3D, opacities, scales, rotations, sh, colors_precomp, transMat_precomp, activations, mask)
        if keep[0].numel() != g.color_channels:
            raise L.SurfelRasterError(f"bg must have {g.color_channels} entries, one per colour channel")
        color = torch.empty((g.color_channels, H, W), dtype=torch.float32, device=dev)
        allmap = torch.empty((7, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty((lib.sr_geom_bytes(P),), dtype=torch.uint8, device=dev)
        img = torch.empty((0 if forward_only else lib.sr_image_bytes(W, H),), dtype=torch.uint8, device=dev)
        stream = _stream(dev)
        D = C.c_uint32(0)
        L.check(lib.sr_forward_plan(C.byref(fr), C.byref(g), _ptr(geom), geom.numel(), _ptr(radii), C.byref(D), stream),
                "sr_forward_plan")
        num_rendered = int(D.value) if binning_capacity is None else (max(1, int(binning_capacity)) if P else 0)
        binning = torch.empty((lib.sr_binning_bytes(P, num_rendered, W, H),), dtype=torch.uint8, device=dev)
        L.check(lib.sr_forward_render(C.byref(fr), C.byref(g), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                      _ptr(img), img.numel(), num_rendered, _ptr(color), _ptr(allmap), stream),
                "sr_forward_render")
        if classes is not None and int(n_classes) > 0:
            if classes.numel() != P:
                raise L.SurfelRasterError("classes must have one entry per Gaussian")
            if forward_only:
                raise L.SurfelRasterError("the shared-plan class pass is a training pass: not with forward_only")
            cls = classes.to(device=dev, dtype=torch.int32).contiguous()
            dist = torch.empty((int(n_classes), H, W), dtype=torch.float32, device=dev)
            cstate = torch.empty((lib.sr_class_shared_bytes(P, W, H, int(n_classes), num_rendered),), dtype=torch.uint8, device=dev)
            with _range("class_forward"):
                L.check(lib.sr_class_forward_shared(C.byref(fr), C.byref(g), int(n_classes), _ptr(cls), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                                    _ptr(cstate), cstate.numel(), num_rendered, _ptr(dist), stream), "sr_class_forward_shared")
            del keep
            return num_rendered, color, allmap, radii, geom, binning, img, dist, cstate
    del keep
    return num_rendered, color, allmap, radii, geom, binning, img


def rasterize_gaussians_backward(bg, means3D, radii, colors_precomp, scales, rotations, scale_modifier, transMat_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dcolor, dL_dallmap, sh, degree, campos,
                                 geomBuffer, num_rendered, binningBuffer, imgBuffer, debug, opacities=None, defer_sh=False,
                                 activations=0, tile=None, after_blend=None, class_state=None, dL_ddist=None, n_classes=0, want_precomp_color_grad=True,
                                 binning_capacity=None, backward_kernel=None):
    """`backward_kernel`: None = the library picks the blend backward by the frame's tile count (the cooperative four-waves-per-tile kernel
    below 2 600 tiles of 16x16, one wave per tile above); "one_wave" / "coop" force one (SR_FLAG_ONE_WAVE_BACKWARD / SR_FLAG_COOP_BACKWARD: A/B, tests).
    `opacities` is not needed (opacity is kept in the packed geometry state); accepted for symmetry.

    `binning_capacity`: the value the forward was given (then `num_rendered` is that capacity): SR_FLAG_BINNING_CAPACITY for the backward too.

    `want_precomp_color_grad=False` (6 / 9 colour channels): dL/dcolors_precomp is not wanted (SR_FLAG_NO_PRECOMP_COLOR_GRAD; an empty
    tensor comes back for it) -- what the autograd shim passes when colors_precomp / extra_colors does not require grad.
    `defer_sh=True` (frame-parallel ranks, streetunveiler_amd.parallel): with SHs as the colour source, dL_dsh is NOT
    expanded (empty tensor returned) and dL_dcolors carries the clamp-masked dL/drgb [P,3] to be all-gathered and expanded
    with `sh_gradient_expand`.  `after_blend(dL_dcolors)`: with `defer_sh`, called between the two halves of the backward
    (sr_backward_blend -> sr_backward_colors -> HERE -> sr_backward_geometry) with the [P,3] colour gradients already final, so that
    a frame-parallel rank can put its all-gather on the wire while K8 still runs."""
    lib = L.load()
    means3D = _f32c(means3D, "means3D")
    colors_precomp = _f32c(colors_precomp, "colors_precomp"); scales = _f32c(scales, "scales")
    rotations = _f32c(rotations, "rotations"); transMat_precomp = _f32c(transMat_precomp, "transMat_precomp")
    sh = _f32c(sh, "sh")
    dL_dcolor = _f32c(dL_dcolor, "dL_dcolor"); dL_dallmap = _f32c(dL_dallmap, "dL_dallmap")
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_dcolor.shape[1]), int(dL_dcolor.shape[2])
    M = int(sh.shape[1]) if sh is not None and sh.numel() else 0
    with torch.cuda.device(dev), _range("backward"):
        skip_cg = (not want_precomp_color_grad) and _channels(colors_precomp) == 6   # ([P,6] precomputed channels: the 6- and the 9-channel pass)
        fr, keep = _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, degree, campos, False, debug, tile,
                          no_precomp_color_grad=skip_cg, binning_capacity=binning_capacity, backward_kernel=backward_kernel)
        # the backward never dereferences opacities (it reads the packed record); pass means3D as a non-NULL stand-in
        g = _gaussians(means3D, means3D, scales, rotations, sh, colors_precomp, transMat_precomp, activations)
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        has = lambda t: t is not None and t.numel() > 0
        # Gradients of inputs that were not provided are not computed (empty tensors, `None` for autograd).  The parameter
        # gradients are carved out of ONE flat allocation (means3D | sh | opacity | scales | rotations, 58 floats per
        # Gaussian with SH degree 3) so that a data-parallel step can all-reduce them with a single collective
        # (streetunveiler_amd.parallel.allreduce_gradients recognises the shared storage).
        defer_sh = bool(defer_sh) and has(sh) and not has(colors_precomp)   # (the 9-channel pass keeps its SH gradient local)
        sizes = [("means3D", (P, 3)), ("sh", (P, M, 3) if has(sh) and not defer_sh else (0, 0, 3)), ("opacity", (P, 1)),
                 ("scales", (P, 2) if has(scales) else (0, 2)), ("rotations", (P, 4) if has(rotations) else (0, 4))]
        numel = lambda shp: int(torch.Size(shp).numel())
        flat = torch.empty(sum(numel(shp) for _, shp in sizes), dtype=torch.float32, device=dev)
        views, off = {}, 0
        for name, shp in sizes:
            n = numel(shp)
            views[name] = flat[off:off + n].view(shp)
            off += n
        dL_dmeans3D, dL_dsh, dL_dopacity, dL_dscales, dL_drotations = (views[k] for k in ("means3D", "sh", "opacity", "scales", "rotations"))
        dL_dmeans2D = e(P, 3)
        NC = g.color_channels
        if int(dL_dcolor.shape[0]) != NC or keep[0].numel() != NC:
            raise L.SurfelRasterError(f"dL_dcolor / bg must have {NC} channels")
        dL_dcolors = e(P, 6 if NC == 9 else NC) if (has(colors_precomp) and not skip_cg) or defer_sh else e(0, 3)
        dL_dtransMat = e(P, 9) if has(transMat_precomp) else e(0, 9)
        ws = torch.empty((lib.sr_backward_workspace_bytes(P, int(num_rendered), NC),), dtype=torch.uint8, device=dev)
        grads = L.SrGradients(_ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dtransMat),
                              _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations))
        if class_state is not None:   # the shared-plan class pass: K7 -> class backward into the same records -> ONE K8
            if defer_sh or after_blend is not None:
                # (a frame-parallel caller would get a locally expanded dL_dsh and no exchange hook -- silently)
                raise L.SurfelRasterError("the shared-plan class pass (class_state) does not combine with the factored SH exchange "
                                          "(defer_sh / after_blend): exchange its gradients with the plain all-reduce")
            dL_ddist = _f32c(dL_ddist, "dL_ddist")
            L.check(lib.sr_backward_blend(C.byref(fr), C.byref(g), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                          binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(), int(num_rendered), _ptr(dL_dcolor),
                                          _ptr(dL_dallmap), _ptr(ws), ws.numel(), _stream(dev)), "sr_backward_blend")
            with _range("class_backward"):
                L.check(lib.sr_class_backward_shared(C.byref(fr), C.byref(g), int(n_classes), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                                     binningBuffer.numel(), _ptr(class_state), class_state.numel(), int(num_rendered), _ptr(dL_ddist),
                                                     _ptr(ws), ws.numel(), _stream(dev)), "sr_class_backward_shared")
            L.check(lib.sr_backward_geometry(C.byref(fr), C.byref(g), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                             binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(), int(num_rendered), _ptr(ws),
                                             ws.numel(), C.byref(grads), _stream(dev)), "sr_backward_geometry")
        elif after_blend is not None and defer_sh and NC == 3:
            L.check(lib.sr_backward_blend(C.byref(fr), C.byref(g), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                          binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(), int(num_rendered), _ptr(dL_dcolor),
                                          _ptr(dL_dallmap), _ptr(ws), ws.numel(), _stream(dev)), "sr_backward_blend")
            L.check(lib.sr_backward_colors(C.byref(fr), C.byref(g), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(), int(num_rendered),
                                           _ptr(ws), ws.numel(), _ptr(dL_dcolors), _stream(dev)), "sr_backward_colors")
            after_blend(dL_dcolors)
            grads.dL_dcolors = None      # already final; K8 need not write it again
            L.check(lib.sr_backward_geometry(C.byref(fr), C.byref(g), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                             binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(), int(num_rendered), _ptr(ws),
                                             ws.numel(), C.byref(grads), _stream(dev)), "sr_backward_geometry")
        else:
            L.check(lib.sr_backward(C.byref(fr), C.byref(g), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(),
                                    _ptr(binningBuffer), binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(),
                                    int(num_rendered), _ptr(dL_dcolor), _ptr(dL_dallmap), _ptr(ws), ws.numel(),
                                    C.byref(grads), _stream(dev)), "sr_backward")
    del keep
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dtransMat, dL_dsh, dL_dscales, dL_drotations


def class_distortions(bg, means3D, classes, opacities, scales, rotations, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                      image_height, image_width, campos, n_classes, debug=False, activations=0, mask=None, tile=None):
    """Per-class distortion pass, forward (sr_forward_plan + sr_class_forward_render): `classes` [P] integer class of every
    Gaussian (negative or >= n_classes: in no class).  -> (num_rendered, dist[n_classes,H,W], radii, class_cols, geom, binning, class_image)."""
    lib = L.load()
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise L.SurfelRasterError("means3D must have dimensions (num_points, 3)")
    means3D = _f32c(means3D, "means3D"); opacities = _f32c(opacities, "opacities"); scales = _f32c(scales, "scales"); rotations = _f32c(rotations, "rotations")
    dev = means3D.device
    P, H, W = int(means3D.shape[0]), int(image_height), int(image_width)
    if classes.numel() != P:
        raise L.SurfelRasterError("classes must have one entry per Gaussian")
    cols = torch.zeros((P, 3), dtype=torch.float32, device=dev)     # class id in the first colour slot of the splat record
    cols[:, 0] = classes.to(device=dev, dtype=torch.float32).reshape(P)
    with torch.cuda.device(dev), _range("class_forward"):
        fr, keep = _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, 0, campos, False, debug, tile)
        mask = _mask(mask, P, dev)
        g = _gaussians(means3D, opacities, scales, rotations, None, cols, None, activations, mask)
        dist = torch.empty((int(n_classes), H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty((lib.sr_geom_bytes(P),), dtype=torch.uint8, device=dev)
        cimg = torch.empty((lib.sr_class_image_bytes(W, H, int(n_classes)),), dtype=torch.uint8, device=dev)
        stream = _stream(dev)
        D = C.c_uint32(0)
        L.check(lib.sr_forward_plan(C.byref(fr), C.byref(g), _ptr(geom), geom.numel(), _ptr(radii), C.byref(D), stream), "sr_forward_plan")
        num_rendered = int(D.value)
        binning = torch.empty((lib.sr_binning_bytes(P, num_rendered, W, H),), dtype=torch.uint8, device=dev)
        L.check(lib.sr_class_forward_render(C.byref(fr), C.byref(g), int(n_classes), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                            _ptr(cimg), cimg.numel(), num_rendered, _ptr(dist), stream), "sr_class_forward_render")
    del keep
    return num_rendered, dist, radii, cols, geom, binning, cimg


def class_distortions_backward(bg, means3D, radii, cols, scales, rotations, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                               dL_ddist, campos, n_classes, geom, num_rendered, binning, cimg, debug=False, activations=0, tile=None):
    """-> (dL_dmeans2D[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dscales[P,2], dL_drotations[P,4]) of the per-class pass."""
    lib = L.load()
    means3D = _f32c(means3D, "means3D"); scales = _f32c(scales, "scales"); rotations = _f32c(rotations, "rotations")
    dL_ddist = _f32c(dL_ddist, "dL_ddist")
    dev = means3D.device
    P = int(means3D.shape[0])
    H, W = int(dL_ddist.shape[1]), int(dL_ddist.shape[2])
    with torch.cuda.device(dev), _range("class_backward"):
        fr, keep = _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, 0, campos, False, debug, tile)
        g = _gaussians(means3D, means3D, scales, rotations, None, cols, None, activations)
        flat = torch.empty(P * 10, dtype=torch.float32, device=dev)     # means3D | opacity | scales | rotations: one buffer, one all-reduce
        dL_dmeans3D, dL_dopacity = flat[:3 * P].view(P, 3), flat[3 * P:4 * P].view(P, 1)
        dL_dscales, dL_drotations = flat[4 * P:6 * P].view(P, 2), flat[6 * P:].view(P, 4)
        dL_dmeans2D = torch.empty((P, 3), dtype=torch.float32, device=dev)
        ws = torch.empty((lib.sr_backward_workspace_bytes(P, int(num_rendered), 3),), dtype=torch.uint8, device=dev)
        grads = L.SrGradients(_ptr(dL_dmeans2D), None, _ptr(dL_dopacity), _ptr(dL_dmeans3D), None, None, _ptr(dL_dscales), _ptr(dL_drotations))
        L.check(lib.sr_class_backward(C.byref(fr), C.byref(g), int(n_classes), _ptr(radii), _ptr(geom), geom.numel(), _ptr(binning), binning.numel(),
                                      _ptr(cimg), cimg.numel(), int(num_rendered), _ptr(dL_ddist), _ptr(ws), ws.numel(), C.byref(grads),
                                      _stream(dev)), "sr_class_backward")
    del keep
    return dL_dmeans2D, dL_dopacity, dL_dmeans3D, dL_dscales, dL_drotations


def pair_decisions(bg, means3D, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos,
                   geomBuffer, num_rendered, binningBuffer, tile=None):
    """Test hook (sr_debug_pair_decisions): -> (valid[D, nq] int64, use3d[D, nq] int64) ballots per (list entry, 8x8 quadrant)."""
    lib = L.load()
    means3D = _f32c(means3D, "means3D")
    dev = means3D.device
    tw, th = tile if tile else (16, 16)
    nq = (tw // 8) * (th // 8)
    valid = torch.zeros((max(int(num_rendered), 1), nq), dtype=torch.int64, device=dev)
    use3d = torch.zeros_like(valid)
    with torch.cuda.device(dev):
        fr, keep = _frame(bg, scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, False, False, tile)
        g = L.SrGaussians(int(means3D.shape[0]), 0, 3, 0, _ptr(means3D), _ptr(means3D), None, None, None, _ptr(means3D), _ptr(means3D), None)
        L.check(lib.sr_debug_pair_decisions(C.byref(fr), C.byref(g), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                                            binningBuffer.numel(), int(num_rendered), _ptr(valid), _ptr(use3d), _stream(dev)),
                "sr_debug_pair_decisions")
    del keep
    return valid[:int(num_rendered)], use3d[:int(num_rendered)]


def sh_gradient_expand(means3D, campos, dL_dcolors, sh_coeffs, degree):
    """dL_dsh [P,M,3] = sum over views v of the SH adjoint of dL_dcolors[v] seen from campos[v] (include/surfel_raster.h)."""
    lib = L.load()
    means3D = _f32c(means3D, "means3D"); campos = _f32c(campos, "campos").reshape(-1, 3); dL_dcolors = _f32c(dL_dcolors, "dL_dcolors")
    P, V = int(means3D.shape[0]), int(campos.shape[0])
    if dL_dcolors.numel() != V * P * 3:
        raise L.SurfelRasterError(f"dL_dcolors must have {V} x {P} x 3 elements")
    out = torch.empty((P, int(sh_coeffs), 3), dtype=torch.float32, device=means3D.device)
    with torch.cuda.device(means3D.device):
        L.check(lib.sr_sh_gradient_expand(P, int(sh_coeffs), int(degree), V, _ptr(means3D), _ptr(campos), _ptr(dL_dcolors),
                                          _ptr(out), _stream(means3D.device)), "sr_sh_gradient_expand")
    return out


def mark_visible(means3D, viewmatrix, projmatrix):
    lib = L.load()
    means3D = _f32c(means3D, "means3D"); viewmatrix = _f32c(viewmatrix, "viewmatrix"); projmatrix = _f32c(projmatrix, "projmatrix")
    P = int(means3D.shape[0])
    present = torch.empty((P,), dtype=torch.bool, device=means3D.device)
    with torch.cuda.device(means3D.device):
        L.check(lib.sr_mark_visible(P, _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix), _ptr(present),
                                    _stream(means3D.device)), "sr_mark_visible")
    return present


# ---- state-buffer views (tests / profiling) ---------------------------------------------------------
def _view(buf: torch.Tensor, ptr, nbytes: int, dtype: torch.dtype) -> torch.Tensor:
    off = int(ptr) - buf.data_ptr()
    return buf[off:off + nbytes].view(dtype)


def forward_status(geom: torch.Tensor, P: int) -> torch.Tensor:
    """Device int32 [3] view into the geometry state: [D (the frame's duplicates), visible Gaussians, overflow].  The third word is written
    by the capacity guard of a `binning_capacity` forward (1 = the frame did not fit: nothing was rendered -- and `visible` was zeroed).  Reading
    it on the host (`.tolist()`) is the caller's choice of when to synchronise."""
    v = L.SrGeomView()
    L.check(L.load().sr_geom_view(_ptr(geom), geom.numel(), int(P), C.byref(v)), "sr_geom_view")
    return _view(geom, v.frame_counts, 12, torch.int32)


def geom_view(geom: torch.Tensor, P: int):
    """Typed views into the geometry state (tests / tools).  `sorted_gid`: only its first frame_counts[1] entries -- the visible Gaussians in
    depth order -- are written (the depth sort drops the culled ones).  `splats`: rows with radii == 0 are UNDEFINED (K1 skips the 128-B
    lines whose rows are all culled: whatever torch.empty left there, possibly NaN) -- index with radii > 0."""
    v = L.SrGeomView()
    L.check(L.load().sr_geom_view(_ptr(geom), geom.numel(), P, C.byref(v)), "sr_geom_view")
    return dict(splats=_view(geom, v.splats, P * 80, torch.float32).view(P, 20),
                depth_keys=_view(geom, v.depth_keys, P * 4, torch.int32), tiles_touched=_view(geom, v.tiles_touched, P * 4, torch.int32),
                clamped=_view(geom, v.clamped, P, torch.uint8), sorted_gid=_view(geom, v.sorted_gid, P * 4, torch.int32),
                frame_counts=_view(geom, v.frame_counts, 8, torch.int32))


def binning_view(binning: torch.Tensor, P: int, D: int, W: int, H: int, tile=(16, 16)):
    v = L.SrBinningView()
    L.check(L.load().sr_binning_view(_ptr(binning), binning.numel(), P, D, W, H, C.byref(v)), "sr_binning_view")
    tiles = ((W + tile[0] - 1) // tile[0]) * ((H + tile[1] - 1) // tile[1])
    ranges = _view(binning, v.ranges, tiles * 8, torch.int32).view(tiles, 2)
    # tile id of every list entry, rebuilt from the ranges (the partition keeps only the permutation)
    counts = (ranges[:, 1] - ranges[:, 0]).long()
    tile_keys = torch.repeat_interleave(torch.arange(tiles, device=binning.device, dtype=torch.int32), counts)
    return dict(tile_keys=tile_keys, point_list=_view(binning, v.point_list, D * 4, torch.int32), ranges=ranges,
                tile_order=_view(binning, v.tile_order, tiles * 4, torch.int32))


def image_view(img: torch.Tensor, W: int, H: int):
    v = L.SrImageView()
    L.check(L.load().sr_image_view(_ptr(img), img.numel(), W, H, C.byref(v)), "sr_image_view")
    return dict(final_T=_view(img, v.final_T, 3 * H * W * 4, torch.float32).view(3, H, W),
                n_contrib=_view(img, v.n_contrib, 2 * H * W * 4, torch.int32).view(2, H, W))
