"""Drop-in `diff_surfel_rasterization` package for StreetUnveiler's render operator.

Exports exactly what the reference imports -- `GaussianRasterizationSettings`, `GaussianRasterizer`
[REF /root/reference/gaussian_renderer/__init__.py:11] -- with the same constructor / call keywords
[REF :39-54, :129-138] and the same 3-tuple result `(color[3,H,W], radii[P] int32, allmap[7,H,W])`
[REF :129, channel meaning :149-165].  The native half (`_C`) is hand-written HIP for gfx950 behind
a C-ABI (include/surfel_raster.h); the upstream CUDA extension it replaces is an un-vendored
submodule of the reference (/root/reference/.gitmodules:9-12).

Gradients flow to means3D, means2D (densification proxy, [P,3] with z = 0), shs, colors_precomp,
opacities, scales, rotations and cov3D_precomp (= a precomputed [P,9] transMat in 2DGS), through
both `color` and `allmap`.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _cpu_snapshot(args):
    return tuple(a.detach().cpu().clone() if isinstance(a, torch.Tensor) else a for a in args)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        activations=0, tile=None, mask=None, probe=None):
    # Will this call be backpropagated?  Decided HERE, on the caller's thread: inside an autograd.Function's forward grad mode is always off
    # and ctx.needs_input_grad ignores torch.no_grad().  The inference callers of the operator all run under no_grad
    # [REF /root/reference/render.py:68; /root/reference/utils/mesh_utils.py:82-100]: the forward then skips the state only a backward reads.
    forward_only = not (torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in
                                                        (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings, activations, tile, mask, probe, forward_only)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                activations=0, tile=None, mask=None, probe=None, forward_only=False):
        s = raster_settings
        ctx.activations = int(activations)
        ctx.tile = tuple(int(t) for t in tile) if tile else None
        fused = {"activations": ctx.activations} if ctx.activations else {}
        if ctx.tile:
            fused["tile"] = ctx.tile
        if mask is not None:   # only the forward looks at it: masked-out Gaussians get radius 0 and, with that, zero gradients
            fused["mask"] = mask
        status_holder = None
        if probe:              # tests / profiling: {"quadrant_cull": bool, "blend_counters": int64[16] device tensor, "ballot_ranking": bool}
            probe = dict(probe)
            status_holder = probe.pop("_status", None)
            ctx.backward_kernel = probe.get("backward_kernel")   # (forward AND backward: the few-tile kernels are picked by one rule)
            fused.update(probe)
        else:
            ctx.backward_kernel = None
        ctx.binning_capacity = fused.get("binning_capacity")
        if forward_only and not (probe and "blend_counters" in probe):   # SR_FLAG_FORWARD_ONLY: no backward will follow -- images bit-identical, backward state not written (the counting variant keeps the full forward)
            fused["forward_only"] = True
        args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp, s.viewmatrix,
                s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh, s.sh_degree, s.campos, s.prefiltered,
                s.debug)
        if s.debug:
            snapshot = _cpu_snapshot(args)  # taken before the call so a crashing kernel cannot corrupt it
            try:
                out = _C.rasterize_gaussians(*args, **fused)
            except Exception:
                torch.save(snapshot, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            out = _C.rasterize_gaussians(*args, **fused)
        num_rendered, color, allmap, radii, geomBuffer, binningBuffer, imgBuffer = out
        if status_holder is not None:
            status_holder["status"] = _C.forward_status(geomBuffer, means3D.shape[0]) if means3D.shape[0] else None
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        # frame-parallel ranks may exchange the SH gradient in factored form (streetunveiler_amd.parallel): the exchange of the enclosing
        # `factored_sh_exchange` block rides on this node -- the backward runs on autograd's thread and consults no global state.
        # (Only when the SHs are the sole colour source: the 9-channel pass keeps its SH gradient local, as _C does.)
        from streetunveiler_amd.parallel import active_sh_exchange
        # ... and only for a call that WILL be backpropagated: a no_grad / eval render inside the block must not draw a frame number (it
        # would shift the frame -> camera pairing of the calls that follow it)
        ctx.sh_exchange = active_sh_exchange() if sh.numel() and not colors_precomp.numel() and not forward_only else None
        # which of the step's frames this call is (row of all_campos[rank]): drawn now, in forward order -- autograd may run the backward
        # nodes of several frames in any order (one summed loss: reverse creation order)
        ctx.sh_frame = ctx.sh_exchange.attach() if ctx.sh_exchange is not None and hasattr(ctx.sh_exchange, "attach") else None
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # no zero tensors for outputs the loss does not touch (backward handles None)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_allmap):
        s = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors
        if grad_out_color is None:
            grad_out_color = torch.zeros((int(s.bg.numel()), s.image_height, s.image_width), dtype=torch.float32, device=means3D.device)
        if grad_allmap is None:
            grad_allmap = torch.zeros((7, s.image_height, s.image_width), dtype=torch.float32, device=means3D.device)
        args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp, s.viewmatrix,
                s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color, grad_allmap, sh, s.sh_degree, s.campos, geomBuffer,
                ctx.num_rendered, binningBuffer, imgBuffer, s.debug)
        exchange = ctx.sh_exchange
        kwargs = {"defer_sh": True} if exchange is not None else {}
        if exchange is not None and hasattr(exchange, "start"):
            # the 12-B colour gradients are final right after the blend backward: their all-gather goes on the wire while K8 runs
            kwargs["after_blend"] = (lambda gc, _ex=exchange, _j=ctx.sh_frame: _ex.start(gc, _j)) if ctx.sh_frame is not None else exchange.start
        if ctx.activations:
            kwargs["activations"] = ctx.activations
        if ctx.tile:
            kwargs["tile"] = ctx.tile
        if ctx.binning_capacity is not None:
            kwargs["binning_capacity"] = ctx.binning_capacity
        if ctx.backward_kernel is not None:
            kwargs["backward_kernel"] = ctx.backward_kernel
        if colors_precomp.numel() and not ctx.needs_input_grad[3]:
            kwargs["want_precomp_color_grad"] = False   # (constant colours -- render_semantic's one-hot channels: K7 skips their sums)
        if s.debug:
            snapshot = _cpu_snapshot(args)
            try:
                out = _C.rasterize_gaussians_backward(*args, **kwargs)
            except Exception:
                torch.save(snapshot, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            out = _C.rasterize_gaussians_backward(*args, **kwargs)
        grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations = out
        if exchange is not None:   # grad_colors_precomp holds the clamp-masked dL/drgb; dL_dsh comes back summed over ranks
            rest = [grad_means3D, grad_opacities, grad_scales, grad_rotations] if exchange.reduce_all else []
            extra = {"frame": ctx.sh_frame} if ctx.sh_frame is not None else {}
            grad_sh = exchange.run(grad_colors_precomp, means3D, s.campos, int(sh.shape[1]), s.sh_degree, also_reduce=rest, **extra)
        none_if_empty = lambda g, ref: g if (g is not None and ref.numel() and g.numel()) else None
        return (grad_means3D, grad_means2D, none_if_empty(grad_sh, sh), none_if_empty(grad_colors_precomp, colors_precomp),
                grad_opacities, none_if_empty(grad_scales, scales), none_if_empty(grad_rotations, rotations),
                none_if_empty(grad_cov3Ds_precomp, cov3Ds_precomp), None, None, None, None, None, None)


class _RasterizeWithClassDistortions(torch.autograd.Function):
    """The operator AND the per-class distortion pass of the same view on ONE plan / binning / K8 (sr_class_forward_shared,
    sr_class_backward_shared): outputs (color, radii, allmap, dist[n_classes,H,W])."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, classes, n_classes, raster_settings, activations, tile, mask):
        s = raster_settings
        ctx.activations, ctx.n_classes = int(activations), int(n_classes)
        ctx.tile = tuple(int(t) for t in tile) if tile else None
        fused = {"activations": ctx.activations} if ctx.activations else {}
        if ctx.tile:
            fused["tile"] = ctx.tile
        if mask is not None:
            fused["mask"] = mask
        empty = torch.empty(0, device=means3D.device)
        num_rendered, color, allmap, radii, geom, binning, img, dist, cstate = _C.rasterize_gaussians(
            s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, empty, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
            s.image_height, s.image_width, sh, s.sh_degree, s.campos, s.prefiltered, s.debug, classes=classes, n_classes=ctx.n_classes, **fused)
        ctx.raster_settings, ctx.num_rendered = s, num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, radii, sh, geom, binning, img, cstate)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        return color, radii, allmap, dist

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_allmap, grad_dist):
        s = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, radii, sh, geom, binning, img, cstate = ctx.saved_tensors
        dev = means3D.device
        z = lambda c: torch.zeros((c, s.image_height, s.image_width), dtype=torch.float32, device=dev)
        grad_color = z(int(s.bg.numel())) if grad_color is None else grad_color
        grad_allmap = z(7) if grad_allmap is None else grad_allmap
        grad_dist = z(ctx.n_classes) if grad_dist is None else grad_dist
        empty = torch.empty(0, device=dev)
        kwargs = {"activations": ctx.activations} if ctx.activations else {}
        if ctx.tile:
            kwargs["tile"] = ctx.tile
        if colors_precomp.numel() and not ctx.needs_input_grad[3]:
            kwargs["want_precomp_color_grad"] = False
        g2d, gcol, gop, g3d, _, gsh, gsc, grot = _C.rasterize_gaussians_backward(
            s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, empty, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
            grad_color, grad_allmap, sh, s.sh_degree, s.campos, geom, ctx.num_rendered, binning, img, s.debug,
            class_state=cstate, dL_ddist=grad_dist, n_classes=ctx.n_classes, **kwargs)
        none_if_empty = lambda g, ref: g if (g is not None and ref.numel() and g.numel()) else None
        return (g3d, g2d, none_if_empty(gsh, sh), none_if_empty(gcol, colors_precomp), gop, gsc, grot, None, None, None, None, None, None)


class _ClassDistortions(torch.autograd.Function):
    """The per-class distortion pass (include/surfel_raster.h, sr_class_forward_render / sr_class_backward)."""

    @staticmethod
    def forward(ctx, means3D, means2D, opacities, scales, rotations, classes, n_classes, raster_settings, activations, mask, tile=None):
        s = raster_settings
        num_rendered, dist, radii, cols, geom, binning, cimg = _C.class_distortions(
            s.bg, means3D, classes, opacities, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
            s.image_height, s.image_width, s.campos, n_classes, s.debug, activations, mask, tile=tile)
        ctx.raster_settings, ctx.num_rendered, ctx.n_classes, ctx.activations, ctx.tile = s, num_rendered, int(n_classes), int(activations), tile
        ctx.save_for_backward(means3D, scales, rotations, radii, cols, geom, binning, cimg)
        ctx.mark_non_differentiable(radii)
        return dist, radii

    @staticmethod
    def backward(ctx, grad_dist, grad_radii):
        s = ctx.raster_settings
        means3D, scales, rotations, radii, cols, geom, binning, cimg = ctx.saved_tensors
        g2d, gop, g3d, gsc, grot = _C.class_distortions_backward(
            s.bg, means3D, radii, cols, scales, rotations, s.scale_modifier, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
            grad_dist, s.campos, ctx.n_classes, geom, ctx.num_rendered, binning, cimg, s.debug, ctx.activations, tile=ctx.tile)
        return g3d, g2d, gop, gsc, grot, None, None, None, None, None, None


def resolve_tile(tile, width, height):
    """`tile` of GaussianRasterizer -> (w, h) or None (= the reference's 16x16).  None reads the process default SURFEL_TILE ("auto", "8x8",
    "16x8", ...; unset = 16x16, the reference's compile-time BLOCK_X x BLOCK_Y).  "auto" picks by frame size: the blend kernels run one
    wave (K7) or two (K6) per tile, and a frame with fewer than ~2 000 tiles of 16x16 leaves most of the GPU's 3 072 wave slots empty --
    the reference's documented runs are such frames (`-r 4`: 480x320 = 600 tiles [REF /root/reference/README.md:195-207]).  Measured fwd+bwd,
    1.5 M Gaussians (tools/time_r4_tiles.py): 480x320 2.19 ms with 16x16, 1.40 with 8x8; 640x480 2.00 / 1.59; 960x640 2.00 / 1.88 with 16x8;
    1280x720 and larger: 16x16 is fastest.  CAUTION: the tile shape is not invisible.  The reference's algorithm truncates a splat at the tiles
    its 3-sigma bounding box touches, while an opaque splat's alpha >= 1/255 footprint reaches 3.3 sigma: smaller tiles drop more of that
    fringe (a scene of large splats: 2.7 % of the pixels differ by more than 1e-4 between 8x8 and 16x16, up to 4e-2).  The tile lists are
    the oracle's lists FOR THAT SHAPE, bit for bit -- not the reference's 16x16 lists.  For small frames WITHOUT that change the default
    path already switches the blend backward to four waves per 16x16 tile (backward_kernel): 480x320 2.16 -> 1.60 ms on the reference's lists."""
    import os
    if tile is None:
        tile = os.environ.get("SURFEL_TILE") or None
    if tile is None:
        return None
    if isinstance(tile, str):
        if tile.lower() == "auto":
            tiles16 = ((int(width) + 15) // 16) * ((int(height) + 15) // 16)
            return (8, 8) if tiles16 < 1600 else ((16, 8) if tiles16 < 3000 else None)
        w, h = tile.lower().split("x")
        tile = (int(w), int(h))
    tile = (int(tile[0]), int(tile[1]))
    return None if tile == (16, 16) else tile


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings, fused_activations: bool = False, tile=None, quadrant_cull: bool = True,
                 blend_counters=None, ballot_ranking: bool = False, row_mapped=None, binning_capacity=None, backward_kernel=None):
        """`binning_capacity=N` (extension, round 6: SR_FLAG_BINNING_CAPACITY): the operator WITHOUT the host read-back of the frame's
        duplicate count between the emission scan and the binning [the reference waits there too: SURVEY.md 7 step 4] -- the binning buffer
        is sized for N duplicates (e.g. 1.25 x the largest count seen: `last_status`), nothing in forward or backward waits for the GPU, and
        the whole step can be captured into a HIP graph.  Whether the frame fitted is decided on the device: after a call `self.last_status`
        is a device int32 [D, visible, overflow]; overflow = 1 means nothing was rendered (background image, zero gradients) and the frame
        must be rendered again with N >= D.  A frame that fits is bit-identical to the default mode.
        `fused_activations=True` (extension): `opacities`, `scales`, `rotations` are the RAW parameters
        (`_opacity`, `_scaling`, `_rotation` of the reference's GaussianModel); sigmoid / exp / normalize run inside the
        preprocess kernel and their adjoints inside its backward, so the returned gradients are w.r.t. the raw values.
        `tile=(w, h)` | "auto" | "8x8" ...: binning tile shape (resolve_tile), default the reference's compile-time 16x16 (BASELINE config 5 sweeps 8x8, 16x8,
        16x16, 32x8, 32x16); images depend on it at the truncation fringe of the 3-sigma bounding box (resolve_tile).
        `quadrant_cull=False` / `blend_counters` (int64[16] device tensor): this call's SrFrame.flags / SrFrame.blend_counters --
        test and profiling switches with identical results (include/surfel_raster.h); `ballot_ranking=True`: SR_FLAG_BALLOT_RANKING,
        the binning's fallback ranking (identical lists); `row_mapped=True` / `False`: force the row-mapped / the quadrant-mapped
        forward blend (SR_FLAG_ROW_MAPPED_FORWARD / SR_FLAG_QUADRANT_MAPPED_FORWARD; bit-identical results; None = picked per frame on the device)."""
        super().__init__()
        self.raster_settings = raster_settings
        self.activations = 7 if fused_activations else 0
        self.tile = resolve_tile(tile, raster_settings.image_width, raster_settings.image_height)
        self.probe = {}
        if not quadrant_cull:
            self.probe["quadrant_cull"] = False
        if blend_counters is not None:
            self.probe["blend_counters"] = blend_counters
        if ballot_ranking:
            self.probe["ballot_ranking"] = True
        if row_mapped is not None:
            self.probe["row_mapped"] = bool(row_mapped)
        if binning_capacity is not None:
            self.probe["binning_capacity"] = int(binning_capacity)
        if backward_kernel is not None:   # "one_wave" / "coop": force one of the two blend-backward kernels (default: picked by the tile count)
            self.probe["backward_kernel"] = backward_kernel
        self.last_status = None   # binning_capacity mode: device int32 [D, visible, overflow] of the last forward (a view into its state)

    def markVisible(self, positions):
        with torch.no_grad():
            s = self.raster_settings
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def class_distortions(self, means3D, means2D, opacities, scales, rotations, classes, n_classes, mask=None):
        """Extension: the distortion maps of the class-filtered renders of ONE view in one pass.
        `classes` [P] integer class per Gaussian (negative / >= n_classes: in no class).  Returns (dist[n_classes,H,W], radii[P]);
        dist[k] == allmap[6] of this operator called on the Gaussians of class k only, differentiable w.r.t. means3D, means2D
        (densification proxy), opacities, scales, rotations.  Every tile shape of the sweep; n_classes <= 6."""
        return _ClassDistortions.apply(means3D, means2D, opacities, scales, rotations, classes, int(n_classes), self.raster_settings,
                                       self.activations, mask, tuple(int(t) for t in self.tile) if self.tile else None)

    def forward_with_class_distortions(self, means3D, means2D, opacities, scales, rotations, classes, n_classes, shs=None, colors_precomp=None,
                                       extra_colors=None, mask=None):
        """Extension (SURVEY 8f N1 in full): this operator's (color, radii, allmap) AND the per-class distortion maps of the same view --
        `class_distortions` -- from ONE preprocess, ONE binning and ONE per-Gaussian backward.  Same colour options as `forward` (shs |
        colors_precomp [P,3] / [P,6] | shs + extra_colors [P,6]); returns (color, radii, allmap, dist[n_classes,H,W]).  The maps are
        bit-identical to the two separate calls; the gradients agree to float summation order."""
        if (shs is None) == (colors_precomp is None) and extra_colors is None:
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if extra_colors is not None:
            if shs is None or colors_precomp is not None or extra_colors.ndim != 2 or extra_colors.shape[1] != 6:
                raise Exception("extra_colors needs SHs as the colour source and must have dimensions (num_points, 6)")
            colors_precomp = extra_colors
        empty = torch.Tensor([]).to(means3D.device)
        return _RasterizeWithClassDistortions.apply(means3D, means2D, empty if shs is None else shs, empty if colors_precomp is None else colors_precomp,
                                                    opacities, scales, rotations, classes, int(n_classes), self.raster_settings, self.activations,
                                                    tuple(int(t) for t in self.tile) if self.tile else None, mask)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, mask=None, extra_colors=None):
        """`mask` (extension): [P] bool; False = leave the Gaussian out, with the same images as boolean-indexing
        every input first (what the reference's render_with_mask / semantic filters do) but without the copies: `radii` and
        all gradients stay full-size, zero where masked out.
        `extra_colors` (extension): [P,6] precomputed channels blended IN ADDITION to the SH colour in the same
        pass (render + render_semantic as one rasterization); needs `shs`, a 9-entry `bg`, returns color[9,H,W]."""
        s = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if extra_colors is not None:
            if shs is None or colors_precomp is not None or extra_colors.ndim != 2 or extra_colors.shape[1] != 6:
                raise Exception("extra_colors needs SHs as the colour source and must have dimensions (num_points, 6)")
            colors_precomp = extra_colors
        empty = torch.Tensor([]).to(means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        if "binning_capacity" in self.probe:
            holder = {}
            out = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, s,
                                      self.activations, self.tile, mask, dict(self.probe, _status=holder))
            self.last_status = holder.get("status")
            return out
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, s,
                                   self.activations, self.tile, mask, self.probe or None)
