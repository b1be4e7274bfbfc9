"""Fused allmap post-processing (HIP): one forward kernel and two backward kernels replace the
~25 full-image torch kernels (and the 25 MB per-call host->device upload of the pixel grid) of
[REF /root/reference/gaussian_renderer/__init__.py:152-177; /root/reference/utils/point_utils.py:9-37]."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class _PostprocessAllmap(torch.autograd.Function):
    @staticmethod
    def forward(ctx, allmap, viewmatrix, W, H, fovx, fovy, depth_ratio):
        if not allmap.is_cuda:
            raise L.SurfelRasterError("postprocess_allmap needs a CUDA (ROCm) tensor; there is no CPU path")
        lib = L.load()
        allmap = allmap.contiguous().float()
        viewmatrix = viewmatrix.contiguous().float()
        dev = allmap.device
        e = lambda c: torch.empty((c, H, W), dtype=torch.float32, device=dev)
        rend_normal, surf_depth, surf_normal, surf_point = e(3), e(1), e(3), e(3)
        with torch.cuda.device(dev):
            L.check(lib.sr_postprocess_forward(W, H, fovx, fovy, depth_ratio, _ptr(viewmatrix), _ptr(allmap), _ptr(rend_normal),
                                               _ptr(surf_depth), _ptr(surf_normal), _ptr(surf_point),
                                               C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "sr_postprocess_forward")
        ctx.save_for_backward(allmap, viewmatrix)
        ctx.cfg = (W, H, fovx, fovy, depth_ratio)
        return rend_normal, surf_depth, surf_normal, surf_point

    @staticmethod
    def backward(ctx, g_rend_normal, g_surf_depth, g_surf_normal, g_surf_point):
        allmap, viewmatrix = ctx.saved_tensors
        W, H, fovx, fovy, depth_ratio = ctx.cfg
        lib = L.load()
        dev = allmap.device
        c = lambda g: None if g is None else g.contiguous().float()
        g_rend_normal, g_surf_depth, g_surf_normal, g_surf_point = c(g_rend_normal), c(g_surf_depth), c(g_surf_normal), c(g_surf_point)
        scratch = torch.empty((6, H, W), dtype=torch.float32, device=dev)
        g_allmap = torch.empty((7, H, W), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.sr_postprocess_backward(W, H, fovx, fovy, depth_ratio, _ptr(viewmatrix), _ptr(allmap), _ptr(g_rend_normal),
                                                _ptr(g_surf_depth), _ptr(g_surf_normal), _ptr(g_surf_point), _ptr(scratch),
                                                _ptr(g_allmap), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                    "sr_postprocess_backward")
        return g_allmap, None, None, None, None, None, None


def postprocess_allmap_fused(viewpoint_camera, depth_ratio: float, allmap: torch.Tensor):
    """-> (rend_normal[3,H,W], surf_depth[1,H,W], surf_normal[3,H,W], surf_point[3,H,W]); differentiable w.r.t. allmap."""
    return _PostprocessAllmap.apply(allmap, viewpoint_camera.world_view_transform, int(viewpoint_camera.image_width),
                                    int(viewpoint_camera.image_height), float(viewpoint_camera.FoVx), float(viewpoint_camera.FoVy),
                                    float(depth_ratio))
