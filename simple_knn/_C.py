"""K-nearest-neighbour distance ops of the reference's simple-knn fork, on gfx950 (include/surfel_raster.h sr_knn_mean_dist2).

The fork (submodules/simple-knn, un-vendored, no pinned SHA) is not available; `dist3knn` / `dist10knn` follow the published
simple-knn definition they descend from -- mean of the 3 / 10 smallest SQUARED distances to the other points, as consumed at
[REF /root/reference/scene/gaussian_model.py:151] (`clamp_min(dist3knn(points), 1e-7)` -> log(sqrt()) initial scales).
`meanDistFromReferencePcd(query, reference, flag)` [REF inpainting_pipeline/2_condition_preparation/2_generate_inpainted_mask.py:71-73]
is restated as the same quantity against another cloud (K = 3); what the fork's third argument does is not observable from the
reference (it is always False there): here True returns the square root of the mean.  There is no CPU path.
"""
import ctypes as C

import torch

from streetunveiler_amd import _lib as L


def _knn(query, reference, K, take_sqrt=False):
    lib = L.load()
    ref = reference
    if not ref.is_cuda:
        raise L.SurfelRasterError("points must be a CUDA (ROCm) tensor; the kNN ops have no CPU path")
    if ref.ndim != 2 or ref.shape[1] != 3 or (query is not None and (query.ndim != 2 or query.shape[1] != 3)):
        raise L.SurfelRasterError("points must have dimensions (num_points, 3)")
    ref = ref.detach().float().contiguous()
    q = None if query is None else query.detach().float().contiguous().to(ref.device)
    n_out = ref.shape[0] if q is None else q.shape[0]
    out = torch.empty((n_out,), dtype=torch.float32, device=ref.device)
    with torch.cuda.device(ref.device):
        ws = torch.empty((lib.sr_knn_workspace_bytes(0 if q is None else q.shape[0], ref.shape[0]),), dtype=torch.uint8, device=ref.device)
        L.check(lib.sr_knn_mean_dist2(0 if q is None else q.shape[0], None if q is None else C.c_void_p(q.data_ptr()), ref.shape[0],
                                      C.c_void_p(ref.data_ptr()), int(K), int(bool(take_sqrt)), C.c_void_p(out.data_ptr()),
                                      C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(torch.cuda.current_stream(ref.device).cuda_stream)),
                "sr_knn_mean_dist2")
    return out


def dist3knn(points: torch.Tensor) -> torch.Tensor:
    """[P] mean squared distance to the 3 nearest other points."""
    return _knn(None, points, 3)


distCUDA2 = dist3knn   # the upstream name of the same op


def dist10knn(points: torch.Tensor) -> torch.Tensor:
    """[P] mean squared distance to the 10 nearest other points."""
    return _knn(None, points, 10)


def meanDistFromReferencePcd(query: torch.Tensor, reference: torch.Tensor, take_sqrt: bool = False) -> torch.Tensor:
    """[Q] mean squared distance from every query point to its 3 nearest points of `reference`."""
    return _knn(query, reference, 3, take_sqrt)
