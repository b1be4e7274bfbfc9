#!/usr/bin/env python
"""Two (or more) frame-parallel ranks: gradients after `factored_sh_exchange` + `allreduce_gradients` must equal the sum of
the gradients of all ranks' frames computed locally.  Launch with torchrun; SURFEL_DIST_BACKEND=gloo lets the ranks share
one GPU (functional check), the default backend on a multi-GPU node is nccl (= RCCL).
SURFEL_CHECK_RIG=yawed (default; the benchmark's camera batch: every camera at the origin) | posed (synthetic.posed_rig: every camera
with its own centre and orientation -- the per-view directions of the expansion then differ from rank to rank)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from streetunveiler_amd.parallel import allreduce_gradients, factored_sh_exchange, init_distributed, reduce_densification_stats

rank, world, local_rank = init_distributed()
dev = torch.device("cuda", local_rank % torch.cuda.device_count())
torch.cuda.set_device(dev)
from diff_surfel_rasterization import GaussianRasterizationSettings, GaussianRasterizer
from streetunveiler_amd.synthetic import posed_rig, synthetic_camera, synthetic_gaussians, synthetic_upstream_grads

P, W, H, deg = 20000, 320, 192, 3
if os.environ.get("SURFEL_CHECK_RIG", "yawed") == "posed":
    rig, g = posed_rig(P, W, H, world, seed=3, scale_lo=2e-3, scale_hi=2e-2, spread=12.0)
    assert len({tuple(c.camera_center.tolist()) for c in rig}) == world
    camera = lambda index: rig[index]
else:
    g = synthetic_gaussians(P, W, H, seed=3)
    camera = lambda index: synthetic_camera(W, H, index=index, n_cams=world)
dc, da = [t.to(dev) for t in synthetic_upstream_grads(W, H, seed=4)]
names = ["means3D", "shs", "opacities", "scales", "rotations"]


def frame(index, exchange, reduce_all=False):
    cam = camera(index)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, False)
    t = {k: g[k].to(dev).requires_grad_() for k in names}
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    # shs reaches the operator through a cat, like GaussianModel.get_features [REF scene/gaussian_model.py:117-121]
    dc_part, rest = t["shs"][:, :1].detach().requires_grad_(), t["shs"][:, 1:].detach().requires_grad_()
    shs = torch.cat([dc_part, rest], dim=1)
    call = lambda: GaussianRasterizer(s)(means3D=t["means3D"], means2D=m2d, shs=shs, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    if exchange:
        with factored_sh_exchange(reduce_all=reduce_all) as ex:   # the FORWARD attaches the exchange to its autograd node ...
            color, radii, allmap = call()
        torch.autograd.backward([color, allmap], [dc, da])        # ... the backward may run outside the block, on autograd's thread
        assert ex is not None and ex.calls == 1 and ex.bytes_sent == P * 12 + (P * 40 if reduce_all else 0) and ex.early_starts == 1
        if not reduce_all:
            allreduce_gradients([t[k].grad for k in names if k != "shs"])
    else:
        color, radii, allmap = call()
        torch.autograd.backward([color, allmap], [dc, da])
    out = {k: t[k].grad for k in names if k != "shs"}
    out["shs"] = torch.cat([dc_part.grad, rest.grad], dim=1)
    return out


expect = None
for k in range(world):
    f = frame(k, False)
    expect = f if expect is None else {n: expect[n] + f[n] for n in f}
torch.cuda.synchronize()
for reduce_all in (False, True):
    got = frame(rank, True, reduce_all)
    torch.cuda.synchronize()
    for n in names:
        scale = float(expect[n].abs().max())
        err = float((got[n] - expect[n]).abs().max())
        assert scale > 0 and err <= 2e-5 * scale, f"rank {rank} reduce_all={reduce_all} {n}: max err {err:.3e} vs scale {scale:.3e}"
# the densification statistics of the ranks' views (one packed exchange) against the per-view values computed locally
def view_stats(index):
    cam = camera(index)
    s = GaussianRasterizationSettings(H, W, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2), torch.zeros(3, device=dev), 1.0,
                                      cam.world_view_transform.to(dev), cam.full_proj_transform.to(dev), deg, cam.camera_center.to(dev), False, False)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True)
    color, radii, allmap = GaussianRasterizer(s)(means3D=g["means3D"].to(dev), means2D=m2d, shs=g["shs"].to(dev), opacities=g["opacities"].to(dev),
                                                 scales=g["scales"].to(dev), rotations=g["rotations"].to(dev))
    torch.autograd.backward([color, allmap], [dc, da])
    return m2d.grad, radii
accum, denom, maxr = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev)
reduce_densification_stats(*view_stats(rank), accum, denom, maxr)
e_acc, e_den, e_max = torch.zeros(P, 1, device=dev), torch.zeros(P, 1, device=dev), torch.zeros(P, device=dev)
for k in range(world):
    vg, rad = view_stats(k)
    vis = rad > 0
    e_acc += torch.where(vis[:, None], vg.norm(dim=-1, keepdim=True), torch.zeros((), device=dev)); e_den += vis[:, None].float()
    e_max = torch.maximum(e_max, torch.where(vis, rad.float(), torch.zeros((), device=dev)))
torch.testing.assert_close(accum, e_acc, rtol=1e-6, atol=1e-6 * float(e_acc.abs().max()))
assert torch.equal(denom, e_den) and torch.equal(maxr, e_max) and float(e_den.max()) >= 1
dist.barrier()
if rank == 0:
    print(f"factored exchange OK (world {world}, backend {dist.get_backend()})")
dist.destroy_process_group()
