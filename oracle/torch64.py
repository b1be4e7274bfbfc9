"""float64 torch-autograd restatement of the surfel operator (forward only; gradients by autograd).

TEST INFRASTRUCTURE ONLY (same rules as surfel_oracle.c).  Purpose: an
independent check of the analytic backward in surfel_oracle.c (K7 + K8): the
forward of SURVEY.md Appendix A.2 / A.4 is written here with differentiable
float64 tensor ops, the discrete structure (tile lists, culling) is taken from
the C oracle, and torch.autograd supplies d/d(means3D, scales, rotations,
opacities, shs | colors_precomp).  Two upstream conventions that are NOT the
true derivative are reproduced explicitly:
  * alpha = min(0.99, opacity*G) is treated as straight-through (A.5: "clamp ignored");
  * the returned means2D gradient is the densification proxy (A.6), computed from
    the blend-only dL/dtransMat.
SH polynomial: /root/reference/utils/sh_utils.py:57-112.
"""
from __future__ import annotations

import numpy as np
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]
NEAR, FAR = 0.2, 100.0
TILE = 16


def eval_sh64(deg, sh, dirs):
    """sh [P,M,3] coefficient-major, dirs [P,3] unit."""
    res = C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
                   + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
                       + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
                       + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def quat_to_R64(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def forward_backward64(fwd, dL_dcolor, dL_dallmap):
    """fwd: dict returned by surfel_oracle.rasterize_forward.  Returns (outputs, grads) as numpy float64."""
    i = fwd["_inputs"]
    dd = torch.float64
    t = lambda a: None if a is None else torch.tensor(np.asarray(a), dtype=dd)
    W, H, deg = i["W"], i["H"], i["deg"]
    means3D = t(i["means3D"]).requires_grad_()
    opac = t(i["opacities"]).requires_grad_()
    use_T_pre = i["transMat_precomp"] is not None
    if use_T_pre:
        Tpre = t(i["transMat_precomp"]).requires_grad_()
    else:
        scales = t(i["scales"]).requires_grad_()
        rots = t(i["rotations"]).requires_grad_()
    shs = t(i["shs"]); cpre = t(i["colors_precomp"])
    if shs is not None: shs.requires_grad_()
    if cpre is not None: cpre.requires_grad_()
    view = t(i["view"]); proj = t(i["proj"]); cam = t(i["cam"]); bg = t(i["bg"])
    m = i["scale_modifier"]
    P = means3D.shape[0]
    vis = torch.tensor(fwd["radii"] > 0)

    V3 = view.view(4, 4)[:3, :3]  # row-major flat v: p_view_j = sum_i v[4i+j] p_i  -> p @ V3 + v[12:15]
    A = proj.view(4, 4).t()       # A[r][c] = f[4c+r]
    Nt = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, 0, 1]], dtype=dd)
    B = Nt @ A                    # 3x4
    p_view = means3D @ V3 + view[12:15]
    if use_T_pre:
        Tm = Tpre.view(P, 3, 3)
        normal = torch.tensor([0.0, 0.0, 1.0], dtype=dd).expand(P, 3).clone().requires_grad_()
    else:
        R = quat_to_R64(rots)
        L0 = R[:, :, 0] * (m * scales[:, 0:1]); L1 = R[:, :, 1] * (m * scales[:, 1:2])
        Mh = torch.stack([torch.cat([L0, torch.zeros(P, 1, dtype=dd)], 1), torch.cat([L1, torch.zeros(P, 1, dtype=dd)], 1),
                          torch.cat([means3D, torch.ones(P, 1, dtype=dd)], 1)], dim=2)  # [P,4,3]
        Tm = torch.einsum("rk,pkc->prc", B, Mh)  # rows Tu,Tv,Tw
        normal = R[:, :, 2] @ V3
    cosv = -(p_view * normal).sum(1, keepdim=True)
    normal = normal * torch.where(cosv > 0, 1.0, -1.0).detach()
    T_r = Tm * 1.0; T_r.retain_grad()   # blend path
    T_c = Tm * 1.0                      # AABB-centre path
    Tu, Tv, Tw = T_c[:, 0], T_c[:, 1], T_c[:, 2]
    tv = torch.tensor([9.0, 9.0, -1.0], dtype=dd)
    dist = (Tw * Tw * tv).sum(1, keepdim=True)
    dist = torch.where(vis[:, None], dist, torch.ones_like(dist))
    f = tv / dist
    centre = torch.stack([(f * Tu * Tw).sum(1), (f * Tv * Tw).sum(1)], 1)
    centre.retain_grad()
    if shs is not None:
        d = means3D - cam
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh64(deg, shs, d) + 0.5, 0.0)
    else:
        rgb = cpre
    rgb_r = rgb * 1.0; rgb_r.retain_grad()
    nrm_r = normal * 1.0; nrm_r.retain_grad()

    gx = (W + TILE - 1) // TILE; gy = (H + TILE - 1) // TILE
    color = torch.zeros(3, H, W, dtype=dd); allmap = torch.zeros(7, H, W, dtype=dd)
    ranges = fwd["ranges"]; plist = torch.tensor(fwd["point_list"].astype(np.int64))
    color_parts = []
    for tile in range(gx * gy):
        r0, r1 = int(ranges[tile, 0]), int(ranges[tile, 1])
        x0, y0 = (tile % gx) * TILE, (tile // gx) * TILE
        xs = torch.arange(x0, min(x0 + TILE, W), dtype=dd); ys = torch.arange(y0, min(y0 + TILE, H), dtype=dd)
        py, px = torch.meshgrid(ys, xs, indexing="ij")
        px = px.reshape(-1); py = py.reshape(-1); Np = px.numel()
        ids = plist[r0:r1]; n = ids.numel()
        if n == 0:
            col = bg[:, None].expand(3, Np); am = torch.zeros(7, Np, dtype=dd)
        else:
            tu, tvv, tw = T_r[ids, 0], T_r[ids, 1], T_r[ids, 2]       # [n,3]
            k = px[:, None, None] * tw[None] - tu[None]; l = py[:, None, None] * tw[None] - tvv[None]
            p = torch.cross(k, l, dim=-1)
            pz_ok = p[..., 2] != 0
            pz = torch.where(pz_ok, p[..., 2], torch.ones_like(p[..., 2]))
            sx, sy = p[..., 0] / pz, p[..., 1] / pz
            rho3 = sx * sx + sy * sy
            dx = centre[ids, 0][None] - px[:, None]; dy = centre[ids, 1][None] - py[:, None]
            rho2 = 2.0 * (dx * dx + dy * dy)
            use3 = rho3 <= rho2
            rho = torch.where(use3, rho3, rho2)
            depth = torch.where(use3, sx * tw[None, :, 0] + sy * tw[None, :, 1] + tw[None, :, 2], tw[None, :, 2].expand(Np, n))
            G = torch.exp(-0.5 * rho)
            a_raw = opac[ids][None] * G
            a_cl = a_raw + (torch.clamp(a_raw, max=0.99) - a_raw).detach()   # straight-through clamp
            valid = pz_ok & (depth >= NEAR) & (-0.5 * rho <= 0) & (a_cl >= 1.0 / 255.0)
            a = torch.where(valid, a_cl, torch.zeros_like(a_cl))
            one_m = 1 - a
            Tincl = torch.cumprod(one_m, dim=1)
            Tex = torch.cat([torch.ones(Np, 1, dtype=dd), Tincl[:, :-1]], 1)
            stop = valid & (Tincl < 1e-4)
            stopped = torch.cumsum(stop.to(torch.int64), 1) > 0          # entries at/after the terminating one
            live = valid & ~stopped
            a = torch.where(live, a, torch.zeros_like(a))
            one_m = 1 - a
            Tincl = torch.cumprod(one_m, dim=1)
            Tex = torch.cat([torch.ones(Np, 1, dtype=dd), Tincl[:, :-1]], 1)
            Tfin = Tincl[:, -1]
            w = a * Tex
            dsafe = torch.where(live, depth, torch.ones_like(depth))
            mm = FAR / (FAR - NEAR) * (1 - NEAR / dsafe)
            mw = torch.where(live, mm * w, torch.zeros_like(w)); m2w = torch.where(live, mm * mm * w, torch.zeros_like(w))
            M1ex = torch.cumsum(mw, 1) - mw; M2ex = torch.cumsum(m2w, 1) - m2w
            Aex = 1 - Tex
            dist_t = torch.where(live, (mm * mm * Aex + M2ex - 2 * mm * M1ex) * w, torch.zeros_like(w)).sum(1)
            Dsum = torch.where(live, dsafe * w, torch.zeros_like(w)).sum(1)
            Nsum = (w[..., None] * nrm_r[ids][None]).sum(1)   # [Np,3]
            Csum = (w[..., None] * rgb_r[ids][None]).sum(1)
            med_mask = live & (Tex > 0.5)
            idx = torch.arange(n)[None].expand(Np, n)
            last = torch.where(med_mask, idx, torch.full_like(idx, -1)).max(1).values
            med = torch.where(last >= 0, torch.gather(dsafe, 1, last.clamp(min=0)[:, None])[:, 0], torch.zeros(Np, dtype=dd))
            col = Csum.t() + Tfin[None] * bg[:, None]
            am = torch.stack([Dsum, 1 - Tfin, Nsum[:, 0], Nsum[:, 1], Nsum[:, 2], med, dist_t], 0)
        hh, ww = ys.numel(), xs.numel()
        color_parts.append((y0, x0, hh, ww, col.reshape(3, hh, ww), am.reshape(7, hh, ww)))
    rows_c, rows_a = [], []
    for ty in range(gy):
        rc = [color_parts[ty * gx + tx][4] for tx in range(gx)]
        ra = [color_parts[ty * gx + tx][5] for tx in range(gx)]
        rows_c.append(torch.cat(rc, 2)); rows_a.append(torch.cat(ra, 2))
    color = torch.cat(rows_c, 1); allmap = torch.cat(rows_a, 1)
    loss = (color * t(dL_dcolor)).sum() + (allmap * t(dL_dallmap)).sum()
    loss.backward()
    z = lambda g, like: (torch.zeros_like(like) if g is None else g).numpy()
    grads = dict(dL_dmeans3D=z(means3D.grad, means3D), dL_dopacity=z(opac.grad, opac).reshape(P, 1),
                 dL_dtransMat_render=z(T_r.grad, T_r).reshape(P, 9), dL_dmean2D_raw=z(centre.grad, centre),
                 dL_dcolors=z(rgb_r.grad, rgb_r), dL_dnormal3D=z(nrm_r.grad, nrm_r))
    if use_T_pre:
        grads["dL_dtransMat"] = z(Tpre.grad, Tpre).reshape(P, 9)
    else:
        grads["dL_dscales"] = z(scales.grad, scales); grads["dL_drotations"] = z(rots.grad, rots)
    if shs is not None: grads["dL_dsh"] = z(shs.grad, shs)
    gT = grads["dL_dtransMat_render"]; Tn = Tm.detach().numpy().reshape(P, 9)
    proxy = np.zeros((P, 3)); proxy[:, 0] = gT[:, 2] * Tn[:, 8] * 0.5 * W; proxy[:, 1] = gT[:, 5] * Tn[:, 8] * 0.5 * H
    grads["dL_dmeans2D"] = proxy * (fwd["radii"] > 0)[:, None]
    outs = dict(color=color.detach().numpy(), allmap=allmap.detach().numpy(), transMat=Tn,
                means2D=centre.detach().numpy(), rgb=rgb.detach().numpy(), normal=normal.detach().numpy())
    return outs, grads
