"""SH evaluation in torch (the python fallback of the render operator, `pipe.convert_SHs_python`).

Real spherical-harmonics basis up to degree 4 in the sign / ordering convention the reference uses
[REF /root/reference/utils/sh_utils.py:26-112] (the native K1 path evaluates the same basis up to degree 3), written as
basis(dir) . coefficients; pinned against the reference function by tests/golden/sh_golden.npz (degrees 0..3) and sh4_golden.npz
(degree 4, which only the reference's python function reaches: the model stops at 3 [REF arguments/__init__.py:44])."""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)
C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
      0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """dirs [..., 3] (unit) -> the (deg + 1)^2 basis values [..., (deg + 1)^2], band by band."""
    assert 0 <= deg <= 4
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    b = [torch.full_like(x, C0)]
    if deg >= 1:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg >= 3:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy), C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
              C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    if deg >= 4:
        b += [C4[0] * xy * (xx - yy), C4[1] * yz * (3 * xx - yy), C4[2] * xy * (7 * zz - 1), C4[3] * yz * (7 * zz - 3),
              C4[4] * (zz * (35 * zz - 30) + 3), C4[5] * xz * (7 * zz - 3), C4[6] * (xx - yy) * (7 * zz - 1),
              C4[7] * xz * (xx - 3 * yy), C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(b, dim=-1)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [..., C, (max_deg + 1)^2], dirs [..., 3] (unit) -> [..., C]; deg in 0..4."""
    n = (deg + 1) ** 2
    assert sh.shape[-1] >= n
    return (sh[..., :n] * sh_basis(deg, dirs).unsqueeze(-2)).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
