"""Real spherical-harmonics colour evaluation on torch tensors (degree <= 3).

Interface of the reference's gaussian_splatting/utils/sh_utils.py (eval_sh :55-118,
RGB2SH :121-122, SH2RGB :125-126), written as a basis-matrix contraction.  The basis
polynomials / constants are those of the rasteriser (cuda_rasterizer/auxiliary.h:22-39).
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_basis(deg, dirs):
    """(..., (deg+1)^2) basis values at unit directions dirs (..., 3)."""
    assert 0 <= deg <= 3
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, C0)]
    if deg > 0:
        cols += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
        if deg > 2:
            cols += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                     C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
                     C3[6] * x * (xx - 3 * yy)]
    return torch.stack(cols, dim=-1)


def eval_sh(deg, sh, dirs):
    """sh: (..., C, (max_deg+1)^2), dirs: (..., 3) unit vectors -> (..., C)."""
    n = (deg + 1) ** 2
    assert sh.shape[-1] >= n
    return (sh[..., :n] * sh_basis(deg, dirs).unsqueeze(-2)).sum(dim=-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
