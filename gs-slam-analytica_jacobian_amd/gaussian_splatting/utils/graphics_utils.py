"""Camera matrix helpers with the reference's names and conventions
(gaussian_splatting/utils/graphics_utils.py: getWorld2View2 :33-46, getProjectionMatrix :49-69,
getProjectionMatrix2 :72-93, fov2focal / focal2fov :96-101)."""
import math

import torch


def getWorld2View2(R, t, translate=None, scale=1.0):
    """4x4 W2C from the rotation block R and translation t; the camera centre can be shifted /
    scaled through `translate` / `scale`.  Uses true matrix inverses (R need not be orthonormal)."""
    Rt = torch.zeros((4, 4), device=R.device, dtype=R.dtype)
    Rt[:3, :3] = R
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    if translate is None and scale == 1.0:
        return torch.linalg.inv(torch.linalg.inv(Rt))  # same round trip as the reference (fp parity)
    C2W = torch.linalg.inv(Rt)
    tr = torch.zeros(3, device=R.device, dtype=R.dtype) if translate is None else translate.to(R.device)
    C2W[:3, 3] = (C2W[:3, 3] + tr) * scale
    return torch.linalg.inv(C2W)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    ty, tx = math.tan(fovY / 2), math.tan(fovX / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1.0 / tx
    P[1, 1] = 1.0 / ty
    P[3, 2] = 1.0
    P[2, 2] = -(zfar + znear) / (zfar - znear)
    P[2, 3] = -2 * (zfar * znear) / (zfar - znear)
    return P


def getProjectionMatrix2(znear, zfar, cx, cy, fx, fy, W, H):
    """Pinhole intrinsics -> clip matrix: P00 = 2fx/W, P11 = 2fy/H, P02 = (2cx-W)/W, P12 = (2cy-H)/H."""
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * fx / W
    P[1, 1] = 2.0 * fy / H
    P[0, 2] = (2.0 * cx - W) / W
    P[1, 2] = (2.0 * cy - H) / H
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))
