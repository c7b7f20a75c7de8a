"""Parameter-space helpers with the reference's names
(gaussian_splatting/utils/general_utils.py: inverse_sigmoid :20-21, strip_lowerdiag /
strip_symmetric :97-110, build_rotation :113-136, build_scaling_rotation :139-148);
device-agnostic (the reference hard-codes device="cuda")."""
import torch


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def strip_lowerdiag(L):
    """(N,3,3) symmetric -> (N,6) as (xx, xy, xz, yy, yz, zz)."""
    return torch.stack([L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]], dim=1)


def strip_symmetric(sym):
    return strip_lowerdiag(sym)


def build_rotation(r):
    """Rotation matrices of (r, x, y, z) quaternions, normalised first."""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(dim=1)
    rows = [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]
    return torch.stack(rows, dim=1).view(-1, 3, 3)


def build_scaling_rotation(s, r):
    """L = R diag(s), so that Sigma = L L^T."""
    return build_rotation(r) * s.unsqueeze(1)
