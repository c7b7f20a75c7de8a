"""SE(3) pose utilities with the reference's API (utils/pose_utils.py:5-93): tau = [rho, theta],
left-multiplicative update T <- Exp(tau) T, small-angle branch below 1e-5."""
import numpy as np
import torch


def rt2mat(R, T):
    out = np.eye(4)
    out[:3, :3], out[:3, 3] = R, T
    return out


def skew_sym_mat(x):
    z = torch.zeros((), device=x.device, dtype=x.dtype)
    return torch.stack([torch.stack([z, -x[2], x[1]]), torch.stack([x[2], z, -x[0]]), torch.stack([-x[1], x[0], z])])


def _series(theta):
    """Coefficients (a, b, c) of Exp: R = I + a W + b W^2, V = I + b W + c W^2."""
    angle = torch.norm(theta)
    if angle < 1e-5:
        one = torch.ones((), device=theta.device, dtype=theta.dtype)
        return one, 0.5 * one, one / 6.0
    return torch.sin(angle) / angle, (1.0 - torch.cos(angle)) / angle**2, (angle - torch.sin(angle)) / angle**3


def SO3_exp(theta):
    W = skew_sym_mat(theta)
    a, b, _ = _series(theta)
    return torch.eye(3, device=theta.device, dtype=theta.dtype) + a * W + b * (W @ W)


def V(theta):
    W = skew_sym_mat(theta)
    _, b, c = _series(theta)
    return torch.eye(3, device=theta.device, dtype=theta.dtype) + b * W + c * (W @ W)


def SE3_exp(tau):
    rho, theta = tau[:3], tau[3:]
    T = torch.eye(4, device=tau.device, dtype=tau.dtype)
    T[:3, :3] = SO3_exp(theta)
    T[:3, 3] = V(theta) @ rho
    return T


def update_pose(camera, converged_threshold=1e-4):
    """W2C <- Exp([rho, theta]) W2C with the camera's accumulated deltas, which are then reset; returns |tau| < threshold."""
    tau = torch.cat((camera.cam_trans_delta, camera.cam_rot_delta))
    pose = torch.eye(4, device=tau.device)
    pose[:3, :3], pose[:3, 3] = camera.R, camera.T
    pose = SE3_exp(tau) @ pose
    small = tau.norm() < converged_threshold
    camera.update_RT(pose[:3, :3], pose[:3, 3])
    for delta in (camera.cam_rot_delta, camera.cam_trans_delta):
        delta.data.zero_()
    return small
