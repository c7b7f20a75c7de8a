"""SE(3) pose utilities with the reference's API (utils/pose_utils.py:5-93): tau = [rho, theta],
left-multiplicative update T <- Exp(tau) T, small-angle branch below 1e-5."""
import numpy as np
import torch


def rt2mat(R, T):
    mat = np.eye(4)
    mat[0:3, 0:3] = R
    mat[0:3, 3] = T
    return mat


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
    """Apply the accumulated pose delta to the camera, zero the deltas, report convergence."""
    tau = torch.cat([camera.cam_trans_delta, camera.cam_rot_delta], axis=0)
    T_w2c = torch.eye(4, device=tau.device)
    T_w2c[0:3, 0:3] = camera.R
    T_w2c[0:3, 3] = camera.T
    new_w2c = SE3_exp(tau) @ T_w2c
    converged = tau.norm() < converged_threshold
    camera.update_RT(new_w2c[0:3, 0:3], new_w2c[0:3, 3])
    camera.cam_rot_delta.data.fill_(0)
    camera.cam_trans_delta.data.fill_(0)
    return converged
