#!/usr/bin/env python3
"""Golden vectors for the on-device pose step: torch.optim.Adam on (cam_rot_delta, cam_trans_delta,
exposure_a, exposure_b) followed by the reference's own utils/pose_utils.update_pose, as slam_frontend.tracking does
(slam_frontend.py:135-193).  Build container only (imports /root/reference/utils/pose_utils.py, torch + numpy);
stores inputs (gradient sequence, initial pose, learning rates) and outputs (pose, deltas, flags per step)."""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


class Cam:
    def __init__(self, R, T):
        self.R, self.T, self.device = R, T, "cpu"
        self.cam_rot_delta = torch.nn.Parameter(torch.zeros(3))
        self.cam_trans_delta = torch.nn.Parameter(torch.zeros(3))
        self.exposure_a = torch.nn.Parameter(torch.tensor([0.0]))
        self.exposure_b = torch.nn.Parameter(torch.tensor([0.0]))

    def update_RT(self, R, t):
        self.R, self.T = R, t


def main():
    spec = importlib.util.spec_from_file_location("ref_pose_utils", "/root/reference/utils/pose_utils.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    w2c = np.loadtxt(os.path.join(HERE, "reference_fixtures", "w2c_gt.txt")).astype(np.float32)
    rng = np.random.default_rng(7)
    steps = 72  # 8 steps with gradients, then zero gradients: the Adam step decays through the convergence
    scale = np.zeros(steps, np.float32)  # threshold (1e-4) and the small-angle branch of SO3_exp / V (1e-5)
    scale[:8] = [1e-1, 1e-2, 1e-3, 1e-4, 3e-2, 1e-6, 1e-3, 1e-2]
    g_tau = (rng.normal(size=(steps, 6)) * scale[:, None]).astype(np.float32)   # [rho, theta] like dL_dtau_sum
    g_exp = (rng.normal(size=(steps, 2)) * 1e-2).astype(np.float32)
    lrs = dict(rot=0.003, trans=0.001, exp_a=0.01, exp_b=0.01)
    cam = Cam(torch.tensor(w2c[:3, :3]), torch.tensor(w2c[:3, 3]))
    opt = torch.optim.Adam([
        {"params": [cam.cam_rot_delta], "lr": lrs["rot"]}, {"params": [cam.cam_trans_delta], "lr": lrs["trans"]},
        {"params": [cam.exposure_a], "lr": lrs["exp_a"]}, {"params": [cam.exposure_b], "lr": lrs["exp_b"]}])
    out_w2c, out_tau, out_conv, out_exp = [], [], [], []
    for k in range(steps):
        opt.zero_grad()
        cam.cam_trans_delta.grad = torch.tensor(g_tau[k, :3])
        cam.cam_rot_delta.grad = torch.tensor(g_tau[k, 3:])
        cam.exposure_a.grad = torch.tensor(g_exp[k, :1])
        cam.exposure_b.grad = torch.tensor(g_exp[k, 1:])
        with torch.no_grad():
            opt.step()
            tau = torch.cat([cam.cam_trans_delta, cam.cam_rot_delta]).clone()
            conv = ref.update_pose(cam, converged_threshold=1e-4)
        m = np.eye(4, dtype=np.float32)
        m[:3, :3], m[:3, 3] = cam.R.numpy(), cam.T.numpy()
        out_w2c.append(m)
        out_tau.append(tau.numpy())
        out_conv.append(bool(conv))
        out_exp.append([float(cam.exposure_a.detach()), float(cam.exposure_b.detach())])
    np.savez_compressed(os.path.join(HERE, "pose_adam_steps.npz"), w2c0=w2c, g_tau=g_tau, g_exp=g_exp,
                        lr=np.array([lrs["rot"], lrs["trans"], lrs["exp_a"], lrs["exp_b"]], np.float32), w2c=np.stack(out_w2c),
                        tau=np.stack(out_tau), converged=np.array(out_conv), exposure=np.array(out_exp, np.float32),
                        threshold=np.float32(1e-4))
    print("converged flags", out_conv)


if __name__ == "__main__":
    main()
