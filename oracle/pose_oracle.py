"""CPU oracle (test infrastructure only) of one tracking pose step: torch.optim.Adam semantics on (cam_trans_delta,
cam_rot_delta, exposure_a, exposure_b) followed by update_pose -- reference utils/pose_utils.py:12-93 and the optimiser
set-up of slam_frontend.py:135-160.  fp32 arithmetic like the reference's tensors.  Pinned against the reference's own
update_pose + torch.optim.Adam (tests/golden/make_pose_goldens.py -> tests/golden/pose_adam_steps.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np

F = np.float32


def skew(x):
    return np.array([[0, -x[2], x[1]], [x[2], 0, -x[0]], [-x[1], x[0], 0]], F)


def so3_exp(theta):                                                     # pose_utils.py:25-40
    W = skew(theta)
    W2 = (W @ W).astype(F)
    angle = F(np.sqrt(np.sum(theta.astype(F) ** 2, dtype=F)))
    I = np.eye(3, dtype=F)
    if angle < 1e-5:
        return (I + W + F(0.5) * W2).astype(F)
    return (I + F(np.sin(angle) / angle) * W + F((F(1) - F(np.cos(angle))) / F(angle ** 2)) * W2).astype(F)


def v_mat(theta):                                                       # pose_utils.py:43-58
    W = skew(theta)
    W2 = (W @ W).astype(F)
    angle = F(np.sqrt(np.sum(theta.astype(F) ** 2, dtype=F)))
    I = np.eye(3, dtype=F)
    if angle < 1e-5:
        return (I + F(0.5) * W + F(1.0 / 6.0) * W2).astype(F)
    return (I + W * F((F(1) - F(np.cos(angle))) / F(angle ** 2)) + W2 * F((angle - F(np.sin(angle))) / F(angle ** 3))).astype(F)


def se3_exp(tau):                                                       # pose_utils.py:61-73
    T = np.eye(4, dtype=F)
    T[:3, :3] = so3_exp(tau[3:])
    T[:3, 3] = v_mat(tau[3:]) @ tau[:3].astype(F)
    return T


class PoseAdam:
    """State of the four parameter groups; step(g_tau [rho, theta], g_exp [a, b]) -> dict."""

    def __init__(self, w2c, lr_rot, lr_trans, lr_exp_a, lr_exp_b, beta1=0.9, beta2=0.999, eps=1e-8, threshold=1e-4):
        self.w2c = np.array(w2c, F)
        self.lr = np.array([lr_trans] * 3 + [lr_rot] * 3 + [lr_exp_a, lr_exp_b], np.float64)
        self.b1, self.b2, self.eps, self.thr = beta1, beta2, eps, threshold
        self.m, self.v, self.t = np.zeros(8, F), np.zeros(8, F), 0
        self.exposure = np.zeros(2, F)

    def step(self, g_tau, g_exp):
        g = np.concatenate([np.asarray(g_tau, F), np.asarray(g_exp, F)])
        self.t += 1
        self.m = (F(self.b1) * self.m + F(1 - self.b1) * g).astype(F)
        self.v = (F(self.b2) * self.v + F(1 - self.b2) * g * g).astype(F)
        bc1, bc2 = 1.0 - self.b1 ** self.t, 1.0 - self.b2 ** self.t
        step_size = (self.lr / bc1).astype(F)
        denom = (np.sqrt(self.v) / F(np.sqrt(bc2)) + F(self.eps)).astype(F)
        delta = (-step_size * (self.m / denom)).astype(F)               # the parameters are zero before the step
        tau = delta[:6]
        self.exposure = (self.exposure + delta[6:]).astype(F)
        self.w2c = (se3_exp(tau) @ self.w2c).astype(F)                  # pose_utils.py:76-84 (left multiplication)
        self.w2c[3] = [0, 0, 0, 1]
        conv = bool(np.sqrt(np.sum(tau ** 2, dtype=F)) < self.thr)      # :89
        return dict(w2c=self.w2c.copy(), tau=tau.copy(), converged=conv, exposure=self.exposure.copy())
