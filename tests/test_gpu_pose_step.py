"""GPU: gsaj_pose_adam_step against the reference-generated goldens (update_pose + torch.optim.Adam) and the oracle."""
import os

import numpy as np
import pytest

from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_adam_steps.npz"))


def test_pose_step_matches_reference_steps():
    import torch
    from gsaj import pose_step, synthetic as syn

    dev = torch.device("cuda:0")
    cam = syn.make_camera(G["w2c0"].astype(np.float64))
    lr = G["lr"]
    pt = pose_step.PoseTracker(G["w2c0"], cam["projmatrix_raw"], dev, lr_rot=float(lr[0]), lr_trans=float(lr[1]),
                               lr_exposure_a=float(lr[2]), lr_exposure_b=float(lr[3]), converged_threshold=float(G["threshold"]))
    np.testing.assert_allclose(pt.viewmatrix.cpu().numpy(), cam["viewmatrix"].reshape(4, 4), atol=1e-6)
    np.testing.assert_allclose(pt.projmatrix.cpu().numpy(), cam["projmatrix"].reshape(4, 4), atol=2e-5)
    np.testing.assert_allclose(pt.campos.cpu().numpy(), cam["campos"], atol=1e-5)
    gt, ge = torch.as_tensor(G["g_tau"], device=dev), torch.as_tensor(G["g_exp"], device=dev)
    for k in range(G["g_tau"].shape[0]):
        pt.step(gt[k].contiguous(), ge[k].contiguous())   # no host sync needed between steps
        np.testing.assert_allclose(pt.tau.cpu().numpy(), G["tau"][k], rtol=3e-5, atol=1e-9, err_msg="tau step %d" % k)
        w = pt.w2c.cpu().numpy()
        np.testing.assert_allclose(w, G["w2c"][k], rtol=0, atol=5e-6, err_msg="w2c step %d" % k)
        assert bool(pt.converged.item() > 0.5) == bool(G["converged"][k]), k
        np.testing.assert_allclose([pt.exposure_a.item(), pt.exposure_b.item()], G["exposure"][k], rtol=3e-5, atol=1e-8)
        # derived camera matrices are consistent with the pose (camera_utils.py:95-109)
        np.testing.assert_allclose(pt.viewmatrix.cpu().numpy(), w.T, atol=0)
        np.testing.assert_allclose(pt.projmatrix.cpu().numpy(), w.T @ cam["projmatrix_raw"].reshape(4, 4), atol=3e-5)
        np.testing.assert_allclose(pt.campos.cpu().numpy(), -np.linalg.inv(w[:3, :3].astype(np.float64)) @ w[:3, 3], atol=2e-5)
    r0, r = G["w2c0"][:3, :3].astype(np.float64), w[:3, :3].astype(np.float64)
    assert np.abs(r @ r.T - r0 @ r0.T).max() < 1e-4  # 72 left-multiplied increments later R R^T (scale^2 I here) is unchanged


def test_batched_pose_step_equals_single_steps_and_honours_the_active_mask():
    """gsaj_pose_adam_step_batch: K keyframe poses in one launch, each with its own Adam state (slam_backend.py:255-262); the
    masked-out pose (the reference never moves keyframe 0) keeps its state bit for bit."""
    import torch
    from gsaj.pose_step import PoseTracker, PoseTrackerBatch

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    K = 5
    cams = syn.keyframe_cameras(K)
    w2cs = [np.ascontiguousarray(c["viewmatrix"].T) for c in cams]
    praw = cams[0]["projmatrix_raw"]
    batch = PoseTrackerBatch(w2cs, praw, dev)
    singles = [PoseTracker(w, praw, dev) for w in w2cs]
    active = torch.tensor([0, 1, 1, 0, 1], dtype=torch.uint8, device=dev)
    for _ in range(7):
        g_tau = torch.as_tensor(rng.normal(scale=0.05, size=(K, 6)), dtype=torch.float32, device=dev)
        g_exp = torch.as_tensor(rng.normal(scale=0.05, size=(K, 2)), dtype=torch.float32, device=dev)
        batch.step(g_tau, g_exp, active)
        for k in range(K):
            if int(active[k]):
                singles[k].step(g_tau[k].contiguous(), g_exp[k].contiguous())
    for k in range(K):
        assert torch.equal(batch.state[k], singles[k].state), k
    vm, pm, cp = batch.matrices()
    assert torch.equal(vm[2], singles[2].viewmatrix) and torch.equal(pm[2], singles[2].projmatrix) and torch.equal(cp[2], singles[2].campos)
    assert float(batch.state[0, 32]) == 0.0 and float(batch.state[1, 32]) == 7.0  # step counters: masked pose never stepped
