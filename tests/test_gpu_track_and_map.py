"""GPU integration: the two loops the hot path sits in, end to end on synthetic data -- the way `slam.py` uses it, without the
reference's frontend / backend processes (out of scope).

  tracking (utils/slam_frontend.py:135-193): every new frame starts from the previous pose and is tracked on a FIXED map with
      gsaj.tracking.DeviceTracker (render -> tracking loss -> analytical dL/dtau -> Adam + update_pose, all on the device);
  mapping  (utils/slam_backend.py:168-232): the window's keyframes rendered against ONE map through BatchContext, mapping-loss seeds
      per view (gsaj_loss_seeds), one batched backward, the per-Gaussian gradients -- summed over the window in the kernel --
      handed to torch.optim.Adam as the parameters' .grad.

Ground truth: renders of the TRUE map at the TRUE poses.  What is checked is behaviour, not bits (the bits are the parity tests'
business): tracking pulls the pose towards the truth frame after frame, mapping lowers the window's loss on a perturbed map."""
import math

import numpy as np
import pytest

from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu

W, H, F = 160, 120, 140.0


def _true_world(P=3000, seed=21, n_frames=4):
    import torch
    from gsaj.rasterizer import FrameContext

    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    cams = syn.keyframe_cameras(n_frames, radius=0.25, W=W, H=H, fx=F, fy=F, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    sc = syn.make_scene(P, seed, cams[n_frames // 2], z_range=(1.0, 4.0), log_scale_range=(math.log(0.02), math.log(0.1)))
    g = dict(means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    bg = torch.zeros(3, device=dev)
    M = sc["shs"].shape[1]
    fc = FrameContext(P, W, H, M, dev)
    frames = []
    for c in cams:  # ground-truth colour / depth of every frame
        fc.forward(bg, g["means3D"], g["opacities"], t(c["viewmatrix"]), t(c["projmatrix"]), t(c["campos"]), c["tanfovx"], c["tanfovy"],
                   sh_degree=3, shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        frames.append((fc.color.clone(), fc.depth[0].clone()))
    return dev, t, cams, sc, g, bg, M, frames


def test_tracking_follows_the_camera_over_consecutive_frames():
    import torch
    from gsaj.tracking import DeviceTracker

    dev, t, cams, sc, g, bg, M, frames = _true_world()
    P = g["means3D"].shape[0]
    w2c_true = [np.ascontiguousarray(c["viewmatrix"].T) for c in cams]
    tr = DeviceTracker(P, W, H, M, dev, w2c_true[0], t(cams[0]["projmatrix_raw"]), cams[0]["tanfovx"], cams[0]["tanfovy"], bg, alpha=0.9,
                       sh_degree=3, lr_rot=0.003, lr_trans=0.002, **g)
    est = w2c_true[0].copy()
    for k in range(1, len(cams)):
        # constant-position prior: start from the previous estimate (slam_frontend.py:142-147 does the same)
        tr.set_frame(frames[k][0], frames[k][1], w2c=est)
        before = float(np.abs(est[:3, 3] - w2c_true[k][:3, 3]).max())
        loss0 = None
        for _ in range(6):
            tr.iterate(25)
            loss0 = float(tr.loss_terms[0]) if loss0 is None else loss0
        est = tr.w2c.cpu().numpy().astype(np.float32)
        after = float(np.abs(est[:3, 3] - w2c_true[k][:3, 3]).max())
        rot_err = float(np.abs(est[:3, :3] @ w2c_true[k][:3, :3].T - np.eye(3)).max())
        assert float(tr.loss_terms[0]) < 0.5 * loss0, (k, loss0, float(tr.loss_terms[0]))
        assert after < 0.35 * before and rot_err < 0.02, (k, before, after, rot_err)
        assert np.allclose(est[:3, :3] @ est[:3, :3].T, np.eye(3), atol=1e-4)  # update_pose keeps the pose a rigid transform


def test_mapping_window_lowers_its_loss_with_adam_on_the_bucket_gradients():
    import torch
    from gsaj.losses import LossSeedsBatch
    from gsaj.rasterizer import BatchContext

    dev, t, cams, sc, g, bg, M, frames = _true_world()
    P, K = g["means3D"].shape[0], len(cams)
    rng = np.random.default_rng(5)
    # the map to refine: the true one, disturbed
    params = dict(
        means3D=torch.nn.Parameter(g["means3D"] + t(rng.normal(scale=0.01, size=(P, 3)))),
        shs=torch.nn.Parameter(g["shs"] + t(rng.normal(scale=0.05, size=tuple(g["shs"].shape)))),
        opac_logit=torch.nn.Parameter(torch.logit(g["opacities"].clamp(0.02, 0.98)) + t(rng.normal(scale=0.3, size=(P, 1)))),
        log_scales=torch.nn.Parameter(torch.log(g["scales"]) + t(rng.normal(scale=0.05, size=(P, 3)))),
        rot=torch.nn.Parameter(g["rotations"] + t(rng.normal(scale=0.01, size=(P, 4)))))
    opt = torch.optim.Adam([dict(params=[params["means3D"]], lr=1e-3), dict(params=[params["shs"]], lr=5e-3),
                            dict(params=[params["opac_logit"]], lr=2e-2), dict(params=[params["log_scales"]], lr=2e-3),
                            dict(params=[params["rot"]], lr=1e-3)])
    views, projs, cps = (t(np.stack([c[k] for c in cams])) for k in ("viewmatrix", "projmatrix", "campos"))
    praw = t(cams[0]["projmatrix_raw"])
    bc = BatchContext(K, P, W, H, M, dev)
    ls = LossSeedsBatch(K, W, H, dev)
    gt_color, gt_depth = torch.stack([f[0] for f in frames]).contiguous(), torch.stack([f[1] for f in frames]).contiguous()
    exp_a, exp_b = torch.zeros(K, device=dev), torch.zeros(K, device=dev)
    tx, ty = cams[0]["tanfovx"], cams[0]["tanfovy"]
    losses = []
    for it in range(40):
        with torch.no_grad():  # the activations of gaussian_model.py:141-177
            opac = torch.sigmoid(params["opac_logit"]).contiguous()
            scales = torch.exp(params["log_scales"]).contiguous()
            rot = torch.nn.functional.normalize(params["rot"]).contiguous()
        geo = dict(sh_degree=3, shs=params["shs"].detach(), scales=scales, rotations=rot)
        bc.forward(bg, params["means3D"].detach(), opac, views, projs, cps, tx, ty, sync=(it == 0), **geo)
        # get_loss_mapping of every keyframe of the window in one launch (slam_utils.py:91-128; flags 0 = RGB-D mapping loss with exposure)
        o = ls(0, 0.95, 0.01, bc.color, bc.depth, bc.opacity, gt_color, gt_depth, None, exp_a, exp_b)
        losses.append(float(o["loss"].sum()))
        gr = bc.backward(bg, params["means3D"].detach(), views, projs, praw, cps, tx, ty, o["dL_dcolor"], o["dL_ddepth"], **geo)
        with torch.no_grad():  # chain through the activations, then the optimiser takes the gradients as they come out of the bucket
            params["means3D"].grad = gr["mean3D"].clone()
            params["shs"].grad = gr["sh"].view_as(params["shs"]).clone()
            params["opac_logit"].grad = (gr["opacity"].view(P, 1) * opac * (1 - opac)).clone()
            params["log_scales"].grad = (gr["scale"] * scales).clone()
            n = params["rot"].norm(dim=1, keepdim=True)
            gq = gr["rot"]
            params["rot"].grad = ((gq - rot * (gq * rot).sum(dim=1, keepdim=True)) / n).clone()
        opt.step()
    assert bc.status()[0][2] is False
    assert losses[-1] < 0.6 * losses[0], (losses[0], losses[-1])
    assert all(b < a * 1.05 for a, b in zip(losses, losses[1:])), "the window loss should go down almost monotonically"


def test_window_pose_refinement_with_the_batched_pose_step():
    """Keyframe poses of the window optimised on a fixed map, everything batched: BatchContext -> LossSeedsBatch ->
    backward -> PoseTrackerBatch.step on the K rows of dL/dtau (keyframe 0 stays fixed, as in slam_backend.py:255-258)."""
    import torch
    from gsaj.losses import LossSeedsBatch
    from gsaj.pose_step import PoseTrackerBatch
    from gsaj.rasterizer import BatchContext

    dev, t, cams, sc, g, bg, M, frames = _true_world()
    P, K = g["means3D"].shape[0], len(cams)
    rng = np.random.default_rng(9)
    w2c_true = [np.ascontiguousarray(c["viewmatrix"].T).astype(np.float32) for c in cams]
    w2c0 = [w.copy() for w in w2c_true]
    for k in range(1, K):
        w2c0[k][:3, 3] += rng.normal(scale=0.01, size=3).astype(np.float32)
    poses = PoseTrackerBatch(w2c0, t(cams[0]["projmatrix_raw"]), dev, lr_rot=0.001, lr_trans=0.001, lr_exposure_a=0.0, lr_exposure_b=0.0)
    active = torch.tensor([0] + [1] * (K - 1), dtype=torch.uint8, device=dev)
    bc = BatchContext(K, P, W, H, M, dev)
    ls = LossSeedsBatch(K, W, H, dev)
    gt_color, gt_depth = torch.stack([f[0] for f in frames]).contiguous(), torch.stack([f[1] for f in frames]).contiguous()
    geo = dict(sh_degree=3, shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    praw = t(cams[0]["projmatrix_raw"])
    tx, ty = cams[0]["tanfovx"], cams[0]["tanfovy"]
    err0 = [float(np.abs(w2c0[k][:3, 3] - w2c_true[k][:3, 3]).max()) for k in range(K)]
    first = None
    for it in range(120):
        views, projs, cps = poses.matrices()
        bc.forward(bg, g["means3D"], g["opacities"], views, projs, cps, tx, ty, sync=(it == 0), **geo)
        o = ls(0, 0.9, 0.01, bc.color, bc.depth, bc.opacity, gt_color, gt_depth, None, poses.exposure[:, 0].contiguous(),
               poses.exposure[:, 1].contiguous())
        first = float(o["loss"].sum()) if first is None else first
        gr = bc.backward(bg, g["means3D"], views, projs, praw, cps, tx, ty, o["dL_dcolor"], o["dL_ddepth"], **geo)
        poses.step(gr["tau_all"].contiguous(), ls.scalars[:, 3:5].contiguous(), active)
    est = poses.w2c.cpu().numpy()
    assert np.array_equal(est[0], w2c_true[0])  # the fixed keyframe did not move
    for k in range(1, K):
        err = float(np.abs(est[k][:3, 3] - w2c_true[k][:3, 3]).max())
        assert err < 0.4 * err0[k], (k, err0[k], err)
    assert float(o["loss"].sum()) < 0.5 * first
