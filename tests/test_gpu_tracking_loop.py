"""GPU integration: a tracking loop run entirely on the device (FrameContext forward -> gsaj_loss_seeds -> backward ->
gsaj_pose_adam_step, no host synchronisation) lands on the same poses as the reference-style host loop on the drop-in
(render() autograd + utils.slam_utils.get_loss_tracking + torch.optim.Adam + utils.pose_utils.update_pose), iteration by
iteration (slam_frontend.py:135-193)."""
import math

import numpy as np
import pytest

import helpers as hp
from gsaj import synthetic as syn

pytestmark = pytest.mark.gpu


def test_device_tracking_loop_matches_host_loop():
    import torch
    from gaussian_splatting.gaussian_renderer import render
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from gsaj import losses, pose_step
    from gsaj.rasterizer import FrameContext
    from utils import pose_utils, slam_utils
    from utils.camera_utils import Camera

    dev = torch.device("cuda:0")
    W, H = 160, 120
    cam_gt = syn.fixture_camera(noisy=False, orthonormal=True, W=W, H=H, fx=140.0, fy=140.0, cx=79.5, cy=59.5)
    cam0 = syn.fixture_camera(noisy=True, orthonormal=True, W=W, H=H, fx=140.0, fy=140.0, cx=79.5, cy=59.5)
    sc = syn.make_scene(3000, 11, cam_gt, z_range=(1.0, 4.0), log_scale_range=(math.log(0.02), math.log(0.1)))
    model = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"], sh_degree=3, device=dev)
    bg = torch.zeros(3, device=dev)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False

    with torch.no_grad():  # ground truth = the render at the un-noised pose
        gt = render(Camera.from_synthetic(cam_gt, device=dev), model, Pipe, bg)
    cfg = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.9}}
    lr = dict(rot=0.003, trans=0.001, a=0.01, b=0.01)
    n_iter = 6

    # ---- host loop on the drop-in (the reference's tracking iteration) ----
    view = Camera.from_synthetic(cam0, color=gt["render"].detach(), depth=gt["depth"].detach()[0].cpu().numpy(), device=dev)
    view.grad_mask = torch.ones((1, H, W), dtype=torch.bool, device=dev)
    opt = torch.optim.Adam([{"params": [view.cam_rot_delta], "lr": lr["rot"]}, {"params": [view.cam_trans_delta], "lr": lr["trans"]},
                            {"params": [view.exposure_a], "lr": lr["a"]}, {"params": [view.exposure_b], "lr": lr["b"]}])
    host_poses, host_losses = [], []
    for _ in range(n_iter):
        opt.zero_grad()
        pkg = render(view, model, Pipe, bg)
        loss = slam_utils.get_loss_tracking(cfg, pkg["render"], pkg["depth"], pkg["opacity"], view)
        loss.backward()
        with torch.no_grad():
            opt.step()
            pose_utils.update_pose(view)
        w = torch.eye(4, device=dev)
        w[:3, :3], w[:3, 3] = view.R, view.T
        host_poses.append(w.cpu().numpy())
        host_losses.append(float(loss.detach()))

    # ---- the same loop with no host round trip ----
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, M = sc["means3D"].shape[0], sc["shs"].shape[1]
    ctx = FrameContext(P, W, H, M, dev)
    pt = pose_step.PoseTracker(cam0["w2c"], cam0["projmatrix_raw"], dev, lr_rot=lr["rot"], lr_trans=lr["trans"],
                               lr_exposure_a=lr["a"], lr_exposure_b=lr["b"])
    ls = losses.LossSeeds(W, H, dev)
    means, opac, shs, scales, rots = (t(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations"))
    gt_c, gt_d = gt["render"].detach().contiguous(), gt["depth"].detach()[0].contiguous()
    dev_poses, dev_losses = [], []
    for it in range(n_iter):
        ctx.forward(bg, means, opac, pt.viewmatrix, pt.projmatrix, pt.campos, cam0["tanfovx"], cam0["tanfovy"], sh_degree=3, shs=shs,
                    scales=scales, rotations=rots, sync=(it == 0))
        o = ls(losses.TRACKING, 0.9, 0.01, ctx.color, ctx.depth, ctx.opacity, gt_c, gt_d, view.grad_mask, pt.exposure_a, pt.exposure_b)
        g = ctx.backward(bg, means, pt.viewmatrix, pt.projmatrix, t(cam0["projmatrix_raw"]), pt.campos, cam0["tanfovx"], cam0["tanfovy"],
                         o["dL_dcolor"], o["dL_ddepth"], sh_degree=3, shs=shs, scales=scales, rotations=rots)
        dev_losses.append(o["loss"].clone())
        pt.step(g["tau_sum"], ls.scalars[3:5])
        dev_poses.append(pt.w2c.clone())
    ctx.status()
    for it in range(n_iter):
        assert abs(float(dev_losses[it]) - host_losses[it]) < 2e-5 * abs(host_losses[it]) + 1e-8, it
        np.testing.assert_allclose(dev_poses[it].cpu().numpy(), host_poses[it], rtol=0, atol=2e-5, err_msg="iteration %d" % it)
