#!/usr/bin/env python3
"""One tracking iteration (render -> loss -> backward to dL/dtau) on cfg2, three ways:
  A. drop-in: gaussian_renderer.render() autograd function + torch losses (utils/slam_utils.get_loss_tracking) + .backward()
  B. FrameContext (no host sync) + torch loss / autograd for the seeds
  C. FrameContext + gsaj_loss_seeds (one kernel for loss + seeds)
Prints ms per iteration (synchronised wall clock over N iterations)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import losses, synthetic as syn  # noqa: E402
from gsaj.rasterizer import FrameContext  # noqa: E402
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from utils import slam_utils  # noqa: E402


class View:
    pass


POSE_ONLY = [False]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene("cfg2")
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    means, opac, shs, scales, rots = t(sc["means3D"]), t(sc["opacities"]), t(sc["shs"]), t(sc["scales"]), t(sc["rotations"])
    view, proj, proj_raw, campos = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["projmatrix_raw"]), t(cam["campos"])
    bg = torch.zeros(3, device=dev)
    rng = np.random.default_rng(0)
    v = View()
    v.original_image = t(rng.uniform(0, 1, (3, H, W)))
    v.depth = rng.uniform(0.5, 4, (H, W)).astype(np.float32)
    v.grad_mask = torch.as_tensor(rng.uniform(size=(1, H, W)) < 0.7, device=dev)
    v.exposure_a = torch.zeros(1, device=dev, requires_grad=True)
    v.exposure_b = torch.zeros(1, device=dev, requires_grad=True)
    cfg = {"Training": {"monocular": False, "rgb_boundary_threshold": 0.01, "alpha": 0.95}}
    gt_depth_dev = t(v.depth)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    # A: drop-in autograd path
    theta = torch.zeros(3, device=dev, requires_grad=True)
    rho = torch.zeros(3, device=dev, requires_grad=True)
    means2D = torch.zeros((P, 3), device=dev, requires_grad=True)
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg,
                                       scale_modifier=1.0, viewmatrix=view, projmatrix=proj, projmatrix_raw=proj_raw, sh_degree=3,
                                       campos=campos, prefiltered=False, debug=False)
    rast = GaussianRasterizer(st)

    def iter_a():
        color, radii, depth, opacity, n_touched = rast(means3D=means, means2D=means2D, opacities=opac, shs=shs, scales=scales,
                                                       rotations=rots, theta=theta, rho=rho)
        loss = slam_utils.get_loss_tracking(cfg, color, depth, opacity, v)
        loss.backward()
        theta.grad = None
        rho.grad = None
        means2D.grad = None

    ctx = FrameContext(P, W, H, M, dev)
    fa = dict(bg=bg, means3D=means, opacities=opac, viewmatrix=view, projmatrix=proj, campos=campos, tanfovx=cam["tanfovx"],
              tanfovy=cam["tanfovy"], sh_degree=3, shs=shs, scales=scales, rotations=rots)
    ctx.forward(**fa)

    def bwd(dc, dd):
        ctx.backward(bg=bg, means3D=means, viewmatrix=view, projmatrix=proj, projmatrix_raw=proj_raw, campos=campos,
                     tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], dL_dcolor=dc, dL_ddepth=dd, sh_degree=3, shs=shs, scales=scales,
                     rotations=rots)

    def iter_b():
        color, depth, opacity = _fwd_b()
        c = color.detach().requires_grad_(True)
        d = depth.detach().requires_grad_(True)
        loss = slam_utils.get_loss_tracking(cfg, c, d, opacity.detach(), v)
        loss.backward()
        bwd(c.grad.contiguous(), d.grad.contiguous())

    def _fwd_b():
        ctx.forward(sync=False, **fa)
        return ctx.color, ctx.depth, ctx.opacity

    ls = losses.LossSeeds(W, H, dev)

    def iter_c():
        ctx.forward(sync=False, **fa)
        o = ls(losses.TRACKING, 0.95, 0.01, ctx.color, ctx.depth, ctx.opacity, v.original_image, gt_depth_dev,
               v.grad_mask, v.exposure_a.detach(), v.exposure_b.detach())
        bwd(o["dL_dcolor"], o["dL_ddepth"])

    # D: the whole iteration on the device: render from the tracker's matrices -> loss seeds -> backward -> Adam + update_pose
    from gsaj import pose_step
    w2c = np.asarray(cam["viewmatrix"], np.float32).reshape(4, 4).T
    # learning rates 0: the ground truth here is noise, a moving pose would drift off the scene and make later sections cheaper
    pt = pose_step.PoseTracker(w2c, cam["projmatrix_raw"], dev, lr_rot=0.0, lr_trans=0.0, lr_exposure_a=0.0, lr_exposure_b=0.0)
    fd = dict(fa)

    def iter_d():
        fd.update(viewmatrix=pt.viewmatrix, projmatrix=pt.projmatrix, campos=pt.campos)
        ctx.forward(sync=False, **fd)
        o = ls(losses.TRACKING, 0.95, 0.01, ctx.color, ctx.depth, ctx.opacity, v.original_image, gt_depth_dev, v.grad_mask,
               pt.exposure_a, pt.exposure_b)
        g = ctx.backward(bg=bg, means3D=means, viewmatrix=pt.viewmatrix, projmatrix=pt.projmatrix, projmatrix_raw=proj_raw,
                         campos=pt.campos, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], dL_dcolor=o["dL_dcolor"],
                         dL_ddepth=o["dL_ddepth"], sh_degree=3, shs=shs, scales=scales, rotations=rots, pose_only=POSE_ONLY[0])
        pt.step(g["tau_sum"], ls.scalars[3:5])

    # A': the drop-in path with the reference's optimiser step, update_pose and converged read-back (slam_frontend.py:163-193)
    from utils import pose_utils

    class CamA:
        pass

    ca = CamA()
    ca.R, ca.T, ca.device = torch.as_tensor(w2c[:3, :3], device=dev), torch.as_tensor(w2c[:3, 3], device=dev), dev
    ca.cam_rot_delta = torch.nn.Parameter(torch.zeros(3, device=dev))
    ca.cam_trans_delta = torch.nn.Parameter(torch.zeros(3, device=dev))
    ca.update_RT = lambda R, t: (setattr(ca, "R", R), setattr(ca, "T", t))
    opt = torch.optim.Adam([{"params": [ca.cam_rot_delta], "lr": 0.003}, {"params": [ca.cam_trans_delta], "lr": 0.001},
                            {"params": [v.exposure_a], "lr": 0.01}, {"params": [v.exposure_b], "lr": 0.01}])

    def iter_a2():
        opt.zero_grad()
        color, radii, depth, opacity, n_touched = rast(means3D=means, means2D=means2D, opacities=opac, shs=shs, scales=scales,
                                                       rotations=rots, theta=ca.cam_rot_delta, rho=ca.cam_trans_delta)
        loss = slam_utils.get_loss_tracking(cfg, color, depth, opacity, v)
        loss.backward()
        with torch.no_grad():
            opt.step()
            converged = pose_utils.update_pose(ca)
        means2D.grad = None
        return bool(converged)  # the host read-back of the reference loop

    ms_a, ms_b, ms_c = timed(iter_a), timed(iter_b), timed(iter_c)
    ms_a2, ms_d = timed(iter_a2), timed(iter_d)
    POSE_ONLY[0] = True
    ms_d_pose = timed(iter_d)
    print("D with the pose-only backward (no per-Gaussian parameter gradients): %.3f ms" % ms_d_pose)
    # E: iteration D captured once into a HIP graph (all buffers are persistent) and replayed
    gm8 = v.grad_mask.to(torch.uint8)
    v.grad_mask = gm8
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        iter_d()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        iter_d()
    ms_e = timed(graph.replay)
    print("E: iteration D as one HIP graph replay: %.3f ms" % ms_e)
    print("full tracking iteration incl. Adam + update_pose: A' drop-in + torch optimiser + converged read-back %.3f ms | "
          "D all on the device (FrameContext + gsaj_loss_seeds + gsaj_pose_adam_step, no host sync) %.3f ms" % (ms_a2, ms_d))
    print("tracking iteration, cfg2 (%d Gaussians, %dx%d): A drop-in autograd %.3f ms | B FrameContext + torch loss %.3f ms | "
          "C FrameContext + gsaj_loss_seeds %.3f ms" % (P, W, H, ms_a, ms_b, ms_c))


if __name__ == "__main__":
    main()
