"""The reference's verification harness on the HIP rasteriser (Jacobian_test.py / Jacobian_test.ipynb, pipeline (D) of SURVEY 3):
camera from w2c_gt @ T_noise with the scripts' hard-coded intrinsics (get_render_settings, Jacobian_test.py:63-141), a
Gaussian model, render() -> compute_loss -> backward, yielding the pose gradient the fork prints as grad_tau
(diff_gaussian_rasterization/__init__.py:167) and the per-Gaussian gradients.

The reference needs optimized_params_small.pt and a NOCS RGB-D frame (both missing blobs); here the model and the ground truth
are arguments, and `synthetic_case()` builds the stand-in of BASELINE config 1 (15 Gaussians, 640x480, ground truth = render
at the un-noised pose, mask = rendered opacity > 0.5).

Two equivalent ways to run it:
  run(model, camera, gt, autograd=True)   exactly the reference's call sequence: gaussian_renderer.render(), the torch
                                          compute_loss of utils/slam_utils.py, loss.backward()  (drop-in autograd function)
  run(..., autograd=False)                no torch graph: FrameContext forward, compute_loss seeds + isotropic term in two
                                          device launches (gsaj_loss_seeds / gsaj_isotropic_loss), C-ABI backward
"""
import numpy as np
import torch

from . import synthetic as syn
from .losses import IsotropicLoss, LossSeeds, compute_loss_seeds
from .rasterizer import FrameContext


def get_render_settings(w=640, h=480, fx=577.5, fy=577.5, cx=319.5, cy=239.5, w2c=None, near=0.01, far=100.0):
    """Camera dictionary of Jacobian_test.py:63-141 (same defaults: the NOCS intrinsics the scripts hard-code)."""
    if w2c is None:
        w2c = syn.W2C_GT @ syn.T_NOISE
    return syn.make_camera(np.asarray(w2c, np.float64), W=w, H=h, fx=fx, fy=fy, cx=cx, cy=cy, znear=near, zfar=far)


def synthetic_case(device="cuda:0", N=15, seed=15):
    """Stand-in for optimized_params_small.pt + the NOCS frame: (scene dict, noisy camera, gt dict)."""
    from gaussian_splatting.gaussian_renderer import render
    from gaussian_splatting.scene.gaussian_model import GaussianModel
    from utils.camera_utils import Camera

    cam_gt = get_render_settings(w2c=syn.W2C_GT)
    cam = get_render_settings()
    sc = syn.make_scene(N, seed, cam_gt, z_range=(0.8, 1.6), log_scale_range=(np.log(0.01), np.log(0.05)))
    model = GaussianModel.from_activated(sc["means3D"], sc["scales"], sc["rotations"], sc["opacities"], sc["shs"], sh_degree=3,
                                         device=device)

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False

    with torch.no_grad():
        pkg = render(Camera.from_synthetic(cam_gt, device=device), model, Pipe, torch.zeros(3, device=device))
    gt = dict(color=pkg["render"].detach().clone(), depth=pkg["depth"].detach().clone(), mask=(pkg["opacity"][0] > 0.5))
    return sc, model, cam, gt


def run(model, cam, gt, autograd=True, device="cuda:0", iso_weight=10.0):
    """-> dict(loss, grad_tau [6] = [rho, theta], grad_xyz [P,3], grad_scaling [P,3] (w.r.t. the ACTIVATED scales), render, depth)."""
    dev = torch.device(device)
    bg = torch.zeros(3, device=dev)
    if autograd:
        from gaussian_splatting.gaussian_renderer import render
        from utils.camera_utils import Camera
        from utils.slam_utils import compute_loss

        class Pipe:
            convert_SHs_python = False
            compute_cov3D_python = False

        view = Camera.from_synthetic(cam, device=device)
        for p in model.parameters():
            p.grad = None
        pkg = render(view, model, Pipe, bg)
        loss = compute_loss(model, pkg["render"], pkg["depth"], gt["color"], gt["depth"], gt["mask"])
        loss.backward()
        # autograd differentiates w.r.t. the stored log-scales; / exp(log s) gives the gradient w.r.t. the activated scales
        return dict(loss=loss.detach(), grad_tau=torch.cat([view.cam_trans_delta.grad, view.cam_rot_delta.grad]),
                    grad_xyz=model._xyz.grad, grad_scaling=model._scaling.grad / model.get_scaling.detach(),
                    render=pkg["render"].detach(), depth=pkg["depth"].detach())
    from utils.camera_utils import Camera

    xyz, opac, shs = model.get_xyz.detach().contiguous(), model.get_opacity.detach().contiguous(), model.get_features.detach().contiguous()
    scales, rots = model.get_scaling.detach().contiguous(), model.get_rotation.detach().contiguous()
    P, M, W, H = xyz.shape[0], shs.shape[1], cam["W"], cam["H"]
    ctx = FrameContext(P, W, H, M, dev)
    c = Camera.from_synthetic(cam, device=device)  # the matrices exactly as render() derives them (camera_utils.py:95-109)
    view, proj = c.world_view_transform.contiguous(), c.full_proj_transform.contiguous()
    praw, campos = c.projection_matrix.contiguous(), c.camera_center.contiguous()
    deg = model.active_sh_degree
    ctx.forward(bg, xyz, opac, view, proj, campos, cam["tanfovx"], cam["tanfovy"], sh_degree=deg, shs=shs, scales=scales, rotations=rots)
    ls = LossSeeds(W, H, dev)
    s = compute_loss_seeds(ls, ctx.color, ctx.depth, gt["color"].contiguous(), gt["depth"].contiguous(), gt["mask"])
    g = ctx.backward(bg, xyz, view, proj, praw, campos, cam["tanfovx"], cam["tanfovy"], s["dL_dcolor"], s["dL_ddepth"], sh_degree=deg,
                     shs=shs, scales=scales, rotations=rots)
    iso, g_scale = IsotropicLoss(P, dev)(scales, iso_weight, grad_out=g["scale"], accumulate=True)
    return dict(loss=s["loss"] + iso, grad_tau=g["tau_sum"], grad_xyz=g["mean3D"], grad_scaling=g_scale, render=ctx.color, depth=ctx.depth)
