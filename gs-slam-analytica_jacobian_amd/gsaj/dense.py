"""The reference's CPU/NumPy analytic path on the GPU ("dense" semantics).

Mirrors the functions of Loss_Derivative_script_compare.py that produce the committed goldens
grad_mu_I_pixel.npy, grad_Sigma_I_pixel.npy, grad_depth_per_gaussian.npy and dL_dtau.npy:

  project_and_sort(...)                        <- OrderGaussiansByDepth (:764-769) + GetImagePlaneMeanAndCovs (:854-971)
                                                  with compute_cov2d (:772-848), ndc2Pix (:851-852), SH colours (:535-588)
  compute_gradients_2D(...)                    <- compute_gradients_2D_vectorized_chunked (:1173-1351);
                                                  naive_guards=True: the naive loop's edge branches (:1050-1169, wrt.py:3-118)
  jacobian_test(...)                           <- the __main__ pipeline (:1354-1706): world Gaussians + camera + ground truth
                                                  -> grad_mu_I, grad_Sigma_I, grad_depth_per_gaussian, dL_dtau
  render_projected(...)                        <- rendered_Image_from_Projected_Gaussians_vectorized (:973-1018)
  compute_analytical_jacobians_all_gaussians   <- same name (:705-760), closed form of GetAnalyticalJcobian (:633-703)
  assemble_dL_dtau(...)                        <- the module-level chain-rule loop (:1587-1695)

All arithmetic runs in libgsaj_hip.so; torch only holds the device buffers.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .rasterizer import _stream

_F, _D = torch.float32, torch.float64


def _dev(x, dtype, device):
    if torch.is_tensor(x):
        return x.to(device=device, dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype, device=device).contiguous()


def l1_seeds(rendered_color, rendered_depth, gt_color, gt_depth, mask):
    """Per-pixel dL/dC (H,W,3), dL/dD (H,W) of the summed masked L1 loss (compare.py:1212-1224)."""
    m = mask.to(rendered_color.dtype)
    gc = torch.sign(rendered_color - gt_color) * m[..., None]
    gd = torch.sign(rendered_depth - gt_depth) * ((gt_depth > 0.0) & mask.bool()).to(rendered_depth.dtype)
    return gc, gd


def project_and_sort(means3D, cov3D6, shs, cam, sh_degree=3, device="cuda:0"):
    """World-frame Gaussians + camera -> the depth-sorted projected Gaussians the NumPy path works on, all on the device
    (fp64 arithmetic on the fp32 inputs, like the reference's Python floats; NO z <= 0.2 cull, global STABLE depth order):
    dict(order [N] int32, mean_2D [N,2], cov_2D [N,2,2], color [N,3], color_raw [N,3], depth [N]) -- every array but `order`
    in sorted order, fp64.  `cam`: mapping with viewmatrix / projmatrix (the rasteriser's transposed 4x4s), campos, fx, fy, W, H."""
    lib = _lib.load()
    dev = torch.device(device)
    m3, c6, sh = _dev(means3D, _F, dev), _dev(cov3D6, _F, dev), _dev(shs, _F, dev)
    vm, pm, cp = _dev(cam["viewmatrix"], _F, dev).reshape(16), _dev(cam["projmatrix"], _F, dev).reshape(16), _dev(cam["campos"], _F, dev)
    N, M = m3.shape[0], sh.shape[1]
    out = dict(order=torch.empty((N,), dtype=torch.int32, device=dev), mean_2D=torch.empty((N, 2), dtype=_D, device=dev),
               cov_2D=torch.empty((N, 2, 2), dtype=_D, device=dev), color=torch.empty((N, 3), dtype=_D, device=dev),
               color_raw=torch.empty((N, 3), dtype=_D, device=dev), depth=torch.empty((N,), dtype=_D, device=dev))
    ws = torch.empty(lib.gsaj_dense_project_workspace_bytes(N), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_dense_project(N, M, int(sh_degree), int(cam["W"]), int(cam["H"]), m3.data_ptr(), c6.data_ptr(),
                                          sh.data_ptr(), vm.data_ptr(), pm.data_ptr(), cp.data_ptr(), float(cam["fx"]),
                                          float(cam["fy"]), out["order"].data_ptr(), out["mean_2D"].data_ptr(),
                                          out["cov_2D"].data_ptr(), out["color"].data_ptr(), out["color_raw"].data_ptr(),
                                          out["depth"].data_ptr(), ws.data_ptr(), _stream(dev)), "gsaj_dense_project")
    return out


def jacobian_test(means3D, cov3D6, opacities, shs, cam, gt_color, gt_depth, mask, sh_degree=3, device="cuda:0"):
    """The reference's Jacobian test end to end on the device (compare.py:1354-1706, pipeline (C) of SURVEY 3): project and
    depth-order the Gaussians, composite them densely, seed the summed masked L1 loss against (gt_color [H,W,3], gt_depth
    [H,W], mask [H,W]), run the dense closed-form backward, the analytical pose Jacobians and the dL/dtau chain rule.
    -> dict with the four arrays the reference saves to Jacob_test_result/ (grad_mu_I_pixel, grad_Sigma_I_pixel,
    grad_depth_per_gaussian in depth-sorted row order, fp32; dL_dtau (6,) fp64) + order, rendered colour / depth."""
    dev = torch.device(device)
    pr = project_and_sort(means3D, cov3D6, shs, cam, sh_degree, device)
    order = pr["order"].long()
    alpha = _dev(opacities, _F, dev).reshape(-1)[order]
    H, W = int(cam["H"]), int(cam["W"])
    img, dep = render_projected(pr["mean_2D"], pr["cov_2D"], pr["color"], pr["depth"], alpha, H, W, device)
    gc, gd = l1_seeds(img, dep, _dev(gt_color, _F, dev), _dev(gt_depth, _F, dev), _dev(mask, torch.bool, dev))
    g_mu, g_S, g_z, g_c = compute_gradients_2D(pr["mean_2D"], pr["cov_2D"], pr["color"], pr["depth"], alpha, gc, gd, device)
    N = order.shape[0]
    mu_h = torch.cat([_dev(means3D, _D, dev), torch.ones((N, 1), dtype=_D, device=dev)], dim=1)
    dmu, dcov = compute_analytical_jacobians_all_gaussians(mu_h, cov3D6, cam["w2c"], cam["fx"], cam["fy"], W, H, device)
    tau, parts = assemble_dL_dtau(pr["order"], g_mu, g_S, g_z, g_c, dmu, dcov, means3D, cam["w2c"], cam["campos"], shs, sh_degree,
                                  device)
    return dict(grad_mu_I_pixel=g_mu, grad_Sigma_I_pixel=g_S, grad_depth_per_gaussian=g_z, grad_color_per_gaussian=g_c,
                dL_dtau=tau, dL_dtau_parts=parts, order=pr["order"], rendered_color=img, rendered_depth=dep, projected=pr)


def compute_gradients_2D(means_2D, covs_2D, colors, depths, alphas, grad_color, grad_depth, device="cuda:0", naive_guards=False,
                         normalised_intrinsics=None):
    """Depth-sorted projected Gaussians + per-pixel seeds -> (grad_mu_I [N,2], grad_Sigma_I [N,2,2],
    grad_depth_per_gaussian [N], grad_color_per_gaussian [N,3]), fp32, rows in sorted order.
    naive_guards=True: edge semantics of the naive per-pixel loop (GSAJ_DENSE_NAIVE_GUARDS).
    normalised_intrinsics=(fx, fy, cx, cy): the variant of Loss_Derivative_script.py:820-979 -- means / covariances in
    normalised image coordinates, pixels at ((col - cx) / fx, (row - cy) / fy) (GSAJ_DENSE_NORMALISED_COORDS)."""
    lib = _lib.load()
    dev = torch.device(device)
    m2, c2 = _dev(means_2D, _F, dev), _dev(covs_2D, _F, dev)
    col, dep, op = _dev(colors, _F, dev), _dev(depths, _F, dev), _dev(alphas, _F, dev).reshape(-1)
    gc, gd = _dev(grad_color, _F, dev), _dev(grad_depth, _F, dev)
    N = m2.shape[0]
    H, W = gd.shape
    assert gc.shape == (H, W, 3) and c2.shape == (N, 2, 2)
    g_mu = torch.empty((N, 2), dtype=_F, device=dev)
    g_S = torch.empty((N, 2, 2), dtype=_F, device=dev)
    g_z = torch.empty((N,), dtype=_F, device=dev)
    g_c = torch.empty((N, 3), dtype=_F, device=dev)
    ws = torch.empty(lib.gsaj_dense_workspace_bytes(N, W, H), dtype=torch.uint8, device=dev)
    flags, intr = (1 if naive_guards else 0), None
    if normalised_intrinsics is not None:
        intr = (ctypes.c_double * 4)(*[float(x) for x in normalised_intrinsics])
        flags |= 2
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_dense_backward(N, W, H, m2.data_ptr(), c2.data_ptr(), col.data_ptr(), dep.data_ptr(),
                                           op.data_ptr(), gc.data_ptr(), gd.data_ptr(), g_mu.data_ptr(), g_S.data_ptr(),
                                           g_z.data_ptr(), g_c.data_ptr(), ws.data_ptr(), flags,
                                           None if intr is None else ctypes.cast(intr, ctypes.c_void_p), _stream(dev)),
                   "gsaj_dense_backward")
    return g_mu, g_S, g_z, g_c


def render_projected(means_2D, covs_2D, colors, depths, alphas, H, W, device="cuda:0"):
    """Dense forward compositor -> colour (H,W,3), depth (H,W)."""
    lib = _lib.load()
    dev = torch.device(device)
    m2, c2 = _dev(means_2D, _F, dev), _dev(covs_2D, _F, dev)
    col, dep, op = _dev(colors, _F, dev), _dev(depths, _F, dev), _dev(alphas, _F, dev).reshape(-1)
    N = m2.shape[0]
    img = torch.empty((H, W, 3), dtype=_F, device=dev)
    d = torch.empty((H, W), dtype=_F, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_dense_render(N, W, H, m2.data_ptr(), c2.data_ptr(), col.data_ptr(), dep.data_ptr(),
                                         op.data_ptr(), img.data_ptr(), d.data_ptr(), _stream(dev)), "gsaj_dense_render")
    return img, d


def compute_analytical_jacobians_all_gaussians(mu_W_all_homo, gaussian_3D_covs, T_cw, fx, fy, W, H, device="cuda:0"):
    """-> dmu_I_dT_all (N,2,6), dcov_I_dT_all (N,4,6), fp64, original index order; the mu-Jacobian is
    in NDC units (x 2fx/W, 2fy/H), the Sigma-Jacobian rows in pixel^2 units (compare.py:724-754)."""
    lib = _lib.load()
    dev = torch.device(device)
    mu = _dev(mu_W_all_homo if torch.is_tensor(mu_W_all_homo) else np.asarray(mu_W_all_homo), _D, dev)[:, :3].contiguous()
    cov = _dev(gaussian_3D_covs, _D, dev)
    T = _dev(np.asarray(T_cw, np.float64).reshape(16), _D, dev)
    N = mu.shape[0]
    dmu = torch.empty((N, 2, 6), dtype=_D, device=dev)
    dcov = torch.empty((N, 4, 6), dtype=_D, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_pose_jacobians(N, T.data_ptr(), mu.data_ptr(), cov.data_ptr(), float(fx), float(fy), int(W),
                                           int(H), dmu.data_ptr(), dcov.data_ptr(), _stream(dev)), "gsaj_pose_jacobians")
    return dmu, dcov


def assemble_dL_dtau(order, grad_mu, grad_Sigma, grad_depth, grad_color, dmu_all, dcov_all, xyz_world, T_cw, campos,
                     shs, sh_degree=3, device="cuda:0"):
    """dL/dtau (6,) fp64 and its four parts (mu, cov, depth, sh) of the NumPy path."""
    lib = _lib.load()
    dev = torch.device(device)
    N = len(order)
    o = _dev(order if torch.is_tensor(order) else np.asarray(order, np.int32), torch.int32, dev)
    gm, gS = _dev(grad_mu, _F, dev), _dev(grad_Sigma, _F, dev)
    gz, gc = _dev(grad_depth, _F, dev), _dev(grad_color, _F, dev)
    dmu, dcov = _dev(dmu_all, _D, dev), _dev(dcov_all, _D, dev)
    mu = _dev(xyz_world if torch.is_tensor(xyz_world) else np.asarray(xyz_world), _D, dev)[:, :3].contiguous()
    T = _dev(np.asarray(T_cw, np.float64).reshape(16), _D, dev)
    cp = _dev(campos, _D, dev)
    sh = _dev(shs, _D, dev)
    M = sh.shape[1]
    out = torch.empty(6, dtype=_D, device=dev)
    parts = torch.empty((4, 6), dtype=_D, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.gsaj_dense_tau(N, M, int(sh_degree), o.data_ptr(), gm.data_ptr(), gS.data_ptr(), gz.data_ptr(),
                                      gc.data_ptr(), dmu.data_ptr(), dcov.data_ptr(), mu.data_ptr(), T.data_ptr(),
                                      cp.data_ptr(), sh.data_ptr(), out.data_ptr(), parts.data_ptr(), _stream(dev)),
                   "gsaj_dense_tau")
    return out, dict(mu=parts[0], cov=parts[1], depth=parts[2], sh=parts[3])
