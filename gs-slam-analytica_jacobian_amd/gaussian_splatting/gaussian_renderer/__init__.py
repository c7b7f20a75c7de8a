"""`render()` with the reference's signature and 7-key result dict
(gaussian_splatting/gaussian_renderer/__init__.py:24-164), on the MI355X rasteriser.

Differences from the reference, all on purpose:
  * no per-call prints of the camera matrices (:73-79);
  * the `mask` branch works: upstream unpacks four values from a five-value result and then
    reads an undefined `n_touched` (:125-138,163); here the masked call returns all five and
    radii / n_touched are scattered back to full length;
  * tensors are created on the model's device instead of a hard-coded "cuda" string.
"""
import math

import torch
from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

from gaussian_splatting.utils.sh_utils import eval_sh


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, mask=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    xyz = pc.get_xyz
    if xyz.shape[0] == 0:
        return None

    # zero tensor whose .grad receives dL/dmean2D (densification reads viewspace_points.grad[:, :2])
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        projmatrix_raw=viewpoint_camera.projection_matrix,
        sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales = pc.get_scaling
        if scales.shape[-1] == 1:  # isotropic model stores one scale per Gaussian
            scales = scales.repeat(1, 3)
        rotations = pc.get_rotation

    shs = colors_precomp = None
    if override_color is not None:
        colors_precomp = override_color
    elif pipe.convert_SHs_python:
        feats = pc.get_features
        shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
        dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
        dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
    else:
        shs = pc.get_features

    def sel(t):
        return t if (t is None or mask is None) else t[mask]

    rendered_image, radii, depth, opacity_img, n_touched = rasterizer(
        means3D=sel(means3D), means2D=sel(means2D), shs=sel(shs), colors_precomp=sel(colors_precomp),
        opacities=sel(opacity), scales=sel(scales), rotations=sel(rotations), cov3D_precomp=sel(cov3D_precomp),
        theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta)

    if mask is not None:
        full_r = torch.zeros(xyz.shape[0], dtype=radii.dtype, device=radii.device)
        full_n = torch.zeros(xyz.shape[0], dtype=n_touched.dtype, device=n_touched.device)
        full_r[mask], full_n[mask] = radii, n_touched
        radii, n_touched = full_r, full_n

    # Gaussians that were frustum-culled or had radius 0 were not visible
    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": radii > 0,
        "radii": radii,
        "depth": depth,
        "opacity": opacity_img,
        "n_touched": n_touched,
    }
