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


_RESULT_KEYS = ("render", "viewspace_points", "visibility_filter", "radii", "depth", "opacity", "n_touched")


def _settings_for(cam, pc, bg, scale_mod):
    half_x, half_y = 0.5 * cam.FoVx, 0.5 * cam.FoVy
    return GaussianRasterizationSettings(int(cam.image_height), int(cam.image_width), math.tan(half_x), math.tan(half_y), bg,
                                         scale_mod, cam.world_view_transform, cam.full_proj_transform, cam.projection_matrix,
                                         pc.active_sh_degree, cam.camera_center, False, False)


def _python_colours(pc, cam, xyz):
    """pipe.convert_SHs_python: evaluate the SH colours on the host side of the rasteriser (reference :103-113)."""
    coeffs = pc.get_features
    per_channel = coeffs.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
    view_dir = xyz - cam.camera_center.repeat(coeffs.shape[0], 1)
    view_dir = view_dir / view_dir.norm(dim=1, keepdim=True)
    return torch.clamp_min(eval_sh(pc.active_sh_degree, per_channel, view_dir) + 0.5, 0.0)


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None, mask=None):
    """Render the Gaussians of `pc` from `viewpoint_camera`; `bg_color` lives on the device.  None for an empty model."""
    xyz = pc.get_xyz
    n_gauss = xyz.shape[0]
    if n_gauss == 0:
        return None
    # its .grad receives dL/dmean2D: densification reads viewspace_points.grad[:, :2] (gaussian_model.py:767-771)
    screen_pts = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    try:
        screen_pts.retain_grad()
    except Exception:
        pass

    geometry = {"scales": None, "rotations": None, "cov3D_precomp": None}
    if pipe.compute_cov3D_python:
        geometry["cov3D_precomp"] = pc.get_covariance(scaling_modifier)
    else:
        s = pc.get_scaling
        geometry["scales"] = s.repeat(1, 3) if s.shape[-1] == 1 else s  # an isotropic model stores one scale per Gaussian
        geometry["rotations"] = pc.get_rotation
    colour = {"shs": None, "colors_precomp": None}
    if override_color is not None:
        colour["colors_precomp"] = override_color
    elif pipe.convert_SHs_python:
        colour["colors_precomp"] = _python_colours(pc, viewpoint_camera, xyz)
    else:
        colour["shs"] = pc.get_features

    per_gaussian = dict(means3D=xyz, means2D=screen_pts, opacities=pc.get_opacity, **geometry, **colour)
    if mask is not None:
        per_gaussian = {k: (v if v is None else v[mask]) for k, v in per_gaussian.items()}
    image, radii, depth, opacity_img, n_touched = GaussianRasterizer(
        raster_settings=_settings_for(viewpoint_camera, pc, bg_color, scaling_modifier))(
            theta=viewpoint_camera.cam_rot_delta, rho=viewpoint_camera.cam_trans_delta, **per_gaussian)
    if mask is not None:  # scatter the per-Gaussian integer outputs back to full length
        full = [torch.zeros(n_gauss, dtype=t.dtype, device=t.device) for t in (radii, n_touched)]
        full[0][mask], full[1][mask] = radii, n_touched  # (boolean mask or index tensor)
        radii, n_touched = full
    # radius 0 = frustum-culled or degenerate: not visible
    return dict(zip(_RESULT_KEYS, (image, screen_pts, radii > 0, radii, depth, opacity_img, n_touched)))
