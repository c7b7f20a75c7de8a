"""Drop-in `diff_gaussian_rasterization` for MI355X.

Public surface of the reference package (submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py):
`GaussianRasterizationSettings` (13 fields, :186-199), `GaussianRasterizer.forward / .markVisible` (:201-259),
`rasterize_gaussians` (:21-46) and an autograd function taking the same 10 tensors + settings, returning the same
5 outputs, with the same gradient order (:170-182).  Everything below the autograd boundary goes through the C ABI of
include/gsaj.h (gsaj.rasterizer) to hand-written HIP kernels.

Not reproduced on purpose: the fork's debug side effects on the hot path (unconditional prints of matrices / means3D /
grad_tau, :90-91,121-122,167; per-call text-file dumps in rasterizer_impl.cu:255-267,451-463).  `debug=True` keeps the
reference's contract of leaving a `snapshot_{fw,bw}.dump` of the arguments behind when a call fails.
"""
from typing import NamedTuple

import torch

from gsaj import rasterizer as _backend

# field order is part of the API: slam code builds the tuple by keyword, notebooks positionally
GaussianRasterizationSettings = NamedTuple("GaussianRasterizationSettings", [
    ("image_height", int), ("image_width", int), ("tanfovx", float), ("tanfovy", float), ("bg", torch.Tensor),
    ("scale_modifier", float), ("viewmatrix", torch.Tensor), ("projmatrix", torch.Tensor), ("projmatrix_raw", torch.Tensor),
    ("sh_degree", int), ("campos", torch.Tensor), ("prefiltered", bool), ("debug", bool)])

_MSG_COLOURS = "Please provide excatly one of either SHs or precomputed colors!"            # (sic) reference :224
_MSG_COV = "Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!"  # reference :227


def _call(fn, args, kwargs, debug, dump_name):
    """Run a backend entry point; in debug mode keep a CPU copy of the arguments and dump it if the call raises."""
    if not debug:
        return fn(*args, **kwargs)
    frozen = tuple(a.detach().cpu().clone() if torch.is_tensor(a) else a for a in args)
    try:
        return fn(*args, **kwargs)
    except Exception:
        torch.save(frozen, dump_name)
        print("\nAn error occured in %s. Please forward %s for debugging." % (dump_name.split("_")[1].split(".")[0], dump_name))
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """inputs: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, settings"""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho, cfg):
        fwd_args = (cfg.bg, means3D, colors_precomp, opacities, scales, rotations, cfg.scale_modifier, cov3Ds_precomp,
                    cfg.viewmatrix, cfg.projmatrix, cfg.projmatrix_raw, cfg.tanfovx, cfg.tanfovy, cfg.image_height,
                    cfg.image_width, sh, cfg.sh_degree, cfg.campos, cfg.prefiltered, cfg.debug)
        R, color, radii, geom, binning, img, depth, opacity, n_touched = _call(
            _backend.rasterize_gaussians, fwd_args, {}, cfg.debug, "snapshot_fw.dump")
        ctx.cfg, ctx.R = cfg, R
        ctx.save_for_backward(means3D, sh, colors_precomp, scales, rotations, cov3Ds_precomp, radii, geom, binning, img)
        return color, radii, depth, opacity, n_touched

    @staticmethod
    def backward(ctx, g_color, _g_radii, g_depth, _g_opacity, _g_touched):
        cfg = ctx.cfg
        means3D, sh, colors_precomp, scales, rotations, cov3Ds_precomp, radii, geom, binning, img = ctx.saved_tensors
        bwd_args = (cfg.bg, means3D, radii, colors_precomp, scales, rotations, cfg.scale_modifier, cov3Ds_precomp,
                    cfg.viewmatrix, cfg.projmatrix, cfg.projmatrix_raw, cfg.tanfovx, cfg.tanfovy, g_color, g_depth, sh,
                    cfg.sh_degree, cfg.campos, geom, ctx.R, binning, img, cfg.debug)
        out = _call(_backend.rasterize_gaussians_backward, bwd_args, dict(per_gaussian_tau=False), cfg.debug, "snapshot_bw.dump")
        d_mean2D, d_colors, d_opac, d_mean3D, d_cov3D, d_sh, d_scale, d_rot, _per_gaussian_tau, tau = out[:10]
        # the reference sums dL/dtau over Gaussians in Python (:162-164); here the kernel already did: tau = [rho, theta]
        return (d_mean3D, d_mean2D, d_sh, d_colors, d_opac, d_scale, d_rot, d_cov3D,
                tau[3:].view(1, -1), tau[:3].view(1, -1), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, theta, rho,
                        raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     theta, rho, raster_settings)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    @torch.no_grad()
    def markVisible(self, positions):
        """bool [P]: inside the view frustum of the settings' camera."""
        cfg = self.raster_settings
        return _backend.mark_visible(positions, cfg.viewmatrix, cfg.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, theta=None, rho=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception(_MSG_COLOURS)
        has_sr, partial_sr = scales is not None and rotations is not None, scales is not None or rotations is not None
        if (not has_sr and cov3D_precomp is None) or (partial_sr and cov3D_precomp is not None):
            raise Exception(_MSG_COV)
        empty = torch.Tensor([])  # what the reference passes for an absent optional
        shs, colors_precomp, scales, rotations, cov3D_precomp, theta, rho = (
            empty if t is None else t for t in (shs, colors_precomp, scales, rotations, cov3D_precomp, theta, rho))
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   theta, rho, self.raster_settings)
