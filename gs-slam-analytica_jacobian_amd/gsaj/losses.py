"""Losses + pixel-gradient seeds in one device pass (C ABI gsaj_loss_seeds).

Mirrors reference utils/slam_utils.py:56-128: `tracking_loss_seeds` = get_loss_tracking, `mapping_loss_seeds` =
get_loss_mapping, with `config["Training"]` keys `monocular`, `rgb_boundary_threshold`, `alpha` (default 0.95).
Instead of a scalar with an autograd graph they return the loss (device scalar) together with what the
rasteriser's backward needs -- dL/dcolor [3,H,W], dL/ddepth [1,H,W] -- and dL/d(exposure a, b), so a tracking
iteration is forward -> this kernel -> backward with no full-frame torch passes in between."""
import torch

from . import _lib

TRACKING, MONOCULAR, NO_EXPOSURE, COMPUTE_LOSS = 1, 2, 4, 8


def _ptr(t):
    return None if t is None else t.data_ptr()


class LossSeeds:
    """Pre-allocated outputs + workspace for one image size; call it once per frame."""

    def __init__(self, W, H, device):
        self.lib = _lib.load()
        self.W, self.H, self.dev = int(W), int(H), torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GsajError("LossSeeds needs a HIP device (there is no CPU path)")
        self.ws = torch.zeros(self.lib.gsaj_loss_workspace_bytes(self.W, self.H), dtype=torch.uint8, device=self.dev)
        self.dL_dcolor = torch.empty((3, self.H, self.W), dtype=torch.float32, device=self.dev)
        self.dL_ddepth = torch.empty((1, self.H, self.W), dtype=torch.float32, device=self.dev)
        self.dL_dopacity = torch.empty((1, self.H, self.W), dtype=torch.float32, device=self.dev)
        self.scalars = torch.zeros(5, dtype=torch.float32, device=self.dev)  # loss, L_rgb, L_depth, dL/da, dL/db

    def __call__(self, flags, alpha, rgb_boundary_threshold, image, depth, opacity, gt_image, gt_depth=None, grad_mask=None,
                 exposure_a=None, exposure_b=None, want_opacity_grad=False):
        for name, t in (("image", image), ("depth", depth), ("opacity", opacity), ("gt_image", gt_image), ("gt_depth", gt_depth),
                        ("exposure_a", exposure_a), ("exposure_b", exposure_b)):
            if t is None:
                continue
            if t.device != self.dev and not (t.device.type == "cuda" and self.dev.index in (None, t.device.index)):
                raise _lib.GsajError("%s must live on %s (got %s)" % (name, self.dev, t.device))
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _lib.GsajError("%s must be a contiguous float32 tensor" % name)
        if tuple(image.shape) != (3, self.H, self.W) or tuple(gt_image.shape) != (3, self.H, self.W):
            raise _lib.GsajError("image / gt_image must be [3, %d, %d]" % (self.H, self.W))
        gm = None
        if grad_mask is not None:
            gm = grad_mask.to(device=self.dev, dtype=torch.uint8).contiguous().view(-1)
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        _lib.check(self.lib.gsaj_loss_seeds(self.W, self.H, int(flags), float(alpha), float(rgb_boundary_threshold), _ptr(image),
                                            _ptr(depth), _ptr(opacity), _ptr(gt_image), _ptr(gt_depth), _ptr(gm),
                                            _ptr(exposure_a), _ptr(exposure_b), _ptr(self.dL_dcolor), _ptr(self.dL_ddepth),
                                            _ptr(self.dL_dopacity) if want_opacity_grad else None, _ptr(self.scalars),
                                            _ptr(self.ws), stream), "gsaj_loss_seeds")
        return {"loss": self.scalars[0], "l1_rgb": self.scalars[1], "l1_depth": self.scalars[2], "dL_dexposure_a": self.scalars[3],
                "dL_dexposure_b": self.scalars[4], "dL_dcolor": self.dL_dcolor, "dL_ddepth": self.dL_ddepth,
                "dL_dopacity": self.dL_dopacity if want_opacity_grad else None}


class LossSeedsBatch:
    """LossSeeds for the K views of a mapping window in one launch (C ABI gsaj_loss_seeds_batch): image / gt_image [K,3,H,W],
    depth / opacity [K,1,H,W], gt_depth / grad_mask [K,H,W], exposure_a / exposure_b [K].  The seeds come out as
    dL_dcolor [K,3,H,W] and dL_ddepth [K,1,H,W] -- what BatchContext.backward takes -- and scalars [K,5] = per keyframe
    (loss, L_rgb, L_depth, dL/da, dL/db); the window's loss is scalars[:, 0].sum() (slam_backend.py:209-232)."""

    def __init__(self, K, W, H, device):
        self.lib = _lib.load()
        self.K, self.W, self.H, self.dev = int(K), int(W), int(H), torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GsajError("LossSeedsBatch needs a HIP device (there is no CPU path)")
        stride = (self.lib.gsaj_loss_workspace_bytes(self.W, self.H) + 255) & ~255
        self.ws = torch.zeros(self.K * stride + 256, dtype=torch.uint8, device=self.dev)
        self._ws_ptr = (self.ws.data_ptr() + 255) & ~255
        f = dict(dtype=torch.float32, device=self.dev)
        self.dL_dcolor = torch.empty((self.K, 3, self.H, self.W), **f)
        self.dL_ddepth = torch.empty((self.K, 1, self.H, self.W), **f)
        self.scalars = torch.zeros((self.K, 5), **f)

    def __call__(self, flags, alpha, rgb_boundary_threshold, image, depth, opacity, gt_image, gt_depth=None, grad_mask=None,
                 exposure_a=None, exposure_b=None):
        K, H, W = self.K, self.H, self.W
        want = {"image": (image, (K, 3, H, W)), "depth": (depth, (K, 1, H, W)), "opacity": (opacity, (K, 1, H, W)),
                "gt_image": (gt_image, (K, 3, H, W)), "gt_depth": (gt_depth, (K, H, W)), "exposure_a": (exposure_a, (K,)),
                "exposure_b": (exposure_b, (K,))}
        for name, (x, shape) in want.items():
            if x is None:
                continue
            if x.device.type != "cuda" or x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != int(torch.tensor(shape).prod()):
                raise _lib.GsajError("%s must be a contiguous float32 device tensor of shape %s" % (name, shape))
        gm = None if grad_mask is None else grad_mask.to(device=self.dev, dtype=torch.uint8).contiguous().view(-1)
        if gm is not None and gm.numel() != K * H * W:
            raise _lib.GsajError("grad_mask must have K*H*W elements")
        _lib.check(self.lib.gsaj_loss_seeds_batch(K, W, H, int(flags), float(alpha), float(rgb_boundary_threshold), _ptr(image), _ptr(depth),
                                                  _ptr(opacity), _ptr(gt_image), _ptr(gt_depth), _ptr(gm), _ptr(exposure_a),
                                                  _ptr(exposure_b), _ptr(self.dL_dcolor), _ptr(self.dL_ddepth), None, _ptr(self.scalars),
                                                  self._ws_ptr, torch.cuda.current_stream(self.dev).cuda_stream), "gsaj_loss_seeds_batch")
        return {"loss": self.scalars[:, 0], "l1_rgb": self.scalars[:, 1], "l1_depth": self.scalars[:, 2],
                "dL_dexposure_a": self.scalars[:, 3], "dL_dexposure_b": self.scalars[:, 4], "dL_dcolor": self.dL_dcolor,
                "dL_ddepth": self.dL_ddepth}


def _cfg(config):
    tr = config["Training"]
    return bool(tr["monocular"]), float(tr["rgb_boundary_threshold"]), float(tr.get("alpha", 0.95))


def _gt_depth(viewpoint, like):
    d = viewpoint.depth
    if d is None:
        return None
    if not torch.is_tensor(d):
        d = torch.from_numpy(d)
    return d.to(dtype=torch.float32, device=like.device).contiguous()


def tracking_loss_seeds(ls, config, image, depth, opacity, viewpoint, want_opacity_grad=False):
    """get_loss_tracking (slam_utils.py:56-88) + its gradients w.r.t. image, depth, exposure."""
    mono, thr, alpha = _cfg(config)
    return ls(TRACKING | (MONOCULAR if mono else 0), alpha, thr, image, depth, opacity, viewpoint.original_image.to(image.device),
              None if mono else _gt_depth(viewpoint, image), getattr(viewpoint, "grad_mask", None), viewpoint.exposure_a,
              viewpoint.exposure_b, want_opacity_grad)


def mapping_loss_seeds(ls, config, image, depth, viewpoint, opacity, initialization=False):
    """get_loss_mapping (slam_utils.py:91-128) + its gradients w.r.t. image, depth, exposure."""
    mono, thr, alpha = _cfg(config)
    flags = (MONOCULAR if mono else 0) | (NO_EXPOSURE if initialization else 0)
    return ls(flags, alpha, thr, image, depth, opacity, viewpoint.original_image.to(image.device),
              None if mono else _gt_depth(viewpoint, image), None, None if initialization else viewpoint.exposure_a,
              None if initialization else viewpoint.exposure_b)


def compute_loss_seeds(ls, color, depth, color_gt, depth_gt, mask, compute_depth_loss=True):
    """The image part of compute_loss (reference Jacobian_test.py:155-196 = compare.py:144-185) + its gradients: masked L1
    colour (mean over 3HW) + L1 depth over the pixels with depth_gt > 0 inside the mask (mean over those pixels).  The third
    term of compute_loss, 10 x the isotropic regulariser, depends on the Gaussians only: IsotropicLoss below."""
    dgt = depth_gt if depth_gt is None or depth_gt.dim() == 2 else depth_gt.squeeze(0)
    flags = COMPUTE_LOSS | (0 if compute_depth_loss else MONOCULAR)
    op = ls.dL_dopacity  # opacity is not an input of compute_loss; any [1,H,W] float buffer satisfies the C ABI's signature
    return ls(flags, 0.0, 0.0, color, depth, op, color_gt, dgt.contiguous() if compute_depth_loss else None, mask)


class IsotropicLoss:
    """weight * mean |s - mean(s, dim=1)| over the scales [P,C] and its gradient, one launch (C ABI gsaj_isotropic_loss):
    the regulariser of compute_loss (weight 10) and of the mapping loss (reference utils/slam_backend.py:229-231)."""

    def __init__(self, P, device):
        self.lib, self.P, self.dev = _lib.load(), int(P), torch.device(device)
        self.ws = torch.zeros(self.lib.gsaj_isotropic_workspace_bytes(self.P), dtype=torch.uint8, device=self.dev)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.dev)

    def __call__(self, scales, weight=10.0, grad_out=None, accumulate=False):
        if scales.dtype != torch.float32 or not scales.is_contiguous() or scales.shape[0] != self.P:
            raise _lib.GsajError("scales must be a contiguous float32 [P, C] tensor")
        g = grad_out if grad_out is not None else torch.empty_like(scales)
        _lib.check(self.lib.gsaj_isotropic_loss(self.P, int(scales.shape[1]), float(weight), scales.data_ptr(), g.data_ptr(),
                                                1 if accumulate else 0, self.loss.data_ptr(), self.ws.data_ptr(),
                                                torch.cuda.current_stream(self.dev).cuda_stream), "gsaj_isotropic_loss")
        return self.loss[0], g
