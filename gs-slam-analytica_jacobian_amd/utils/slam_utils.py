"""Losses that seed the backward of the render path, with the reference's names and semantics
(utils/slam_utils.py:56-128 get_loss_tracking* / get_loss_mapping*; compute_loss of
Jacobian_test.py:155-196).  Device-agnostic restatement."""
import torch


def _as_depth_tensor(d, like):
    if not torch.is_tensor(d):
        d = torch.from_numpy(d)
    return d.to(dtype=torch.float32, device=like.device)[None]


def _rgb_mask(config, gt_image, shape):
    thr = config["Training"]["rgb_boundary_threshold"]
    return (gt_image.sum(dim=0) > thr).view(*shape)


def get_loss_tracking(config, image, depth, opacity, viewpoint, initialization=False):
    image_ab = torch.exp(viewpoint.exposure_a) * image + viewpoint.exposure_b
    if config["Training"]["monocular"]:
        return get_loss_tracking_rgb(config, image_ab, depth, opacity, viewpoint)
    return get_loss_tracking_rgbd(config, image_ab, depth, opacity, viewpoint)


def get_loss_tracking_rgb(config, image, depth, opacity, viewpoint):
    gt = viewpoint.original_image.to(image.device)
    _, h, w = gt.shape
    mask = _rgb_mask(config, gt, (1, h, w)) * viewpoint.grad_mask
    return (opacity * torch.abs(image * mask - gt * mask)).mean()


def get_loss_tracking_rgbd(config, image, depth, opacity, viewpoint, initialization=False):
    alpha = config["Training"].get("alpha", 0.95)
    gt_depth = _as_depth_tensor(viewpoint.depth, image)
    depth_mask = (gt_depth > 0.01).view(*depth.shape) * (opacity > 0.95).view(*depth.shape)
    l1_rgb = get_loss_tracking_rgb(config, image, depth, opacity, viewpoint)
    l1_depth = torch.abs(depth * depth_mask - gt_depth * depth_mask)
    return alpha * l1_rgb + (1 - alpha) * l1_depth.mean()


def get_loss_mapping(config, image, depth, viewpoint, opacity, initialization=False):
    image_ab = image if initialization else torch.exp(viewpoint.exposure_a) * image + viewpoint.exposure_b
    if config["Training"]["monocular"]:
        return get_loss_mapping_rgb(config, image_ab, depth, viewpoint)
    return get_loss_mapping_rgbd(config, image_ab, depth, viewpoint)


def get_loss_mapping_rgb(config, image, depth, viewpoint):
    gt = viewpoint.original_image.to(image.device)
    _, h, w = gt.shape
    mask = _rgb_mask(config, gt, (1, h, w))
    return torch.abs(image * mask - gt * mask).mean()


def get_loss_mapping_rgbd(config, image, depth, viewpoint, initialization=False):
    alpha = config["Training"].get("alpha", 0.95)
    gt = viewpoint.original_image.to(image.device)
    gt_depth = _as_depth_tensor(viewpoint.depth, image)
    rgb_mask = _rgb_mask(config, gt, depth.shape)
    depth_mask = (gt_depth > 0.01).view(*depth.shape)
    l1_rgb = torch.abs(image * rgb_mask - gt * rgb_mask)
    l1_depth = torch.abs(depth * depth_mask - gt_depth * depth_mask)
    return alpha * l1_rgb.mean() + (1 - alpha) * l1_depth.mean()


def compute_loss(gaussian_model, color, depth, color_gt, depth_gt, mask, compute_depth_loss=True):
    """Masked L1 colour (mean over 3HW) + L1 depth over valid pixels + 10 x isotropic regulariser."""
    m = mask.unsqueeze(0)
    loss = torch.nn.functional.l1_loss(color * m, color_gt * m)
    scales = gaussian_model.get_scaling
    loss = loss + 10.0 * torch.abs(scales - scales.mean(dim=1, keepdim=True)).mean()
    if compute_depth_loss:
        dgt = depth_gt if depth_gt.dim() == 2 else depth_gt.squeeze(0)
        valid = (dgt > 0.0) & mask
        loss = loss + torch.nn.functional.l1_loss(depth.squeeze(0)[valid], dgt[valid])
    return loss
