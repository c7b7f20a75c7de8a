"""CPU oracle (test infrastructure only) of the tracking / mapping photometric + depth L1 losses and the
pixel-gradient seeds they hand to the rasteriser backward -- reference utils/slam_utils.py:56-128
(get_loss_tracking{,_rgb,_rgbd}, get_loss_mapping{,_rgb,_rgbd}).  Pinned against outputs of those reference
functions themselves (CPU autograd, tests/golden/make_loss_goldens.py -> tests/golden/loss_*.npz).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np

TRACKING, MONOCULAR, NO_EXPOSURE = 1, 2, 4


def loss_and_seeds(flags, image, depth, opacity, gt, gt_depth, grad_mask, exposure_a, exposure_b, alpha, rgb_thr):
    """image/gt [3,H,W], depth/opacity [1,H,W], gt_depth [H,W], grad_mask [1,H,W] bool or None.
    Returns dict(loss, l_rgb, l_depth, dL_dimage, dL_ddepth, dL_dopacity, dL_da, dL_db) -- float64 scalars, float32 images."""
    image, gt = np.asarray(image, np.float64), np.asarray(gt, np.float64)
    depth, opacity = np.asarray(depth, np.float64), np.asarray(opacity, np.float64)
    gt_depth = np.asarray(gt_depth, np.float64)[None]
    _, H, W = image.shape
    tracking, mono, noexp = bool(flags & TRACKING), bool(flags & MONOCULAR), bool(flags & NO_EXPOSURE)
    ea = 1.0 if noexp else float(np.exp(np.float64(exposure_a)))
    eb = 0.0 if noexp else float(exposure_b)
    m = (gt.sum(axis=0, keepdims=True) > rgb_thr).astype(np.float64)            # slam_utils.py:68,106
    if tracking and grad_mask is not None:
        m = m * np.asarray(grad_mask, np.float64).reshape(1, H, W)               # :69
    wrgb = opacity if tracking else np.ones_like(opacity)                        # :70 (tracking weights by opacity)
    # the residual (and so its sign) is formed in fp32, as the reference's torch tensors are
    r32 = (np.float32(ea) * image.astype(np.float32) + np.float32(eb)) * m.astype(np.float32) - gt.astype(np.float32) * m.astype(np.float32)
    r = r32.astype(np.float64)
    s = np.sign(r)
    n_rgb = 3.0 * H * W
    l_rgb = float((wrgb * np.abs(r)).sum() / n_rgb)
    k_rgb = (1.0 if mono else alpha) / n_rgb
    dimg = k_rgb * wrgb * m * s * ea
    dop = (k_rgb * np.abs(r).sum(axis=0, keepdims=True)) if tracking else np.zeros_like(opacity)
    da = 0.0 if noexp else float((k_rgb * wrgb * m * s * ea * image).sum())
    db = 0.0 if noexp else float((k_rgb * wrgb * m * s).sum())
    l_d, ddep = 0.0, np.zeros_like(depth)
    if not mono:
        dm = (gt_depth > 0.01).astype(np.float64)                                # :82,122
        if tracking:
            dm = dm * (opacity > 0.95)                                           # :83
        rd = (depth.astype(np.float32) * dm.astype(np.float32) - gt_depth.astype(np.float32) * dm.astype(np.float32)).astype(np.float64)
        l_d = float(np.abs(rd).sum() / (H * W))
        ddep = (1.0 - alpha) / (H * W) * dm * np.sign(rd)
    loss = l_rgb if mono else alpha * l_rgb + (1.0 - alpha) * l_d
    return dict(loss=loss, l_rgb=l_rgb, l_depth=l_d, dL_dimage=dimg.astype(np.float32), dL_ddepth=ddep.astype(np.float32),
                dL_dopacity=dop.astype(np.float32), dL_da=da, dL_db=db)
