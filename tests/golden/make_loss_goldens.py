#!/usr/bin/env python3
"""Golden vectors for the tracking / mapping losses and their pixel-gradient seeds.

Runs ONLY in the build container (needs /root/reference): imports the reference's own
utils/slam_utils.py (torch only) and evaluates get_loss_tracking / get_loss_mapping with CPU
autograd on seeded synthetic images.  The reference calls `.cuda()` on the ground-truth image; the
stand-in viewpoint hands it an object whose `.cuda()` returns the CPU tensor, so the reference code runs
unmodified.  Only inputs and outputs are stored (tests/golden/loss_*.npz).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils/slam_utils.py"


class _Img:
    def __init__(self, t):
        self.t = t

    def cuda(self):
        return self.t


class _View:
    pass


def main():
    spec = importlib.util.spec_from_file_location("ref_slam_utils", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    H, W = 48, 64
    for seed, monocular, alpha in ((1, False, 0.95), (2, True, 0.95), (3, False, 0.9)):
        rng = np.random.default_rng(seed)
        image = rng.uniform(0, 1, (3, H, W)).astype(np.float32)
        depth = rng.uniform(0.5, 4, (1, H, W)).astype(np.float32)
        opacity = rng.uniform(0.6, 1.0, (1, H, W)).astype(np.float32)
        gt = np.clip(image + rng.normal(0, 0.1, image.shape), 0, 1).astype(np.float32)
        gt[:, :6, :] = 0.0  # below the rgb boundary threshold
        gt_depth = (depth[0] + rng.normal(0, 0.05, (H, W))).astype(np.float32)
        gt_depth[rng.uniform(size=(H, W)) < 0.1] = 0.0  # invalid depth
        grad_mask = (rng.uniform(size=(1, H, W)) < 0.7)
        a, b = np.float32(0.03), np.float32(-0.02)
        cfg = {"Training": {"monocular": monocular, "rgb_boundary_threshold": 0.01, "alpha": alpha}}
        out = dict(image=image, depth=depth, opacity=opacity, gt=gt, gt_depth=gt_depth, grad_mask=grad_mask,
                   exposure_a=a, exposure_b=b, alpha=np.float32(alpha), rgb_boundary_threshold=np.float32(0.01),
                   monocular=np.bool_(monocular))
        for kind in ("tracking", "mapping", "mapping_init"):
            ti = torch.tensor(image, requires_grad=True)
            td = torch.tensor(depth, requires_grad=True)
            to = torch.tensor(opacity, requires_grad=True)
            v = _View()
            v.original_image = _Img(torch.tensor(gt))
            v.depth = gt_depth
            v.grad_mask = torch.tensor(grad_mask)
            v.exposure_a = torch.tensor([a], requires_grad=True)
            v.exposure_b = torch.tensor([b], requires_grad=True)
            if kind == "tracking":
                loss = ref.get_loss_tracking(cfg, ti, td, to, v)
            else:
                loss = ref.get_loss_mapping(cfg, ti, td, v, to, initialization=(kind == "mapping_init"))
            loss.backward()
            z = lambda t, like: (t.grad if t.grad is not None else torch.zeros_like(like)).numpy()  # noqa: E731
            out[kind + "_loss"] = np.float32(loss.item())
            out[kind + "_dL_dimage"] = z(ti, ti)
            out[kind + "_dL_ddepth"] = z(td, td)
            out[kind + "_dL_dopacity"] = z(to, to)
            out[kind + "_dL_da"] = z(v.exposure_a, v.exposure_a)
            out[kind + "_dL_db"] = z(v.exposure_b, v.exposure_b)
        np.savez_compressed(os.path.join(HERE, "loss_seed%d_%dx%d.npz" % (seed, W, H)), **out)
        print("wrote seed", seed, {k: float(out[k]) for k in out if k.endswith("_loss")})


if __name__ == "__main__":
    sys.exit(main())
