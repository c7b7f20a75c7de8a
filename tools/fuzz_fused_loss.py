#!/usr/bin/env python3
"""GPU box: random scenes (the fuzz cases of tests/test_gpu_random.py) through the loss-fused entry points
(gsaj_rasterize_forward_loss / _backward_loss) and through forward -> gsaj_loss_seeds -> backward: every gradient output must be
identical bit for bit (tests/test_gpu_device_tracker.py does this on one scene).  usage: fuzz_fused_loss.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import test_gpu_random as tr  # noqa: E402
import torch  # noqa: E402
from gsaj.losses import MONOCULAR, TRACKING, LossSeeds  # noqa: E402
from gsaj.rasterizer import FrameContext  # noqa: E402

dev = torch.device("cuda:0")
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    P, W, H, deg, cam, sc, bg3, bits = tr._fuzz_case(seed)
    rng = np.random.default_rng(seed + 1)
    M = sc["shs"].shape[1]
    means, opac = t(sc["means3D"]), t(sc["opacities"])
    kw = dict(sh_degree=deg, shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]))
    view, proj, cp, praw = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), t(cam["projmatrix_raw"])
    bg = t(np.array(bg3))
    gt_c = t(rng.uniform(0, 1, (3, H, W)))
    gd = rng.uniform(0.2, 6.0, (1, H, W))
    gd[rng.uniform(size=gd.shape) < 0.1] = 0.0  # invalid depth pixels
    monocular, masked = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    gtd = None if monocular else t(gd)
    mask = t(rng.uniform(size=H * W) > 0.3).to(torch.uint8) if masked else None
    ea, eb = torch.tensor([float(rng.normal(0, 0.1))], device=dev), torch.tensor([float(rng.normal(0, 0.05))], device=dev)
    flags = TRACKING | (MONOCULAR if monocular else 0)
    try:
        a = FrameContext(P, W, H, M, dev, per_gaussian_tau=True, record_bits=bits)
        a.forward(bg, means, opac, view, proj, cp, cam["tanfovx"], cam["tanfovy"], sync=True, **kw)
        ls = LossSeeds(W, H, dev)
        L = ls(flags, 0.9, 0.01, a.color, a.depth, a.opacity, gt_c, gtd, mask, ea, eb)
        ga = a.backward(bg, means, view, proj, praw, cp, cam["tanfovx"], cam["tanfovy"], L["dL_dcolor"], L["dL_ddepth"], **kw)
        want = {n: x.clone() for n, x in ga.items() if torch.is_tensor(x)}
        want_scalars = ls.scalars.clone()
        b = FrameContext(P, W, H, M, dev, per_gaussian_tau=True, record_bits=bits)
        b.forward(bg, means, opac, view, proj, cp, cam["tanfovx"], cam["tanfovy"], sync=True, **kw)
        scalars = torch.zeros(5, device=dev)
        FL = dict(flags=flags, alpha=0.9, rgb_boundary_threshold=0.01, gt_color=gt_c, gt_depth=gtd, grad_mask=mask, exposure_a=ea,
                  exposure_b=eb, scalars=scalars)
        b.forward_loss(FL, bg, means, opac, view, proj, cp, cam["tanfovx"], cam["tanfovy"], **kw)
        gb = b.backward_loss(FL, bg, means, view, proj, praw, cp, cam["tanfovx"], cam["tanfovy"], **kw)
        assert torch.equal(a.color, b.color) and torch.equal(a.depth, b.depth) and torch.equal(a.n_touched, b.n_touched), "images"
        for n, x in want.items():
            assert torch.equal(gb[n], x), "dL/d%s of the fused path differs (max %g of %g)" % (n, float((gb[n] - x).abs().max()), float(x.abs().max()))
        assert float((scalars - want_scalars).abs().max()) <= 4e-6 * float(want_scalars.abs().max()) + 1e-12, ("scalars", scalars.tolist(), want_scalars.tolist())
    except AssertionError as e:
        bad += 1
        print(seed, P, W, H, "mono" if monocular else "rgbd", "masked" if masked else "", str(e)[:300])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
