#!/usr/bin/env python3
"""One mapping iteration of a window (utils/slam_backend.py:168-262) on the device, cfg2 x 8 keyframes:
  A. the path alone: BatchContext.forward + backward (what bench.py times)
  B. + the window's losses (LossSeedsBatch, one launch) and the keyframe pose step (PoseTrackerBatch, one launch)
  C. + torch.optim.Adam(fused=True) on the Gaussian parameters, fed from the gradient bucket
usage: mapping_iter_bench.py [iterations=100] [workload=cfg2]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import synthetic as syn  # noqa: E402
from gsaj.losses import LossSeedsBatch  # noqa: E402
from gsaj.pose_step import PoseTrackerBatch  # noqa: E402
from gsaj.rasterizer import BatchContext  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    K = 8
    dev = torch.device("cuda:0")
    cam, sc = syn.config_scene(wl)
    cams = syn.keyframe_cameras(K, W=cam["W"], H=cam["H"], fx=cam["fx"], fy=cam["fy"], cx=cam["cx"], cy=cam["cy"])
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    deg = int(round(M ** 0.5)) - 1
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    par = {k: torch.nn.Parameter(t(sc[k])) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    bg, praw = torch.zeros(3, device=dev), t(cams[0]["projmatrix_raw"])
    rng = np.random.default_rng(0)
    gt_color, gt_depth = t(rng.uniform(0, 1, (K, 3, H, W))), t(rng.uniform(0.5, 4, (K, H, W)))
    # learning rates 0: noise for ground truth would walk the poses off the scene
    poses = PoseTrackerBatch([np.ascontiguousarray(c["viewmatrix"].T) for c in cams], praw, dev, lr_rot=0.0, lr_trans=0.0, lr_exposure_a=0.0,
                             lr_exposure_b=0.0)
    opt = torch.optim.Adam(list(par.values()), lr=0.0, fused=True)
    bc, ls = BatchContext(K, P, W, H, M, dev), LossSeedsBatch(K, W, H, dev)
    dLc0, dLd0 = t(rng.normal(size=(K, 3, H, W)) / (3 * H * W)), t(rng.normal(size=(K, 1, H, W)) / (H * W))
    tx, ty = cam["tanfovx"], cam["tanfovy"]
    state = {"n": 0}

    def geo():
        return dict(sh_degree=deg, shs=par["shs"].detach(), scales=par["scales"].detach(), rotations=par["rotations"].detach())

    def it(level):
        views, projs, cps = poses.matrices()
        bc.forward(bg, par["means3D"].detach(), par["opacities"].detach(), views, projs, cps, tx, ty, sync=(state["n"] == 0), **geo())
        state["n"] += 1
        if level == "A":
            bc.backward(bg, par["means3D"].detach(), views, projs, praw, cps, tx, ty, dLc0, dLd0, **geo())
            return
        o = ls(0, 0.95, 0.01, bc.color, bc.depth, bc.opacity, gt_color, gt_depth, None, poses.exposure[:, 0].contiguous(),
               poses.exposure[:, 1].contiguous())
        g = bc.backward(bg, par["means3D"].detach(), views, projs, praw, cps, tx, ty, o["dL_dcolor"], o["dL_ddepth"], **geo())
        poses.step(g["tau_all"], ls.scalars[:, 3:5].contiguous())
        if level == "C":
            par["means3D"].grad, par["opacities"].grad = g["mean3D"], g["opacity"].view_as(par["opacities"])
            par["shs"].grad, par["scales"].grad, par["rotations"].grad = g["sh"].view_as(par["shs"]), g["scale"], g["rot"]
            opt.step()

    out = {"workload": wl, "keyframes": K, "P": P, "W": W, "H": H, "iterations": n}
    for level in "ABC":
        for _ in range(5):
            it(level)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            it(level)
        torch.cuda.synchronize()
        out["ms_" + level] = round(1e3 * (time.perf_counter() - t0) / n, 4)
    out["what"] = {"A": "forward + backward", "B": "A + window losses (1 launch) + keyframe pose step (1 launch)", "C": "B + fused torch Adam on the Gaussians"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
