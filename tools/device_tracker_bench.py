#!/usr/bin/env python3
"""gsaj.tracking.DeviceTracker: ms per tracking iteration (best of 3 runs) -- eager C-ABI calls against the captured hipGraph, and the
loss fused into the compositors against the separate loss kernel -- for a small frame (launch-bound) and the cfg2 frame
(kernel-bound).  usage: device_tracker_bench.py [iterations=300]"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-slam-analytica_jacobian_amd"))
import torch  # noqa: E402
from gsaj import synthetic as syn  # noqa: E402
from gsaj.rasterizer import FrameContext  # noqa: E402
from gsaj.tracking import DeviceTracker  # noqa: E402


def case(name, cam, sc, n):
    dev = torch.device("cuda:0")
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)  # noqa: E731
    P, W, H, M = sc["means3D"].shape[0], cam["W"], cam["H"], sc["shs"].shape[1]
    deg = int(round(M ** 0.5)) - 1
    g = dict(means3D=t(sc["means3D"]), opacities=t(sc["opacities"]), shs=t(sc["shs"]), scales=t(sc["scales"]), rotations=t(sc["rotations"]),
             sh_degree=deg)
    bg = torch.zeros(3, device=dev)
    gt = FrameContext(P, W, H, M, dev)
    gt.forward(bg, g["means3D"], g["opacities"], t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), cam["tanfovx"], cam["tanfovy"],
               sh_degree=deg, shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    w2c = np.ascontiguousarray(cam["viewmatrix"].T).copy()
    w2c[:3, 3] += np.array([0.01, -0.008, 0.012], np.float32)
    out = {"workload": name, "P": P, "W": W, "H": H}
    # eager_ms / graph_ms: the loss fused into the compositors (the default); unfused_ms: forward -> gsaj_loss_seeds -> backward
    for key, use_graph, fused in (("eager_ms", False, True), ("graph_ms", True, True), ("unfused_ms", False, False)):
        tr = DeviceTracker(P, W, H, M, dev, w2c, t(cam["projmatrix_raw"]), cam["tanfovx"], cam["tanfovy"], bg, alpha=0.9, use_graph=use_graph,
                           fused=fused, **g)
        tr.set_frame(gt.color, gt.depth[0])
        tr.iterate(20)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.iterate(n)
            torch.cuda.synchronize()
            best = min(best, 1e3 * (time.perf_counter() - t0) / n)
        out[key] = round(best, 4)
    out["speedup"] = round(out["eager_ms"] / out["graph_ms"], 2)
    out["fused_over_unfused"] = round(out["unfused_ms"] / out["eager_ms"], 3)
    return out


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    W, H = 160, 120
    cam = syn.fixture_camera(noisy=False, orthonormal=True, W=W, H=H, fx=140.0, fy=140.0, cx=79.5, cy=59.5)
    small = syn.make_scene(3000, 11, cam, z_range=(1.0, 4.0), log_scale_range=(math.log(0.02), math.log(0.1)))
    res = [case("160x120, 3000 Gaussians SH-3", cam, small, n)]
    cam2, sc2 = syn.config_scene("cfg2")
    res.append(case("cfg2", cam2, sc2, n))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
