#!/usr/bin/env python3
"""GPU box: the dense (NumPy-semantics) backward and forward compositors against oracle/dense_oracle on random N, image sizes and
scenes (tests/test_gpu_dense.py::test_dense_large_n_vs_numpy_oracle_and_render is one such case).  usage: fuzz_dense.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
from gsaj import dense, synthetic as syn  # noqa: E402
from oracle import dense_oracle as dor  # noqa: E402


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    N = int(rng.choice([1, 2, 15, 63, 64, 65, 127, 128, 129, 256, 300, 500]))
    W, H = int(rng.integers(8, 200)), int(rng.integers(8, 150))
    f = float(rng.uniform(0.6, 1.4)) * W
    cam = syn.fixture_camera(noisy=bool(rng.integers(0, 2)), orthonormal=True, W=W, H=H, fx=f, fy=f, cx=W / 2 - 0.5, cy=H / 2 - 0.5)
    lo = float(np.log(rng.uniform(0.005, 0.1)))
    sc = syn.make_scene(N, seed, cam, z_range=(float(rng.uniform(0.3, 1.5)), float(rng.uniform(2.0, 6.0))),
                        log_scale_range=(lo, lo + float(rng.uniform(0.2, 2.5))), margin=float(rng.uniform(-0.1, 0.3)))
    cov6 = syn.covariance6(sc["scales"], sc["rotations"])
    m2, c2, dep = dor.project_gaussians(sc["means3D"], cov6, cam["w2c"], cam["fx"], cam["fy"], cam["cx"], cam["cy"], W, H)
    order = np.argsort(dep, kind="stable")
    dirs = dor.view_dirs(sc["means3D"].astype(np.float64), cam["campos"].astype(np.float64))
    col, _ = dor.colors_from_sh(sc["shs"].astype(np.float64), dirs, 3)
    m2, c2, dep, col, op = m2[order], c2[order], dep[order], col[order], sc["opacities"][order, 0]
    gc = rng.choice([-1.0, 0.0, 1.0], size=(H, W, 3)).astype(np.float32)
    gd = rng.choice([-1.0, 0.0, 1.0], size=(H, W)).astype(np.float32)
    want = dor.dense_backward(m2, c2, col, dep, op, gc, gd)
    got = dense.compute_gradients_2D(m2, c2, col, dep, op, gc, gd)
    errs = [rel(a.cpu().numpy(), b) for a, b in zip(got, want)]
    img, d = dense.render_projected(m2, c2, col, dep, op, H, W)
    img_ref, d_ref = dor.dense_render(m2, c2, col, dep, op, H, W)
    errs += [rel(img.cpu().numpy(), img_ref), rel(d.cpu().numpy(), d_ref)]
    if not (max(errs[:-2]) < 2e-4 and max(errs[-2:]) < 1e-4):
        bad += 1
        print(seed, N, W, H, ["%.1e" % e for e in errs])
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
