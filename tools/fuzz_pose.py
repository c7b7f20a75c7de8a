#!/usr/bin/env python3
"""GPU box: gsaj_pose_adam_step against oracle/pose_oracle.PoseAdam on random poses and gradient magnitudes from 1e-14 to 1e2
(Adam turns |g| << eps = 1e-8 into tiny steps: both branches of SO3_exp / V at 1e-5, pose_utils.py:25-58).  usage: fuzz_pose.py LO HI"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import conftest  # noqa: F401,E402
import torch  # noqa: E402
from gsaj import pose_step, synthetic as syn  # noqa: E402
from oracle import pose_oracle  # noqa: E402

dev = torch.device("cuda:0")
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    w2c = np.eye(4); w2c[:3, :3] = R; w2c[:3, 3] = rng.normal(size=3) * 3
    w2c = w2c.astype(np.float32)
    cam = syn.make_camera(w2c.astype(np.float64))
    pt = pose_step.PoseTracker(w2c, cam["projmatrix_raw"], dev)
    ref = pose_oracle.PoseAdam(w2c, 0.003, 0.001, 0.01, 0.01)
    mag = float(10 ** rng.uniform(-14, 2))
    ok = True
    for k in range(int(rng.integers(1, 12))):
        g_tau = (rng.normal(size=6) * mag * float(10 ** rng.uniform(-2, 2))).astype(np.float32)
        g_exp = (rng.normal(size=2) * mag).astype(np.float32)
        pt.step(torch.as_tensor(g_tau, device=dev), torch.as_tensor(g_exp, device=dev))
        r = ref.step(g_tau, g_exp)
        tau = pt.tau.cpu().numpy()
        try:
            np.testing.assert_allclose(tau, r["tau"], rtol=1e-4, atol=3e-10)  # (a component whose first moment cancels: rounding at the size of lr * eps-ish)
            np.testing.assert_allclose(pt.w2c.cpu().numpy(), r["w2c"], rtol=0, atol=2e-5 * max(1.0, float(np.abs(r["w2c"]).max())))
            np.testing.assert_allclose([pt.exposure_a.item(), pt.exposure_b.item()], r["exposure"], rtol=1e-4, atol=3e-7)  # (a running sum of steps of ~1e-2)
            nt = float(np.sqrt((r["tau"].astype(np.float64) ** 2).sum()))
            if abs(nt - 1e-4) > 1e-8:  # (the converged flag may fall either way only within rounding of the threshold)
                assert bool(pt.converged.item() > 0.5) == bool(r["converged"])
        except AssertionError as e:
            ok = False
            print(seed, k, "mag %.1e" % mag, str(e).replace("\n", " ")[:260])
            break
    bad += not ok
print("failed", bad, "of", int(sys.argv[2]) - int(sys.argv[1]))
